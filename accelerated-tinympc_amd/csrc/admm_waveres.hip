// admm_waveres.hip — state-on-chip kernel (exact and fma arithmetic) for problem classes with 16 < nx + nu <= 64 and N <= 50
// (BASELINE.json configs[3]: nx = 32, nu = 16, N = 50): ONE WAVEFRONT = ONE INSTANCE, the loop-carried state in registers/LDS.
//
// Same mapping, arithmetic and results as admm_wave.hip (lane r owns row r of [x ; u]; wave_math.h: bitwise equal to the
// compiled reference), but the streaming kernel moves ~140 KB per instance and iteration through L2/HBM — 2 048 instances
// already exceed the L2, so configs[3] was bound by that traffic (21 GB per launch at 2 048 instances, 172 GB at 16 384).
// Here, like in admm_rowloop.hip,
//   * the duals a = [g ; y] and c = [-(Xref.*Q) ; d] of the 50 steps live in VGPRs, as register vectors indexed dynamically
//     by the horizon step (32 + 16 + 2 registers each: a gfx950 register tuple has at most 32 entries);
//   * the slack is ONE LDS word per step, updated in place: entering a forward sweep b[i] = v_i | z_i, the sweep reads it for
//     the dual residual and overwrites it with vnew_i | znew_i;
//   * what leaves the chip per iteration is write-only and off the dependent chain: the replaced slack (the live-out v | z
//     should this iteration converge, admm.cpp:135-142) and [p ; d] of the backward sweep (live-out only), 512 B per step.
// The bounds come from their table in global memory one step ahead (shared by the batch or per instance).
// Two waves per SIMD, 13 KB of LDS per wave.
//
// Round 3: the sweeps are software pipelined around their LDS broadcasts.  Counters had shown a wave parked
// in s_waitcnt for 27 % of its life — two (forward) and three (backward) broadcast round trips per horizon step, each waited out in
// place — and the vector pipe of a SIMD 58 % busy.  Now
//   * a broadcast is issued (store + 16-byte reads) as soon as its input exists and consumed as late as possible: in the forward step
//     the slack / dual arithmetic of the step sits behind the issue of the u broadcast and the residual maxima behind that of x_{i+1};
//     in the backward step the broadcast of p_i is issued first, then the linear cost of step i - 1 with ITS broadcast and
//     Kinf^T r_{i-1} (none of which depends on p), then Quu_inv (Bdyn^T p + r) of step i — three broadcasts in flight together and only
//     the products of the next step wait for the first;
//   * the gains of a sweep are loaded at its head (48 registers, L1/L2 resident, once per 49 steps) instead of holding both sweeps'
//     96 for the whole solve: that is what makes room for the values in flight (and removes the scratch spills).
// The arithmetic and its order are untouched (wave_math.h), results stay bitwise.
#include "wave_math.h"

namespace tinympc
{

typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int WAVERES_MAX_N = 50;

// per-step state of the 50 steps: [0,32) in a 32-register vector, [32,48) in a 16-register vector, 48 and 49 in scalars
struct StepRegs
{
    v32f lo;
    v16f mid;
    float t0, t1;
    __device__ __forceinline__ float get(int i) const { return i < 32 ? lo[i] : (i < 48 ? mid[i - 32] : (i == 48 ? t0 : t1)); }
};

template <int NX, int NU, bool EXACT>
__global__ __launch_bounds__(WAVE, 2) void admm_waveres_kernel(const RowParams P)
{
    using PL = WavePlans<NX, NU>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const vec = lds;               // [64] x_i | u_i broadcast of the forward sweep (and of the terminal term)
    float *const vecP = lds + WAVE;       // [64] p broadcast of the backward sweep
    float *const vecL = lds + 2 * WAVE;   // [64] linear cost [q ; r] broadcast
    float *const vecW = lds + 3 * WAVE;   // [64] q + AmBKt p | Bdyn^T p + r broadcast
    const int lane = threadIdx.x;
    float *b = lds + 4 * WAVE + lane;     // b[i * WAVE]: slack of step i, in place
    const int inst = blockIdx.x;
    const bool is_x = lane < NX, is_u = (lane >= NX) && (lane < NX + NU);
    const int N = P.N;
    const int rowbase = (inst * N) * WAVE + lane;
    const float rho = P.rho;
    const float2 *bnd = reinterpret_cast<const float2 *>(P.bounds) + (size_t)inst * P.bounds_inst_stride + lane; // bnd[i * WAVE]
    const float qrow = P.mats[(2 * NX + 2 * NU) * WAVE + lane];
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[inst];
    const int xref_off = inst * (int)P.xref_inst_stride + lane;
    auto xref_at = [&](int i) {
        if (P.xref_mode == 1)
        {
            int row = wstart + i;
            row = row < P.table_rows ? row : P.table_rows - 1;
            return P.xref_table[row * WAVE + lane];
        }
        return P.xref[xref_off + i * WAVE];
    };
    const bool cold = P.cold_start != 0, zdual = cold || (P.duals_zero != 0);

    // ---- live-in: a, c into registers, the slack into LDS ----
    StepRegs a, c;
    float xrN = 0.f;
    {
        auto live_in = [&](int i, float &ai, float &ci) {
            const int o = rowbase + i * WAVE;
            const float xr = xref_at(i);
            const float pd = cold ? 0.f : P.pd[o];
            ci = is_x ? -(xr * qrow) : pd; // admm.cpp:81 | d_i
            ai = zdual ? 0.f : P.gy[o];
            b[i * WAVE] = cold ? 0.f : P.vz[o];
            xrN = xr;
        };
        a.t0 = a.t1 = c.t0 = c.t1 = 0.f;
#pragma unroll 1
        for (int i = 0; i < (N < 32 ? N : 32); i++) { float ai, ci; live_in(i, ai, ci); a.lo[i] = ai; c.lo[i] = ci; }
#pragma unroll 1
        for (int i = 32; i < (N < 48 ? N : 48); i++) { float ai, ci; live_in(i, ai, ci); a.mid[i - 32] = ai; c.mid[i - 32] = ci; }
        if (N > 48) live_in(48, a.t0, c.t0);
        if (N > 49) live_in(49, a.t1, c.t1);
    }
    const float x0 = P.xu[rowbase];
    float pterm;
    {
        float PT[NX], t[NX]; // -(Xref_{N-1}^T Pinf) (admm.cpp:83), x rows; PT[k] = Pinf(k, r)
#pragma unroll
        for (int k = 0; k < NX; k++) PT[k] = P.mats[(2 * NX + 2 * NU + 1 + k) * WAVE + lane];
        if constexpr (EXACT)
        {
            lane_products<0, NX>(t, xrN, PT, vec, lane);
            pterm = -wreduce<PL::TERM>(t);
        }
        else pterm = -lane_fma_dot<0, NX>(0.f, xrN, PT, vec, lane);
    }
    int st = TINY_STATUS_UNSOLVED_, itn = 1;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (!P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    auto wave_max = [](float v) { // max over the lanes of the wave, every lane gets the result
        v = fmaxf(v, dpp_mov<0x128>(v)); v = fmaxf(v, dpp_mov<0x124>(v)); v = fmaxf(v, dpp_mov<0x122>(v)); v = fmaxf(v, dpp_mov<0x121>(v));
        float m = v;
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)));
        m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
        return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)));
    };
    float pN = 0.f;
    bool ran_bwd = false;

    for (int it = 0; it < P.max_iter; ++it)
    {
        // the last permitted iteration must not overwrite d in c: x,u of an instance that exhausts max_iter come from the d
        // its last forward sweep used (regenerated in the epilogue); the final d itself is in the pd array
        const bool keep_d = (it == P.max_iter - 1);
        float t1 = 0.f;
        float pri = 0.f, dua = 0.f;
        // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
        {
            WaveGainsF<NX, NU> GF;
            int oz; // an opaque zero: the loads are loop invariant, and hoisted out of the iteration loop they would be live across both sweeps again
            asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
            GF.load(P.mats + oz, lane);
            float s = x0;
            float2 lh = bnd[0];
            float b_cur = b[0];
            int o = rowbase;
            float xv[NX]; // x_i as every lane sees it
            vec[lane] = s;
            bcast_fetch<0, NX>(xv, vec);
            auto fwd_step = [&](int i, float ai, float ci) {
                float sv, xn = 0.f, acc = 0.f;
                float uv[NU];
                if (i < N - 1) // u_i = -Kinf x_i - d_i (admm.cpp:31): products of the broadcast x_i, summed in the reference's order
                {
                    float un;
                    if constexpr (EXACT)
                    {
                        float t[NX];
                        products_of(t, GF.M1, xv);
                        if constexpr (PL::FWD_U == PL::FWD_XA) acc = wreduce<PL::FWD_XA>(t);
                        else acc = is_x ? wreduce<PL::FWD_XA>(t) : wreduce<PL::FWD_U>(t);
                        un = -acc - ci; // u rows of M1 hold +Kinf: -(K x) - d with the sum negated, like the reference
                    }
                    else
                    {
                        acc = fma_dot_of(0.f, GF.M1, xv); // u rows of M1 hold -Kinf (pack_gains, fast)
                        un = acc - ci;
                    }
                    vec[lane] = un;               // broadcast of u_i: issued here ...
                    bcast_fetch<NX, NU>(uv, vec);
                    sv = is_u ? un : s;
                }
                else sv = is_x ? s : 0.f;
                // ... and the slack / dual update of the step runs while it is in flight
                const float t0 = sv + ai;                                   // admm.cpp:47-48 and the sum of :69-70
                const float t = __builtin_amdgcn_fmed3f(t0, lh.x, lh.y);    // admm.cpp:51-60 (lo := min(lo, hi) on the host)
                const float an = t0 - t;                                    // admm.cpp:69-70  (a + sv) - t
                b[i * WAVE] = t;
                if (i < N - 1) // x_{i+1} = Adyn x_i + Bdyn u_i (admm.cpp:35)
                {
                    if constexpr (EXACT)
                    {
                        float t2[NU];
                        products_of(t2, GF.M2, uv);
                        xn = acc + wreduce<PL::FWD_XB>(t2);
                    }
                    else xn = fma_dot_of(acc, GF.M2, uv);
                    vec[lane] = xn;               // broadcast of x_{i+1}: consumed by the next step,
                    bcast_fetch<0, NX>(xv, vec);
                }
                pri = fmaxf(pri, fabsf(sv - t));                            // ... behind the residual maxima (admm.cpp:95-98)
                dua = fmaxf(dua, fabsf(b_cur - t));
                P.vz[o] = b_cur; // v_i | z_i, should this iteration converge
                t1 = t - an;
                const int inext = i + 1 < N ? i + 1 : i;
                lh = bnd[inext * WAVE];
                b_cur = b[inext * WAVE];
                o += WAVE;
                s = xn;
                return an;
            };
#pragma unroll 1
            for (int i = 0; i < (N < 32 ? N : 32); i++) a.lo[i] = fwd_step(i, a.lo[i], c.lo[i]);
#pragma unroll 1
            for (int i = 32; i < (N < 48 ? N : 48); i++) a.mid[i - 32] = fwd_step(i, a.mid[i - 32], c.mid[i - 32]);
            if (N > 48) a.t0 = fwd_step(48, a.t0, c.t0);
            if (N > 49) a.t1 = fwd_step(49, a.t1, c.t1);
        }
        pN = EXACT ? pterm - rho * t1 : __builtin_fmaf(-rho, t1, pterm); // admm.cpp:83-84
        const float pri_x = wave_max(is_x ? pri : 0.f), dua_x = wave_max(is_x ? dua : 0.f);
        const float pri_u = wave_max(is_u ? pri : 0.f), dua_u = wave_max(is_u ? dua : 0.f);
        itn = it + 1;
        bool conv = false;
        if ((it + 1) % P.check_termination == 0) // admm.cpp:91-109
        {
            r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
            conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
        }
        if (conv) // wave-uniform: the instance is the wave
        {
            st = TINY_STATUS_SOLVED_;
            break;
        }
        // ---------------- backward sweep: (v = vnew is the in-place slack) linear cost + backward_pass_grad ----------------
        ran_bwd = true;
        {
            WaveGainsB<NX, NU> GB;
            int oz;
            asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
            GB.load(P.mats + oz, lane);
            const bool upd_d = is_u && !keep_d;
            const int top = N - 2;
            int o = rowbase + top * WAVE;
            // carried from step to step: the broadcast of p_{i+1}, the linear cost of step i and (exact) Kinf^T r_i, (fma) the broadcast r_i
            float pv[NX], lv[NU];
            float lin, tks = 0.f;
            vecP[lane] = pN;
            bcast_fetch<0, NX>(pv, vecP);
            auto prepare = [&](float ai, float ci, float sni) { // [q ; r] of a step and what of p_i depends on it alone (admm.cpp:80-82, :20)
                const float cq = is_x ? ci : -0.f; // x rows: -(Xref.*Q); u rows: -0 (r = -rho*(znew - y) keeps the sign of a zero difference)
                lin = EXACT ? cq - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, cq);
                vecL[lane] = lin;
                bcast_fetch<NX, NU>(lv, vecL);
            };
            auto finish_prepare = [&]() {
                if constexpr (EXACT)
                {
                    float tk[NU];
                    products_of(tk, GB.M45, lv); // Kinf^T * r (x rows)
                    tks = wreduce<PL::BWD_PK>(tk);
                }
            };
            prepare(a.get(top), c.get(top), b[top * WAVE]);
            finish_prepare();
            float sn_n = b[(top > 0 ? top - 1 : 0) * WAVE]; // slack of the NEXT step of the sweep (i - 1), fetched a step ahead
            // one step: (an_, cn_) are the registers of step i - 1
            auto bwd_step = [&](int i, float ci, float an_, float cn_, bool has_next) {
                float wv, pn;
                if constexpr (EXACT)
                {
                    float t[NX];
                    products_of(t, GB.M3, pv);
                    float dot;
                    if constexpr (PL::BWD_PA == PL::BWD_TMP) dot = wreduce<PL::BWD_PA>(t);
                    else dot = is_x ? wreduce<PL::BWD_PA>(t) : wreduce<PL::BWD_TMP>(t);
                    wv = lin + dot;   // q + AmBKt*p  |  Bdyn^T*p + r
                    pn = wv - tks;    // admm.cpp:20
                }
                else
                {
                    wv = fma_dot_of(lin, GB.M3, pv);
                    pn = fma_dot_of(wv, GB.M45, lv); // x rows hold -Kinf^T (pack_gains, fast)
                }
                vecP[lane] = pn;                  // broadcast of p_i for the next step: first in the queue,
                bcast_fetch<0, NX>(pv, vecP);
                if (has_next) prepare(an_, cn_, sn_n); // then everything of step i - 1 that does not depend on p,
                float wvv[NU];
                vecW[lane] = wv;                  // then d_i = Quu_inv (Bdyn^T p + r) (admm.cpp:19)
                bcast_fetch<NX, NU>(wvv, vecW);
                if (has_next) finish_prepare();
                float dd;
                if constexpr (EXACT)
                {
                    float td[NU];
                    products_of(td, GB.M45, wvv);
                    if constexpr (PL::GEMV) dd = 0.f + (0.f + wreduce<PLAN_SEQ>(td)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
                    else dd = wreduce<PL::BWD_D>(td);
                }
                else dd = fma_dot_of(0.f, GB.M45, wvv); // u rows: Quu_inv
                P.pd[o] = is_u ? dd : pn; // [p_i ; d_i] of this sweep (live-out only)
                sn_n = b[(i > 1 ? i - 2 : 0) * WAVE];
                o -= WAVE;
                return upd_d ? dd : ci;
            };
            // the register of step i - 1 next to that of step i: chunk by chunk (a register tuple has at most 32 entries)
            if (top >= 48) c.t0 = bwd_step(48, c.t0, a.mid[15], c.mid[15], true);
#pragma unroll 1
            for (int i = (top < 47 ? top : 47); i >= 33; i--) c.mid[i - 32] = bwd_step(i, c.mid[i - 32], a.mid[i - 33], c.mid[i - 33], true);
            if (top >= 32) c.mid[0] = bwd_step(32, c.mid[0], a.lo[31], c.lo[31], true);
#pragma unroll 1
            for (int i = (top < 31 ? top : 31); i >= 1; i--) c.lo[i] = bwd_step(i, c.lo[i], a.lo[i - 1], c.lo[i - 1], true);
            c.lo[0] = bwd_step(0, c.lo[0], 0.f, 0.f, false);
        }
    }
    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (lane == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    {
        // ---------------- live-out ----------------
        const bool solved = (st == TINY_STATUS_SOLVED_);
        WaveGains<NX, NU> G;
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        G.load(P.mats + oz, lane);
        float s = x0;
        int o = rowbase;
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            const float ci = c.get(i), ai = a.get(i);
            float sv, xn = 0.f;
            if (i < N - 1) wave_lqr_step<NX, NU, EXACT>(G, vec, lane, is_x, is_u, s, ci, sv, xn);
            else sv = is_x ? s : 0.f;
            P.xu[o] = sv;
            s = xn;
            const float sni = b[i * WAVE];
            const float lin = EXACT ? (is_x ? ci : -0.f) - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, is_x ? ci : -0.f);
            P.qr[o] = (i < N - 1 || is_x) ? lin : 0.f;
            if (i == N - 1) P.pd[o] = is_x ? pN : 0.f;
            else if (cold && !ran_bwd) P.pd[o] = 0.f;
            if (!solved) P.vz[o] = sni; // v = vnew happened; a solved instance keeps the stash
            P.vzn[o] = sni;
            P.gy[o] = ai;
            o += WAVE;
        }
        if (lane == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

bool waveres_supported(int nx, int nu, int N) { return wavedims_supported(nx, nu) && N <= WAVERES_MAX_N; }

hipError_t launch_admm_waveres(int nx, int nu, bool exact, const RowParams &P, hipStream_t stream)
{
    const size_t ldsb = (size_t)(4 * WAVE + P.N * WAVE) * sizeof(float);
#define TINY_WAVERES_DISPATCH(NX, NU)                                                                             \
    if (nx == NX && nu == NU)                                                                                     \
    {                                                                                                             \
        if (exact) hipLaunchKernelGGL((admm_waveres_kernel<NX, NU, true>), dim3(P.batch), dim3(WAVE), ldsb, stream, P);  \
        else hipLaunchKernelGGL((admm_waveres_kernel<NX, NU, false>), dim3(P.batch), dim3(WAVE), ldsb, stream, P);       \
        return hipGetLastError();                                                                                 \
    }
    TINY_FOR_EACH_WAVEDIMS(TINY_WAVERES_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
