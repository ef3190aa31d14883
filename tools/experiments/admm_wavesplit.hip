// admm_wavesplit.hip — state-on-chip kernel for the one-wave-per-instance classes (16 < nx + nu <= 64, N <= 50; BASELINE.json
// configs[3]: nx = 32, nu = 16, N = 50) with the ROW CLASSES SPLIT OVER WAVES: a workgroup solves G instances with XW "x waves"
// (each holds the x rows of 64 / nx instances, one row per lane) and one "u wave" (the u rows of all G instances).
//
// Why: admm_waveres.hip gives an instance one wave, lane r = row r of [x ; u].  Its counters show the kernel
// bound by vector-instruction ISSUE — 148 instructions per horizon step and wave at the ~6 clocks a wave gets per instruction, two
// waves per SIMD — and most of those instructions work for a fraction of the lanes: a stage whose outputs are the x rows (Bdyn u,
// Kinf^T r) runs with 32 of 64 lanes, one for the u rows (Quu_inv (Bdyn^T p + r)) with 16, and the backward products are summed twice,
// once in the x rows' order and once in the u rows' (Eigen's GEMV order), each lane keeping one.  Here every instruction of an x wave
// serves 64 / nx instances and every instruction of the u wave serves G: for nx = 32, nu = 16, G = 4 the instruction count per instance
// and horizon step pair falls from 296 to 154, and the two classes run side by side instead of one after the other.
//
// The price is that the broadcasts (x_i, u_i, p_i, r_i) now cross waves: they go through LDS as before, with a workgroup barrier
// where a wave needs the other class's rows — two per forward step (u_i for Bdyn u_i; x_{i+1}), one per backward step (p_i and
// r_{i-1}, double buffered).  The barrier is `s_waitcnt lgkmcnt(0); s_barrier`: the write-through to HBM is not waited for.
//
// Arithmetic, summation orders and results are those of admm_waveres.hip / admm_wave.hip (wave_math.h): bitwise equal to the
// compiled reference in exact mode.  HBM layout, RowParams and the packed gains are unchanged (row width 64).
#include "wave_math.h"
#include <atomic>
#include <cstdlib>

namespace tinympc
{

typedef float v32f_s __attribute__((ext_vector_type(32)));
typedef float v16f_s __attribute__((ext_vector_type(16)));
constexpr int WAVESPLIT_MAX_N = 50;

struct SplitStepRegs // per-step state of the 50 steps (admm_waveres.hip: StepRegs)
{
    v32f_s lo;
    v16f_s mid;
    float t0, t1;
    __device__ __forceinline__ float get(int i) const { return i < 32 ? lo[i] : (i < 48 ? mid[i - 32] : (i == 48 ? t0 : t1)); }
};

// LDS of one workgroup, in floats
template <int NX, int NU, int XW>
struct SplitLayout
{
    static constexpr int XI = WAVE / NX;   // instances per x wave
    static constexpr int G = XW * XI;      // instances per workgroup
    static constexpr int NWAVES = XW + 1;
    static constexpr int XV = 0;                      // [2][G][NX] x_i (forward sweep), parity of the step
    static constexpr int PV = XV + 2 * G * NX;        // [2][G][NX] p_i (backward sweep)
    static constexpr int UV = PV + 2 * G * NX;        // [G][NU] u_i
    static constexpr int RV = UV + G * NU;            // [2][G][NU] r_i
    static constexpr int WV = RV + 2 * G * NU;        // [G][NU] Bdyn^T p + r (u wave only)
    static constexpr int RS = WV + G * NU;            // [2][G][4] residual maxima {pri_x, pri_u, dua_x, dua_u} as int bits
    static constexpr int HEAD = ((RS + 2 * G * 4 + WAVE - 1) / WAVE) * WAVE;
    static_assert(NX % 4 == 0 && NU % 4 == 0, "16-byte broadcast groups");
    static_assert(XI >= 1 && G * NU <= WAVE, "the u rows of the workgroup's instances fill at most one wave");
    // slack of every wave's rows [NWAVES][N][64], then -(Xref.*Q) of the x waves' rows [XW][N][64]
    static constexpr size_t bytes(int N) { return (size_t)(HEAD + (NWAVES + XW) * N * WAVE) * sizeof(float); }
};

__device__ __forceinline__ void lds_barrier() // LDS traffic of this wave done, then the workgroup barrier; global memory is not waited for
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// what both roles share: identity of a lane, the broadcast vectors, the settings
template <int NX, int NU, int XW>
struct SplitCtx
{
    using L = SplitLayout<NX, NU, XW>;
    int lane, wid, gi, r, row, inst, N, rowbase;
    bool valid;
    float rho;
    float *b;                                  // b[i * WAVE]: slack of step i, in place
    float *xvec, *pvec, *uvec, *rvec, *wvec;   // this lane's instance; + parity * G * NX (xvec, pvec) / G * NU (rvec)
    const float2 *bnd;                         // bnd[i * WAVE]
    const float *mats_row;
    bool cold, zdual;
    __device__ __forceinline__ void init(const RowParams &P, float *lds, bool xrole)
    {
        lane = threadIdx.x & (WAVE - 1);
        wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gi = xrole ? wid * L::XI + lane / NX : lane / NU;
        r = xrole ? lane % NX : lane % NU;
        row = xrole ? r : NX + r;
        const bool lane_ok = xrole ? (lane < L::XI * NX) : (lane < L::G * NU);
        const int inst_raw = blockIdx.x * L::G + gi;
        valid = lane_ok && inst_raw < P.batch;
        inst = valid ? inst_raw : 0;
        N = P.N;
        rowbase = (inst * N) * WAVE + row;
        rho = P.rho;
        b = lds + L::HEAD + wid * (N * WAVE) + lane;
        xvec = lds + L::XV + gi * NX; pvec = lds + L::PV + gi * NX;
        uvec = lds + L::UV + gi * NU; rvec = lds + L::RV + gi * NU; wvec = lds + L::WV + gi * NU;
        bnd = reinterpret_cast<const float2 *>(P.bounds) + (size_t)inst * P.bounds_inst_stride + row;
        mats_row = P.mats + row;
        cold = P.cold_start != 0;
        zdual = cold || (P.duals_zero != 0);
    }
};

// termination_condition for the workgroup's instances (admm.cpp:91-109): the classes' maxima meet in LDS; every lane derives the
// same set of instances still iterating.  Returns false when none is left.
template <int NX, int NU, int XW>
__device__ __forceinline__ bool split_check(const RowParams &P, float *lds, const SplitCtx<NX, NU, XW> &C, bool xrole, float pri, float dua, int &rp, unsigned &live,
                                            bool &act, int &st, float &r_ps, float &r_pi, float &r_ds, float &r_di)
{
    using L = SplitLayout<NX, NU, XW>;
    constexpr int G = L::G;
    int *const rs = reinterpret_cast<int *>(lds + L::RS);
    if (act)
    {
        atomicMax(rs + rp * (G * 4) + C.gi * 4 + (xrole ? 0 : 1), __builtin_bit_cast(int, pri)); // non-negative floats order like ints
        atomicMax(rs + rp * (G * 4) + C.gi * 4 + (xrole ? 2 : 3), __builtin_bit_cast(int, dua));
    }
    lds_barrier();
    unsigned nlive = live;
#pragma unroll
    for (int g = 0; g < G; g++)
    {
        const float4 q = reinterpret_cast<const float4 *>(lds + L::RS + rp * (G * 4))[g];
        const float ps = q.x, pi = q.y, ds = q.z * C.rho, di = q.w * C.rho;
        const bool conv = (ps < P.abs_pri_tol) && (pi < P.abs_pri_tol) && (ds < P.abs_dua_tol) && (di < P.abs_dua_tol);
        if (conv) nlive &= ~(1u << g);
        if (act && g == C.gi)
        {
            r_ps = ps; r_pi = pi; r_ds = ds; r_di = di;
            if (conv) st = TINY_STATUS_SOLVED_;
        }
    }
    live = __builtin_amdgcn_readfirstlane(nlive);
    rp ^= 1;
    if (threadIdx.x < G * 4) rs[rp * (G * 4) + threadIdx.x] = 0; // its last readers passed a barrier since
    act = C.valid && ((live >> C.gi) & 1u);
    return live != 0;
}

template <int G>
__device__ __forceinline__ unsigned split_live0(const RowParams &P)
{
    unsigned live = 0;
#pragma unroll
    for (int g = 0; g < G; g++) live |= ((int)(blockIdx.x * G + g) < P.batch) ? (1u << g) : 0u;
    return live;
}

// ------------------------------------------------------------------------------------------------------------------------------
// x wave: the x rows of 64 / NX instances.  Registers: a = g (50); -(Xref.*Q) sits in LDS (cx), it is read once per backward step.
// ------------------------------------------------------------------------------------------------------------------------------
template <int NX, int NU, bool EXACT, int XW>
__device__ __forceinline__ void split_x_role(const RowParams &P, float *lds)
{
    using PL = WavePlans<NX, NU>;
    using L = SplitLayout<NX, NU, XW>;
    constexpr int G = L::G;
    SplitCtx<NX, NU, XW> C;
    C.init(P, lds, true);
    const int N = C.N, r = C.r;
    const bool valid = C.valid;
    const float rho = C.rho;
    float *const b = C.b;
    float *const cx = lds + L::HEAD + (L::NWAVES + C.wid) * (N * WAVE) + C.lane; // cx[i * WAVE] = -(Xref_i .* Q) (admm.cpp:81)
    const float qrow = C.mats_row[(2 * NX + 2 * NU) * WAVE];
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[C.inst];
    const int xref_off = C.inst * (int)P.xref_inst_stride + C.row;
    auto xref_at = [&](int i) {
        if (P.xref_mode == 1)
        {
            int rw = wstart + i;
            rw = rw < P.table_rows ? rw : P.table_rows - 1;
            return P.xref_table[rw * WAVE + C.row];
        }
        return P.xref[xref_off + i * WAVE];
    };
    SplitStepRegs a;
    float xrN = 0.f;
    {
        auto live_in = [&](int i) {
            const int o = C.rowbase + i * WAVE;
            float xr = 0.f, gy = 0.f, vz = 0.f;
            if (valid)
            {
                xr = xref_at(i);
                gy = C.zdual ? 0.f : P.gy[o];
                vz = C.cold ? 0.f : P.vz[o];
            }
            cx[i * WAVE] = -(xr * qrow);
            b[i * WAVE] = vz;
            xrN = xr;
            return gy;
        };
        a.t0 = a.t1 = 0.f;
#pragma unroll 1
        for (int i = 0; i < (N < 32 ? N : 32); i++) a.lo[i] = live_in(i);
#pragma unroll 1
        for (int i = 32; i < (N < 48 ? N : 48); i++) a.mid[i - 32] = live_in(i);
        if (N > 48) a.t0 = live_in(48);
        if (N > 49) a.t1 = live_in(49);
    }
    const float x0 = valid ? P.xu[C.rowbase] : 0.f;
    float pterm;
    {
        float PT[NX], xv[NX]; // -(Xref_{N-1}^T Pinf) (admm.cpp:83); PT[k] = Pinf(k, r).  This broadcast stays inside the wave.
#pragma unroll
        for (int k = 0; k < NX; k++) PT[k] = C.mats_row[(2 * NX + 2 * NU + 1 + k) * WAVE];
        if (valid) C.xvec[r] = xrN; // (the lanes past the wave's last instance would land in another wave's vector)
        bcast_fetch<0, NX>(xv, C.xvec);
        if constexpr (EXACT)
        {
            float t[NX];
            products_of(t, PT, xv);
            pterm = -wreduce<PL::TERM>(t);
        }
        else pterm = -fma_dot_of(0.f, PT, xv);
    }
    int st = TINY_STATUS_UNSOLVED_, itn = 1;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (!P.cold_start && valid)
    {
        r_ps = P.res[4 * C.inst + 0]; r_pi = P.res[4 * C.inst + 1];
        r_ds = P.res[4 * C.inst + 2]; r_di = P.res[4 * C.inst + 3];
    }
    float pN = 0.f;
    bool ran_bwd = false;
    unsigned live = split_live0<G>(P);
    bool act = valid;
    int rp = 0;
    lds_barrier(); // rs zeroed, the terminal-term broadcast read

    for (int it = 0; it < P.max_iter; ++it)
    {
        float t1 = 0.f;
        float pri = 0.f, dua = 0.f;
        // ---------------- forward sweep ----------------
        {
            float M1[NX], M2[NU]; // Adyn row | Bdyn row
            {
                int oz; // opaque zero: keeps the (loop invariant) loads inside the sweep, where their registers are free again afterwards
                asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
                const float *m = C.mats_row + oz;
#pragma unroll
                for (int k = 0; k < NX; k++) M1[k] = m[k * WAVE];
#pragma unroll
                for (int k = 0; k < NU; k++) M2[k] = m[(NX + k) * WAVE];
            }
            float s = x0;
            float2 lh = C.bnd[0];
            float b_cur = b[0];
            int o = C.rowbase;
            int par = 0;
            float xv[NX];
            if (act) C.xvec[r] = s;
            lds_barrier();
            if (act) bcast_fetch<0, NX>(xv, C.xvec);
            auto fwd_step = [&](int i, float ai) {
                const bool inner = i < N - 1;
                float xn = 0.f, acc = 0.f, an = ai;
                // the next step's bounds: issued BEFORE this step's store, so that waiting for them (vmcnt counts in order) does not
                // wait for that store's acknowledgement as well
                const int inext = i + 1 < N ? i + 1 : i;
                float2 lh_n = lh;
                if (act) lh_n = C.bnd[inext * WAVE];
                if (inner)
                {
                    if (act) // Adyn x_i (admm.cpp:35)
                    {
                        if constexpr (EXACT)
                        {
                            float t[NX];
                            products_of(t, M1, xv);
                            acc = wreduce<PL::FWD_XA>(t);
                        }
                        else acc = fma_dot_of(0.f, M1, xv);
                    }
                    lds_barrier(); // u_i is there
                }
                if (act)
                {
                    float uv[NU];
                    if (inner) bcast_fetch<0, NU>(uv, C.uvec);
                    const float t0 = s + ai;                                    // admm.cpp:47-48 and the sum of :69-70
                    const float t = __builtin_amdgcn_fmed3f(t0, lh.x, lh.y);    // admm.cpp:51-60 (lo := min(lo, hi) on the host)
                    an = t0 - t;                                                // admm.cpp:69-70  (a + sv) - t
                    b[i * WAVE] = t;
                    if (inner) // x_{i+1} = Adyn x_i + Bdyn u_i
                    {
                        if constexpr (EXACT)
                        {
                            float t2[NU];
                            products_of(t2, M2, uv);
                            xn = acc + wreduce<PL::FWD_XB>(t2);
                        }
                        else xn = fma_dot_of(acc, M2, uv);
                        C.xvec[(par ^ 1) * (G * NX) + r] = xn;
                    }
                    pri = fmaxf(pri, fabsf(s - t));                             // admm.cpp:95-98
                    dua = fmaxf(dua, fabsf(b_cur - t));
                    P.vz[o] = b_cur; // v_i, should this iteration converge
                    t1 = t - an;
                }
                if (inner)
                {
                    lds_barrier(); // x_{i+1} is there
                    par ^= 1;
                    if (act) bcast_fetch<0, NX>(xv, C.xvec + par * (G * NX));
                }
                lh = lh_n;
                if (act) b_cur = b[inext * WAVE];
                o += WAVE;
                s = xn;
                return an;
            };
#pragma unroll 1
            for (int i = 0; i < (N < 32 ? N : 32); i++) a.lo[i] = fwd_step(i, a.lo[i]);
#pragma unroll 1
            for (int i = 32; i < (N < 48 ? N : 48); i++) a.mid[i - 32] = fwd_step(i, a.mid[i - 32]);
            if (N > 48) a.t0 = fwd_step(48, a.t0);
            if (N > 49) a.t1 = fwd_step(49, a.t1);
        }
        if (act)
        {
            pN = EXACT ? pterm - rho * t1 : __builtin_fmaf(-rho, t1, pterm); // admm.cpp:83-84
            itn = it + 1;
        }
        if ((it + 1) % P.check_termination == 0)
            if (!split_check<NX, NU, XW>(P, lds, C, true, pri, dua, rp, live, act, st, r_ps, r_pi, r_ds, r_di)) break;
        // ---------------- backward sweep ----------------
        if (act) ran_bwd = true;
        {
            float M3[NX], M45[NU]; // AmBKt row | Kinf^T row
            {
                int oz;
                asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
                const float *m = C.mats_row + oz;
#pragma unroll
                for (int k = 0; k < NX; k++) M3[k] = m[(NX + NU + k) * WAVE];
#pragma unroll
                for (int k = 0; k < NU; k++) M45[k] = m[(2 * NX + NU + k) * WAVE];
            }
            const int top = N - 2;
            int o = C.rowbase + top * WAVE;
            int q = 0;
            float pv[NX], lv[NU];
            float lin = 0.f, tks = 0.f;
            auto linear = [&](float ai, float ci, float sni) { // q_i (admm.cpp:81-82)
                return EXACT ? ci - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, ci);
            };
            auto fetch_r = [&](const float *src) { // r_i and, in exact arithmetic, Kinf^T r_i (admm.cpp:20)
                bcast_fetch<0, NU>(lv, src);
                if constexpr (EXACT)
                {
                    float tk[NU];
                    products_of(tk, M45, lv);
                    tks = wreduce<PL::BWD_PK>(tk);
                }
            };
            if (act)
            {
                lin = linear(a.get(top), cx[top * WAVE], b[top * WAVE]);
                C.pvec[r] = pN;
            }
            lds_barrier();
            if (act)
            {
                bcast_fetch<0, NX>(pv, C.pvec);
                fetch_r(C.rvec);
            }
            auto bwd_step = [&](int i, float an_, bool has_next) { // an_: a of step i - 1
                if (act)
                {
                    float wv, pn;
                    if constexpr (EXACT)
                    {
                        float t[NX];
                        products_of(t, M3, pv);
                        wv = lin + wreduce<PL::BWD_PA>(t); // q + AmBKt*p
                        pn = wv - tks;                     // admm.cpp:20
                    }
                    else
                    {
                        wv = fma_dot_of(lin, M3, pv);
                        pn = fma_dot_of(wv, M45, lv); // -Kinf^T (pack_gains, fast)
                    }
                    C.pvec[(q ^ 1) * (G * NX) + r] = pn;
                    P.pd[o] = pn; // p_i of this sweep (live-out only)
                    if (has_next) lin = linear(an_, cx[(i - 1) * WAVE], b[(i - 1) * WAVE]);
                }
                lds_barrier(); // p_i and r_{i-1} are there
                q ^= 1;
                if (act)
                {
                    bcast_fetch<0, NX>(pv, C.pvec + q * (G * NX));
                    if (has_next) fetch_r(C.rvec + q * (G * NU));
                }
                o -= WAVE;
            };
            if (top >= 48) bwd_step(48, a.mid[15], true);
#pragma unroll 1
            for (int i = (top < 47 ? top : 47); i >= 33; i--) bwd_step(i, a.mid[i - 33], true);
            if (top >= 32) bwd_step(32, a.lo[31], true);
#pragma unroll 1
            for (int i = (top < 31 ? top : 31); i >= 1; i--) bwd_step(i, a.lo[i - 1], true);
            bwd_step(0, 0.f, false);
        }
    }
    {
        // ---------------- live-out: x regenerated from the d of the last executed forward sweep by the same instruction sequence ----------------
        const bool solved = (st == TINY_STATUS_SOLVED_);
        float M1[NX], M2[NU];
        {
            int oz;
            asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
            const float *m = C.mats_row + oz;
#pragma unroll
            for (int k = 0; k < NX; k++) M1[k] = m[k * WAVE];
#pragma unroll
            for (int k = 0; k < NU; k++) M2[k] = m[(NX + k) * WAVE];
        }
        float s = x0;
        int o = C.rowbase;
        int par = 0;
        float xv[NX];
        lds_barrier(); // the loop's last reads of the broadcast vectors
        if (valid) C.xvec[r] = s;
        lds_barrier();
        if (valid) bcast_fetch<0, NX>(xv, C.xvec);
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            const float ai = a.get(i);
            const bool inner = i < N - 1;
            float xn = 0.f, acc = 0.f;
            if (inner)
            {
                if (valid)
                {
                    if constexpr (EXACT)
                    {
                        float t[NX];
                        products_of(t, M1, xv);
                        acc = wreduce<PL::FWD_XA>(t);
                    }
                    else acc = fma_dot_of(0.f, M1, xv);
                }
                lds_barrier();
                if (valid)
                {
                    float uv[NU];
                    bcast_fetch<0, NU>(uv, C.uvec);
                    if constexpr (EXACT)
                    {
                        float t2[NU];
                        products_of(t2, M2, uv);
                        xn = acc + wreduce<PL::FWD_XB>(t2);
                    }
                    else xn = fma_dot_of(acc, M2, uv);
                    C.xvec[(par ^ 1) * (G * NX) + r] = xn;
                }
                lds_barrier();
                par ^= 1;
                if (valid) bcast_fetch<0, NX>(xv, C.xvec + par * (G * NX));
            }
            if (valid)
            {
                P.xu[o] = s;
                const float sni = b[i * WAVE], ci = cx[i * WAVE];
                P.qr[o] = EXACT ? ci - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, ci);
                if (i == N - 1) P.pd[o] = pN;
                else if (C.cold && !ran_bwd) P.pd[o] = 0.f;
                if (!solved) P.vz[o] = sni; // v = vnew happened; a solved instance keeps the stash
                P.vzn[o] = sni;
                P.gy[o] = ai;
            }
            s = xn;
            o += WAVE;
        }
        if (valid && r == 0)
        {
            P.res[4 * C.inst + 0] = r_ps; P.res[4 * C.inst + 1] = r_pi;
            P.res[4 * C.inst + 2] = r_ds; P.res[4 * C.inst + 3] = r_di;
            P.status[C.inst] = st;
            P.iter[C.inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// u wave: the u rows of the workgroup's G instances.  Registers: a = y (50), c = d (50).
// ------------------------------------------------------------------------------------------------------------------------------
template <int NX, int NU, bool EXACT, int XW>
__device__ __forceinline__ void split_u_role(const RowParams &P, float *lds)
{
    using PL = WavePlans<NX, NU>;
    using L = SplitLayout<NX, NU, XW>;
    constexpr int G = L::G;
    SplitCtx<NX, NU, XW> C;
    C.init(P, lds, false);
    const int N = C.N, r = C.r;
    const bool valid = C.valid;
    const float rho = C.rho;
    float *const b = C.b;
    SplitStepRegs a, c;
    {
        auto live_in = [&](int i, float &ai, float &ci) {
            const int o = C.rowbase + i * WAVE;
            float pd = 0.f, gy = 0.f, vz = 0.f;
            if (valid)
            {
                pd = C.cold ? 0.f : P.pd[o];
                gy = C.zdual ? 0.f : P.gy[o];
                vz = C.cold ? 0.f : P.vz[o];
            }
            ci = pd; // d_i
            ai = gy;
            b[i * WAVE] = vz;
        };
        a.t0 = a.t1 = c.t0 = c.t1 = 0.f;
#pragma unroll 1
        for (int i = 0; i < (N < 32 ? N : 32); i++) { float ai, ci; live_in(i, ai, ci); a.lo[i] = ai; c.lo[i] = ci; }
#pragma unroll 1
        for (int i = 32; i < (N < 48 ? N : 48); i++) { float ai, ci; live_in(i, ai, ci); a.mid[i - 32] = ai; c.mid[i - 32] = ci; }
        if (N > 48) live_in(48, a.t0, c.t0);
        if (N > 49) live_in(49, a.t1, c.t1);
    }
    int st = TINY_STATUS_UNSOLVED_;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    bool ran_bwd = false;
    unsigned live = split_live0<G>(P);
    bool act = valid;
    int rp = 0;
    lds_barrier();

    for (int it = 0; it < P.max_iter; ++it)
    {
        // the last permitted iteration must not overwrite d in c: u of an instance that exhausts max_iter comes from the d
        // its last forward sweep used (regenerated in the epilogue); the final d itself is in the pd array
        const bool keep_d = (it == P.max_iter - 1);
        float pri = 0.f, dua = 0.f;
        // ---------------- forward sweep ----------------
        {
            float M1[NX]; // Kinf row (exact) / -Kinf row (fast)
            {
                int oz;
                asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
                const float *m = C.mats_row + oz;
#pragma unroll
                for (int k = 0; k < NX; k++) M1[k] = m[k * WAVE];
            }
            float2 lh = C.bnd[0];
            float b_cur = b[0];
            int o = C.rowbase;
            int par = 0;
            float xv[NX];
            lds_barrier();
            if (act) bcast_fetch<0, NX>(xv, C.xvec);
            auto fwd_step = [&](int i, float ai, float ci) {
                const bool inner = i < N - 1;
                float un = 0.f, an = ai;
                const int inext = i + 1 < N ? i + 1 : i;
                float2 lh_n = lh; // see the x wave
                if (act) lh_n = C.bnd[inext * WAVE];
                if (inner)
                {
                    if (act) // u_i = -Kinf x_i - d_i (admm.cpp:31)
                    {
                        if constexpr (EXACT)
                        {
                            float t[NX];
                            products_of(t, M1, xv);
                            un = -wreduce<PL::FWD_U>(t) - ci;
                        }
                        else un = fma_dot_of(0.f, M1, xv) - ci;
                        C.uvec[r] = un;
                    }
                    lds_barrier(); // u_i is there
                }
                if (act)
                {
                    const float t0 = un + ai;
                    const float t = __builtin_amdgcn_fmed3f(t0, lh.x, lh.y);
                    an = t0 - t;
                    b[i * WAVE] = t;
                    pri = fmaxf(pri, fabsf(un - t));
                    dua = fmaxf(dua, fabsf(b_cur - t));
                    P.vz[o] = b_cur; // z_i, should this iteration converge
                }
                if (inner)
                {
                    lds_barrier(); // x_{i+1} is there
                    par ^= 1;
                    if (act) bcast_fetch<0, NX>(xv, C.xvec + par * (G * NX));
                }
                lh = lh_n;
                if (act) b_cur = b[inext * WAVE];
                o += WAVE;
                return an;
            };
#pragma unroll 1
            for (int i = 0; i < (N < 32 ? N : 32); i++) a.lo[i] = fwd_step(i, a.lo[i], c.lo[i]);
#pragma unroll 1
            for (int i = 32; i < (N < 48 ? N : 48); i++) a.mid[i - 32] = fwd_step(i, a.mid[i - 32], c.mid[i - 32]);
            if (N > 48) a.t0 = fwd_step(48, a.t0, c.t0);
            if (N > 49) a.t1 = fwd_step(49, a.t1, c.t1);
        }
        if ((it + 1) % P.check_termination == 0)
            if (!split_check<NX, NU, XW>(P, lds, C, false, pri, dua, rp, live, act, st, r_ps, r_pi, r_ds, r_di)) break;
        // ---------------- backward sweep ----------------
        if (act) ran_bwd = true;
        {
            float M3[NX], M45[NU]; // Bdyn^T row | Quu_inv row
            {
                int oz;
                asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
                const float *m = C.mats_row + oz;
#pragma unroll
                for (int k = 0; k < NX; k++) M3[k] = m[(NX + NU + k) * WAVE];
#pragma unroll
                for (int k = 0; k < NU; k++) M45[k] = m[(2 * NX + NU + k) * WAVE];
            }
            const bool upd_d = !keep_d;
            const int top = N - 2;
            int o = C.rowbase + top * WAVE;
            int q = 0;
            float pv[NX];
            float lin = 0.f;
            auto linear = [&](float ai, float sni) { // r_i = -rho*(znew - y): -0 + ... keeps the sign of a zero difference (admm.cpp:80)
                return EXACT ? -0.f - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, -0.f);
            };
            if (act)
            {
                lin = linear(a.get(top), b[top * WAVE]);
                C.rvec[r] = lin;
            }
            lds_barrier();
            if (act) bcast_fetch<0, NX>(pv, C.pvec);
            auto bwd_step = [&](int i, float ci, float an_, bool has_next) { // an_: a of step i - 1
                float ret = ci;
                if (act)
                {
                    float wv;
                    if constexpr (EXACT)
                    {
                        float t[NX];
                        products_of(t, M3, pv);
                        wv = lin + wreduce<PL::BWD_TMP>(t); // Bdyn^T*p + r
                    }
                    else wv = fma_dot_of(lin, M3, pv);
                    float wvv[NU];
                    C.wvec[r] = wv; // d_i = Quu_inv (Bdyn^T p + r) (admm.cpp:19): this broadcast stays inside the wave
                    bcast_fetch<0, NU>(wvv, C.wvec);
                    if (has_next)
                    {
                        lin = linear(an_, b[(i - 1) * WAVE]);
                        C.rvec[(q ^ 1) * (G * NU) + r] = lin;
                    }
                    float dd;
                    if constexpr (EXACT)
                    {
                        float td[NU];
                        products_of(td, M45, wvv);
                        if constexpr (PL::GEMV) dd = 0.f + (0.f + wreduce<PLAN_SEQ>(td)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
                        else dd = wreduce<PL::BWD_D>(td);
                    }
                    else dd = fma_dot_of(0.f, M45, wvv);
                    P.pd[o] = dd; // d_i of this sweep (live-out only)
                    if (upd_d) ret = dd;
                }
                lds_barrier(); // p_i and r_{i-1} are there
                q ^= 1;
                if (act) bcast_fetch<0, NX>(pv, C.pvec + q * (G * NX));
                o -= WAVE;
                return ret;
            };
            if (top >= 48) c.t0 = bwd_step(48, c.t0, a.mid[15], true);
#pragma unroll 1
            for (int i = (top < 47 ? top : 47); i >= 33; i--) c.mid[i - 32] = bwd_step(i, c.mid[i - 32], a.mid[i - 33], true);
            if (top >= 32) c.mid[0] = bwd_step(32, c.mid[0], a.lo[31], true);
#pragma unroll 1
            for (int i = (top < 31 ? top : 31); i >= 1; i--) c.lo[i] = bwd_step(i, c.lo[i], a.lo[i - 1], true);
            c.lo[0] = bwd_step(0, c.lo[0], 0.f, false);
        }
    }
    {
        // ---------------- live-out: u regenerated from the d of the last executed forward sweep ----------------
        const bool solved = (st == TINY_STATUS_SOLVED_);
        float M1[NX];
        {
            int oz;
            asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
            const float *m = C.mats_row + oz;
#pragma unroll
            for (int k = 0; k < NX; k++) M1[k] = m[k * WAVE];
        }
        int o = C.rowbase;
        int par = 0;
        float xv[NX];
        lds_barrier();
        lds_barrier();
        if (valid) bcast_fetch<0, NX>(xv, C.xvec);
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            const float ci = c.get(i), ai = a.get(i);
            const bool inner = i < N - 1;
            float un = 0.f;
            if (inner)
            {
                if (valid)
                {
                    if constexpr (EXACT)
                    {
                        float t[NX];
                        products_of(t, M1, xv);
                        un = -wreduce<PL::FWD_U>(t) - ci;
                    }
                    else un = fma_dot_of(0.f, M1, xv) - ci;
                    C.uvec[r] = un;
                }
                lds_barrier();
                lds_barrier();
                par ^= 1;
                if (valid) bcast_fetch<0, NX>(xv, C.xvec + par * (G * NX));
            }
            if (valid)
            {
                P.xu[o] = un;
                const float sni = b[i * WAVE];
                const float lin = EXACT ? -0.f - rho * (sni - ai) : __builtin_fmaf(-rho, sni - ai, -0.f);
                P.qr[o] = inner ? lin : 0.f;
                if (i == N - 1) P.pd[o] = 0.f;
                else if (C.cold && !ran_bwd) P.pd[o] = 0.f;
                if (!solved) P.vz[o] = sni;
                P.vzn[o] = sni;
                P.gy[o] = ai;
            }
            o += WAVE;
        }
    }
}

template <int NX, int NU, bool EXACT, int XW>
__global__ __launch_bounds__(WAVE *(XW + 1), 2) void admm_wavesplit_kernel(const RowParams P)
{
    using L = SplitLayout<NX, NU, XW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        const int inst = blockIdx.x * L::G + threadIdx.x;
        if (threadIdx.x < L::G && inst < P.batch)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    if (threadIdx.x < 2 * L::G * 4) reinterpret_cast<int *>(lds + L::RS)[threadIdx.x] = 0;
    if (wid < XW) split_x_role<NX, NU, EXACT, XW>(P, lds);
    else split_u_role<NX, NU, EXACT, XW>(P, lds);
}

// x waves per workgroup for a class: as many as make the u wave full (G * nu <= 64), at most 3
template <int NX, int NU>
struct SplitShape
{
    static constexpr int XI = WAVE / NX;
    static constexpr int by_u = (WAVE / NU) / XI; // x waves whose instances' u rows fit one wave
    static constexpr int XW = by_u < 1 ? 0 : (by_u > 2 ? 2 : by_u);
};

bool wavesplit_supported(int nx, int nu, int N)
{
    if (N > WAVESPLIT_MAX_N || N < 2) return false;
#define TINY_WAVESPLIT_OK(NX, NU) \
    if (nx == NX && nu == NU) return (NX % 4 == 0) && (NU % 4 == 0) && SplitShape<NX, NU>::XW >= 1;
    TINY_FOR_EACH_WAVEDIMS(TINY_WAVESPLIT_OK)
    return false;
}

template <int NX, int NU, bool EXACT>
static hipError_t launch_split(const RowParams &P, hipStream_t stream)
{
    if constexpr ((NX % 4 == 0) && (NU % 4 == 0) && SplitShape<NX, NU>::XW >= 1)
    {
        constexpr int XW = SplitShape<NX, NU>::XW;
        using L = SplitLayout<NX, NU, XW>;
        const size_t ldsb = L::bytes(P.N);
        if (ldsb > 64 * 1024) // opt in once per device
        {
            static std::atomic<unsigned long long> done{0};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            if (dev >= 64 || !((done.load() >> dev) & 1ull))
            {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&admm_wavesplit_kernel<NX, NU, EXACT, XW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)L::bytes(WAVESPLIT_MAX_N));
                if (e != hipSuccess) return e;
                if (dev < 64) done.fetch_or(1ull << dev);
            }
        }
        hipLaunchKernelGGL((admm_wavesplit_kernel<NX, NU, EXACT, XW>), dim3((P.batch + L::G - 1) / L::G), dim3(WAVE * L::NWAVES), ldsb, stream, P);
        return hipGetLastError();
    }
    else return hipErrorInvalidValue;
}

hipError_t launch_admm_wavesplit(int nx, int nu, bool exact, const RowParams &P, hipStream_t stream)
{
#define TINY_WAVESPLIT_DISPATCH(NX, NU) \
    if (nx == NX && nu == NU) return exact ? launch_split<NX, NU, true>(P, stream) : launch_split<NX, NU, false>(P, stream);
    TINY_FOR_EACH_WAVEDIMS(TINY_WAVESPLIT_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
