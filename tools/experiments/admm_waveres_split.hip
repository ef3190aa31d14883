// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3).  admm_waveres.hip with a lane-split, software-pipelined
// backward sweep for the nx = 32, nu = 16 class (template parameter SPLIT): two lanes per x row for the reference's halving tree,
// four lanes per u row for Eigen's GEMV accumulators, d in LDS, the linear cost one step ahead, Quu_inv one step behind.
// Bitwise equal to the compiled reference (the GPU parity tests of the class pass with it) and SLOWER than the row-per-lane
// sweep it was meant to replace: 43.4 ms against 38.3 ms at 16 384 instances, 5.9 against 5.2 ms at 2 048 (MI355X) — 177 -> 141
// vector instructions per backward step, but four LDS hand-offs between the two lane mappings instead of three broadcasts, on an LDS
// unit that the CU's eight waves already keep half busy.  Kept as the record of the measurement; to try it, copy it over
// csrc/admm_waveres.hip and rebuild (TINYMPC_WAVERES_SPLIT=0 then selects the shipped sweep at run time).
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
// Two waves per SIMD (gains 96 + state 100 VGPRs), 13 KB of LDS per wave.
#include "wave_math.h"
#include <cstdlib>
#ifndef TINY_WAVERES_ABLATE_STORES
#define TINY_WAVERES_ABLATE_STORES 0 // timing experiment only (results are wrong): no per-iteration write-through
#endif

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

// SPLIT (round 3, exact arithmetic of the nx = 32, nu = 16 class): the backward sweep spreads every reduction over the lanes the
// row-per-lane mapping leaves idle, at the reference's bits.  The reference's orders for these sizes are trees and packet
// accumulators, which cut cleanly:
//   AmBKt p  (x rows, halving tree over 32)      T(0,32) = T(0,16) + T(16,16): two lanes per row, 16 products + 15 adds each, one
//                                                cross-lane add (v_add_f32 with a quad_perm DPP operand; fp32 addition commutes)
//   Bdyn^T p (u rows, Eigen's row-major GEMV)    four packet accumulators c_j = 0 + t[j] + t[4+j] + ...: four lanes per row, 8 products
//                                                + 8 adds each, then (c0+c2)+(c1+c3) as two cross-lane adds, then 0 + res
//   Kinf^T r (x rows, packet tree over 16)       s_j = (t[j]+t[4+j]) + (t[8+j]+t[12+j]); lane h of the pair takes j = h, h+2 and forms
//                                                s_h + s_{h+2}; (s0+s2)+(s1+s3) is one cross-lane add
// 64 of 64 lanes work in each pass instead of 32 / 16, and the x-row and u-row reductions no longer run one after the other under two
// EXEC masks: 112 vector instructions per backward step instead of 177.  (The forward sweep's sums are sequential in the
// reference — lazy products evaluated per packet of rows — and cannot be cut: it keeps the row-per-lane mapping.)  Values move between
// the two mappings through four small LDS vectors (p double buffered, the linear cost, Bdyn^T p + r, d); LDS operations of one wave
// execute in order, so no barrier is needed.
template <int NX, int NU, bool EXACT, bool SPLIT = false>
__global__ __launch_bounds__(WAVE, 2) void admm_waveres_kernel(const RowParams P)
{
    using PL = WavePlans<NX, NU>;
    static_assert(!SPLIT || (EXACT && NX == 32 && NU == 16), "the split backward sweep is written for the (32, 16) class in exact arithmetic");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *vec = lds;                 // [64] broadcast buffer of lane_products
    const int lane = threadIdx.x;
    // SPLIT: + vecP[2][32] (p, double buffered), vecL[2][64] (linear cost, one step ahead), vecW[2][16] (Bdyn^T p + r, consumed one
    // step later), then dbuf[N][16]: the feed-forward d of every step (it lives in LDS in this variant: the backward sweep produces
    // it in the four-lanes-per-row mapping, the forward sweep reads it one step ahead in the row-per-lane mapping)
    constexpr int LDS_HEAD = SPLIT ? 5 * WAVE : WAVE;
    float *const vecP = lds + WAVE, *const vecL = lds + 2 * WAVE, *const vecW = lds + 4 * WAVE, *const dbuf = lds + LDS_HEAD;
    const int N = P.N;
    float *b = lds + LDS_HEAD + (SPLIT ? N * 16 : 0) + lane; // b[i * WAVE]: slack of step i, in place
    const int ul_ = (lane - NX) & (NU - 1);                   // u row of this lane in the row-per-lane mapping
    const int inst = blockIdx.x;
    const bool is_x = lane < NX, is_u = (lane >= NX) && (lane < NX + NU);
    const int rowbase = (inst * N) * WAVE + lane;
    const float rho = P.rho;
    const float2 *bnd = reinterpret_cast<const float2 *>(P.bounds) + (size_t)inst * P.bounds_inst_stride + lane; // bnd[i * WAVE]
    WaveGains<NX, NU> G;
    G.load(P.mats, lane);
    const float qrow = P.mats[(2 * NX + 2 * NU) * WAVE + lane];
    // split backward sweep: lane l serves x row l >> 1 (half l & 1) in the X pass and u row l >> 2 (accumulator l & 3) in the U pass
    const int xr_ = lane >> 1, xh_ = lane & 1, um_ = lane >> 2, uj_ = lane & 3;
    float G3x[SPLIT ? 16 : 1], GKx[SPLIT ? 8 : 1], G3u[SPLIT ? 8 : 1], GQ[SPLIT ? 16 : 1];
    if constexpr (SPLIT)
    {
#pragma unroll
        for (int k = 0; k < 16; k++) G3x[k] = P.mats[(NX + NU + 16 * xh_ + k) * WAVE + xr_];            // AmBKt(r, 16h + k)
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            GKx[q] = P.mats[(2 * NX + NU + 4 * q + xh_) * WAVE + xr_];                                    // Kinf^T(r, 4q + h)
            GKx[4 + q] = P.mats[(2 * NX + NU + 4 * q + xh_ + 2) * WAVE + xr_];                            // Kinf^T(r, 4q + h + 2)
        }
#pragma unroll
        for (int q = 0; q < 8; q++) G3u[q] = P.mats[(NX + NU + 4 * q + uj_) * WAVE + NX + um_];          // Bdyn^T(m, 4q + j)
#pragma unroll
        for (int k = 0; k < 16; k++) GQ[k] = P.mats[(2 * NX + NU + k) * WAVE + NX + um_];                 // Quu_inv(m, k)
    }
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
            if constexpr (SPLIT) { if (is_u) dbuf[i * 16 + ul_] = pd; }
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
        // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
        float s = x0, pri = 0.f, dua = 0.f, t1 = 0.f;
        float2 lh = bnd[0];
        float b_cur = b[0];
        float d_cur = SPLIT ? dbuf[ul_] : 0.f;
        int o = rowbase;
        // one horizon step: ai/ci = this step's dual and feed-forward; returns the new dual
        auto fwd_step = [&](int i, float ai, float ci) {
            float sv, xn = 0.f;
            if (i < N - 1) wave_lqr_step<NX, NU, EXACT>(G, vec, lane, is_x, is_u, s, SPLIT ? d_cur : ci, sv, xn);
            else sv = is_x ? s : 0.f;
            const float t0 = sv + ai;                                   // admm.cpp:47-48 and the sum of :69-70
            const float t = __builtin_amdgcn_fmed3f(t0, lh.x, lh.y);    // admm.cpp:51-60 (lo := min(lo, hi) on the host)
            const float an = t0 - t;                                    // admm.cpp:69-70  (a + sv) - t
            pri = fmaxf(pri, fabsf(sv - t));                            // admm.cpp:95,97
            dua = fmaxf(dua, fabsf(b_cur - t));                         // admm.cpp:96,98
            b[i * WAVE] = t;
#if !TINY_WAVERES_ABLATE_STORES
            P.vz[o] = b_cur; // v_i | z_i, should this iteration converge
#endif
            t1 = t - an;
            const int inext = i + 1 < N ? i + 1 : i;
            lh = bnd[inext * WAVE];
            b_cur = b[inext * WAVE];
            if constexpr (SPLIT) d_cur = dbuf[inext * 16 + ul_];
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
        if constexpr (SPLIT)
        {
            // Software pipelined over the horizon: step i runs the X pass and the first half of the U pass (both need p_{i+1}, the
            // one broadcast round trip that is on the dependent chain), then finishes d_{i+1} (whose input was written a step
            // ago) and prepares the linear cost of step i - 1 (which does not depend on p at all).
            const bool wr_d = !keep_d; // the last permitted iteration leaves d as its forward sweep used it (x, u are regenerated from it)
            auto finish_d = [&](int k) { // d_k = Quu_inv (Bdyn^T p_{k+1} + r_k), admm.cpp:19
                float td[16];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++)
                {
                    const float4 v = reinterpret_cast<const float4 *>(vecW + 16 * (k & 1))[k4];
                    td[4 * k4 + 0] = GQ[4 * k4 + 0] * v.x; td[4 * k4 + 1] = GQ[4 * k4 + 1] * v.y;
                    td[4 * k4 + 2] = GQ[4 * k4 + 2] * v.z; td[4 * k4 + 3] = GQ[4 * k4 + 3] * v.w;
                }
                const float dsum = 0.f + (0.f + reduce<PLAN_SEQ>(td)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
                if (uj_ == 0)
                {
                    if (wr_d) dbuf[k * 16 + um_] = dsum;
                    P.pd[(inst * N + k) * WAVE + NX + um_] = dsum; // live-out only
                }
            };
            if (is_x) vecP[lane] = pN;
            int cur = 0; // which half of vecP holds p_{i+1}
            auto xu_pass = [&](int i) {
                const float *pc = vecP + 32 * cur, *lc = vecL + WAVE * (i & 1);
                const int obase = (inst * N + i) * WAVE;
                // ---- X pass: p_i = q_i + AmBKt p_{i+1} - Kinf^T r_i (admm.cpp:20), two lanes per row
                float t[16];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++)
                {
                    const float4 v = reinterpret_cast<const float4 *>(pc + 16 * xh_)[k4];
                    t[4 * k4 + 0] = G3x[4 * k4 + 0] * v.x; t[4 * k4 + 1] = G3x[4 * k4 + 1] * v.y;
                    t[4 * k4 + 2] = G3x[4 * k4 + 2] * v.z; t[4 * k4 + 3] = G3x[4 * k4 + 3] * v.w;
                }
                float sx = tree_sum<0, 16>(t);
                sx = sx + dpp_mov<0xB1>(sx); // quad_perm:[1,0,3,2]: T(0,16) + T(16,16), in both lanes of the pair
                const float wvx = lc[xr_] + sx;
                float tA[4], tB[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const float4 v = reinterpret_cast<const float4 *>(lc + NX)[q]; // r_i[4q .. 4q+3]
                    tA[q] = GKx[q] * (xh_ ? v.y : v.x);
                    tB[q] = GKx[4 + q] * (xh_ ? v.w : v.z);
                }
                const float part = ((tA[0] + tA[1]) + (tA[2] + tA[3])) + ((tB[0] + tB[1]) + (tB[2] + tB[3])); // s_h + s_{h+2}
                const float pnx = wvx - (part + dpp_mov<0xB1>(part));                                       // (s0+s2) + (s1+s3)
                if (xh_ == 0)
                {
                    vecP[32 * (cur ^ 1) + xr_] = pnx; // p_i for the next step of the sweep
                    P.pd[obase + xr_] = pnx;          // live-out only
                }
                // ---- U pass, first half: Bdyn^T p_{i+1} + r_i, four lanes per row
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 8; q++) acc = acc + G3u[q] * pc[4 * q + uj_]; // c_j = 0 + t[j] + t[4+j] + ... (Eigen's GEMV accumulators)
                const float a02 = acc + dpp_mov<0x4E>(acc);                       // quad_perm:[2,3,0,1]: c0+c2 | c1+c3
                const float dotu = 0.f + (a02 + dpp_mov<0xB1>(a02));              // (c0+c2) + (c1+c3), then 0 + 1*acc
                if (uj_ == 0) vecW[16 * (i & 1) + um_] = lc[NX + um_] + dotu;
                cur ^= 1;
            };
            // loop index j: the linear cost of step j (state of step j, straight from its registers), after the passes of step j + 1
            // and the rest of d_{j+2}
            float sn_cur = b[(N - 2) * WAVE];
            auto step_j = [&](int j, float aj, float cj) {
                if (j < N - 2) xu_pass(j + 1);
                if (j < N - 3) finish_d(j + 2);
                const float cq = is_x ? cj : -0.f; // u rows: -0 (r = -rho*(znew - y) keeps the sign of a zero difference)
                vecL[WAVE * (j & 1) + lane] = cq - rho * (sn_cur - aj); // [q_j ; r_j] (admm.cpp:80-82), row-per-lane mapping
                sn_cur = b[(j > 0 ? j - 1 : 0) * WAVE];
            };
            if (N - 2 >= 48) step_j(48, a.t0, c.t0);
#pragma unroll 1
            for (int j = (N - 2 < 47 ? N - 2 : 47); j >= 32; j--) step_j(j, a.mid[j - 32], c.mid[j - 32]);
#pragma unroll 1
            for (int j = (N - 2 < 31 ? N - 2 : 31); j >= 0; j--) step_j(j, a.lo[j], c.lo[j]);
            xu_pass(0);
            if (N - 2 >= 1) finish_d(1);
            finish_d(0);
        }
        else
        {
        float p = pN;
        const bool upd_d = is_u && !keep_d;
        o = rowbase + (N - 2) * WAVE;
        float sn_cur = b[(N - 2) * WAVE];
        auto bwd_step = [&](int i, float ai, float ci) {
            const float cq = is_x ? ci : -0.f; // x rows: -(Xref.*Q); u rows: -0 (r = -rho*(znew - y) keeps the sign of a zero difference)
            float pn, dd;
            const float lin = EXACT ? cq - rho * (sn_cur - ai) : __builtin_fmaf(-rho, sn_cur - ai, cq);
            wave_riccati_step<NX, NU, EXACT>(G, vec, lane, is_x, p, lin, pn, dd); // admm.cpp:19-20,80-82
#if !TINY_WAVERES_ABLATE_STORES
            P.pd[o] = is_u ? dd : pn; // [p_i ; d_i] of this sweep (live-out only)
#endif
            p = pn;
            sn_cur = b[(i > 0 ? i - 1 : 0) * WAVE];
            o -= WAVE;
            return upd_d ? dd : ci;
        };
        if (N - 2 >= 48) c.t0 = bwd_step(48, a.t0, c.t0);
#pragma unroll 1
        for (int i = (N - 2 < 47 ? N - 2 : 47); i >= 32; i--) c.mid[i - 32] = bwd_step(i, a.mid[i - 32], c.mid[i - 32]);
#pragma unroll 1
        for (int i = (N - 2 < 31 ? N - 2 : 31); i >= 0; i--) c.lo[i] = bwd_step(i, a.lo[i], c.lo[i]);
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
        float s = x0;
        int o = rowbase;
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            const float ci = (SPLIT && !is_x) ? dbuf[i * 16 + ul_] : c.get(i), ai = a.get(i);
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
    // the split backward sweep (nx = 32, nu = 16, exact) is the default; TINYMPC_WAVERES_SPLIT=0 selects the row-per-lane sweep (A/B timing)
    static const bool split_on = []() { const char *e = getenv("TINYMPC_WAVERES_SPLIT"); return !(e && e[0] == '0'); }();
    if (nx == 32 && nu == 16 && exact && split_on)
    {
        const size_t ldss = (size_t)(5 * WAVE + P.N * 16 + P.N * WAVE) * sizeof(float);
        hipLaunchKernelGGL((admm_waveres_kernel<32, 16, true, true>), dim3(P.batch), dim3(WAVE), ldss, stream, P);
        return hipGetLastError();
    }
    const size_t ldsb = (size_t)(WAVE + P.N * WAVE) * sizeof(float);
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
