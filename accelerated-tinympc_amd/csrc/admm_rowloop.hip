// admm_rowloop.hip — state-on-chip batched TinyMPC ADMM kernel with ROLLED horizon loops (any N <= 64, nx + nu <= 16:
// admm_rowloop_kernel for N <= 32, admm_rowloop64_kernel beyond).
//
// Same mapping, same arithmetic and same results as admm_rowlane.hip (one DPP row of 16 lanes = one instance, lane r owns
// row r of [x ; u], rowlane_math.h), but built for occupancy instead of for the fewest instructions:
//   * the per-step state a = [g ; y] and c = [-(Xref.*Q) ; d] lives in two 32-register vectors indexed DYNAMICALLY by the
//     horizon step (s_set_gpr_idx_on / v_mov), so the sweeps are loops of one step, not 30 unrolled copies;
//   * the slack is ONE LDS word per step, updated in place: entering a forward sweep b[i] = v_i | z_i, the sweep reads it
//     for the dual residual and overwrites it with vnew_i | znew_i (what the backward sweep and the next iteration need).
//     The replaced value is streamed to the vz array: if THIS iteration converges tiny_solve returns before v = vnew
//     (admm.cpp:135-142) and the stash is the live-out v | z, otherwise the epilogue overwrites it.  Like [p ; d] the
//     repeated overwrites of the same lines are absorbed by L2 / Infinity Cache.
// Register and LDS footprint (~145 VGPRs, 384 B of LDS per horizon step and wave) allow 3 waves per SIMD where the
// unrolled kernel fits 2, but the loop and index bookkeeping cost ~45 % more instructions per step: at N = 30 it is 13 %
// slower than the unrolled kernel, which therefore stays the default where it is instantiated.
// N is a run-time value: one instantiation per (nx, nu) serves every horizon up to 32 — 1.9x faster than streaming the
// state through HBM (admm_rowstream_kernel).
#include "rowlane_math.h"

namespace tinympc
{

typedef float v32f __attribute__((ext_vector_type(32)));
constexpr int ROWLOOP_MAX_N = 32;

template <int NX, int NU, bool EXACT, bool H16>
__global__ __launch_bounds__(WAVE, 3) void admm_rowloop_kernel(const RowParams P)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int r16 = lane & 15;
    const int grp = P.order ? P.order[blockIdx.x] : (int)blockIdx.x; // dispatch order (tiny_batch_set_dispatch)
    const int inst = grp * 4 + (lane >> 4);
    const bool valid = (unsigned)inst < (unsigned)P.batch;
    const int inst_a = valid ? inst : P.batch - 1; // load index of a row that stores nothing (padding, or a bad entry of a caller's order)
    const bool is_x = r16 < NX;
    const bool is_u = (r16 >= NX) && (r16 < NX + NU);
    const int N = P.N;
    const float rho = P.rho;

    float2 *bnd = reinterpret_cast<float2 *>(lds); // [N][16] {lo, hi}, shared by the batch
    float *b = lds + N * 32 + lane;                 // b[i * WAVE]
    // batch-shared bounds are staged in LDS once; per-instance bounds (bounds_inst_stride != 0: a [B][N][16] table) are read
    // from global memory one step ahead of their use
    const bool bpi = P.bounds_inst_stride != 0;
    const int bbase = inst_a * (int)P.bounds_inst_stride + r16;
    if (!bpi)
        for (int e = lane; e < N * 16; e += WAVE) bnd[e] = ld_bounds<H16>(P.bounds, e);
    __syncthreads();
    auto bounds_at = [&](int i) { return bpi ? ld_bounds<H16>(P.bounds, bbase + i * 16) : bnd[i * 16 + r16]; };

    RowGains<NX, NU> G;
    G.load(P.mats, r16);

    v32f a, c; // a[i] = g_i | y_i ;  c[i] = -(Xref_i .* Q) | d_i
    const int rowbase = (inst_a * N) * 16 + r16;
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
    const int xref_off = inst_a * (int)P.xref_inst_stride + r16;
    const bool cold = P.cold_start != 0;
    const bool zdual = cold || (P.duals_zero != 0);
    float xrN = 0.f;
    {
        const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16];
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            float xr;
            if (P.xref_mode == 1)
            {
                int row = wstart + i;
                row = row < P.table_rows ? row : P.table_rows - 1;
                xr = ldw<H16>(P.xref_table, row * 16 + r16);
            }
            else
                xr = ldw<H16>(P.xref, xref_off + i * 16);
            const int o = rowbase + i * 16;
            const float pd = cold ? 0.f : ldw<H16>(P.pd, o);
            c[i] = is_x ? rnd<H16>(-(xr * qrow)) : pd; // admm.cpp:81
            b[i * WAVE] = cold ? 0.f : ldw<H16>(P.vz, o);
            a[i] = zdual ? 0.f : ldw<H16>(P.gy, o);
            xrN = xr;
        }
    }
    const float x0 = ldw<H16>(P.xu, rowbase);
    const float pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xrN); // admm.cpp:83

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN = 0.f;
    bool ran_bwd = false;

    bool active = valid && (P.max_iter > 0);
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        // the last permitted iteration must not overwrite d: x,u of an instance that exhausts max_iter come from the
        // d its last forward sweep used; the final d itself is in the pd array
        const bool keep_d = (it == P.max_iter - 1);
        if (active)
        {
            // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
            float s = x0, pri = 0.f, dua = 0.f, t1 = 0.f;
            float2 lh = bounds_at(0);
            float b_cur = b[0];
            int o = rowbase;
            // slack, dual and residual part of step i (sv = [x_i ; u_i]); reloads lh / b_cur for the next step AFTER their
            // last use, so the loop carries them without register rotation
            auto elementwise = [&](int i, int inext, float sv) {
                const float t0 = sv + a[i];                                        // admm.cpp:47-48 and the sum of :69-70
                const float t = __builtin_amdgcn_fmed3f(rnd<H16>(t0), lh.x, lh.y); // admm.cpp:51-60 (lo := min(lo, hi) on the host)
                const float an = rnd<H16>(t0 - t);                                 // admm.cpp:69-70  (a + sv) - t
                a[i] = an;
                pri = max_abs(pri, sv - t);                                        // admm.cpp:95,97
                dua = max_abs(dua, b_cur - t);                                     // admm.cpp:96,98
                b[i * WAVE] = t;
                stw<H16>(P.vz, o, b_cur); // v_i | z_i, should this iteration converge
                t1 = t - an;
                lh = bounds_at(inext);
                b_cur = b[inext * WAVE];
                o += 16;
            };
#pragma unroll 1
            for (int i = 0; i < N - 1; i++)
            {
                float sv, xn;
                lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, c[i], sv, xn);
                elementwise(i, i + 1, sv);
                s = xn;
            }
            elementwise(N - 1, N - 1, is_x ? s : 0.f);
            pN = lin_cost<EXACT, H16>(pterm, rho, t1); // admm.cpp:83-84
            // ---------------- termination_condition (admm.cpp:91-109) ----------------
            const float pri_x = row_max(is_x ? pri : 0.f), dua_x = row_max(is_x ? dua : 0.f);
            const float pri_u = row_max(is_u ? pri : 0.f), dua_u = row_max(is_u ? dua : 0.f);
            itn = it + 1;
            bool conv = false;
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv)
            {
                st = TINY_STATUS_SOLVED_;
                active = false;
            }
            else
            {
                // ---------------- backward sweep: (v = vnew is the in-place slack) linear cost + backward_pass_grad ----------------
                float p = pN;
                ran_bwd = true;
                const bool upd_d = is_u && !keep_d;
                float sn_cur = b[(N - 2) * WAVE];
                o = rowbase + (N - 2) * 16;
#pragma unroll 1
                for (int i = N - 2; i >= 0; i--)
                {
                    const float ci = c[i];
                    const float tb = sn_cur - a[i];
                    const float cq = cost_term(ci, is_x); // x rows: -(Xref.*Q) ; u rows: -0
                    float pn, dd;
                    riccati_step<NX, NU, EXACT, H16>(G, is_x, p, lin_cost<EXACT, H16>(cq, rho, tb), pn, dd); // admm.cpp:19-20,80-82
                    c[i] = upd_d ? dd : ci;
                    stw<H16>(P.pd, o, is_u ? dd : pn); // [p_i ; d_i] of this sweep
                    p = pn;
                    sn_cur = b[(i > 0 ? i - 1 : 0) * WAVE]; // next step's slack, loaded after this step's last use
                    o -= 16;
                }
            }
        }
    }

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && r16 == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }

    // ---------------- live-out ----------------
    if (valid)
    {
        const bool solved = (st == TINY_STATUS_SOLVED_);
        float s = x0;
        int o = rowbase;
#pragma unroll 1
        for (int i = 0; i < N; i++)
        {
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            const float ci = c[i];
            float sv, xn = 0.f;
            if (i < N - 1) lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, ci, sv, xn);
            else sv = is_x ? s : 0.f;
            stw<H16>(P.xu, o, sv);
            s = xn;
            const float sni = b[i * WAVE];
            const float lin = lin_cost<EXACT, H16>(cost_term(ci, is_x), rho, sni - a[i]);
            stw<H16>(P.qr, o, (i < N - 1 || is_x) ? lin : 0.f);
            if (i == N - 1) stw<H16>(P.pd, o, is_x ? pN : 0.f);
            else if (cold && !ran_bwd) stw<H16>(P.pd, o, 0.f);
            if (!solved) stw<H16>(P.vz, o, sni); // v = vnew happened; a solved instance keeps the stash
            stw<H16>(P.vzn, o, sni);
            stw<H16>(P.gy, o, a[i]);
            o += 16;
        }
        if (r16 == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// admm_rowloop64_kernel: the same kernel for 32 < N <= 64.  A gfx950 register tuple has at most 32 entries, so a and c are
// two 32-register vectors each and every sweep is two loops, one per vector (the step body is a lambda that takes and
// returns the step's registers).  128 state registers + 32 of gains: two waves per SIMD; 384 B of LDS per step and wave.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ROWLOOP64_MAX_N = 64;

template <int NX, int NU, bool EXACT, bool H16>
__global__ __launch_bounds__(WAVE, 2) void admm_rowloop64_kernel(const RowParams P)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int r16 = lane & 15;
    const int grp = P.order ? P.order[blockIdx.x] : (int)blockIdx.x;
    const int inst = grp * 4 + (lane >> 4);
    const bool valid = (unsigned)inst < (unsigned)P.batch;
    const int inst_a = valid ? inst : P.batch - 1;
    const bool is_x = r16 < NX;
    const bool is_u = (r16 >= NX) && (r16 < NX + NU);
    const int N = P.N;                      // 32 < N <= 64
    const int NLO = 32, NHI = N - 32;       // steps held by the two vectors
    const float rho = P.rho;

    float2 *bnd = reinterpret_cast<float2 *>(lds); // [N][16] {lo, hi}, shared by the batch
    float *b = lds + N * 32 + lane;                 // b[i * WAVE]
    // batch-shared bounds are staged in LDS once; per-instance bounds (bounds_inst_stride != 0: a [B][N][16] table) are read
    // from global memory one step ahead of their use
    const bool bpi = P.bounds_inst_stride != 0;
    const int bbase = inst_a * (int)P.bounds_inst_stride + r16;
    if (!bpi)
        for (int e = lane; e < N * 16; e += WAVE) bnd[e] = ld_bounds<H16>(P.bounds, e);
    __syncthreads();
    auto bounds_at = [&](int i) { return bpi ? ld_bounds<H16>(P.bounds, bbase + i * 16) : bnd[i * 16 + r16]; };

    RowGains<NX, NU> G;
    G.load(P.mats, r16);

    v32f alo, ahi, clo, chi; // a[i] = g_i | y_i ;  c[i] = -(Xref_i .* Q) | d_i ;  steps [0,32) and [32,N)
    const int rowbase = (inst_a * N) * 16 + r16;
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
    const int xref_off = inst_a * (int)P.xref_inst_stride + r16;
    const bool cold = P.cold_start != 0;
    const bool zdual = cold || (P.duals_zero != 0);
    float xrN = 0.f;
    {
        const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16];
        auto live_in = [&](int i, float &ai, float &ci) {
            float xr;
            if (P.xref_mode == 1)
            {
                int row = wstart + i;
                row = row < P.table_rows ? row : P.table_rows - 1;
                xr = ldw<H16>(P.xref_table, row * 16 + r16);
            }
            else
                xr = ldw<H16>(P.xref, xref_off + i * 16);
            const int o = rowbase + i * 16;
            const float pd = cold ? 0.f : ldw<H16>(P.pd, o);
            ci = is_x ? rnd<H16>(-(xr * qrow)) : pd; // admm.cpp:81
            b[i * WAVE] = cold ? 0.f : ldw<H16>(P.vz, o);
            ai = zdual ? 0.f : ldw<H16>(P.gy, o);
            xrN = xr;
        };
#pragma unroll 1
        for (int i = 0; i < NLO; i++) { float ai, ci; live_in(i, ai, ci); alo[i] = ai; clo[i] = ci; }
#pragma unroll 1
        for (int i = 0; i < NHI; i++) { float ai, ci; live_in(32 + i, ai, ci); ahi[i] = ai; chi[i] = ci; }
    }
    const float x0 = ldw<H16>(P.xu, rowbase);
    const float pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xrN); // admm.cpp:83

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN = 0.f;
    bool ran_bwd = false;

    bool active = valid && (P.max_iter > 0);
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        const bool keep_d = (it == P.max_iter - 1); // see admm_rowloop_kernel
        if (active)
        {
            // ---------------- forward sweep ----------------
            float s = x0, pri = 0.f, dua = 0.f, t1 = 0.f;
            float2 lh = bounds_at(0);
            float b_cur = b[0];
            int o = rowbase;
            auto fwd_step = [&](int i, float ai, float ci) {
                float sv, xn = 0.f;
                if (i < N - 1) lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, ci, sv, xn);
                else sv = is_x ? s : 0.f;
                const float t0 = sv + ai;                                          // admm.cpp:47-48 and the sum of :69-70
                const float t = __builtin_amdgcn_fmed3f(rnd<H16>(t0), lh.x, lh.y); // admm.cpp:51-60
                const float an = rnd<H16>(t0 - t);                                 // admm.cpp:69-70
                pri = max_abs(pri, sv - t);                                        // admm.cpp:95,97
                dua = max_abs(dua, b_cur - t);                                     // admm.cpp:96,98
                b[i * WAVE] = t;
                stw<H16>(P.vz, o, b_cur); // v_i | z_i, should this iteration converge
                t1 = t - an;
                const int inext = i + 1 < N ? i + 1 : i;
                lh = bounds_at(inext);
                b_cur = b[inext * WAVE];
                o += 16;
                s = xn;
                return an;
            };
#pragma unroll 1
            for (int i = 0; i < NLO; i++) alo[i] = fwd_step(i, alo[i], clo[i]);
#pragma unroll 1
            for (int i = 0; i < NHI; i++) ahi[i] = fwd_step(32 + i, ahi[i], chi[i]);
            pN = lin_cost<EXACT, H16>(pterm, rho, t1); // admm.cpp:83-84
            // ---------------- termination_condition (admm.cpp:91-109) ----------------
            const float pri_x = row_max(is_x ? pri : 0.f), dua_x = row_max(is_x ? dua : 0.f);
            const float pri_u = row_max(is_u ? pri : 0.f), dua_u = row_max(is_u ? dua : 0.f);
            itn = it + 1;
            bool conv = false;
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv)
            {
                st = TINY_STATUS_SOLVED_;
                active = false;
            }
            else
            {
                // ---------------- backward sweep ----------------
                float p = pN;
                ran_bwd = true;
                const bool upd_d = is_u && !keep_d;
                float sn_cur = b[(N - 2) * WAVE];
                o = rowbase + (N - 2) * 16;
                auto bwd_step = [&](int i, float ai, float ci) {
                    const float tb = sn_cur - ai;
                    const float cq = cost_term(ci, is_x); // x rows: -(Xref.*Q) ; u rows: -0
                    float pn, dd;
                    riccati_step<NX, NU, EXACT, H16>(G, is_x, p, lin_cost<EXACT, H16>(cq, rho, tb), pn, dd); // admm.cpp:19-20,80-82
                    stw<H16>(P.pd, o, is_u ? dd : pn); // [p_i ; d_i] of this sweep
                    p = pn;
                    sn_cur = b[(i > 0 ? i - 1 : 0) * WAVE];
                    o -= 16;
                    return upd_d ? dd : ci;
                };
#pragma unroll 1
                for (int i = NHI - 2; i >= 0; i--) chi[i] = bwd_step(32 + i, ahi[i], chi[i]); // steps N-2 .. 32
#pragma unroll 1
                for (int i = (NHI >= 2 ? 31 : N - 2); i >= 0; i--) clo[i] = bwd_step(i, alo[i], clo[i]); // steps min(N-2, 31) .. 0
            }
        }
    }

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && r16 == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }

    // ---------------- live-out ----------------
    if (valid)
    {
        const bool solved = (st == TINY_STATUS_SOLVED_);
        float s = x0;
        int o = rowbase;
        auto live_out = [&](int i, float ai, float ci) {
            float sv, xn = 0.f;
            if (i < N - 1) lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, ci, sv, xn);
            else sv = is_x ? s : 0.f;
            stw<H16>(P.xu, o, sv);
            s = xn;
            const float sni = b[i * WAVE];
            const float lin = lin_cost<EXACT, H16>(cost_term(ci, is_x), rho, sni - ai);
            stw<H16>(P.qr, o, (i < N - 1 || is_x) ? lin : 0.f);
            if (i == N - 1) stw<H16>(P.pd, o, is_x ? pN : 0.f);
            else if (cold && !ran_bwd) stw<H16>(P.pd, o, 0.f);
            if (!solved) stw<H16>(P.vz, o, sni); // v = vnew happened; a solved instance keeps the stash
            stw<H16>(P.vzn, o, sni);
            stw<H16>(P.gy, o, ai);
            o += 16;
        };
#pragma unroll 1
        for (int i = 0; i < NLO; i++) live_out(i, alo[i], clo[i]);
#pragma unroll 1
        for (int i = 0; i < NHI; i++) live_out(32 + i, ahi[i], chi[i]);
        if (r16 == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

bool rowloop_supported(int nx, int nu, int N)
{
    return rowdims_supported(nx, nu) && N <= ROWLOOP64_MAX_N;
}

hipError_t launch_admm_rowloop(int nx, int nu, bool exact, bool h16, const RowParams &P, hipStream_t stream)
{
    const int nblocks = (P.batch + 3) / 4;
    const size_t lds = (size_t)P.N * (16 * sizeof(float2) + WAVE * sizeof(float));
    const bool big = P.N > ROWLOOP_MAX_N; // two state vectors per array
    if (big)
    {
        // beyond the default dynamic-LDS limit at N > 42
#define TINY_ROWLOOP64_ATTR(NX, NU)                                                                                                 \
    if (nx == NX && nu == NU)                                                                                                       \
    {                                                                                                                               \
        (void)hipFuncSetAttribute((const void *)admm_rowloop64_kernel<NX, NU, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        (void)hipFuncSetAttribute((const void *)admm_rowloop64_kernel<NX, NU, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        (void)hipFuncSetAttribute((const void *)admm_rowloop64_kernel<NX, NU, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        (void)hipFuncSetAttribute((const void *)admm_rowloop64_kernel<NX, NU, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
    }
        TINY_FOR_EACH_ROWDIMS(TINY_ROWLOOP64_ATTR)
    }
#define TINY_ROWLOOP_LAUNCH(NX, NU, EX, H)                                                                                   \
    do                                                                                                                       \
    {                                                                                                                        \
        if (big) hipLaunchKernelGGL((admm_rowloop64_kernel<NX, NU, EX, H>), dim3(nblocks), dim3(WAVE), lds, stream, P);      \
        else hipLaunchKernelGGL((admm_rowloop_kernel<NX, NU, EX, H>), dim3(nblocks), dim3(WAVE), lds, stream, P);            \
    } while (0)
#define TINY_ROWLOOP_DISPATCH(NX, NU)                             \
    if (nx == NX && nu == NU)                                     \
    {                                                             \
        if (exact && !h16) TINY_ROWLOOP_LAUNCH(NX, NU, true, false);   \
        else if (exact) TINY_ROWLOOP_LAUNCH(NX, NU, true, true);       \
        else if (!h16) TINY_ROWLOOP_LAUNCH(NX, NU, false, false);      \
        else TINY_ROWLOOP_LAUNCH(NX, NU, false, true);                 \
        return hipGetLastError();                                 \
    }
    TINY_FOR_EACH_ROWDIMS(TINY_ROWLOOP_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
