// tinympc_batch64.hip — the batched TinyMPC ADMM solver for `typedef double tinytype` (the reference as shipped,
// src/tinympc/glob_opts.hpp:3): C-ABI of include/tinympc_batch64.h, device workspace, and the kernel.
//
// One thread per instance, state in HBM.  Every per-instance array is stored instance-minor — element (step, row) of
// instance b at ((step * dim + row) * Bpad + b) — so that the 64 instances of a wavefront read and write 512 contiguous
// bytes per element.  One launch runs all ADMM iterations; a converged instance stops storing exactly where the reference
// returns (admm.cpp:135-137: before the v/z copy and the backward pass).
//
// Arithmetic: tiny_solve() of the reference (admm.cpp:15-152) statement by statement, every product and sum a separately
// rounded fp64 operation (the library is built with -ffp-contract=off), summed in the order of the reference's SSE2 Eigen
// build, whose packets hold PS = 2 doubles: sequential for packet-evaluated lazy products (result rows a multiple of 2),
// halving tree for coefficient-evaluated ones, packet tree + s0 + s1 for vectorised reductions.  Results are bitwise equal
// to the compiled reference (tests/test_parity_gpu.py: golden vectors of the as-shipped fp64 N = 10 hovering run).
// There is no CPU fallback.
#include "../../include/tinympc_batch64.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

namespace
{

constexpr int PS = 2; // doubles per SSE2 packet
constexpr int WAVE64 = 64;
constexpr int ST_SOLVED = 1, ST_UNSOLVED = 11; // admm.cpp:136, :114

// ---- reductions in the reference's orders (Eigen 3.4.90: redux_novec_unroller, redux_vec_unroller, etor_product_packet_impl) ----
template <int LO, int CNT, int NN>
__device__ __forceinline__ double tree_sum(const double (&t)[NN])
{
    if constexpr (CNT == 1) return t[LO];
    else
    {
        constexpr int H = CNT / 2;
        return tree_sum<LO, H>(t) + tree_sum<LO + H, CNT - H>(t);
    }
}
template <int PLO, int PCNT, int L, int NN>
__device__ __forceinline__ double ptree_sum(const double (&t)[NN]) // lane L of the packets [PLO, PLO + PCNT)
{
    if constexpr (PCNT == 1) return t[PS * PLO + L];
    else
    {
        constexpr int H = PCNT / 2;
        return ptree_sum<PLO, H, L>(t) + ptree_sum<PLO + H, PCNT - H, L>(t);
    }
}
template <int NN>
__device__ __forceinline__ double seq_sum(const double (&t)[NN])
{
    double acc = t[0];
#pragma unroll
    for (int k = 1; k < NN; k++) acc = acc + t[k];
    return acc;
}
template <int NN>
__device__ __forceinline__ double novec_sum(const double (&t)[NN])
{
    if constexpr (NN == 1) return t[0];
    else if constexpr (3 * NN - 1 <= 110) return tree_sum<0, NN>(t);
    else return seq_sum(t);
}
template <int NN>
__device__ __forceinline__ double vec_sum(const double (&t)[NN])
{
    static_assert(3 * NN - 1 <= 110 * PS, "beyond Eigen's unrolling limit the order is not restated");
    if constexpr (NN < PS) return novec_sum(t);
    else
    {
        constexpr int NPK = NN / PS;
        double res = ptree_sum<0, NPK, 0>(t) + ptree_sum<0, NPK, 1>(t); // predux of a 2-double packet
        if constexpr (NN % PS != 0) res = res + tree_sum<PS * NPK, NN - PS * NPK>(t);
        return res;
    }
}
// (row i of a column-major ROWS x COLS matrix) . xin, as a lazy product whose result has ROWS rows
template <int ROWS, int COLS>
__device__ __forceinline__ double row_dot(const double *M, int i, const double (&xin)[COLS])
{
    static_assert(ROWS <= PS || ROWS % PS == 0, "results with rows > 2 and odd: the reference's order depends on alignment");
    double t[COLS];
#pragma unroll
    for (int k = 0; k < COLS; k++) t[k] = M[k * ROWS + i] * xin[k];
    if constexpr (ROWS > 1 && ROWS % PS == 0) return seq_sum(t);
    else if constexpr (ROWS == 1) return vec_sum(t);
    else return novec_sum(t);
}

struct Params64
{
    int nx, nu, N, batch, bpad;
    double rho, abs_pri_tol, abs_dua_tol;
    int max_iter, check_termination, en_state_bound, en_input_bound;
    double *arr[TINY_ARR_COUNT];
    const double *xref, *xmin, *xmax, *umin, *umax; // [steps][dim][stride]; stride = bpad (per instance) or 1 (shared)
    int xref_stride, xb_stride, ub_stride;
    const double *mats; // Kinf | Pinf | Quu_inv | AmBKt | Adyn | Bdyn | Q, column-major
    double *res;        // [4][bpad]
    int *status, *iter, *n_unsolved;
};

template <int NX, int NU>
__global__ __launch_bounds__(WAVE64) void admm_f64_kernel(const Params64 P)
{
    static_assert(!(NX >= 8 && NU >= 8), "both dims >= 8: Eigen switches to its GEMV kernel, not restated here");
    constexpr int NMAT = NU * NX + NX * NX + NU * NU + NX * NX + NX * NX + NX * NU + NX;
    __shared__ double mats[NMAT];
    for (int e = threadIdx.x; e < NMAT; e += WAVE64) mats[e] = P.mats[e];
    __syncthreads();
    const double *K = mats, *Pinf = K + NU * NX, *Quu = Pinf + NX * NX, *Am = Quu + NU * NU, *A = Am + NX * NX, *Bm = A + NX * NX,
                 *Q = Bm + NX * NU;

    const int b = blockIdx.x * WAVE64 + threadIdx.x;
    const bool valid = b < P.batch;
    const int N = P.N;
    const size_t bp = (size_t)P.bpad;
    const double rho = P.rho;
    auto at = [&](int id, int step, int row, int dim) -> double & { return P.arr[id][((size_t)step * dim + row) * bp + b]; };
    auto in = [&](const double *base, int stride, int step, int row, int dim) {
        return base[((size_t)step * dim + row) * (size_t)stride + (stride > 1 ? b : 0)];
    };

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid)
        {
            P.status[b] = ST_UNSOLVED; P.iter[b] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    double x0[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) x0[j] = valid ? at(TINY_ARR_X, 0, j, NX) : 0.0;
    double r_ps = 0, r_pi = 0, r_ds = 0, r_di = 0;
    if (valid)
    {
        r_ps = P.res[0 * bp + b]; r_pi = P.res[1 * bp + b]; r_ds = P.res[2 * bp + b]; r_di = P.res[3 * bp + b];
    }
    int st = ST_UNSOLVED, itn = 1; // admm.cpp:114-115
    bool active = valid;
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        if (active)
        {
            itn = it + 1; // admm.cpp:120
            // ---- forward_pass (:27-37) + update_slack (:45-61) + update_dual (:67-71) + update_linear_cost (:77-85) + residuals (:95-98) ----
            double x[NX], pN[NX];
#pragma unroll
            for (int j = 0; j < NX; j++) x[j] = x0[j];
            double pri_x = 0, dua_x = 0, pri_u = 0, dua_u = 0;
            for (int i = 0; i < N; i++)
            {
                double u[NU], xn[NX];
                if (i < N - 1)
                {
#pragma unroll
                    for (int j = 0; j < NU; j++) u[j] = -row_dot<NU, NX>(K, j, x) - at(TINY_ARR_D, i, j, NU);           // :31
#pragma unroll
                    for (int j = 0; j < NX; j++) xn[j] = row_dot<NX, NX>(A, j, x) + row_dot<NX, NU>(Bm, j, u);           // :35
#pragma unroll
                    for (int j = 0; j < NU; j++)
                    {
                        const double y = at(TINY_ARR_Y, i, j, NU);
                        double zn = u[j] + y;                                                                             // :47
                        if (P.en_input_bound)                                                                             // :51-54
                        {
                            const double lo = in(P.umin, P.ub_stride, i, j, NU), hi = in(P.umax, P.ub_stride, i, j, NU);
                            zn = (lo < zn) ? zn : lo;
                            zn = (zn < hi) ? zn : hi;
                        }
                        const double yn = y + u[j] - zn;                                                                  // :69
                        const double pu = fabs(u[j] - zn), du = fabs(at(TINY_ARR_Z, i, j, NU) - zn);                      // :97-98
                        pri_u = (i == 0 && j == 0) ? pu : (pu > pri_u ? pu : pri_u);
                        dua_u = (i == 0 && j == 0) ? du : (du > dua_u ? du : dua_u);
                        at(TINY_ARR_U, i, j, NU) = u[j];
                        at(TINY_ARR_ZNEW, i, j, NU) = zn;
                        at(TINY_ARR_Y, i, j, NU) = yn;
                        at(TINY_ARR_R, i, j, NU) = -rho * (zn - yn);                                                      // :80
                    }
                }
#pragma unroll
                for (int j = 0; j < NX; j++)
                {
                    const double g = at(TINY_ARR_G, i, j, NX);
                    double vn = x[j] + g;                                                                                 // :48
                    if (P.en_state_bound)                                                                                 // :57-60
                    {
                        const double lo = in(P.xmin, P.xb_stride, i, j, NX), hi = in(P.xmax, P.xb_stride, i, j, NX);
                        vn = (lo < vn) ? vn : lo;
                        vn = (vn < hi) ? vn : hi;
                    }
                    const double gn = g + x[j] - vn;                                                                      // :70
                    const double px = fabs(x[j] - vn), dx = fabs(at(TINY_ARR_V, i, j, NX) - vn);                          // :95-96
                    pri_x = (i == 0 && j == 0) ? px : (px > pri_x ? px : pri_x);
                    dua_x = (i == 0 && j == 0) ? dx : (dx > dua_x ? dx : dua_x);
                    at(TINY_ARR_X, i, j, NX) = x[j];
                    at(TINY_ARR_VNEW, i, j, NX) = vn;
                    at(TINY_ARR_G, i, j, NX) = gn;
                    double q = -(in(P.xref, P.xref_stride, i, j, NX) * Q[j]);                                             // :81
                    q = q - rho * (vn - gn);                                                                              // :82
                    at(TINY_ARR_Q, i, j, NX) = q;
                    if (i == N - 1) pN[j] = rho * (vn - gn); // second half of :84, applied below
                }
                if (i < N - 1)
                {
#pragma unroll
                    for (int j = 0; j < NX; j++) x[j] = xn[j];
                }
            }
            {
                // p.col(N-1) = -(Xref.col(N-1)^T * Pinf)  (:83), then -= rho * (vnew - g)  (:84)
                double xr[NX];
#pragma unroll
                for (int k = 0; k < NX; k++) xr[k] = in(P.xref, P.xref_stride, N - 1, k, NX);
#pragma unroll
                for (int j = 0; j < NX; j++)
                {
                    double t[NX];
#pragma unroll
                    for (int k = 0; k < NX; k++) t[k] = xr[k] * Pinf[j * NX + k];
                    const double pt = -vec_sum(t);
                    pN[j] = pt - pN[j];
                    at(TINY_ARR_P, N - 1, j, NX) = pN[j];
                }
            }
            // ---- termination_condition (:91-109) ----
            bool conv = false;
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv)
            {
                st = ST_SOLVED; // :136, returns before the v/z copy and the backward pass
                active = false;
            }
            else
            {
                // ---- v = vnew, z = znew (:141-142) + backward_pass_grad (:15-22) ----
#pragma unroll
                for (int j = 0; j < NX; j++) at(TINY_ARR_V, N - 1, j, NX) = at(TINY_ARR_VNEW, N - 1, j, NX);
                double pn[NX];
#pragma unroll
                for (int j = 0; j < NX; j++) pn[j] = pN[j];
                for (int i = N - 2; i >= 0; i--)
                {
                    double r[NU], tmp[NU], dn[NU], pi[NX];
#pragma unroll
                    for (int j = 0; j < NU; j++)
                    {
                        at(TINY_ARR_Z, i, j, NU) = at(TINY_ARR_ZNEW, i, j, NU);
                        r[j] = at(TINY_ARR_R, i, j, NU);
                    }
#pragma unroll
                    for (int j = 0; j < NU; j++) // Bdyn^T p_{i+1} + r_i: column j of Bdyn is contiguous -> vectorised reduction
                    {
                        double t[NX];
#pragma unroll
                        for (int k = 0; k < NX; k++) t[k] = Bm[j * NX + k] * pn[k];
                        tmp[j] = vec_sum(t) + r[j];
                    }
#pragma unroll
                    for (int j = 0; j < NU; j++) dn[j] = row_dot<NU, NU>(Quu, j, tmp);                                    // :19
#pragma unroll
                    for (int j = 0; j < NX; j++)
                    {
                        at(TINY_ARR_V, i, j, NX) = at(TINY_ARR_VNEW, i, j, NX);
                        double t[NX], tk[NU];
#pragma unroll
                        for (int k = 0; k < NX; k++) t[k] = Am[k * NX + j] * pn[k];
                        const double a = (NU == 1 && NX % PS == 0) ? seq_sum(t) : novec_sum(t);
#pragma unroll
                        for (int m = 0; m < NU; m++) tk[m] = K[j * NU + m] * r[m]; // Kinf^T r: column j of Kinf is contiguous
                        pi[j] = at(TINY_ARR_Q, i, j, NX) + a - vec_sum(tk);                                               // :20
                    }
#pragma unroll
                    for (int j = 0; j < NU; j++) at(TINY_ARR_D, i, j, NU) = dn[j];
#pragma unroll
                    for (int j = 0; j < NX; j++)
                    {
                        at(TINY_ARR_P, i, j, NX) = pi[j];
                        pn[j] = pi[j];
                    }
                }
            }
        }
    }
    if (valid)
    {
        P.res[0 * bp + b] = r_ps; P.res[1 * bp + b] = r_pi; P.res[2 * bp + b] = r_ds; P.res[3 * bp + b] = r_di;
        P.status[b] = st;
        P.iter[b] = itn;
        if (st != ST_SOLVED) atomicAdd(P.n_unsolved, 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// admm_f64_rows_kernel<NX,NU,N>: the 16-lanes-per-instance mapping of admm_rowlane.hip in double.  Lane r of a DPP row owns
// row r of the stacked vector [x ; u] (nx + nu <= 16), four instances per wavefront, the horizon unrolled, and the whole
// loop-carried state of the solve in registers: per step a = [g;y], c = [-(Xref.*Q) ; d], pd = [p;d] of the last backward
// sweep, b = [v;z], sn = [vnew;znew] (five doubles per lane and step; one wave per SIMD owns the 512-entry register file).
// The workspace arrays are read once before the first and written once after the last iteration — the thread-per-instance
// kernel above moves every array through HBM in every iteration (3.7 GB per iteration of 65 536 quadrotor instances).
// A state element is broadcast within its row by two v_mov_b32_dpp row_newbcast (low and high word), products and sums
// are separately rounded v_mul_f64 / v_add_f64 in the same orders as above.  Results are bitwise equal to the
// thread-per-instance kernel and to the compiled fp64 reference.
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// t[k] = M[k] * s[K0 + k], s[j] = the value lane j of this lane's row holds (row_newbcast:j = 0x150 + j)
template <int K0, int CNT, int K = 0>
__device__ __forceinline__ void row_products(double (&t)[CNT], double s, const double (&M)[CNT])
{
    if constexpr (K < CNT)
    {
        t[K] = M[K] * dpp64<0x150 + K0 + K>(s);
        row_products<K0, CNT, K + 1>(t, s, M);
    }
}
__device__ __forceinline__ double row_max(double v) // max over the 16 lanes of a row, every lane gets it
{
    v = fmax(v, dpp64<0x128>(v)); v = fmax(v, dpp64<0x124>(v)); v = fmax(v, dpp64<0x122>(v)); v = fmax(v, dpp64<0x121>(v));
    return v;
}
// the reference's order for a lazy product whose result has ROWS rows (see row_dot above)
template <int ROWS, int CNT>
__device__ __forceinline__ double lazy_sum(const double (&t)[CNT])
{
    static_assert(ROWS <= PS || ROWS % PS == 0, "results with rows > 2 and odd: the reference's order depends on alignment");
    if constexpr (ROWS > 1 && ROWS % PS == 0) return seq_sum(t);
    else if constexpr (ROWS == 1) return vec_sum(t);
    else return novec_sum(t);
}

// problem classes of the fp64 library (the thread-per-instance kernels; any N).  nx and nu must each be <= 2 or even (the
// reference's order for other sizes depends on alignment, see row_dot) and not both >= 8 (Eigen's GEMV kernel is not restated here)
#define TINY_FOR_EACH_F64DIMS(X) X(12, 4) X(4, 1) X(8, 4) X(12, 2) X(4, 2) X(4, 4) X(16, 4)
#define TINY_FOR_EACH_F64ROWS(X) X(12, 4, 10) X(12, 4, 30) X(12, 4, 20) X(4, 1, 10) X(8, 4, 9)
constexpr int F64_AHEAD = 4; // bounds are fetched this many steps ahead of their use

// RT = false: the horizon is the template parameter N.  RT = true ("any horizon", round 3): N is the CAPACITY of the unrolled body
// (32 or 64 steps: registers are indexed statically, so the body stays unrolled) and the horizon n = P.N <= N is a launch parameter;
// the steps past it are skipped by wave-uniform branches.  With a capacity of 64 the backward sweep's [p ; d] (live-out only) is
// written through to its arrays instead of being held in 2 x 64 registers.
template <int NX, int NU, int N, bool RT = false>
__global__ __launch_bounds__(WAVE64, (N <= 12 ? 2 : 1)) void admm_f64_rows_kernel(const Params64 P, const double *__restrict__ gains)
{
    static_assert(NX + NU <= 16 && !(NX >= 8 && NU >= 8), "16-lane mapping; both dims >= 8 would take Eigen's GEMV kernel");
    static_assert(!RT || N > 20, "the runtime-horizon variant indexes the slack by the horizon: it keeps it in LDS");
    const int n = RT ? P.N : N; // the horizon
    constexpr bool PD_REG = !(RT && N > 32);
    const int lane = threadIdx.x, r16 = lane & 15;
    const int inst = blockIdx.x * 4 + (lane >> 4);
    const bool valid = inst < P.batch;
    const size_t b = valid ? inst : P.batch - 1; // loads of an out-of-range row address the last instance; it stores nothing
    const bool is_x = r16 < NX, is_u = r16 >= NX && r16 < NX + NU;
    const int row = is_x ? r16 : (is_u ? r16 - NX : 0);
    const size_t bp = (size_t)P.bpad;
    const double rho = P.rho;
    // element (step, this lane's row) of the pair [x-type array ; u-type array]; u-type arrays have N - 1 steps
    auto ld = [&](int idx, int idu, int i) -> double {
        if (is_x) return P.arr[idx][((size_t)i * NX + row) * bp + b];
        if (is_u && i < n - 1) return P.arr[idu][((size_t)i * NU + row) * bp + b];
        return 0.0;
    };
    auto st = [&](int idx, int idu, int i, double v) {
        if (!valid) return;
        if (is_x) P.arr[idx][((size_t)i * NX + row) * bp + b] = v;
        else if (is_u && i < n - 1) P.arr[idu][((size_t)i * NU + row) * bp + b] = v;
    };
    const bool en_b = is_x ? (P.en_state_bound != 0) : (is_u && P.en_input_bound != 0);
    // this lane's bounds of step i sit at pl0[i * bstep], ph0[i * bstep] (u rows have N - 1 steps; the other lanes read x row 0, unused)
    const size_t bstr = is_u ? (size_t)P.ub_stride : (size_t)P.xb_stride;
    const size_t boff = (size_t)row * bstr + (bstr > 1 ? b : 0);
    const double *const pl0 = (is_u ? P.umin : P.xmin) + boff, *const ph0 = (is_u ? P.umax : P.xmax) + boff;
    const size_t bstep = (is_u ? NU : NX) * bstr;
    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && r16 == 0)
        {
            P.status[inst] = ST_UNSOLVED; P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    // gains, one register per matrix column: [reg][16] doubles (pack_row_gains on the host)
    double M1[NX], M2[NU], M3[NX], M45[NU];
#pragma unroll
    for (int k = 0; k < NX; k++) { M1[k] = gains[k * 16 + r16]; M3[k] = gains[(NX + NU + k) * 16 + r16]; }
#pragma unroll
    for (int m = 0; m < NU; m++) { M2[m] = gains[(NX + m) * 16 + r16]; M45[m] = gains[(2 * NX + NU + m) * 16 + r16]; }
    const double qrow = gains[(2 * NX + 2 * NU) * 16 + r16];

    // ---- live-in ----
    // long horizons keep the two slack words of a step in LDS (lane-linear, 1 KB per step and wave), short ones in registers
    constexpr bool LDS_SLACK = N > 20;
    constexpr int NREG = LDS_SLACK ? 1 : N;
    __shared__ double slack_lds[LDS_SLACK ? 2 * N * WAVE64 : 1];
    // (two halves: hipcc leaves a statically indexed array of more than 32 doubles in scratch memory)
    constexpr int NLO = N < 32 ? N : 32, NHI = N > 32 ? N - 32 : 1;
    double a_lo[NLO], a_hi[NHI], c_lo[NLO], c_hi[NHI], pd[PD_REG ? N : 1], bb_r[NREG], sn_r[NREG];
#define A_(i) ((i) < 32 ? a_lo[(i) < 32 ? (i) : 0] : a_hi[(i) < 32 ? 0 : (i) - 32])
#define C_(i) ((i) < 32 ? c_lo[(i) < 32 ? (i) : 0] : c_hi[(i) < 32 ? 0 : (i) - 32])
    double *const sl = slack_lds + lane;
#define BB_GET(i) (LDS_SLACK ? sl[(2 * (i)) * WAVE64] : bb_r[LDS_SLACK ? 0 : (i)])
#define SN_GET(i) (LDS_SLACK ? sl[(2 * (i) + 1) * WAVE64] : sn_r[LDS_SLACK ? 0 : (i)])
#define BB_SET(i, v) do { if constexpr (LDS_SLACK) sl[(2 * (i)) * WAVE64] = (v); else bb_r[LDS_SLACK ? 0 : (i)] = (v); } while (0)
#define SN_SET(i, v) do { if constexpr (LDS_SLACK) sl[(2 * (i) + 1) * WAVE64] = (v); else sn_r[LDS_SLACK ? 0 : (i)] = (v); } while (0)
    double xrN = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        if (RT && i >= n) continue;
        double xr = 0.0;
        if (is_x) xr = P.xref[((size_t)i * NX + row) * (size_t)P.xref_stride + (P.xref_stride > 1 ? b : 0)];
        const double pdi = ld(TINY_ARR_P, TINY_ARR_D, i);
        if constexpr (PD_REG) pd[i] = pdi;
        C_(i) = is_x ? -(xr * qrow) : pdi;            // admm.cpp:81 | d_i
        A_(i) = ld(TINY_ARR_G, TINY_ARR_Y, i);
        BB_SET(i, ld(TINY_ARR_V, TINY_ARR_Z, i));
        SN_SET(i, 0.0);
        if (i == n - 1) xrN = xr;
    }
    const double x0 = ld(TINY_ARR_X, TINY_ARR_U, 0); // x.col(0) on the x rows
    double pterm;
    {
        double PT[NX], t[NX]; // p.col(N-1) = -(Xref.col(N-1)^T * Pinf)  (admm.cpp:83): a vectorised reduction over k
#pragma unroll
        for (int k = 0; k < NX; k++) PT[k] = gains[(2 * NX + 2 * NU + 1 + k) * 16 + r16];
        row_products<0, NX>(t, xrN, PT);
        pterm = -vec_sum(t);
    }
    double r_ps = 0, r_pi = 0, r_ds = 0, r_di = 0;
    {
        const size_t bi = b;
        r_ps = P.res[0 * bp + bi]; r_pi = P.res[1 * bp + bi]; r_ds = P.res[2 * bp + bi]; r_di = P.res[3 * bp + bi];
    }
    int status = ST_UNSOLVED, itn = 1; // admm.cpp:114-115
    bool active = valid;
    double pN = 0.0;

    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        // No divergent region around the body: a row that has converged keeps executing with its wave and every update of its
        // state is a select on `active` (a compiler-visible `if (active)` would make hipcc keep the old and the new value of
        // all 3N state registers alive across the region: 264 spilled registers at N = 30).
        itn = active ? it + 1 : itn; // admm.cpp:120
        // the last permitted backward sweep must not overwrite d in c: x, u of an instance that exhausts max_iter come from
        // the d its last forward sweep used (regenerated below); the final d itself goes to pd
        const bool keep_d = (it == P.max_iter - 1);
        // ---- forward_pass + update_slack + update_dual + residual maxima (admm.cpp:27-71, 95-98) ----
        double s = x0, pri = 0.0, dua = 0.0, t1 = 0.0;
        double lo[F64_AHEAD], hi[F64_AHEAD];
        // an offset the compiler cannot see through, renewed every iteration: without it LICM hoists the 2N 64-bit bound
        // addresses out of the iteration loop and keeps them in 4N registers
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        const double *const pl = pl0 + oz, *const ph = ph0 + oz;
        auto ld_bounds = [&](int i, double &l, double &h) {
            const size_t o = (size_t)((is_u && i >= n - 1) ? 0 : i) * bstep;
            l = pl[o]; h = ph[o];
        };
#pragma unroll
        for (int k = 0; k < F64_AHEAD; k++) ld_bounds(k < n ? k : n - 1, lo[k], hi[k]);
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            if (RT && i >= n) continue;
            double sv, xn = 0.0;
            if (i < n - 1)
            {
                double t[NX], t2[NU];
                row_products<0, NX>(t, s, M1);
                const double acc = is_x ? lazy_sum<NX>(t) : lazy_sum<NU>(t);   // Adyn*x | Kinf*x
                const double un = -acc - C_(i);                                   // admm.cpp:31
                row_products<NX, NU>(t2, un, M2);
                xn = acc + lazy_sum<NX>(t2);                                     // admm.cpp:35
                sv = is_u ? un : s;
            }
            else sv = is_x ? s : 0.0;
            const double lo_i = lo[i % F64_AHEAD], hi_i = hi[i % F64_AHEAD];
            if (i + F64_AHEAD < n) ld_bounds(i + F64_AHEAD, lo[i % F64_AHEAD], hi[i % F64_AHEAD]);
            const double t0 = sv + A_(i);                                         // admm.cpp:47-48
            double tc = t0;
            if (en_b && (i < n - 1 || is_x))                                     // admm.cpp:51-60: min(max, max(min, t)); u has N - 1 steps
            {
                tc = (lo_i < tc) ? tc : lo_i;
                tc = (tc < hi_i) ? tc : hi_i;
            }
            const double an = (A_(i) + sv) - tc;                                  // admm.cpp:69-70
            pri = fmax(pri, fabs(sv - tc));                                      // admm.cpp:95,97
            dua = fmax(dua, fabs(BB_GET(i) - tc));                               // admm.cpp:96,98
            A_(i) = active ? an : A_(i);
            if constexpr (LDS_SLACK) { if (active) SN_SET(i, tc); }
            else sn_r[LDS_SLACK ? 0 : i] = active ? tc : sn_r[LDS_SLACK ? 0 : i];
            t1 = tc - an;
            s = xn;
        }
        pN = active ? pterm - rho * t1 : pN; // admm.cpp:83-84
        const double pri_x = row_max(is_x ? pri : 0.0), dua_x = row_max(is_x ? dua : 0.0);
        const double pri_u = row_max(is_u ? pri : 0.0), dua_u = row_max(is_u ? dua : 0.0);
        bool conv = false;
        if ((it + 1) % P.check_termination == 0) // admm.cpp:91-109 (wave-uniform condition)
        {
            r_ps = active ? pri_x : r_ps; r_ds = active ? dua_x * rho : r_ds; r_pi = active ? pri_u : r_pi; r_di = active ? dua_u * rho : r_di;
            conv = active && (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
        }
        status = conv ? ST_SOLVED : status; // admm.cpp:136: returns before the v/z copy and the backward pass
        active = active && !conv;
        if (!__any(active)) break;
        // ---- v = vnew, z = znew (admm.cpp:141-142), update_linear_cost + backward_pass_grad (admm.cpp:15-22, 80-82) ----
        double p = pN;
        if constexpr (LDS_SLACK) { if (active) BB_SET(n - 1, SN_GET(n - 1)); }
        else bb_r[LDS_SLACK ? 0 : N - 1] = active ? sn_r[LDS_SLACK ? 0 : N - 1] : bb_r[LDS_SLACK ? 0 : N - 1];
#pragma unroll
        for (int i = N - 2; i >= 0; i--)
        {
            if (RT && i > n - 2) continue;
            const double sni = SN_GET(i);
            if constexpr (LDS_SLACK) { if (active) BB_SET(i, sni); }
            else bb_r[LDS_SLACK ? 0 : i] = active ? sni : bb_r[LDS_SLACK ? 0 : i];
            const double cq = is_x ? C_(i) : -0.0; // u rows: r = -rho*(znew - y) keeps the sign of a zero difference
            const double lin = cq - rho * (sni - A_(i));
            double t[NX], tk[NU], td[NU];
            row_products<0, NX>(t, p, M3);
            // x rows: AmBKt*p (coefficient-evaluated: halving tree, or sequential when nu = 1); u rows: Bdyn^T*p (vectorised reduction)
            const double dot = is_x ? ((NU == 1 && NX % PS == 0) ? seq_sum(t) : novec_sum(t)) : vec_sum(t);
            const double wv = lin + dot;               // q + AmBKt*p | Bdyn^T*p + r
            row_products<NX, NU>(tk, lin, M45);        // Kinf^T * r
            row_products<NX, NU>(td, wv, M45);         // Quu_inv * (Bdyn^T p + r)
            const double pn = wv - vec_sum(tk);        // admm.cpp:20
            const double dd = lazy_sum<NU>(td);        // admm.cpp:19
            if constexpr (PD_REG) pd[i] = active ? (is_u ? dd : pn) : pd[i];
            else { if (active) st(TINY_ARR_P, TINY_ARR_D, i, is_u ? dd : pn); }
            C_(i) = (active && is_u && !keep_d) ? dd : C_(i);
            p = pn;
        }
    }
    // ---- live-out: every work array once ----
    {
        const bool solved = status == ST_SOLVED;
        double s = x0;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            if (RT && i >= n) continue;
            double sv, xn = 0.0;
            if (i < n - 1) // x, u regenerated from the d of the last executed forward sweep by the same instruction sequence
            {
                double t[NX], t2[NU];
                row_products<0, NX>(t, s, M1);
                const double acc = is_x ? lazy_sum<NX>(t) : lazy_sum<NU>(t);
                const double un = -acc - C_(i);
                row_products<NX, NU>(t2, un, M2);
                xn = acc + lazy_sum<NX>(t2);
                sv = is_u ? un : s;
            }
            else sv = is_x ? s : 0.0;
            s = xn;
            st(TINY_ARR_X, TINY_ARR_U, i, sv);
            const double sni = SN_GET(i);
            st(TINY_ARR_Q, TINY_ARR_R, i, (is_x ? C_(i) : -0.0) - rho * (sni - A_(i)));
            if constexpr (PD_REG) st(TINY_ARR_P, TINY_ARR_D, i, i == n - 1 ? pN : pd[i]);
            else { if (i == n - 1) st(TINY_ARR_P, TINY_ARR_D, i, pN); }
            st(TINY_ARR_V, TINY_ARR_Z, i, BB_GET(i));
            st(TINY_ARR_VNEW, TINY_ARR_ZNEW, i, sni);
            st(TINY_ARR_G, TINY_ARR_Y, i, A_(i));
        }
        if (valid && r16 == 0)
        {
            P.res[0 * bp + inst] = r_ps; P.res[1 * bp + inst] = r_pi; P.res[2 * bp + inst] = r_ds; P.res[3 * bp + inst] = r_di;
            P.status[inst] = status;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
#undef A_
#undef C_
#undef BB_GET
#undef SN_GET
#undef BB_SET
#undef SN_SET
}

// ---------------------------------------------------------------------------------------------------------------------
// The six functions tiny_solve() is made of (admm.hpp:12-18), one launch each, one thread per instance: each reads and writes
// exactly the members the reference function does, in the same arithmetic as the fused kernels above.
// ---------------------------------------------------------------------------------------------------------------------
enum { F64_FORWARD_PASS = 0, F64_UPDATE_SLACK, F64_UPDATE_DUAL, F64_UPDATE_LINEAR_COST, F64_TERMINATION_CONDITION, F64_BACKWARD_PASS_GRAD };

template <int NX, int NU, int FN>
__global__ __launch_bounds__(WAVE64) void admm_f64_step_kernel(const Params64 P, int *__restrict__ conv_out)
{
    constexpr int NMAT = NU * NX + NX * NX + NU * NU + NX * NX + NX * NX + NX * NU + NX;
    __shared__ double mats[NMAT];
    for (int e = threadIdx.x; e < NMAT; e += WAVE64) mats[e] = P.mats[e];
    __syncthreads();
    const double *K = mats, *Pinf = K + NU * NX, *Quu = Pinf + NX * NX, *Am = Quu + NU * NU, *A = Am + NX * NX, *Bm = A + NX * NX,
                 *Q = Bm + NX * NU;
    const int b = blockIdx.x * WAVE64 + threadIdx.x;
    if (b >= P.batch) return;
    const int N = P.N;
    const size_t bp = (size_t)P.bpad;
    const double rho = P.rho;
    auto at = [&](int id, int step, int row, int dim) -> double & { return P.arr[id][((size_t)step * dim + row) * bp + b]; };
    auto in = [&](const double *base, int stride, int step, int row, int dim) {
        return base[((size_t)step * dim + row) * (size_t)stride + (stride > 1 ? b : 0)];
    };
    if constexpr (FN == F64_FORWARD_PASS) // admm.cpp:27-37
    {
        double x[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) x[j] = at(TINY_ARR_X, 0, j, NX);
        for (int i = 0; i < N - 1; i++)
        {
            double u[NU], xn[NX];
#pragma unroll
            for (int j = 0; j < NU; j++) u[j] = -row_dot<NU, NX>(K, j, x) - at(TINY_ARR_D, i, j, NU);
#pragma unroll
            for (int j = 0; j < NX; j++) xn[j] = row_dot<NX, NX>(A, j, x) + row_dot<NX, NU>(Bm, j, u);
#pragma unroll
            for (int j = 0; j < NU; j++) at(TINY_ARR_U, i, j, NU) = u[j];
#pragma unroll
            for (int j = 0; j < NX; j++) { at(TINY_ARR_X, i + 1, j, NX) = xn[j]; x[j] = xn[j]; }
        }
    }
    else if constexpr (FN == F64_UPDATE_SLACK) // admm.cpp:45-61
    {
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = 0; j < NU; j++)
            {
                double zn = at(TINY_ARR_U, i, j, NU) + at(TINY_ARR_Y, i, j, NU);
                if (P.en_input_bound)
                {
                    const double lo = in(P.umin, P.ub_stride, i, j, NU), hi = in(P.umax, P.ub_stride, i, j, NU);
                    zn = (lo < zn) ? zn : lo;
                    zn = (zn < hi) ? zn : hi;
                }
                at(TINY_ARR_ZNEW, i, j, NU) = zn;
            }
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < NX; j++)
            {
                double vn = at(TINY_ARR_X, i, j, NX) + at(TINY_ARR_G, i, j, NX);
                if (P.en_state_bound)
                {
                    const double lo = in(P.xmin, P.xb_stride, i, j, NX), hi = in(P.xmax, P.xb_stride, i, j, NX);
                    vn = (lo < vn) ? vn : lo;
                    vn = (vn < hi) ? vn : hi;
                }
                at(TINY_ARR_VNEW, i, j, NX) = vn;
            }
    }
    else if constexpr (FN == F64_UPDATE_DUAL) // admm.cpp:67-71
    {
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = 0; j < NU; j++) at(TINY_ARR_Y, i, j, NU) = at(TINY_ARR_Y, i, j, NU) + at(TINY_ARR_U, i, j, NU) - at(TINY_ARR_ZNEW, i, j, NU);
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < NX; j++) at(TINY_ARR_G, i, j, NX) = at(TINY_ARR_G, i, j, NX) + at(TINY_ARR_X, i, j, NX) - at(TINY_ARR_VNEW, i, j, NX);
    }
    else if constexpr (FN == F64_UPDATE_LINEAR_COST) // admm.cpp:77-85
    {
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = 0; j < NU; j++) at(TINY_ARR_R, i, j, NU) = -rho * (at(TINY_ARR_ZNEW, i, j, NU) - at(TINY_ARR_Y, i, j, NU));
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < NX; j++)
            {
                double q = -(in(P.xref, P.xref_stride, i, j, NX) * Q[j]);
                q = q - rho * (at(TINY_ARR_VNEW, i, j, NX) - at(TINY_ARR_G, i, j, NX));
                at(TINY_ARR_Q, i, j, NX) = q;
            }
        double xr[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) xr[k] = in(P.xref, P.xref_stride, N - 1, k, NX);
#pragma unroll
        for (int j = 0; j < NX; j++)
        {
            double t[NX];
#pragma unroll
            for (int k = 0; k < NX; k++) t[k] = xr[k] * Pinf[j * NX + k];
            const double pt = -vec_sum(t);
            at(TINY_ARR_P, N - 1, j, NX) = pt - rho * (at(TINY_ARR_VNEW, N - 1, j, NX) - at(TINY_ARR_G, N - 1, j, NX));
        }
    }
    else if constexpr (FN == F64_TERMINATION_CONDITION) // admm.cpp:91-109
    {
        bool conv = false;
        if (P.iter[b] % P.check_termination == 0)
        {
            double pri_x = 0, dua_x = 0, pri_u = 0, dua_u = 0;
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int j = 0; j < NX; j++)
                {
                    const double vn = at(TINY_ARR_VNEW, i, j, NX);
                    const double px = fabs(at(TINY_ARR_X, i, j, NX) - vn), dx = fabs(at(TINY_ARR_V, i, j, NX) - vn);
                    pri_x = (i == 0 && j == 0) ? px : (px > pri_x ? px : pri_x);
                    dua_x = (i == 0 && j == 0) ? dx : (dx > dua_x ? dx : dua_x);
                }
            for (int i = 0; i < N - 1; i++)
#pragma unroll
                for (int j = 0; j < NU; j++)
                {
                    const double zn = at(TINY_ARR_ZNEW, i, j, NU);
                    const double pu = fabs(at(TINY_ARR_U, i, j, NU) - zn), du = fabs(at(TINY_ARR_Z, i, j, NU) - zn);
                    pri_u = (i == 0 && j == 0) ? pu : (pu > pri_u ? pu : pri_u);
                    dua_u = (i == 0 && j == 0) ? du : (du > dua_u ? du : dua_u);
                }
            const double r_ps = pri_x, r_ds = dua_x * rho, r_pi = pri_u, r_di = dua_u * rho;
            P.res[0 * bp + b] = r_ps; P.res[1 * bp + b] = r_pi; P.res[2 * bp + b] = r_ds; P.res[3 * bp + b] = r_di;
            conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
        }
        conv_out[b] = conv ? 1 : 0;
    }
    else // F64_BACKWARD_PASS_GRAD, admm.cpp:15-22
    {
        double pn[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) pn[j] = at(TINY_ARR_P, N - 1, j, NX);
        for (int i = N - 2; i >= 0; i--)
        {
            double r[NU], tmp[NU], dn[NU], pi[NX];
#pragma unroll
            for (int j = 0; j < NU; j++) r[j] = at(TINY_ARR_R, i, j, NU);
#pragma unroll
            for (int j = 0; j < NU; j++)
            {
                double t[NX];
#pragma unroll
                for (int k = 0; k < NX; k++) t[k] = Bm[j * NX + k] * pn[k];
                tmp[j] = vec_sum(t) + r[j];
            }
#pragma unroll
            for (int j = 0; j < NU; j++) dn[j] = row_dot<NU, NU>(Quu, j, tmp);
#pragma unroll
            for (int j = 0; j < NX; j++)
            {
                double t[NX], tk[NU];
#pragma unroll
                for (int k = 0; k < NX; k++) t[k] = Am[k * NX + j] * pn[k];
                const double a = (NU == 1 && NX % PS == 0) ? seq_sum(t) : novec_sum(t);
#pragma unroll
                for (int m = 0; m < NU; m++) tk[m] = K[j * NU + m] * r[m];
                pi[j] = at(TINY_ARR_Q, i, j, NX) + a - vec_sum(tk);
            }
#pragma unroll
            for (int j = 0; j < NU; j++) at(TINY_ARR_D, i, j, NU) = dn[j];
#pragma unroll
            for (int j = 0; j < NX; j++) { at(TINY_ARR_P, i, j, NX) = pi[j]; pn[j] = pi[j]; }
        }
    }
}

// host layout [cnt][steps][dim] (cnt = 1: shared, stored once with stride 1)  <->  device [steps][dim][stride]
__global__ void pack64_kernel(const double *__restrict__ src, double *__restrict__ dst, int nb, int steps, int dim, int stride, int step0, int nsteps)
{
    const long long total = (long long)nb * nsteps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int b = (int)(e % nb);
        const long long t = e / nb;
        const int row = (int)(t % dim), s = (int)(t / dim);
        dst[((long long)(step0 + s) * dim + row) * stride + (stride > 1 ? b : 0)] = src[((long long)b * steps + step0 + s) * dim + row];
    }
}
__global__ void unpack64_kernel(const double *__restrict__ src, double *__restrict__ dst, int nb, int steps, int dim, int stride)
{
    const long long total = (long long)nb * steps * dim;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
    {
        const int b = (int)(e % nb);
        const long long t = e / nb;
        const int row = (int)(t % dim), s = (int)(t / dim);
        dst[((long long)b * steps + s) * dim + row] = src[((long long)s * dim + row) * stride + b];
    }
}

// Plant step of the examples' closed loop (quadrotor_hovering.cpp:110-111): x.col(0) <- Adyn * x.col(0) + Bdyn * u.col(0), in
// Eigen's order for that expression over tiny_VectorNx: a product whose rows and depth are both >= 8 runs through the
// column-major GEMV kernel (row accumulator from +0, then alpha * acc + result), anything smaller is the lazy product's
// sequential sum (tests/test_oracle.py: test_plant_step_bit_exact_vs_compiled_reference pins the oracle's restatement of
// this against the compiled expression, fp64 configurations included).
template <int NX, int NU>
__global__ void plant64_kernel(double *__restrict__ X, const double *__restrict__ U, const double *__restrict__ mats, int batch, int bpad)
{
    static_assert(!(NX >= 8 && NU >= 8), "Bdyn*u would take the GEMV kernel too");
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    constexpr int OFF_A = NU * NX + NX * NX + NU * NU + NX * NX, OFF_B = OFF_A + NX * NX;
    const double *A = mats + OFF_A, *Bm = mats + OFF_B;
    double x[NX], u[NU], xn[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = X[(size_t)j * bpad + b];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = U[(size_t)j * bpad + b];
#pragma unroll
    for (int i = 0; i < NX; i++)
    {
        double a;
        if constexpr (NX >= 8)
        {
            double c = 0.0;
#pragma unroll
            for (int j = 0; j < NX; j++) c = A[j * NX + i] * x[j] + c;
            a = c * 1.0 + 0.0;
        }
        else a = row_dot<NX, NX>(A, i, x);
        xn[i] = a + row_dot<NX, NU>(Bm, i, u);
    }
#pragma unroll
    for (int i = 0; i < NX; i++) X[(size_t)i * bpad + b] = xn[i];
}

thread_local std::string g_err64;
int fail64(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err64 = buf;
    return code;
}
#define HIP64(expr)                                                                                         \
    do                                                                                                      \
    {                                                                                                       \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail64(TINY_BATCH_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define CHECK64(c, ...) \
    if (!(c)) return fail64(TINY_BATCH_EINVAL, __VA_ARGS__)

bool xfam(int id) { return id == TINY_ARR_X || id == TINY_ARR_Q || id == TINY_ARR_P || id == TINY_ARR_V || id == TINY_ARR_VNEW || id == TINY_ARR_G; }
int grid64(long long total)
{
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

} // namespace

struct TinyBatch64
{
    int nx = 0, nu = 0, N = 0, batch = 0, bpad = 0, device = 0;
    double *arr[TINY_ARR_COUNT] = {};
    double *in[5] = {};        // xref, xmin, xmax, umin, umax
    int in_stride[5] = {1, 1, 1, 1, 1};
    bool in_set[5] = {};
    double *mats = nullptr, *res = nullptr, *staging = nullptr;
    double *row_gains = nullptr; // [3nx + 2nu + 1 + nx][16]: the matrices as one register per column and lane (admm_f64_rows_kernel)
    int kernel_choice = 0;       // tiny_batch64_select_kernel: 0 auto, 1 thread per instance, 2 sixteen lanes per instance
    int *status = nullptr, *iter = nullptr, *n_unsolved = nullptr;
    std::vector<double> hm; // host copy of the packed matrices
    bool have_cache = false, have_dyn = false, have_settings = false, mats_dirty = true;
    double rho = 0, abs_pri_tol = 0, abs_dua_tol = 0;
    int max_iter = 0, check_termination = 1, en_state_bound = 0, en_input_bound = 0;
};

namespace
{
size_t steps_of(const TinyBatch64 *tb, int id) { return xfam(id) ? tb->N : tb->N - 1; }
int dim_of(const TinyBatch64 *tb, int id) { return xfam(id) ? tb->nx : tb->nu; }
size_t mat_off(const TinyBatch64 *tb, int which) // Kinf, Pinf, Quu_inv, AmBKt, Adyn, Bdyn, Q
{
    const size_t nx = tb->nx, nu = tb->nu;
    const size_t sz[7] = {nu * nx, nx * nx, nu * nu, nx * nx, nx * nx, nx * nu, nx};
    size_t o = 0;
    for (int k = 0; k < which; k++) o += sz[k];
    return o;
}
// classes of the sixteen-lane kernel with a runtime horizon (any N <= 64): the unrolled body has a capacity of 32 or 64 steps
#define TINY_FOR_EACH_F64ROWS_RT(X) X(12, 4) X(4, 1) X(8, 4) X(12, 2) X(4, 2) X(4, 4)
constexpr int F64ROWS_RT_MAX_N = 64;
bool rows_unrolled(int nx, int nu, int N)
{
#define TINY_F64ROWS_CHECK(NX, NU, NN) \
    if (nx == NX && nu == NU && N == NN) return true;
    TINY_FOR_EACH_F64ROWS(TINY_F64ROWS_CHECK)
    return false;
}
bool rows_supported(int nx, int nu, int N)
{
    if (rows_unrolled(nx, nu, N)) return true;
#define TINY_F64ROWS_RT_CHECK(NX, NU) \
    if (nx == NX && nu == NU && N >= 2 && N <= F64ROWS_RT_MAX_N) return true;
    TINY_FOR_EACH_F64ROWS_RT(TINY_F64ROWS_RT_CHECK)
    return false;
}
// lane r of a row holds, per matrix column, the element of the row it owns (x rows r < nx, u rows nx <= r < nx + nu):
//   M1[k]  x rows Adyn(r,k)   | u rows Kinf(m,k)        M2[m]  x rows Bdyn(r,m)
//   M3[k]  x rows AmBKt(r,k)  | u rows Bdyn(k,m)        M45[m] x rows Kinf(m,r) | u rows Quu_inv(mr,m)
//   Q      x rows Q(r)                                  PT[k]  x rows Pinf(k,r)
std::vector<double> pack_row_gains(const TinyBatch64 *tb)
{
    const int nx = tb->nx, nu = tb->nu;
    const double *K = tb->hm.data() + mat_off(tb, 0), *Pinf = tb->hm.data() + mat_off(tb, 1), *Quu = tb->hm.data() + mat_off(tb, 2),
                 *Am = tb->hm.data() + mat_off(tb, 3), *A = tb->hm.data() + mat_off(tb, 4), *B = tb->hm.data() + mat_off(tb, 5),
                 *Q = tb->hm.data() + mat_off(tb, 6);
    std::vector<double> g((size_t)(3 * nx + 2 * nu + 1) * 16, 0.0);
    for (int r = 0; r < 16; r++)
    {
        const bool isx = r < nx, isu = r >= nx && r < nx + nu;
        const int mr = r - nx;
        for (int k = 0; k < nx; k++)
        {
            g[(size_t)k * 16 + r] = isx ? A[k * nx + r] : (isu ? K[k * nu + mr] : 0.0);
            g[(size_t)(nx + nu + k) * 16 + r] = isx ? Am[k * nx + r] : (isu ? B[mr * nx + k] : 0.0);
            g[(size_t)(2 * nx + 2 * nu + 1 + k) * 16 + r] = isx ? Pinf[r * nx + k] : 0.0;
        }
        for (int m = 0; m < nu; m++)
        {
            g[(size_t)(nx + m) * 16 + r] = isx ? B[m * nx + r] : 0.0;
            g[(size_t)(2 * nx + nu + m) * 16 + r] = isx ? K[r * nu + m] : (isu ? Quu[m * nu + mr] : 0.0);
        }
        g[(size_t)(2 * nx + 2 * nu) * 16 + r] = isx ? Q[r] : 0.0;
    }
    return g;
}
int set_input(TinyBatch64 *tb, int which, const double *src, int shared)
{
    CHECK64(tb && src, "NULL argument");
    HIP64(hipSetDevice(tb->device));
    const int dim = which < 3 ? tb->nx : tb->nu, steps = which < 3 ? tb->N : tb->N - 1;
    const int stride = shared ? 1 : tb->bpad, nb = shared ? 1 : tb->batch;
    if (tb->in[which] && tb->in_stride[which] != stride) { (void)hipFree(tb->in[which]); tb->in[which] = nullptr; }
    if (!tb->in[which])
    {
        HIP64(hipMalloc((void **)&tb->in[which], (size_t)steps * dim * stride * sizeof(double)));
        HIP64(hipMemset(tb->in[which], 0, (size_t)steps * dim * stride * sizeof(double)));
    }
    HIP64(hipMemcpy(tb->staging, src, (size_t)nb * steps * dim * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack64_kernel, dim3(grid64((long long)nb * steps * dim)), dim3(256), 0, 0, tb->staging, tb->in[which], nb, steps, dim, stride, 0, steps);
    HIP64(hipGetLastError());
    HIP64(hipDeviceSynchronize());
    tb->in_stride[which] = stride;
    tb->in_set[which] = true;
    return 0;
}
int prepare64(TinyBatch64 *tb, Params64 &P)
{
    if (!tb->have_cache || !tb->have_dyn || !tb->have_settings)
        return fail64(TINY_BATCH_ENOTREADY, "set_cache, set_dynamics and set_settings must be called first");
    if (tb->in_stride[1] != tb->in_stride[2] || tb->in_stride[3] != tb->in_stride[4])
        return fail64(TINY_BATCH_EINVAL, "min and max bounds must both be shared or both be per-instance");
    HIP64(hipSetDevice(tb->device));
    if (tb->mats_dirty)
    {
        HIP64(hipMemcpy(tb->mats, tb->hm.data(), tb->hm.size() * sizeof(double), hipMemcpyHostToDevice));
        const std::vector<double> g = pack_row_gains(tb);
        if (!tb->row_gains) HIP64(hipMalloc((void **)&tb->row_gains, g.size() * sizeof(double)));
        HIP64(hipMemcpy(tb->row_gains, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
        tb->mats_dirty = false;
    }
    P.nx = tb->nx; P.nu = tb->nu; P.N = tb->N; P.batch = tb->batch; P.bpad = tb->bpad;
    P.rho = tb->rho; P.abs_pri_tol = tb->abs_pri_tol; P.abs_dua_tol = tb->abs_dua_tol;
    P.max_iter = tb->max_iter; P.check_termination = tb->check_termination;
    P.en_state_bound = tb->en_state_bound; P.en_input_bound = tb->en_input_bound;
    for (int id = 0; id < TINY_ARR_COUNT; id++) P.arr[id] = tb->arr[id];
    P.xref = tb->in[0]; P.xmin = tb->in[1]; P.xmax = tb->in[2]; P.umin = tb->in[3]; P.umax = tb->in[4];
    P.xref_stride = tb->in_stride[0]; P.xb_stride = tb->in_stride[1]; P.ub_stride = tb->in_stride[3];
    P.mats = tb->mats; P.res = tb->res; P.status = tb->status; P.iter = tb->iter; P.n_unsolved = tb->n_unsolved;
    return 0;
}

// one of the six step functions over the whole batch
template <int FN>
int run_step64(TinyBatch64 *tb, int *conv_dev)
{
    CHECK64(tb, "NULL handle");
    Params64 P;
    int rc = prepare64(tb, P);
    if (rc < 0) return rc;
    const int nblocks = tb->bpad / WAVE64;
#define TINY_F64_STEP_LAUNCH(NX, NU) \
    if (tb->nx == NX && tb->nu == NU) hipLaunchKernelGGL((admm_f64_step_kernel<NX, NU, FN>), dim3(nblocks), dim3(WAVE64), 0, 0, P, conv_dev);
    TINY_FOR_EACH_F64DIMS(TINY_F64_STEP_LAUNCH)
    HIP64(hipGetLastError());
    HIP64(hipDeviceSynchronize());
    return 0;
}

} // namespace

extern "C"
{

int tiny_batch64_create(TinyBatch64 **out, int nx, int nu, int N, int batch, int device)
{
    CHECK64(out, "NULL out pointer");
    *out = nullptr;
    CHECK64(nx >= 1 && nu >= 1 && N >= 2 && batch >= 1, "need nx>=1, nu>=1, N>=2, batch>=1 (got %d,%d,%d,%d)", nx, nu, N, batch);
    bool have = false;
#define TINY_F64_HAVE(NX, NU) have = have || (nx == NX && nu == NU);
    TINY_FOR_EACH_F64DIMS(TINY_F64_HAVE)
    if (!have)
        return fail64(TINY_BATCH_EUNSUPPORTED, "no fp64 kernel instantiation for nx=%d nu=%d (add it to TINY_FOR_EACH_F64DIMS)", nx, nu);
    int ndev = 0;
    HIP64(hipGetDeviceCount(&ndev));
    CHECK64(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    HIP64(hipSetDevice(device));
    TinyBatch64 *tb = new TinyBatch64();
    tb->nx = nx; tb->nu = nu; tb->N = N; tb->batch = batch; tb->device = device;
    tb->bpad = (batch + WAVE64 - 1) / WAVE64 * WAVE64;
    auto zalloc = [&](void **p, size_t bytes) {
        if (hipMalloc(p, bytes) != hipSuccess) return false;
        return hipMemset(*p, 0, bytes) == hipSuccess;
    };
    bool ok = true;
    for (int id = 0; id < TINY_ARR_COUNT && ok; id++)
        ok = zalloc((void **)&tb->arr[id], steps_of(tb, id) * dim_of(tb, id) * tb->bpad * sizeof(double));
    ok = ok && zalloc((void **)&tb->res, 4 * (size_t)tb->bpad * sizeof(double)) && zalloc((void **)&tb->status, tb->bpad * sizeof(int)) &&
         zalloc((void **)&tb->iter, tb->bpad * sizeof(int)) && zalloc((void **)&tb->n_unsolved, sizeof(int)) &&
         zalloc((void **)&tb->staging, (size_t)batch * N * (nx > nu ? nx : nu) * sizeof(double)) &&
         zalloc((void **)&tb->mats, (mat_off(tb, 6) + nx) * sizeof(double));
    // inputs that were never set read as zero (the reference's zero-initialised members)
    for (int w = 0; w < 5 && ok; w++)
        ok = zalloc((void **)&tb->in[w], (size_t)(w < 3 ? N * nx : (N - 1) * nu) * sizeof(double));
    if (!ok || hipDeviceSynchronize() != hipSuccess)
    {
        tiny_batch64_destroy(tb);
        return fail64(TINY_BATCH_EHIP, "device allocation failed");
    }
    tb->hm.assign(mat_off(tb, 6) + nx, 0.0);
    *out = tb;
    return 0;
}

void tiny_batch64_destroy(TinyBatch64 *tb)
{
    if (!tb) return;
    (void)hipSetDevice(tb->device);
    for (int id = 0; id < TINY_ARR_COUNT; id++) (void)hipFree(tb->arr[id]);
    for (int w = 0; w < 5; w++) (void)hipFree(tb->in[w]);
    (void)hipFree(tb->mats); (void)hipFree(tb->res); (void)hipFree(tb->staging); (void)hipFree(tb->row_gains);
    (void)hipFree(tb->status); (void)hipFree(tb->iter); (void)hipFree(tb->n_unsolved);
    delete tb;
}

int tiny_batch64_set_cache(TinyBatch64 *tb, double rho, const double *Kinf, const double *Pinf, const double *Quu_inv, const double *AmBKt)
{
    CHECK64(tb && Kinf && Pinf && Quu_inv && AmBKt, "NULL argument");
    const size_t nx = tb->nx, nu = tb->nu;
    tb->rho = rho;
    std::copy(Kinf, Kinf + nu * nx, tb->hm.begin() + mat_off(tb, 0));
    std::copy(Pinf, Pinf + nx * nx, tb->hm.begin() + mat_off(tb, 1));
    std::copy(Quu_inv, Quu_inv + nu * nu, tb->hm.begin() + mat_off(tb, 2));
    std::copy(AmBKt, AmBKt + nx * nx, tb->hm.begin() + mat_off(tb, 3));
    tb->have_cache = true; tb->mats_dirty = true;
    return 0;
}

int tiny_batch64_set_dynamics(TinyBatch64 *tb, const double *Adyn, const double *Bdyn, const double *Q)
{
    CHECK64(tb && Adyn && Bdyn && Q, "NULL argument");
    const size_t nx = tb->nx, nu = tb->nu;
    std::copy(Adyn, Adyn + nx * nx, tb->hm.begin() + mat_off(tb, 4));
    std::copy(Bdyn, Bdyn + nx * nu, tb->hm.begin() + mat_off(tb, 5));
    std::copy(Q, Q + nx, tb->hm.begin() + mat_off(tb, 6));
    tb->have_dyn = true; tb->mats_dirty = true;
    return 0;
}

int tiny_batch64_set_settings(TinyBatch64 *tb, double abs_pri_tol, double abs_dua_tol, int max_iter, int check_termination,
                              int en_state_bound, int en_input_bound)
{
    CHECK64(tb, "NULL handle");
    CHECK64(check_termination >= 1, "check_termination must be >= 1 (the reference computes iter %% check_termination, admm.cpp:93)");
    tb->abs_pri_tol = abs_pri_tol; tb->abs_dua_tol = abs_dua_tol; tb->max_iter = max_iter; tb->check_termination = check_termination;
    tb->en_state_bound = en_state_bound; tb->en_input_bound = en_input_bound;
    tb->have_settings = true;
    return 0;
}

int tiny_batch64_set_x0(TinyBatch64 *tb, const double *x0)
{
    CHECK64(tb && x0, "NULL argument");
    HIP64(hipSetDevice(tb->device));
    HIP64(hipMemcpy(tb->staging, x0, (size_t)tb->batch * tb->nx * sizeof(double), hipMemcpyHostToDevice));
    // [B][1][nx] -> step 0 of x
    hipLaunchKernelGGL(pack64_kernel, dim3(grid64((long long)tb->batch * tb->nx)), dim3(256), 0, 0, tb->staging, tb->arr[TINY_ARR_X], tb->batch, 1,
                       tb->nx, tb->bpad, 0, 1);
    HIP64(hipGetLastError());
    HIP64(hipDeviceSynchronize());
    return 0;
}

int tiny_batch64_set_xref(TinyBatch64 *tb, const double *xref, int shared) { return set_input(tb, 0, xref, shared); }
int tiny_batch64_set_xmin(TinyBatch64 *tb, const double *v, int shared) { return set_input(tb, 1, v, shared); }
int tiny_batch64_set_xmax(TinyBatch64 *tb, const double *v, int shared) { return set_input(tb, 2, v, shared); }
int tiny_batch64_set_umin(TinyBatch64 *tb, const double *v, int shared) { return set_input(tb, 3, v, shared); }
int tiny_batch64_set_umax(TinyBatch64 *tb, const double *v, int shared) { return set_input(tb, 4, v, shared); }

int tiny_batch64_reset_dual_variables(TinyBatch64 *tb)
{
    CHECK64(tb, "NULL handle");
    HIP64(hipSetDevice(tb->device));
    HIP64(hipMemset(tb->arr[TINY_ARR_Y], 0, (size_t)(tb->N - 1) * tb->nu * tb->bpad * sizeof(double)));
    HIP64(hipMemset(tb->arr[TINY_ARR_G], 0, (size_t)tb->N * tb->nx * tb->bpad * sizeof(double)));
    return 0;
}

int tiny_batch64_forward_pass(TinyBatch64 *tb) { return run_step64<F64_FORWARD_PASS>(tb, nullptr); }
int tiny_batch64_update_slack(TinyBatch64 *tb) { return run_step64<F64_UPDATE_SLACK>(tb, nullptr); }
int tiny_batch64_update_dual(TinyBatch64 *tb) { return run_step64<F64_UPDATE_DUAL>(tb, nullptr); }
int tiny_batch64_update_linear_cost(TinyBatch64 *tb) { return run_step64<F64_UPDATE_LINEAR_COST>(tb, nullptr); }
int tiny_batch64_backward_pass_grad(TinyBatch64 *tb) { return run_step64<F64_BACKWARD_PASS_GRAD>(tb, nullptr); }
int tiny_batch64_termination_condition(TinyBatch64 *tb, int *converged)
{
    CHECK64(tb && converged, "NULL argument");
    HIP64(hipSetDevice(tb->device));
    int *dev = nullptr;
    HIP64(hipMalloc((void **)&dev, (size_t)tb->bpad * sizeof(int)));
    int rc = run_step64<F64_TERMINATION_CONDITION>(tb, dev);
    if (rc >= 0 && hipMemcpy(converged, dev, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail64(TINY_BATCH_EHIP, "copying the termination flags back failed");
    (void)hipFree(dev);
    return rc;
}

int tiny_batch64_solve(TinyBatch64 *tb)
{
    CHECK64(tb, "NULL handle");
    Params64 P;
    {
        const int rc = prepare64(tb, P);
        if (rc < 0) return rc;
    }
    const bool rows = tb->kernel_choice == 2 || (tb->kernel_choice == 0 && rows_supported(tb->nx, tb->nu, tb->N));
    if (rows && !rows_supported(tb->nx, tb->nu, tb->N))
        return fail64(TINY_BATCH_EUNSUPPORTED, "the sixteen-lane fp64 kernel has no instantiation for nx=%d nu=%d N=%d", tb->nx, tb->nu, tb->N);
    HIP64(hipMemset(tb->n_unsolved, 0, sizeof(int)));
    const int nblocks = tb->bpad / WAVE64;
    if (rows)
    {
        const int nrow_blocks = (tb->batch + 3) / 4;
#define TINY_F64ROWS_LAUNCH(NX, NU, NN)                                                                                            \
    if (tb->nx == NX && tb->nu == NU && tb->N == NN)                                                                               \
        hipLaunchKernelGGL((admm_f64_rows_kernel<NX, NU, NN>), dim3(nrow_blocks), dim3(WAVE64), 0, 0, P, (const double *)tb->row_gains);
        if (rows_unrolled(tb->nx, tb->nu, tb->N)) { TINY_FOR_EACH_F64ROWS(TINY_F64ROWS_LAUNCH) }
        else
        {
#define TINY_F64ROWS_RT_LAUNCH(NX, NU)                                                                                              \
    if (tb->nx == NX && tb->nu == NU)                                                                                               \
    {                                                                                                                               \
        if (tb->N <= 32) hipLaunchKernelGGL((admm_f64_rows_kernel<NX, NU, 32, true>), dim3(nrow_blocks), dim3(WAVE64), 0, 0, P, (const double *)tb->row_gains); \
        else hipLaunchKernelGGL((admm_f64_rows_kernel<NX, NU, 64, true>), dim3(nrow_blocks), dim3(WAVE64), 0, 0, P, (const double *)tb->row_gains);             \
    }
            TINY_FOR_EACH_F64ROWS_RT(TINY_F64ROWS_RT_LAUNCH)
        }
    }
    else
    {
#define TINY_F64_LAUNCH(NX, NU) \
    if (tb->nx == NX && tb->nu == NU) hipLaunchKernelGGL((admm_f64_kernel<NX, NU>), dim3(nblocks), dim3(WAVE64), 0, 0, P);
        TINY_FOR_EACH_F64DIMS(TINY_F64_LAUNCH)
    }
    HIP64(hipGetLastError());
    int n = 0;
    HIP64(hipMemcpy(&n, tb->n_unsolved, sizeof(int), hipMemcpyDeviceToHost));
    return n > 0 ? 1 : 0;
}

int tiny_batch64_mpc_step(TinyBatch64 *tb)
{
    CHECK64(tb, "NULL handle");
    int rc = tiny_batch64_reset_dual_variables(tb); // quadrotor_hovering.cpp:100-101
    if (rc < 0) return rc;
    rc = tiny_batch64_solve(tb);                    // :104
    if (rc < 0) return rc;
    const int nb = (tb->batch + 127) / 128;          // :110-111
#define TINY_F64_PLANT(NX, NU)                                                                                                       \
    if (tb->nx == NX && tb->nu == NU)                                                                                                \
        hipLaunchKernelGGL((plant64_kernel<NX, NU>), dim3(nb), dim3(128), 0, 0, tb->arr[TINY_ARR_X], tb->arr[TINY_ARR_U], tb->mats, tb->batch, tb->bpad);
    TINY_FOR_EACH_F64DIMS(TINY_F64_PLANT)
    HIP64(hipGetLastError());
    return rc;
}

int tiny_batch64_get_first_columns(TinyBatch64 *tb, double *x0, double *u0)
{
    CHECK64(tb, "NULL handle");
    HIP64(hipSetDevice(tb->device));
    for (int w = 0; w < 2; w++)
    {
        double *dst = w ? u0 : x0;
        if (!dst) continue;
        const int dim = w ? tb->nu : tb->nx;
        hipLaunchKernelGGL(unpack64_kernel, dim3(grid64((long long)tb->batch * dim)), dim3(256), 0, 0, tb->arr[w ? TINY_ARR_U : TINY_ARR_X], tb->staging,
                           tb->batch, 1, dim, tb->bpad);
        HIP64(hipGetLastError());
        HIP64(hipMemcpy(dst, tb->staging, (size_t)tb->batch * dim * sizeof(double), hipMemcpyDeviceToHost));
    }
    return 0;
}

int tiny_batch64_select_kernel(TinyBatch64 *tb, int which)
{
    CHECK64(tb, "NULL handle");
    CHECK64(which >= 0 && which <= 2, "kernel must be 0 (auto), 1 (one thread per instance) or 2 (sixteen lanes per instance)");
    if (which == 2 && !rows_supported(tb->nx, tb->nu, tb->N))
        return fail64(TINY_BATCH_EUNSUPPORTED, "the sixteen-lane fp64 kernel has no instantiation for nx=%d nu=%d N=%d", tb->nx, tb->nu, tb->N);
    tb->kernel_choice = which;
    return 0;
}

const char *tiny_batch64_kernel_name(TinyBatch64 *tb)
{
    static thread_local char nm[64];
    if (!tb) return "";
    const bool rows = tb->kernel_choice == 2 || (tb->kernel_choice == 0 && rows_supported(tb->nx, tb->nu, tb->N));
    if (rows && !rows_unrolled(tb->nx, tb->nu, tb->N)) snprintf(nm, sizeof nm, "rows64<%d,%d,n<=%d>", tb->nx, tb->nu, tb->N <= 32 ? 32 : 64);
    else if (rows) snprintf(nm, sizeof nm, "rows64<%d,%d,%d>", tb->nx, tb->nu, tb->N);
    else snprintf(nm, sizeof nm, "thread64<%d,%d>", tb->nx, tb->nu);
    return nm;
}

int tiny_batch64_set_array(TinyBatch64 *tb, int id, const double *src)
{
    CHECK64(tb && src, "NULL argument");
    CHECK64(id >= 0 && id < TINY_ARR_COUNT, "bad array id %d", id);
    HIP64(hipSetDevice(tb->device));
    const int steps = (int)steps_of(tb, id), dim = dim_of(tb, id);
    HIP64(hipMemcpy(tb->staging, src, (size_t)tb->batch * steps * dim * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack64_kernel, dim3(grid64((long long)tb->batch * steps * dim)), dim3(256), 0, 0, tb->staging, tb->arr[id], tb->batch, steps, dim,
                       tb->bpad, 0, steps);
    HIP64(hipGetLastError());
    HIP64(hipDeviceSynchronize());
    return 0;
}

int tiny_batch64_get_array(TinyBatch64 *tb, int id, double *dst)
{
    CHECK64(tb && dst, "NULL argument");
    CHECK64(id >= 0 && id < TINY_ARR_COUNT, "bad array id %d", id);
    HIP64(hipSetDevice(tb->device));
    const int steps = (int)steps_of(tb, id), dim = dim_of(tb, id);
    hipLaunchKernelGGL(unpack64_kernel, dim3(grid64((long long)tb->batch * steps * dim)), dim3(256), 0, 0, tb->arr[id], tb->staging, tb->batch, steps, dim,
                       tb->bpad);
    HIP64(hipGetLastError());
    HIP64(hipMemcpy(dst, tb->staging, (size_t)tb->batch * steps * dim * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int tiny_batch64_get_status(TinyBatch64 *tb, int *iter, int *status, double *residuals)
{
    CHECK64(tb, "NULL handle");
    HIP64(hipSetDevice(tb->device));
    if (iter) HIP64(hipMemcpy(iter, tb->iter, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost));
    if (status) HIP64(hipMemcpy(status, tb->status, (size_t)tb->batch * sizeof(int), hipMemcpyDeviceToHost));
    if (residuals)
    {
        std::vector<double> tmp(4 * (size_t)tb->bpad);
        HIP64(hipMemcpy(tmp.data(), tb->res, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < tb->batch; b++)
            for (int k = 0; k < 4; k++) residuals[4 * (size_t)b + k] = tmp[(size_t)k * tb->bpad + b];
    }
    return 0;
}

int tiny_batch64_set_status(TinyBatch64 *tb, const int *iter, const int *status, const double *residuals)
{
    CHECK64(tb, "NULL handle");
    HIP64(hipSetDevice(tb->device));
    if (iter) HIP64(hipMemcpy(tb->iter, iter, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice));
    if (status) HIP64(hipMemcpy(tb->status, status, (size_t)tb->batch * sizeof(int), hipMemcpyHostToDevice));
    if (residuals)
    {
        std::vector<double> tmp(4 * (size_t)tb->bpad, 0.0);
        for (int b = 0; b < tb->batch; b++)
            for (int k = 0; k < 4; k++) tmp[(size_t)k * tb->bpad + b] = residuals[4 * (size_t)b + k];
        HIP64(hipMemcpy(tb->res, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

const char *tiny_batch64_last_error(void) { return g_err64.c_str(); }

} // extern "C"
