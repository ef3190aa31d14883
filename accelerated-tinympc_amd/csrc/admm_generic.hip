// admm_generic.hip — EXACT arithmetic for any problem class the reference's orders are defined for, without a rebuild.
//
// The reference takes NSTATES / NINPUTS / NHORIZON as macros (src/tinympc/glob_opts.hpp:3-9); the fast exact kernels of this library
// are compiled per (nx, nu) class (TINY_FOR_EACH_ROWDIMS / _WAVEDIMS).  Until round 4 a class outside those lists ran in fma
// arithmetic on the padded MFMA kernel (admm_stream.hip).  This kernel closes the gap: ONE THREAD PER INSTANCE, dimensions and the
// reduction orders chosen at RUN time, every product and sum a separately rounded fp32 operation in the order Eigen 3.4.90 (SSE2, packets
// of four floats) evaluates the expressions of src/tinympc/admm.cpp:15-152 — the same rules rowlane_math.h (RowPlans) and wave_math.h
// (WavePlans) instantiate at compile time:
//   lazy product into a result with R rows:  R > 1 and R % 4 == 0 -> packet-evaluated = sequential over k
//                                            R == 1               -> vectorised redux = packet tree (below) over the contiguous row
//                                            else                 -> coefficient-evaluated = halving tree (complete unrolling: 3n - 1 <= 110)
//   halving tree  T(lo, n) = T(lo, n/2) + T(lo + n/2, n - n/2)
//   packet tree   products in packets of 4, packets summed by the halving tree, (s0 + s2) + (s1 + s3), then the n % 4 leftover (halving tree)
//   nx >= 8 and nu >= 8: Bdyn^T p goes through Eigen's row-major GEMV kernel (four accumulators from zero, (c0+c2)+(c1+c3), 0 + 1*acc) and
//                        Quu_inv (.) through the column-major one (0 + 1*(0 + sequential))
// Defined — as in the compiled classes — for nx, nu each <= 4 or a multiple of 4 (for other sizes the reference's own order depends on the
// 16-byte alignment of the destination column) and nx <= 36 (beyond Eigen's complete-unrolling limit, 3n - 1 <= 110, the
// coefficient-evaluated products are not the plain loop one would expect: measured against the compiled reference for nx = 40 and 64, not
// restated); outside that tiny_batch_create offers fma arithmetic only.
// Results are bitwise equal to the compiled reference of the class (tests/test_parity_gpu.py::test_generic_exact_kernel_*).
//
// It is the any-class fallback, not a fast path: the state goes through HBM / L2 in every iteration, one lane per instance, in the TILE
// layout of admm_stream.hip (so that a handle can switch between this kernel and the fma kernel without converting its workspace).
#include "tinympc_internal.h"

namespace tinympc
{

namespace
{
constexpr int GEN_MAX = 36; // nx <= 36 (nu <= 32): beyond Eigen's complete-unrolling limit (3n - 1 <= 110) the orders are not pinned (measured: nx = 40, 64 differ)

__device__ inline float gen_tree(const float *v, int n) // T(0, n), evaluated with an explicit stack (depth <= 7)
{
    int lo_s[8], n_s[8], ph[8];
    float left[8];
    int sp = 0;
    lo_s[0] = 0; n_s[0] = n; ph[0] = 0;
    float ret = 0.f;
    while (sp >= 0)
    {
        const int lo = lo_s[sp], m = n_s[sp];
        if (m == 1) { ret = v[lo]; sp--; continue; }
        const int h = m / 2;
        if (ph[sp] == 0) { ph[sp] = 1; lo_s[sp + 1] = lo; n_s[sp + 1] = h; ph[sp + 1] = 0; sp++; }
        else if (ph[sp] == 1) { left[sp] = ret; ph[sp] = 2; lo_s[sp + 1] = lo + h; n_s[sp + 1] = m - h; ph[sp + 1] = 0; sp++; }
        else { ret = left[sp] + ret; sp--; }
    }
    return ret;
}
// element l of the packets [0, npk) summed by the halving tree over the packets
__device__ inline float gen_ptree_lane(const float *v, int npk, int l)
{
    int lo_s[8], n_s[8], ph[8];
    float left[8];
    int sp = 0;
    lo_s[0] = 0; n_s[0] = npk; ph[0] = 0;
    float ret = 0.f;
    while (sp >= 0)
    {
        const int lo = lo_s[sp], m = n_s[sp];
        if (m == 1) { ret = v[4 * lo + l]; sp--; continue; }
        const int h = m / 2;
        if (ph[sp] == 0) { ph[sp] = 1; lo_s[sp + 1] = lo; n_s[sp + 1] = h; ph[sp + 1] = 0; sp++; }
        else if (ph[sp] == 1) { left[sp] = ret; ph[sp] = 2; lo_s[sp + 1] = lo + h; n_s[sp + 1] = m - h; ph[sp + 1] = 0; sp++; }
        else { ret = left[sp] + ret; sp--; }
    }
    return ret;
}
__device__ inline float gen_seq(const float *t, int n)
{
    float acc = t[0];
    for (int k = 1; k < n; k++) acc = acc + t[k];
    return acc;
}
__device__ inline float gen_novec(const float *t, int n) { return gen_tree(t, n); } // (n <= 36: always inside Eigen's complete-unrolling limit)
__device__ inline float gen_vec(const float *t, int n)
{
    if (n < 4) return gen_novec(t, n);
    const int npk = n / 4, vs = 4 * npk;
    const float s0 = gen_ptree_lane(t, npk, 0), s1 = gen_ptree_lane(t, npk, 1), s2 = gen_ptree_lane(t, npk, 2), s3 = gen_ptree_lane(t, npk, 3);
    float res = (s0 + s2) + (s1 + s3);
    if (vs != n) res = res + gen_tree(t + vs, n - vs);
    return res;
}
__device__ inline float gen_gemv_rm(const float *t, int n) // Eigen's row-major GEMV inner product over the products t[k]
{
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    const int vs = (n / 4) * 4;
    for (int k = 0; k < vs; k += 4) { c0 = c0 + t[k]; c1 = c1 + t[k + 1]; c2 = c2 + t[k + 2]; c3 = c3 + t[k + 3]; }
    float res = (c0 + c2) + (c1 + c3);
    for (int k = vs; k < n; k++) res = res + t[k];
    return 0.f + res;
}
// (row i of a column-major rows x cols matrix) . xin for a lazy product whose result has `rows` rows
__device__ inline float gen_row_dot(const float *M, int rows, int cols, int i, const float *xin, float *t)
{
    if (rows == 1)
    {
        for (int k = 0; k < cols; k++) t[k] = M[k] * xin[k];
        return gen_vec(t, cols);
    }
    for (int k = 0; k < cols; k++) t[k] = M[(size_t)k * rows + i] * xin[k];
    return (rows % 4 == 0) ? gen_seq(t, cols) : gen_novec(t, cols);
}

// element (step, row) of instance (tile, c) in the TILE layout: x-family [ntiles][N][64][NXC], u-family [ntiles][N-1][64][NUC]
struct GenIdx
{
    int tile, c, N, NXC, NUC;
    __device__ size_t x(int step, int row) const { return (((size_t)tile * N + step) * WAVE + ((row & 3) * 16 + c)) * NXC + (row >> 2); }
    __device__ size_t u(int step, int row) const { return (((size_t)tile * (N - 1) + step) * WAVE + ((row & 3) * 16 + c)) * NUC + (row >> 2); }
};

__global__ __launch_bounds__(64) void admm_generic_kernel(const SolveParams P, const float *__restrict__ G, int NXC, int NUC)
{
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= P.batch) return;
    const int nx = P.nx, nu = P.nu, N = P.N;
    const float rho = P.rho;
    const float *Kinf = G, *Pinf = Kinf + nu * nx, *Quu = Pinf + nx * nx, *AmBKt = Quu + nu * nu, *Adyn = AmBKt + nx * nx, *Bdyn = Adyn + nx * nx,
                *Qd = Bdyn + nx * nu;
    GenIdx I{inst / TILE, inst % TILE, N, NXC, NUC};
    // shared inputs live in tile 0 (tile stride 0), per-instance ones in the instance's own tile
    GenIdx Ib = I, Ibu = I, Ir = I;
    if (P.xb_tile_stride == 0) Ib.tile = 0;
    if (P.ub_tile_stride == 0) Ibu.tile = 0;
    if (P.xref_tile_stride == 0) Ir.tile = 0;
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[inst];
    auto xref = [&](int i, int row) -> float {
        if (P.xref_mode == 1)
        {
            int rw = wstart + i;
            rw = rw < P.table_rows ? rw : P.table_rows - 1;
            return P.xref_table[((size_t)rw * 4 + (row & 3)) * NXC + (row >> 2)];
        }
        return P.xref[Ir.x(i, row)];
    };
    float t[GEN_MAX], xi[GEN_MAX], ui[32], xn[GEN_MAX], tmp[32], pn[GEN_MAX], pi[GEN_MAX], ri[32]; // (t: products of up to max(nx, nu) <= 36 terms)
    const bool gemv = nu >= 8 && nx >= 8;
    const bool p_packet = nu == 1 && nx % 4 == 0;

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        P.status[inst] = TINY_STATUS_UNSOLVED_;
        P.iter[inst] = 1;
        atomicAdd(P.n_unsolved, 1);
        return;
    }
    // reset_workspace() / reset_dual_variables() folded into the launch: the arrays they zero are zeroed here (x.col(0) carries x0)
    if (P.cold_start || P.duals_zero)
    {
        for (int i = 0; i < N; i++)
            for (int r = 0; r < nx; r++)
            {
                P.g[I.x(i, r)] = 0.f;
                if (P.cold_start) { P.v[I.x(i, r)] = 0.f; P.p[I.x(i, r)] = 0.f; }
            }
        for (int i = 0; i < N - 1; i++)
            for (int r = 0; r < nu; r++)
            {
                P.y[I.u(i, r)] = 0.f;
                if (P.cold_start) { P.z[I.u(i, r)] = 0.f; P.d[I.u(i, r)] = 0.f; }
            }
    }
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (!P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    for (int it = 0; it < P.max_iter; ++it)
    {
        itn = it + 1; // :120
        // ---- forward_pass (admm.cpp:27-37) ----
        for (int r = 0; r < nx; r++) xi[r] = P.x[I.x(0, r)];
        for (int i = 0; i < N - 1; i++)
        {
            for (int j = 0; j < nu; j++) ui[j] = -gen_row_dot(Kinf, nu, nx, j, xi, t) - P.d[I.u(i, j)]; // :31
            for (int j = 0; j < nx; j++)
            {
                const float a = gen_row_dot(Adyn, nx, nx, j, xi, t);
                xn[j] = a + gen_row_dot(Bdyn, nx, nu, j, ui, t); // :35
            }
            for (int j = 0; j < nu; j++) P.u[I.u(i, j)] = ui[j];
            for (int j = 0; j < nx; j++) { P.x[I.x(i + 1, j)] = xn[j]; xi[j] = xn[j]; }
        }
        // ---- update_slack (:45-61), update_dual (:67-71), update_linear_cost (:77-85), the residual maxima of termination_condition (:95-98) ----
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        bool first_x = true, first_u = true;
        for (int i = 0; i < N - 1; i++)
            for (int j = 0; j < nu; j++)
            {
                const size_t o = I.u(i, j);
                const float uu = P.u[o], yy = P.y[o];
                float zn = uu + yy; // :47
                if (P.en_input_bound)
                {
                    const float lo = P.umin[Ibu.u(i, j)], hi = P.umax[Ibu.u(i, j)];
                    zn = (lo < zn) ? zn : lo; // u_min.cwiseMax(znew)
                    zn = (zn < hi) ? zn : hi; // u_max.cwiseMin(.)
                }
                P.znew[o] = zn;
                const float yn = yy + uu - zn; // :69
                P.y[o] = yn;
                P.r[o] = -rho * (zn - yn); // :80
                const float a = fabsf(uu - zn), b = fabsf(P.z[o] - zn);
                pri_u = (first_u || a > pri_u) ? a : pri_u;
                dua_u = (first_u || b > dua_u) ? b : dua_u;
                first_u = false;
            }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < nx; j++)
            {
                const size_t o = I.x(i, j);
                const float xx = P.x[o], gg = P.g[o];
                float vn = xx + gg; // :48
                if (P.en_state_bound)
                {
                    const float lo = P.xmin[Ib.x(i, j)], hi = P.xmax[Ib.x(i, j)];
                    vn = (lo < vn) ? vn : lo;
                    vn = (vn < hi) ? vn : hi;
                }
                P.vnew[o] = vn;
                const float gn = gg + xx - vn; // :70
                P.g[o] = gn;
                float q = -(xref(i, j) * Qd[j]); // :81
                q = q - rho * (vn - gn);         // :82
                P.q[o] = q;
                const float a = fabsf(xx - vn), b = fabsf(P.v[o] - vn);
                pri_x = (first_x || a > pri_x) ? a : pri_x;
                dua_x = (first_x || b > dua_x) ? b : dua_x;
                first_x = false;
            }
        // p.col(N-1) = -(Xref.col(N-1)^T Pinf) - rho (vnew - g)   (:83-84): a row-vector lazy product, coefficient-wise, both operands contiguous
        for (int k = 0; k < nx; k++) xi[k] = xref(N - 1, k);
        for (int j = 0; j < nx; j++)
        {
            for (int k = 0; k < nx; k++) t[k] = xi[k] * Pinf[(size_t)j * nx + k];
            float pv = -gen_vec(t, nx);
            pv = pv - rho * (P.vnew[I.x(N - 1, j)] - P.g[I.x(N - 1, j)]);
            P.p[I.x(N - 1, j)] = pv;
        }
        // ---- termination_condition (:91-109) ----
        if ((it + 1) % P.check_termination == 0)
        {
            r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
            if (r_ps < P.abs_pri_tol && r_pi < P.abs_pri_tol && r_ds < P.abs_dua_tol && r_di < P.abs_dua_tol)
            {
                st = TINY_STATUS_SOLVED_; // returns BEFORE the v/z copy and the backward pass (:135-137)
                break;
            }
        }
        // ---- v = vnew; z = znew (:141-142) ----
        for (int i = 0; i < N; i++)
            for (int j = 0; j < nx; j++) P.v[I.x(i, j)] = P.vnew[I.x(i, j)];
        for (int i = 0; i < N - 1; i++)
            for (int j = 0; j < nu; j++) P.z[I.u(i, j)] = P.znew[I.u(i, j)];
        // ---- backward_pass_grad (:15-22) ----
        for (int j = 0; j < nx; j++) pn[j] = P.p[I.x(N - 1, j)];
        for (int i = N - 2; i >= 0; i--)
        {
            for (int j = 0; j < nu; j++) ri[j] = P.r[I.u(i, j)];
            // d_i = Quu_inv (Bdyn^T p_{i+1} + r_i)   (:19): the inner product into a temporary first (column j of Bdyn is contiguous)
            for (int j = 0; j < nu; j++)
            {
                for (int k = 0; k < nx; k++) t[k] = Bdyn[(size_t)j * nx + k] * pn[k];
                tmp[j] = (gemv ? gen_gemv_rm(t, nx) : gen_vec(t, nx)) + ri[j];
            }
            for (int j = 0; j < nu; j++)
            {
                float dd;
                if (gemv)
                {
                    for (int k = 0; k < nu; k++) t[k] = Quu[(size_t)k * nu + j] * tmp[k];
                    dd = 0.f + (0.f + gen_seq(t, nu)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
                }
                else dd = gen_row_dot(Quu, nu, nu, j, tmp, t);
                P.d[I.u(i, j)] = dd;
            }
            // p_i = q_i + AmBKt p_{i+1} - Kinf^T r_i   (:20): Kinf^T is a row-major view, coefficient-wise unless nu == 1
            for (int j = 0; j < nx; j++)
            {
                for (int k = 0; k < nx; k++) t[k] = AmBKt[(size_t)k * nx + j] * pn[k];
                const float a = p_packet ? gen_seq(t, nx) : gen_novec(t, nx);
                for (int k = 0; k < nu; k++) t[k] = Kinf[(size_t)j * nu + k] * ri[k];
                const float kk = gen_vec(t, nu);
                pi[j] = P.q[I.x(i, j)] + a - kk;
            }
            for (int j = 0; j < nx; j++) { P.p[I.x(i, j)] = pi[j]; pn[j] = pi[j]; }
        }
    }
    P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
    P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
    P.status[inst] = st;
    P.iter[inst] = itn;
    if (st != TINY_STATUS_SOLVED_) atomicAdd(P.n_unsolved, 1);
}
} // namespace

// exact arithmetic is defined for these dimensions (see the header comment)
bool generic_exact_supported(int nx, int nu) { return nx >= 1 && nu >= 1 && nx <= GEN_MAX && nu <= 32 && (nx <= 4 || nx % 4 == 0) && (nu <= 4 || nu % 4 == 0); }

hipError_t launch_admm_generic(const SolveParams &P, const float *gains, int nxc, int nuc, hipStream_t stream)
{
    if (!generic_exact_supported(P.nx, P.nu)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(admm_generic_kernel, dim3((P.batch + 63) / 64), dim3(64), 0, stream, P, gains, nxc, nuc);
    return hipGetLastError();
}

} // namespace tinympc
