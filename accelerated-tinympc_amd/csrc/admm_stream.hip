// admm_stream.hip — generic batched TinyMPC ADMM kernel, loop-carried state streamed through HBM/L2.
//
// Restates tiny_solve() of the reference (src/tinympc/admm.cpp:111-152) for 16 instances per
// wavefront.  One launch runs ALL ADMM iterations; instances that converge are frozen (their
// stores are masked) exactly where the reference returns (admm.cpp:135-137: before the v/z copy
// and the backward pass).
//
// Per iteration the reference makes seven passes over the workspace; here they are fused into
// two horizon sweeps:
//   forward sweep  i = 0..N-1 : forward_pass (:27-37) + update_slack (:45-61) + update_dual (:67-71)
//                               + the residual maxima of termination_condition (:95-98)
//                               + the terminal p of update_linear_cost (:83-84)
//   backward sweep i = N-2..0 : r,q of update_linear_cost (:80-82) recomputed on the fly,
//                               v=vnew / z=znew (:141-142), backward_pass_grad (:15-22)
// The gain x state products are v_mfma_f32_16x16x4_f32 issues (exact fp32 fma chains): the
// instances are the 16 MFMA columns, the gain slices are single-VGPR A operands packed by the
// host (tinympc_batch.hip: pack_operands), and the stacked vector [x;u] is both B operand and D
// result layout, so no cross-lane traffic occurs inside a sweep.
#include "tile_math.h"

namespace tinympc
{

template <int NXC, int NUC>
__global__ __launch_bounds__(WAVE) void admm_stream_kernel(const SolveParams P)
{
    using D = Dims<NXC, NUC>;
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    const int c = lane & 15, gq = lane >> 4;
    const int inst = tile * TILE + c;
    const bool valid = inst < P.batch;
    const int N = P.N;
    const float rho = P.rho;

    Operands<D> op;
    op.load(P.opnd, lane);
    float Qv[NXC];
    ldv<NXC>(P.qvec + lane * NXC, Qv);

    const size_t xstep = (size_t)WAVE * NXC, ustep = (size_t)WAVE * NUC;
    const size_t xbase = (size_t)tile * N * xstep + (size_t)lane * NXC;
    const size_t ubase = (size_t)tile * (N - 1) * ustep + (size_t)lane * NUC;
    const float *xmin_p = P.xmin + (size_t)tile * P.xb_tile_stride + (size_t)lane * NXC;
    const float *xmax_p = P.xmax + (size_t)tile * P.xb_tile_stride + (size_t)lane * NXC;
    const float *umin_p = P.umin + (size_t)tile * P.ub_tile_stride + (size_t)lane * NUC;
    const float *umax_p = P.umax + (size_t)tile * P.ub_tile_stride + (size_t)lane * NUC;
    const float *xref_p = P.xref + (size_t)tile * P.xref_tile_stride + (size_t)lane * NXC;
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];

    auto load_xref = [&](int i, float(&o)[NXC]) {
        if (P.xref_mode == 1)
        {
            int row = wstart + i;
            row = row < P.table_rows ? row : P.table_rows - 1;
            ldv<NXC>(P.xref_table + ((size_t)row * 4 + gq) * NXC, o);
        }
        else
            ldv<NXC>(xref_p + (size_t)i * xstep, o);
    };

    float x0[NXC];
    ldv<NXC>(P.x + xbase, x0);

    // per-instance scalars (work->status, work->iter and the four residual fields are live across solves)
    int st = 0, itn = 0;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start) // reset_workspace() zeroes the residual fields too
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    st = TINY_STATUS_UNSOLVED_; // admm.cpp:114
    itn = 1;                    // admm.cpp:115

    // -(Xref_{N-1}^T Pinf): constant during a solve (admm.cpp:83)
    float pterm[NXC];
    {
        float xr[NXC];
        load_xref(N - 1, xr);
        f32x4 acc[D::NTX];
#pragma unroll
        for (int t = 0; t < D::NTX; t++)
        {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NXC; k++) acc[t] = MFMA(op.AP[t][k], xr[k], acc[t]);
        }
#pragma unroll
        for (int k = 0; k < NXC; k++) pterm[k] = acc[k / 4][k % 4];
    }

    bool active = valid && (P.max_iter > 0);

    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        const bool zero_duals = (it == 0) && ((P.duals_zero | P.cold_start) != 0);
        const bool zero_state = (it == 0) && (P.cold_start != 0);
        // an instance that exhausts max_iter has its d overwritten by the last backward sweep, so its x,u (which
        // belong to the last FORWARD sweep) are stored here; converged instances regenerate theirs at the end.
        const bool keep_xu = active && (it == P.max_iter - 1);
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        float xs[NXC], pN[NXC];
#pragma unroll
        for (int k = 0; k < NXC; k++) xs[k] = x0[k];

        // ---------------- forward sweep ----------------
        for (int i = 0; i < N; i++)
        {
            const size_t xo = xbase + (size_t)i * xstep;
            // state part of step i: vnew = clip(x+g), g += x - vnew, residual maxima   (admm.cpp:48,57-60,70,95-96)
            float g[NXC], v[NXC], vn[NXC];
            if (zero_duals) {
#pragma unroll
                for (int k = 0; k < NXC; k++) g[k] = 0.f;
            } else
                ldv<NXC>(P.g + xo, g);
            if (zero_state) {
#pragma unroll
                for (int k = 0; k < NXC; k++) v[k] = 0.f;
            } else
                ldv<NXC>(P.v + xo, v);
            float lo[NXC], hi[NXC];
            if (P.en_state_bound)
            {
                ldv<NXC>(xmin_p + (size_t)i * xstep, lo);
                ldv<NXC>(xmax_p + (size_t)i * xstep, hi);
            }
            float us[NUC], xn[NXC], dd[NUC];
            if (i < N - 1)
            {
                if (zero_state) {
#pragma unroll
                    for (int m = 0; m < NUC; m++) dd[m] = 0.f;
                } else
                    ldv<NUC>(P.d + ubase + (size_t)i * ustep, dd);
                lqr_step<D>(op, xs, dd, us, xn);
            }
#pragma unroll
            for (int k = 0; k < NXC; k++)
            {
                float t = xs[k] + g[k];
                if (P.en_state_bound) t = fminf(hi[k], fmaxf(lo[k], t));
                vn[k] = t;
                g[k] = (g[k] + xs[k]) - t;
                pri_x = fmaxf(pri_x, fabsf(xs[k] - t));
                dua_x = fmaxf(dua_x, fabsf(v[k] - t));
            }
            stv<NXC>(P.vnew + xo, vn, active);
            stv<NXC>(P.g + xo, g, active);
            stv<NXC>(P.x + xo, xs, keep_xu);
            if (i < N - 1)
            {
                const size_t uo = ubase + (size_t)i * ustep;
                float y[NUC], z[NUC], zn[NUC], ulo[NUC], uhi[NUC];
                if (zero_duals) {
#pragma unroll
                    for (int m = 0; m < NUC; m++) y[m] = 0.f;
                } else
                    ldv<NUC>(P.y + uo, y);
                if (zero_state) {
#pragma unroll
                    for (int m = 0; m < NUC; m++) z[m] = 0.f;
                } else
                    ldv<NUC>(P.z + uo, z);
                if (P.en_input_bound)
                {
                    ldv<NUC>(umin_p + (size_t)i * ustep, ulo);
                    ldv<NUC>(umax_p + (size_t)i * ustep, uhi);
                }
#pragma unroll
                for (int m = 0; m < NUC; m++)
                {
                    float t = us[m] + y[m];
                    if (P.en_input_bound) t = fminf(uhi[m], fmaxf(ulo[m], t));
                    zn[m] = t;
                    y[m] = (y[m] + us[m]) - t;
                    pri_u = fmaxf(pri_u, fabsf(us[m] - t));
                    dua_u = fmaxf(dua_u, fabsf(z[m] - t));
                }
                stv<NUC>(P.znew + uo, zn, active);
                stv<NUC>(P.y + uo, y, active);
                stv<NUC>(P.u + uo, us, keep_xu);
#pragma unroll
                for (int k = 0; k < NXC; k++) xs[k] = xn[k];
            }
            else
            {
                // p_{N-1} = -(Xref^T Pinf)^T - rho (vnew - g)     (admm.cpp:83-84)
#pragma unroll
                for (int k = 0; k < NXC; k++) pN[k] = pterm[k] - rho * (vn[k] - g[k]);
                stv<NXC>(P.p + xo, pN, active);
            }
        }

        // ---------------- termination_condition (admm.cpp:91-109) ----------------
        pri_x = inst_max(pri_x); dua_x = inst_max(dua_x);
        pri_u = inst_max(pri_u); dua_u = inst_max(dua_u);
        bool conv = false;
        if (active)
        {
            itn = it + 1; // admm.cpp:120
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv) st = TINY_STATUS_SOLVED_; // admm.cpp:136
        }
        active = active && !conv;
        if (!__any(active)) break;

        // ---------------- backward sweep ----------------
        {
            float p[NXC];
#pragma unroll
            for (int k = 0; k < NXC; k++) p[k] = pN[k];
            // v.col(N-1) = vnew.col(N-1)   (admm.cpp:141)
            {
                const size_t xo = xbase + (size_t)(N - 1) * xstep;
                float vn[NXC];
                ldv<NXC>(P.vnew + xo, vn);
                stv<NXC>(P.v + xo, vn, active);
            }
            for (int i = N - 2; i >= 0; i--)
            {
                const size_t xo = xbase + (size_t)i * xstep, uo = ubase + (size_t)i * ustep;
                float g[NXC], vn[NXC], xr[NXC], y[NUC], zn[NUC], q[NXC], r[NUC], d[NUC];
                ldv<NXC>(P.g + xo, g);
                ldv<NXC>(P.vnew + xo, vn);
                ldv<NUC>(P.y + uo, y);
                ldv<NUC>(P.znew + uo, zn);
                load_xref(i, xr);
#pragma unroll
                for (int m = 0; m < NUC; m++) r[m] = -rho * (zn[m] - y[m]); // admm.cpp:80
#pragma unroll
                for (int k = 0; k < NXC; k++)
                {
                    float t = -(xr[k] * Qv[k]);       // admm.cpp:81
                    q[k] = t - rho * (vn[k] - g[k]);  // admm.cpp:82
                }
                riccati_step<D>(op, p, q, r, d);
                stv<NUC>(P.d + uo, d, active);
                stv<NXC>(P.p + xo, p, active);
                stv<NXC>(P.v + xo, vn, active); // admm.cpp:141
                stv<NUC>(P.z + uo, zn, active); // admm.cpp:142
            }
        }
    }

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && gq == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }

    // ---------------- outputs: x,u of the last executed iteration, r and q ----------------
    // For converged instances x and u are regenerated from the frozen d (the same instruction sequence as the
    // sweep that produced them => bit-identical) instead of being stored on every iteration.
    {
        // A cold start (reset_workspace() folded into this launch: the arrays still hold the previous solve) that converged in
        // its FIRST iteration ran no backward sweep, which is what writes p, d, v, z: for such an instance they are the zeros
        // the workspace was reset to — d reads as zero below and the four arrays are materialised here.
        const bool fresh = valid && (P.cold_start != 0) && st == TINY_STATUS_SOLVED_ && itn == 1;
        float zx[NXC], zu[NUC];
#pragma unroll
        for (int k = 0; k < NXC; k++) zx[k] = 0.f;
#pragma unroll
        for (int m = 0; m < NUC; m++) zu[m] = 0.f;
        float xs[NXC];
#pragma unroll
        for (int k = 0; k < NXC; k++) xs[k] = x0[k];
        for (int i = 0; i < N; i++)
        {
            const size_t xo = xbase + (size_t)i * xstep;
            float g[NXC], vn[NXC], xr[NXC], q[NXC];
            stv<NXC>(P.v + xo, zx, fresh);
            if (i < N - 1) stv<NXC>(P.p + xo, zx, fresh);
            ldv<NXC>(P.g + xo, g);
            ldv<NXC>(P.vnew + xo, vn);
            load_xref(i, xr);
#pragma unroll
            for (int k = 0; k < NXC; k++)
            {
                float t = -(xr[k] * Qv[k]);
                q[k] = t - rho * (vn[k] - g[k]);
            }
            stv<NXC>(P.q + xo, q, valid);
            stv<NXC>(P.x + xo, xs, valid && st == TINY_STATUS_SOLVED_);
            if (i < N - 1)
            {
                const size_t uo = ubase + (size_t)i * ustep;
                float dd[NUC], us[NUC], xn[NXC], y[NUC], zn[NUC], r[NUC];
                ldv<NUC>(P.d + uo, dd);
#pragma unroll
                for (int m = 0; m < NUC; m++) dd[m] = fresh ? 0.f : dd[m];
                stv<NUC>(P.d + uo, zu, fresh);
                stv<NUC>(P.z + uo, zu, fresh);
                ldv<NUC>(P.y + uo, y);
                ldv<NUC>(P.znew + uo, zn);
                lqr_step<D>(op, xs, dd, us, xn);
#pragma unroll
                for (int m = 0; m < NUC; m++) r[m] = -rho * (zn[m] - y[m]);
                stv<NUC>(P.u + uo, us, valid && st == TINY_STATUS_SOLVED_);
                stv<NUC>(P.r + uo, r, valid);
#pragma unroll
                for (int k = 0; k < NXC; k++) xs[k] = xn[k];
            }
        }
    }
    if (valid && gq == 0)
    {
        P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
        P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
        P.status[inst] = st;
        P.iter[inst] = itn;
        if (st != TINY_STATUS_SOLVED_) atomicAdd(P.n_unsolved, 1);
    }
}

hipError_t launch_admm_stream(int nxc, int nuc, const SolveParams &P, hipStream_t stream)
{
#define TINY_DISPATCH_STREAM(NXC, NUC)                                                         \
    if (nxc == NXC && nuc == NUC)                                                              \
    {                                                                                          \
        hipLaunchKernelGGL((admm_stream_kernel<NXC, NUC>), dim3(P.ntiles), dim3(WAVE), 0, stream, P); \
        return hipGetLastError();                                                              \
    }
    TINY_FOR_EACH_DIMS(TINY_DISPATCH_STREAM)
    return hipErrorInvalidValue;
}

} // namespace tinympc
