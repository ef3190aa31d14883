// admm_steps.hip — batched twins of the six step functions the reference exports next to tiny_solve()
// (src/tinympc/admm.hpp:10-18): forward_pass, update_slack, update_dual, update_linear_cost, termination_condition,
// backward_pass_grad.  Each is one launch over the row layout (one DPP row of 16 lanes = one instance, any horizon N,
// nx + nu <= 16) that reads and writes the device-resident workspace exactly as the reference function reads and writes
// TinyWorkspace.  They share the arithmetic of the fused solver (rowlane_math.h), so in exact mode each function is
// bit-identical to the reference's.  The fused kernels do not call these; they exist for callers that drive the
// algorithm step by step (e.g. only roll out a trajectory with forward_pass).
#include "rowlane_math.h"

namespace tinympc
{

// gain rows of the optional terms (pack_gains): R on the u rows, then coeff_d2p(r, m) on the x rows
template <int NX, int NU>
__device__ __forceinline__ float input_cost_row(const float *mats, int r16) { return mats[(3 * NX + 2 * NU + 1) * 16 + r16]; }
template <int NX, int NU>
__device__ __forceinline__ void load_d2p(float (&CD)[NU], const float *mats, int r16)
{
#pragma unroll
    for (int m = 0; m < NU; m++) CD[m] = mats[(3 * NX + 2 * NU + 2 + m) * 16 + r16];
}

// c term of a step (what lin_cost() subtracts rho*(snew - dual) from): x rows -(Xref .* Q) (admm.cpp:81); u rows -0, so that
// r = -rho*(znew - y) keeps the sign of a zero difference, or -(Uref .* R) when the optional input reference is on
template <bool H16>
__device__ __forceinline__ float cost_of(const RowParams &P, bool is_x, bool is_u, float xr, float qrow, float rrow, int uoff)
{
    if (is_x) return rnd<H16>(-(xr * qrow));
    if (P.uref != nullptr && is_u) return rnd<H16>(-(ldw<H16>(P.uref, uoff) * rrow));
    return -0.f;
}

template <int NX, int NU, bool EXACT, bool H16>
__global__ __launch_bounds__(WAVE) void admm_step_kernel(const RowParams P, const int fn, int *__restrict__ conv_out)
{
    const int lane = threadIdx.x, r16 = lane & 15;
    const int inst = blockIdx.x * 4 + (lane >> 4);
    const bool valid = inst < P.batch;
    const bool is_x = r16 < NX, is_u = (r16 >= NX) && (r16 < NX + NU);
    const int N = P.N;
    const int rowbase = (inst * N) * 16 + r16;
    const float rho = P.rho;

    if (fn == STEP_FORWARD_PASS) // admm.cpp:27-37  reads x.col(0), d; writes u, x.col(1..N-1)
    {
        RowGains<NX, NU> G;
        G.load(P.mats, r16);
        float s = ldw<H16>(P.xu, rowbase);
        for (int i = 0; i < N - 1; i++)
        {
            float sv, xn;
            lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, ldw<H16>(P.pd, rowbase + i * 16), sv, xn);
            if (valid) stw<H16>(P.xu, rowbase + i * 16, sv);
            s = xn;
        }
        if (valid) stw<H16>(P.xu, rowbase + (N - 1) * 16, is_x ? s : 0.f);
    }
    else if (fn == STEP_UPDATE_SLACK) // admm.cpp:45-61  znew = clip(u + y), vnew = clip(x + g)
    {
        for (int i = 0; i < N; i++)
        {
            const float2 lh = ld_bounds<H16>(P.bounds, inst * (int)P.bounds_inst_stride + i * 16 + r16);
            const float t = ldw<H16>(P.xu, rowbase + i * 16) + ldw<H16>(P.gy, rowbase + i * 16);
            if (valid) stw<H16>(P.vzn, rowbase + i * 16, __builtin_amdgcn_fmed3f(t, lh.x, lh.y));
        }
    }
    else if (fn == STEP_UPDATE_DUAL) // admm.cpp:67-71  y += u - znew, g += x - vnew
    {
        for (int i = 0; i < N; i++)
        {
            const float a = ldw<H16>(P.gy, rowbase + i * 16);
            if (valid) stw<H16>(P.gy, rowbase + i * 16, (a + ldw<H16>(P.xu, rowbase + i * 16)) - ldw<H16>(P.vzn, rowbase + i * 16));
        }
    }
    else if (fn == STEP_UPDATE_LINEAR_COST) // admm.cpp:77-85  r, q, p.col(N-1)
    {
        const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16], rrow = input_cost_row<NX, NU>(P.mats, r16);
        int wstart = 0;
        if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
        const int xref_off = inst * (int)P.xref_inst_stride + r16, uref_off = inst * (int)P.uref_inst_stride + r16;
        float xr = 0.f, t1 = 0.f;
        for (int i = 0; i < N; i++)
        {
            if (P.xref_mode == 1)
            {
                int row = wstart + i;
                row = row < P.table_rows ? row : P.table_rows - 1;
                xr = ldw<H16>(P.xref_table, row * 16 + r16);
            }
            else
                xr = ldw<H16>(P.xref, xref_off + i * 16);
            const float cq = cost_of<H16>(P, is_x, is_u, xr, qrow, rrow, uref_off + i * 16);
            t1 = ldw<H16>(P.vzn, rowbase + i * 16) - ldw<H16>(P.gy, rowbase + i * 16);
            const float lin = lin_cost<EXACT, H16>(cq, rho, t1);
            if (valid) stw<H16>(P.qr, rowbase + i * 16, (i < N - 1 || is_x) ? lin : 0.f);
        }
        const float pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xr); // xr, t1 are those of step N-1 here
        if (valid && is_x) stw<H16>(P.pd, rowbase + (N - 1) * 16, lin_cost<EXACT, H16>(pterm, rho, t1));
    }
    else if (fn == STEP_TERMINATION_CONDITION) // admm.cpp:91-109  residual fields + the boolean it returns
    {
        bool conv = false;
        const int itn = valid ? P.iter[inst] : 1;
        if (itn % P.check_termination == 0) // row-uniform
        {
            float pri = 0.f, dua = 0.f;
            for (int i = 0; i < N; i++)
            {
                const float sv = ldw<H16>(P.xu, rowbase + i * 16), t = ldw<H16>(P.vzn, rowbase + i * 16);
                pri = fmaxf(pri, fabsf(sv - t));
                dua = fmaxf(dua, fabsf(ldw<H16>(P.vz, rowbase + i * 16) - t));
            }
            const float r_ps = row_max(is_x ? pri : 0.f), r_ds = row_max(is_x ? dua : 0.f) * rho;
            const float r_pi = row_max(is_u ? pri : 0.f), r_di = row_max(is_u ? dua : 0.f) * rho;
            conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            if (valid && r16 == 0)
            {
                P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
                P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            }
        }
        if (valid && r16 == 0)
        {
            conv_out[inst] = conv ? 1 : 0;
            if (!conv) atomicAdd(P.n_unsolved, 1);
        }
    }
    else if (fn == STEP_BACKWARD_PASS_GRAD) // admm.cpp:15-22  reads p.col(N-1), q, r; writes d, p.col(0..N-2)
    {
        RowGains<NX, NU> G;
        G.load(P.mats, r16);
        float CD[NU];
        load_d2p<NX, NU>(CD, P.mats, r16);
        float p = ldw<H16>(P.pd, rowbase + (N - 1) * 16);
        for (int i = N - 2; i >= 0; i--)
        {
            float pn, dd;
            riccati_step<NX, NU, EXACT, H16>(G, is_x, p, ldw<H16>(P.qr, rowbase + i * 16), pn, dd);
            if (P.en_d2p) pn = d2p_term<NX, NU, EXACT, H16>(CD, pn, dd);
            if (valid) stw<H16>(P.pd, rowbase + i * 16, is_u ? dd : pn);
            p = pn;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// admm_rowstream_kernel — the fused solve (tiny_solve, admm.cpp:111-152) in the row mapping for ANY horizon N: same
// arithmetic and same per-instance early exit as admm_rowlane_kernel, but the loop-carried state lives in the workspace
// arrays (streamed through L2/HBM every iteration) instead of registers, so nothing is unrolled over N.  It is the
// fallback for (nx, nu) classes whose N has no register-resident instantiation, and keeps exact mode bit-identical
// to the reference there too.
// ---------------------------------------------------------------------------------------------------------------------
template <int NX, int NU, bool EXACT, bool H16>
__global__ __launch_bounds__(WAVE) void admm_rowstream_kernel(const RowParams P)
{
    const int lane = threadIdx.x, r16 = lane & 15;
    const int inst = blockIdx.x * 4 + (lane >> 4);
    const bool valid = inst < P.batch;
    const bool is_x = r16 < NX, is_u = (r16 >= NX) && (r16 < NX + NU);
    const int N = P.N;
    const int rowbase = (inst * N) * 16 + r16;
    const float rho = P.rho;
    RowGains<NX, NU> G;
    G.load(P.mats, r16);
    const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16], rrow = input_cost_row<NX, NU>(P.mats, r16);
    float CD[NU];
    load_d2p<NX, NU>(CD, P.mats, r16);
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
    const int xref_off = inst * (int)P.xref_inst_stride + r16, uref_off = inst * (int)P.uref_inst_stride + r16;
    auto xref_at = [&](int i) {
        if (P.xref_mode == 1)
        {
            int row = wstart + i;
            row = row < P.table_rows ? row : P.table_rows - 1;
            return ldw<H16>(P.xref_table, row * 16 + r16);
        }
        return ldw<H16>(P.xref, xref_off + i * 16);
    };
    const float x0 = ldw<H16>(P.xu, rowbase);
    const float pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xref_at(N - 1));
    int st = TINY_STATUS_UNSOLVED_, itn = 1;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start) // reset_workspace() zeroes the residual fields too
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    bool active = valid && (P.max_iter > 0);
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        const bool last_iter = (it == P.max_iter - 1);
        const bool zero_state = (it == 0) && (P.cold_start != 0);
        const bool zero_duals = (it == 0) && ((P.cold_start | P.duals_zero) != 0);
        if (active)
        {
            float s = x0, pri = 0.f, dua = 0.f, t1 = 0.f;
            for (int i = 0; i < N; i++)
            {
                const int o = rowbase + i * 16;
                float sv, xn = 0.f;
                if (i < N - 1) lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, zero_state ? 0.f : ldw<H16>(P.pd, o), sv, xn);
                else sv = is_x ? s : 0.f;
                const float2 lh = ld_bounds<H16>(P.bounds, inst * (int)P.bounds_inst_stride + i * 16 + r16);
                const float a = zero_duals ? 0.f : ldw<H16>(P.gy, o);
                const float bprev = zero_state ? 0.f : ldw<H16>(P.vz, o);
                const float t = __builtin_amdgcn_fmed3f(rnd<H16>(sv + a), lh.x, lh.y);
                const float an = rnd<H16>((a + sv) - t);
                pri = fmaxf(pri, fabsf(sv - t));
                dua = fmaxf(dua, fabsf(bprev - t));
                stw<H16>(P.vzn, o, t);
                stw<H16>(P.gy, o, an);
                if (last_iter) stw<H16>(P.xu, o, sv);
                t1 = t - an;
                s = xn;
            }
            const float pN = lin_cost<EXACT, H16>(pterm, rho, t1);
            stw<H16>(P.pd, rowbase + (N - 1) * 16, is_x ? pN : 0.f);
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
                float p = pN;
                stw<H16>(P.vz, rowbase + (N - 1) * 16, ldw<H16>(P.vzn, rowbase + (N - 1) * 16));
                for (int i = N - 2; i >= 0; i--)
                {
                    const int o = rowbase + i * 16;
                    const float sni = ldw<H16>(P.vzn, o);
                    const float cq = cost_of<H16>(P, is_x, is_u, xref_at(i), qrow, rrow, uref_off + i * 16);
                    float pn, dd;
                    riccati_step<NX, NU, EXACT, H16>(G, is_x, p, lin_cost<EXACT, H16>(cq, rho, sni - ldw<H16>(P.gy, o)), pn, dd);
                    if (P.en_d2p) pn = d2p_term<NX, NU, EXACT, H16>(CD, pn, dd);
                    stw<H16>(P.pd, o, is_u ? dd : pn);
                    stw<H16>(P.vz, o, sni);
                    p = pn;
                }
            }
        }
    }
    if (P.max_iter <= 0)
    {
        if (valid && r16 == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    {
        // live-out: q, r (admm.cpp:80-82) and, for converged instances, x,u regenerated from the frozen d
        const bool solved = (st == TINY_STATUS_SOLVED_);
        // reset_workspace() folded into this launch (cold start): an instance that converged in its FIRST iteration ran no
        // backward sweep, which is what writes [p;d] and [v;z] — they are the zeros of the reset, materialised here
        const bool fresh = valid && (P.cold_start != 0) && solved && itn == 1;
        float s = x0;
        for (int i = 0; i < N; i++)
        {
            const int o = rowbase + i * 16;
            float sv, xn = 0.f;
            if (i < N - 1) lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s, fresh ? 0.f : ldw<H16>(P.pd, o), sv, xn);
            else sv = is_x ? s : 0.f;
            if (fresh)
            {
                if (i < N - 1) stw<H16>(P.pd, o, 0.f);
                stw<H16>(P.vz, o, 0.f);
            }
            if (valid && solved) stw<H16>(P.xu, o, sv);
            s = xn;
            const float cq = cost_of<H16>(P, is_x, is_u, xref_at(i), qrow, rrow, uref_off + i * 16);
            const float lin = lin_cost<EXACT, H16>(cq, rho, ldw<H16>(P.vzn, o) - ldw<H16>(P.gy, o));
            if (valid) stw<H16>(P.qr, o, (i < N - 1 || is_x) ? lin : 0.f);
        }
        if (valid && r16 == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

hipError_t launch_admm_rowstream(int nx, int nu, bool exact, bool h16, const RowParams &P, hipStream_t stream)
{
    const int nblocks = (P.batch + 3) / 4;
#define TINY_ROWSTREAM_LAUNCH(NX, NU, EX, H) \
    hipLaunchKernelGGL((admm_rowstream_kernel<NX, NU, EX, H>), dim3(nblocks), dim3(WAVE), 0, stream, P)
#define TINY_ROWSTREAM_DISPATCH(NX, NU)                                                                                \
    if (nx == NX && nu == NU)                                                                                          \
    {                                                                                                                  \
        if (exact && !h16) TINY_ROWSTREAM_LAUNCH(NX, NU, true, false);                                                 \
        else if (exact) TINY_ROWSTREAM_LAUNCH(NX, NU, true, true);                                                     \
        else if (!h16) TINY_ROWSTREAM_LAUNCH(NX, NU, false, false);                                                    \
        else TINY_ROWSTREAM_LAUNCH(NX, NU, false, true);                                                               \
        return hipGetLastError();                                                                                      \
    }
    TINY_FOR_EACH_ROWDIMS(TINY_ROWSTREAM_DISPATCH)
    return hipErrorInvalidValue;
}

bool rowdims_supported(int nx, int nu)
{
#define TINY_ROWDIMS_CHECK(NX, NU) \
    if (nx == NX && nu == NU) return true;
    TINY_FOR_EACH_ROWDIMS(TINY_ROWDIMS_CHECK)
    return false;
}

hipError_t launch_admm_step(int nx, int nu, bool exact, bool h16, int fn, const RowParams &P, int *conv_out, hipStream_t stream)
{
    const int nblocks = (P.batch + 3) / 4;
#define TINY_STEP_LAUNCH(NX, NU, EX, H) \
    hipLaunchKernelGGL((admm_step_kernel<NX, NU, EX, H>), dim3(nblocks), dim3(WAVE), 0, stream, P, fn, conv_out)
#define TINY_ROWDIMS_DISPATCH(NX, NU)                                                                                       \
    if (nx == NX && nu == NU)                                                                                               \
    {                                                                                                                       \
        if (exact && !h16) TINY_STEP_LAUNCH(NX, NU, true, false);                                                           \
        else if (exact) TINY_STEP_LAUNCH(NX, NU, true, true);                                                               \
        else if (!h16) TINY_STEP_LAUNCH(NX, NU, false, false);                                                              \
        else TINY_STEP_LAUNCH(NX, NU, false, true);                                                                         \
        return hipGetLastError();                                                                                           \
    }
    TINY_FOR_EACH_ROWDIMS(TINY_ROWDIMS_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
