// admm_rowlane.hip — state-on-chip batched TinyMPC ADMM kernel for small problems (nx + nu <= 16).
//
// Restates tiny_solve() (src/tinympc/admm.cpp:111-152) with the WHOLE loop-carried state of an instance kept on chip for
// the entire solve (three VGPRs and two LDS words per horizon step and lane, see below): HBM is touched only to read the
// live-in arrays once and to write the live-out arrays once.
//
// Mapping ("row lanes"): a DPP row = 16 lanes = ONE instance, 4 instances per wavefront, one wavefront per workgroup.
// Lane r of the row owns row r of the stacked vector [x ; u]: r < NX -> x(r), NX <= r < NX+NU -> u(r-NX) (quadrotor:
// 12 + 4 = exactly 16).  A gain x state product y = M s is NX (or NU) instructions
//      t_k = M[r][k] * s[k]      with s[k] fetched by the DPP operand modifier row_newbcast:k,
// the gain row sits in the lane's own VGPRs, and the x rows (A, AmBKt) and the u rows (Kinf, Bdyn^T) of a sweep step
// execute in the same instructions.  (On gfx950 a DPP instruction issues in 4 cycles against 2 for a plain fp32
// mul/add/fma — tools/micro/*.hip — so the broadcast costs one issue slot, not zero.)  The horizon is fully unrolled
// so that the per-step state indexes registers statically.
//
// Two arithmetic modes (template parameter EXACT), shared with the other row kernels through rowlane_math.h:
//   EXACT = true : every product and every sum is a separately rounded fp32 operation, summed in the order the
//                  reference's SSE2 Eigen build uses (RowPlans).  Results are BITWISE identical to the compiled
//                  reference, iteration counts included.
//   EXACT = false: one v_fmac_f32_dpp per multiply-add (k-ascending fma chain), about half the instructions.
//
// Converged instances are frozen by running the iteration body under `if (active)`: the 16 lanes of an instance leave
// EXEC together, so row broadcasts and row reductions of the remaining instances only ever read active lanes.
#include "rowlane_math.h"

namespace tinympc
{

// MPC = true: the closed-loop variant (P.mpc_steps MPC steps inside one launch); a separate instantiation so that the
// ordinary solve keeps its register allocation
// BPI = true: box bounds per instance (every reference workspace owns its u_min .. x_max, types.hpp:88-91): the {lo, hi}
// pair of a lane-step comes from the [B][N][16] table in global memory, BPI_AHEAD steps ahead of its use (8 bytes per lane
// and step; the 3.8 KB of an instance are re-read every iteration and stay in L2 / the Infinity Cache), instead of from the
// one table the batch shares in LDS
#ifndef TINY_BPI_AHEAD
#define TINY_BPI_AHEAD 4
#endif
constexpr int BPI_AHEAD = TINY_BPI_AHEAD;
// OPT = true (round 4): the two terms the reference ships commented out (admm.cpp:20 "+ coeff_d2p * d.col(i)", :79 Uref; off by default,
// tiny_batch_set_optional_terms) on this register-resident kernel — until now a handle that enabled one was routed to the kernel that streams
// its state.  The input rows of the cost register c hold d, so -(Uref .* R) is not kept: an input lane reads its Uref element of step i
// OPT_AHEAD steps ahead from the [B or 1][N][16] array (4 bytes per lane-step, L2 resident) where the linear cost is formed; coeff_d2p * d_i is
// one more DPP product group behind the Riccati step (rowlane_math.h d2p_term: added to the STORED p_i, in sequential order, as Eigen does).
// fp32 storage, batch-shared bounds, one solve per launch.
constexpr int OPT_AHEAD = 4;
template <int NX, int NU, int N, bool EXACT, bool H16, bool MPC = false, bool BPI = false, bool D32 = false, bool OPT = false>
__global__ __launch_bounds__(WAVE, (N > 32 && EXACT && !H16) ? 1 : 2) void admm_rowlane_kernel(const RowParams P)
{
    static_assert(!OPT || (!H16 && !MPC && !BPI && !D32), "the optional terms are instantiated for fp32 storage, shared bounds, one solve per launch");
    constexpr bool HD = H16 && !D32; // storage precision of the duals (gy)
    const int lane = threadIdx.x;
    const int r16 = lane & 15;
    // dispatch order: workgroups start in index order, so order[] decides which instance groups start first
    const int grp = P.order ? P.order[blockIdx.x] : (int)blockIdx.x;
    const int inst = grp * 4 + (lane >> 4);
    // also rejects a negative or too large entry of a caller-supplied order: such a row stores nothing, and its live-in
    // loads below address the last instance instead (inst_a), so a bad permutation cannot read out of bounds
    const bool valid = (unsigned)inst < (unsigned)P.batch;
    const int inst_a = valid ? inst : P.batch - 1;
    const bool is_x = r16 < NX;
    const bool is_u = (r16 >= NX) && (r16 < NX + NU);
    const float rho = P.rho;

    // ---- box bounds of the whole horizon, shared by the batch: LDS table [N][16] of {lo, hi} ----
    __shared__ float2 bnd[BPI ? 1 : N * 16];
    // vnew/znew of the current sweep (sn) and the previous slack v/z (b) live in LDS (lane-linear => conflict free):
    // each is written once and read once per iteration, so they do not need a VGPR per horizon step.
    __shared__ float sn_lds[N * WAVE];
    __shared__ float b_lds[N * WAVE];
    if constexpr (!BPI)
    {
        for (int e = lane; e < N * 16; e += WAVE) bnd[e] = ld_bounds<H16>(P.bounds, e);
        __syncthreads();
    }
    const int bbase = BPI ? inst_a * (int)P.bounds_inst_stride + r16 : 0; // {lo,hi} entry of step i: bbase + i*16
    float *sn = sn_lds + lane;   // sn[i * WAVE]
    float *b = b_lds + lane;     // b[i * WAVE]

    // ---- gain rows of this lane -----------------------------------------------------------------
    RowGains<NX, NU> G;
    G.load(P.mats, r16);
    // optional terms (OPT): R(r) on the input rows, coeff_d2p(r, m) on the state rows (pack_gains), this lane's Uref column
    [[maybe_unused]] float rrow = 0.f, CD[NU];
    [[maybe_unused]] const bool uref_on = OPT && P.uref != nullptr, d2p_on = OPT && P.en_d2p != 0;
    [[maybe_unused]] const int uref_off = inst_a * (int)P.uref_inst_stride + r16;
    if constexpr (OPT)
    {
        rrow = P.mats[(3 * NX + 2 * NU + 1) * 16 + r16];
#pragma unroll
        for (int m = 0; m < NU; m++) CD[m] = P.mats[(3 * NX + 2 * NU + 2 + m) * 16 + r16];
    }
    // cost term of step i as lin_cost() wants it: -(Xref_i .* Q) on the state rows; on the input rows -0 (so that r = -rho (znew - y) keeps the
    // sign of a zero difference) or, with the Uref term on, -(Uref_i .* R)   (admm.cpp:79-82)
    auto cost_of_step = [&](float ci, [[maybe_unused]] float ur) -> float {
        float cq = cost_term(ci, is_x);
        if constexpr (OPT)
            if (uref_on) cq = is_u ? -(ur * rrow) : cq;
        return cq;
    };

    // ---- per-instance state ------------------------------------------------------------------------------
    //   a[i]  : g_i (x rows) | y_i (u rows)           duals                                (VGPR)
    //   c[i]  : -(Xref_i.*Q) | d_i                    reference cost term | feed-forward   (VGPR)
    //   b[i]  : v_i          | z_i                    previous slack                       (LDS)
    //   sn[i] : vnew_i       | znew_i                 current slack                        (LDS)
    //   pd[i] : p_i          | d_i of the last executed backward sweep (live-out only; live-in until then)   (VGPR)
    float a[N], c[N], pd[N];
    // 32-bit element offset: lets the compiler address every array as SGPR base + one shared VGPR offset instead of
    // keeping a 64-bit address pair per array alive through the loop (the host checks batch*N*16 < 2^30)
    const int rowbase = (inst_a * N) * 16 + r16;
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
    const int xref_off = inst_a * (int)P.xref_inst_stride + r16;
    const bool cold = P.cold_start != 0;
    const bool zdual = cold || (P.duals_zero != 0);
    float xrN = 0.f; // Xref_{N-1}(r)
    {
        const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16]; // Q(r) on x rows, 0 elsewhere
#pragma unroll
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
            pd[i] = cold ? 0.f : ldw<H16>(P.pd, rowbase + i * 16); // live-in [p_i ; d_i]: kept by an instance that runs no backward sweep
            c[i] = is_x ? rnd<H16>(-(xr * qrow)) : pd[i];    // admm.cpp:81  q(i,j) = -(Xref(i,j) * Q(i))
            b[i * WAVE] = cold ? 0.f : ldw<H16>(P.vz, rowbase + i * 16);
            a[i] = zdual ? 0.f : ldw<HD>(P.gy, rowbase + i * 16);
            sn[i * WAVE] = 0.f;
            if (i == N - 1) xrN = xr;
        }
    }
    float x0 = ldw<H16>(P.xu, rowbase); // x.col(0) on x rows (u rows hold stale u_0, never used as x)

    // -(Xref_{N-1}^T Pinf): constant during a solve (admm.cpp:83)
    float pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xrN);

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start) // reset_workspace() zeroes the residual fields too
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN = 0.f; // p_{N-1} of the last executed forward sweep (x rows)
    bool ran_bwd = false;

    auto lqr = [&](float s_, float ci, float &sv, float &xn) { lqr_step<NX, NU, EXACT, H16>(G, is_x, is_u, s_, ci, sv, xn); };

    // Closed loop on chip: P.mpc_steps > 1 repeats {solve; advance} with the whole state staying in registers/LDS.
    // An ordinary solve is mpc_steps == 1.
    for (int ms = 0;; ++ms)
    {
    bool active = valid && (P.max_iter > 0);
    st = TINY_STATUS_UNSOLVED_; itn = 1;
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        // the last permitted iteration must not overwrite d in registers: x,u of an instance that exhausts max_iter come
        // from the d its last forward sweep used (regenerated in the epilogue); the final d itself is kept in pd[]
        const bool keep_d = (it == P.max_iter - 1);
        if (active)
        {
            // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
            float s = x0, pri = 0.f, dua = 0.f;
            float2 lh, lhq[BPI_AHEAD];
            if constexpr (BPI)
            {
#pragma unroll
                for (int k = 0; k < BPI_AHEAD; k++) lhq[k] = ld_bounds<H16>(P.bounds, bbase + (k < N ? k : N - 1) * 16);
                lh = lhq[0];
            }
            else lh = bnd[r16];
            float b_pref = b[0];
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                float2 lh_next;
                if constexpr (BPI)
                {
                    if (i + BPI_AHEAD < N) lhq[i % BPI_AHEAD] = ld_bounds<H16>(P.bounds, bbase + (i + BPI_AHEAD) * 16); // its slot was consumed by step i
                    lh_next = lhq[(i + 1) % BPI_AHEAD];
                }
                else lh_next = bnd[(i + 1 < N ? i + 1 : i) * 16 + r16]; // LDS reads one step ahead
                const float b_next = b[(i + 1 < N ? i + 1 : i) * WAVE];
                float sv, xn = 0.f;
                if (i < N - 1) lqr(s, c[i], sv, xn);
                else sv = is_x ? s : 0.f;
                float t = rnd<H16>(sv + a[i]);             // admm.cpp:47-48
                // admm.cpp:51-60: min(hi, max(lo, t)).  The host stores lo := min(lo, hi), which makes the median
                // identical to that expression for every t (and +-inf where a bound is disabled).
                t = __builtin_amdgcn_fmed3f(t, lh.x, lh.y);
                a[i] = rnd<HD>((a[i] + sv) - t);          // admm.cpp:69-70
                pri = fmaxf(pri, fabsf(sv - t));           // admm.cpp:95,97
                dua = fmaxf(dua, fabsf(b_pref - t));       // admm.cpp:96,98
                sn[i * WAVE] = t;
                s = xn;
                lh = lh_next;
                b_pref = b_next;
            }
            {
                const float t1 = sn[(N - 1) * WAVE] - a[N - 1];
                pN = lin_cost<EXACT, H16>(pterm, rho, t1); // admm.cpp:83-84
            }
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
                // ---------------- backward sweep: v=vnew, z=znew, linear cost, backward_pass_grad ----------------
                float p = pN;
                b[(N - 1) * WAVE] = sn[(N - 1) * WAVE];
                ran_bwd = true;
                const bool upd_d = is_u && !keep_d;
                float sn_pref = sn[(N - 2) * WAVE];
                [[maybe_unused]] float ur_ring[OPT_AHEAD];
                if constexpr (OPT)
                    if (uref_on)
                    {
#pragma unroll
                        for (int k = 0; k < OPT_AHEAD; k++) ur_ring[k] = P.uref[uref_off + (N - 2 - k > 0 ? N - 2 - k : 0) * 16];
                    }
#pragma unroll
                for (int i = N - 2; i >= 0; i--)
                {
                    const float sni = sn_pref;
                    sn_pref = sn[(i > 0 ? i - 1 : 0) * WAVE]; // LDS read one step ahead
                    const float t1 = sni - a[i];
                    float ur = 0.f;
                    if constexpr (OPT)
                        if (uref_on)
                        {
                            ur = ur_ring[(N - 2 - i) % OPT_AHEAD];
                            if (i - OPT_AHEAD >= 0) ur_ring[(N - 2 - i) % OPT_AHEAD] = P.uref[uref_off + (i - OPT_AHEAD) * 16]; // its slot was consumed just now
                        }
                    const float cq = cost_of_step(c[i], ur); // x rows: -(Xref.*Q) ; u rows: -0 (or -(Uref.*R))
                    float pn, dd;
                    riccati_step<NX, NU, EXACT, H16>(G, is_x, p, lin_cost<EXACT, H16>(cq, rho, t1), pn, dd); // admm.cpp:19-20,80-82
                    if constexpr (OPT)
                        if (d2p_on) pn = d2p_term<NX, NU, EXACT, H16>(CD, pn, dd); // "+ coeff_d2p * d.col(i)" (admm.cpp:20)
                    c[i] = upd_d ? dd : c[i];
                    b[i * WAVE] = sni;                     // admm.cpp:141-142
                    pd[i] = is_u ? dd : pn;                // [p_i ; d_i] of this sweep (live-out only: stays in registers)
                    p = pn;
                }
            }
        }
    }

    if (!MPC || ms + 1 >= P.mpc_steps) break;
    // ---------------- advance to the next MPC step (quadrotor_tracking.cpp:101-118), nothing leaves the chip ----------------
    {
        float sv0, x1;
        lqr(x0, c[0], sv0, x1); // [x_0 ; u_0] of the solve that just finished, in the solver's own arithmetic
        if (P.u0_traj && valid && is_u) P.u0_traj[((long long)ms * P.batch + inst) * NU + (r16 - NX)] = sv0;
        if constexpr (MPC) x0 = plant_step<NX, NU>(G, sv0); // x_1 = Adyn x0 + Bdyn u_0 (:110), the plant kernel's arithmetic
        wstart += P.window_advance;
        const float qrow = P.mats[(2 * NX + 2 * NU) * 16 + r16];
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            float cqn = c[i];
            if (P.xref_mode == 1)
            {
                int row = wstart + i;
                row = row < P.table_rows ? row : P.table_rows - 1;
                const float xr = ldw<H16>(P.xref_table, row * 16 + r16);
                cqn = rnd<H16>(-(xr * qrow));
                if (i == N - 1) xrN = xr;
            }
            c[i] = is_x ? cqn : pd[i]; // d of the workspace = d of the last executed backward sweep
            a[i] = 0.f;                // y = g = 0 (:106-107)
        }
        if (P.xref_mode == 1) pterm = terminal_term<NX, NU, EXACT, H16>(P.mats, r16, xrN);
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

    // ---------------- live-out: every work array written once ----------------
    if (valid)
    {
        const bool solved = (st == TINY_STATUS_SOLVED_);
        float s = x0;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            const int o = rowbase + i * 16;
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            float sv, xn = 0.f;
            if (i < N - 1) lqr(s, c[i], sv, xn);
            else sv = is_x ? s : 0.f;
            stw<H16>(P.xu, o, sv);
            s = xn;
            const float sni = sn[i * WAVE];
            const float t1 = sni - a[i];
            float ur = 0.f;
            if constexpr (OPT)
                if (uref_on) ur = P.uref[uref_off + i * 16];
            const float cq = cost_of_step(c[i], ur);
            const float lin = lin_cost<EXACT, H16>(cq, rho, t1);
            stw<H16>(P.qr, o, (i < N - 1 || is_x) ? lin : 0.f);
            // p.col(N-1) is rewritten by every forward sweep (admm.cpp:83-84); the other columns and d were stored by the
            // last backward sweep this instance executed.  An instance that never ran one keeps its live-in p,d
            // (all zero after reset_workspace, which only marked them so).
            stw<H16>(P.pd, o, i == N - 1 ? (is_x ? pN : 0.f) : pd[i]);
            stw<H16>(P.vz, o, b[i * WAVE]);
            stw<H16>(P.vzn, o, sni);
            stw<HD>(P.gy, o, a[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MPC) // the host's plant step continues from here
        {
            if (is_x) P.x0buf[inst * NX + r16] = x0;
            if (r16 == 0 && P.xref_mode == 1) P.xref_start[inst] = wstart;
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

bool rowlane_supported(int nx, int nu, int N)
{
#define TINY_ROWLANE_CHECK(NX, NU, NN) \
    if (nx == NX && nu == NU && N == NN) return true;
    TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_CHECK)
    return false;
}

hipError_t launch_admm_rowlane(int nx, int nu, int N, bool exact, bool h16, const RowParams &P, hipStream_t stream)
{
    const int nblocks = (P.batch + 3) / 4;
    if (P.mpc_steps > 1) // closed loop on chip: fp32 storage and batch-shared bounds only
    {
        if (h16 || P.dual32 || P.bounds_inst_stride != 0) return hipErrorInvalidValue;
#define TINY_ROWLANE_MPC_DISPATCH(NX, NU, NN)                                                               \
    if (nx == NX && nu == NU && N == NN)                                                                    \
    {                                                                                                       \
        if (exact) hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, true, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);     \
        return hipGetLastError();                                                                           \
    }
        TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_MPC_DISPATCH)
        return hipErrorInvalidValue;
    }
    if (P.bounds_inst_stride != 0) // per-instance bounds: fp32 storage only (fp16 storage with them runs on the streaming row kernel)
    {
        if (h16 || P.dual32) return hipErrorInvalidValue;
#define TINY_ROWLANE_BPI_DISPATCH(NX, NU, NN)                                                                            \
    if (nx == NX && nu == NU && N == NN)                                                                                 \
    {                                                                                                                    \
        if (exact) hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, true, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, false, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);      \
        return hipGetLastError();                                                                                        \
    }
        TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_BPI_DISPATCH)
        return hipErrorInvalidValue;
    }
    if (P.uref != nullptr || P.en_d2p) // the optional terms (round 4): fp32 storage, batch-shared bounds
    {
        if (h16 || P.dual32) return hipErrorInvalidValue;
#define TINY_ROWLANE_OPT_DISPATCH(NX, NU, NN)                                                                            \
    if (nx == NX && nu == NU && N == NN)                                                                                 \
    {                                                                                                                    \
        if (exact) hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, true, false, false, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, false, false, false, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);      \
        return hipGetLastError();                                                                                        \
    }
        TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_OPT_DISPATCH)
        return hipErrorInvalidValue;
    }
    if (P.dual32) // fp16 storage with fp32 duals
    {
        if (!h16) return hipErrorInvalidValue;
#define TINY_ROWLANE_D32_DISPATCH(NX, NU, NN)                                                                            \
    if (nx == NX && nu == NU && N == NN)                                                                                 \
    {                                                                                                                    \
        if (exact) hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, true, true, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, false, true, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);      \
        return hipGetLastError();                                                                                        \
    }
        TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_D32_DISPATCH)
        return hipErrorInvalidValue;
    }
#define TINY_ROWLANE_LAUNCH(NX, NU, NN, EX, H) \
    hipLaunchKernelGGL((admm_rowlane_kernel<NX, NU, NN, EX, H>), dim3(nblocks), dim3(WAVE), 0, stream, P)
#define TINY_ROWLANE_DISPATCH(NX, NU, NN)                                                                   \
    if (nx == NX && nu == NU && N == NN)                                                                    \
    {                                                                                                       \
        if (exact && !h16) TINY_ROWLANE_LAUNCH(NX, NU, NN, true, false);                                    \
        else if (exact) TINY_ROWLANE_LAUNCH(NX, NU, NN, true, true);                                        \
        else if (!h16) TINY_ROWLANE_LAUNCH(NX, NU, NN, false, false);                                       \
        else TINY_ROWLANE_LAUNCH(NX, NU, NN, false, true);                                                  \
        return hipGetLastError();                                                                           \
    }
    TINY_FOR_EACH_ROWLANE(TINY_ROWLANE_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
