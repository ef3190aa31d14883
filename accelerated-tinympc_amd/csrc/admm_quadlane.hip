// admm_quadlane.hip — state-on-chip batched TinyMPC ADMM kernel for nx = 4, nu = 1 (the cartpole class of
// examples/codegen_cartpole.cpp): FOUR LANES = ONE INSTANCE, 16 instances per wavefront.
//
// The 16-lanes-per-instance mapping of admm_rowlane.hip leaves 11 of 16 lanes idle for this class (5 rows of [x ; u]).
// Here lane j of a quad owns row j of x; the single input row is computed redundantly by all four lanes of the quad, so
// it needs no lane of its own.  State broadcasts use the DPP quad_perm modifier (v_mul_f32_dpp ... quad_perm:[k,k,k,k]).
// The whole loop-carried state sits in registers (8 words per horizon step and lane, horizon unrolled), nothing in LDS
// but the shared bounds table.  Same arithmetic modes and same storage option as the other row kernels (rowlane_math.h):
// exact arithmetic follows the reference's orders for these sizes — u = -(K x) - d through the packet reduction
// (s0+s2)+(s1+s3), A x and AmBKt p sequentially — and is bitwise identical to the compiled reference.
// Restates tiny_solve() (src/tinympc/admm.cpp:111-152); row layout in HBM as for the other row kernels.
#include "rowlane_math.h"

namespace tinympc
{

// two gain x state products of the same source in one block: tA[k] = MA[k]*s[k], tB[k] = MB[k]*s[k], s[k] = lane k of the quad
__device__ __forceinline__ void quad_products2(float (&tA)[4], float (&tB)[4], float s, const float (&MA)[4], const float (&MB)[4])
{
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %8, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %8, %10 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %8, %11 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %8, %12 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %4, %8, %13 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %5, %8, %14 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %6, %8, %15 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %7, %8, %16 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(tA[0]), "=&v"(tA[1]), "=&v"(tA[2]), "=&v"(tA[3]), "=&v"(tB[0]), "=&v"(tB[1]), "=&v"(tB[2]), "=&v"(tB[3])
        : "v"(s), "v"(MA[0]), "v"(MA[1]), "v"(MA[2]), "v"(MA[3]), "v"(MB[0]), "v"(MB[1]), "v"(MB[2]), "v"(MB[3]));
}
__device__ __forceinline__ void quad_products(float (&t)[4], float s, const float (&M)[4])
{
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %5 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %4, %6 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %4, %7 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %4, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
        : "v"(s), "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]));
}
// fma arithmetic: accA = sum_k MA[k]*s[k], accB = sum_k MB[k]*s[k] (k ascending chains)
__device__ __forceinline__ void quad_fma2(float &accA, float &accB, float s, const float (&MA)[4], const float (&MB)[4])
{
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %2, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %2, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %5 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %2, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %10 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(accA), "=&v"(accB)
        : "v"(s), "v"(MA[0]), "v"(MA[1]), "v"(MA[2]), "v"(MA[3]), "v"(MB[0]), "v"(MB[1]), "v"(MB[2]), "v"(MB[3]));
}
// max over the four lanes of a quad, every lane gets the result
__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v)); // quad_perm:[1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v)); // quad_perm:[2,3,0,1]
    return v;
}

// MPC = true: closed-loop variant (P.mpc_steps MPC steps in one launch, the state staying in registers; see admm_rowlane.hip)
template <int N, bool EXACT, bool H16, bool MPC = false, bool D32 = false>
__global__ __launch_bounds__(WAVE) void admm_quadlane_kernel(const RowParams P)
{
    constexpr int NX = 4;
    constexpr bool HD = H16 && !D32; // storage precision of the duals (gy)
    const int lane = threadIdx.x;
    const int j = lane & 3;
    const int inst = blockIdx.x * 16 + (lane >> 2);
    const bool valid = inst < P.batch;
    const bool lead = (j == 0); // the lane of a quad that stores the input-type row (row NX of the row layout)
    const float rho = P.rho;

    __shared__ float2 bnd[N * 16];
    for (int e = lane; e < N * 16; e += WAVE) bnd[e] = ld_bounds<H16>(P.bounds, e);
    __syncthreads();

    // gain registers of pack_gains() (tinympc_batch.hip): M1[k] = regs 0..3, M2[0] = 4, M3[k] = 5..8, M45[0] = 9, Q = 10, PT[k] = 11..14
    float Arow[4], Kneg[4], Am[4], Bcol[4], PT[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        Arow[k] = P.mats[k * 16 + j];        // Adyn(j, k)
        Kneg[k] = P.mats[k * 16 + NX];       // Kinf(0, k) (exact) / -Kinf(0, k) (fast)
        Am[k] = P.mats[(5 + k) * 16 + j];    // AmBKt(j, k)
        Bcol[k] = P.mats[(5 + k) * 16 + NX]; // Bdyn(k, 0)
        PT[k] = P.mats[(11 + k) * 16 + j];   // Pinf(k, j)
    }
    const float Bj = P.mats[4 * 16 + j];     // Bdyn(j, 0)
    const float Kj = P.mats[9 * 16 + j];     // Kinf(0, j) (exact) / -Kinf(0, j) (fast)
    const float Quu = P.mats[9 * 16 + NX];   // Quu_inv
    const float qrow = P.mats[10 * 16 + j];  // Q(j)

    // per-instance state, all in registers: duals a, reference cost term / feed-forward, previous and current slack
    float ax[N], ay[N], cq[N], dd[N], bx[N], bz[N], sx[N], sz[N];
    float pp[N], dl[N]; // [p_i ; d_i] of the last executed backward sweep (live-out only; the live-in values until then)
    // 16 instances per wave but the arrays are padded to a multiple of 4 instances only: the quads beyond the batch read
    // the last instance's rows (and store nothing)
    const int inst_a = valid ? inst : P.batch - 1;
    const int rowx = (inst_a * N) * 16 + j, rowu = (inst_a * N) * 16 + NX;
    int wstart = 0;
    if (P.xref_mode == 1 && valid) wstart = P.xref_start[inst];
    const int xref_off = inst_a * (int)P.xref_inst_stride + j;
    const bool cold = P.cold_start != 0;
    const bool zdual = cold || (P.duals_zero != 0);
    float xrN = 0.f;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
        float xr;
        if (P.xref_mode == 1)
        {
            int row = wstart + i;
            row = row < P.table_rows ? row : P.table_rows - 1;
            xr = ldw<H16>(P.xref_table, row * 16 + j);
        }
        else
            xr = ldw<H16>(P.xref, xref_off + i * 16);
        cq[i] = rnd<H16>(-(xr * qrow)); // admm.cpp:81
        dd[i] = (cold || i == N - 1) ? 0.f : ldw<H16>(P.pd, rowu + i * 16);
        dl[i] = dd[i];
        pp[i] = cold ? 0.f : ldw<H16>(P.pd, rowx + i * 16);
        bx[i] = cold ? 0.f : ldw<H16>(P.vz, rowx + i * 16);
        bz[i] = (cold || i == N - 1) ? 0.f : ldw<H16>(P.vz, rowu + i * 16);
        ax[i] = zdual ? 0.f : ldw<HD>(P.gy, rowx + i * 16);
        ay[i] = (zdual || i == N - 1) ? 0.f : ldw<HD>(P.gy, rowu + i * 16);
        sx[i] = 0.f; sz[i] = 0.f;
        if (i == N - 1) xrN = xr;
    }
    float x0 = ldw<H16>(P.xu, rowx);
    auto terminal = [&](float xr_last) { // -(Xref_{N-1}^T Pinf) (admm.cpp:83): packet reduction
        float t[4];
        quad_products(t, xr_last, PT);
        if constexpr (EXACT) return rnd<H16>(-((t[0] + t[2]) + (t[1] + t[3])));
        else return rnd<H16>(-(((t[0] + t[1]) + t[2]) + t[3]));
    };
    float pterm = terminal(xrN);

    // one forward_pass step (admm.cpp:31,35): from x_i (row j) and d_i -> u_i (all lanes) and x_{i+1} (row j)
    auto lqr = [&](float s, float di, float &un, float &xn) {
        if constexpr (EXACT)
        {
            float tK[4], tA[4];
            quad_products2(tK, tA, s, Kneg, Arow);
            un = rnd<H16>(-((tK[0] + tK[2]) + (tK[1] + tK[3])) - di); // -(K x) - d: the sum is negated, like the reference
            xn = rnd<H16>((((tA[0] + tA[1]) + tA[2]) + tA[3]) + Bj * un);
        }
        else
        {
            float aK, aA;
            quad_fma2(aK, aA, s, Kneg, Arow);
            un = rnd<H16>(aK - di);
            xn = rnd<H16>(__builtin_fmaf(Bj, un, aA));
        }
    };

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN = 0.f;
    bool ran_bwd = false;
    for (int ms = 0;; ++ms) // MPC steps of the closed-loop variant; an ordinary solve runs the body once
    {
    bool active = valid && (P.max_iter > 0);
    st = TINY_STATUS_UNSOLVED_; itn = 1;
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        const bool keep_d = (it == P.max_iter - 1); // see admm_rowlane.hip: x,u of an exhausted instance are regenerated from this d
        if (active)
        {
            // ---------------- forward sweep ----------------
            float s = x0, prx = 0.f, dux = 0.f, pru = 0.f, duu = 0.f;
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                float un = 0.f, xn = 0.f;
                if (i < N - 1)
                {
                    lqr(s, dd[i], un, xn);
                    const float2 lu = bnd[i * 16 + NX];
                    const float t0 = un + ay[i];                                         // admm.cpp:47
                    const float tz = __builtin_amdgcn_fmed3f(rnd<H16>(t0), lu.x, lu.y); // admm.cpp:51-54
                    ay[i] = rnd<HD>(t0 - tz);                                           // admm.cpp:69
                    pru = fmaxf(pru, fabsf(un - tz));                                    // admm.cpp:97
                    duu = fmaxf(duu, fabsf(bz[i] - tz));                                 // admm.cpp:98
                    sz[i] = tz;
                }
                const float2 lx = bnd[i * 16 + j];
                const float t0 = s + ax[i];                                              // admm.cpp:48
                const float tx = __builtin_amdgcn_fmed3f(rnd<H16>(t0), lx.x, lx.y);     // admm.cpp:57-60
                ax[i] = rnd<HD>(t0 - tx);                                               // admm.cpp:70
                prx = fmaxf(prx, fabsf(s - tx));                                         // admm.cpp:95
                dux = fmaxf(dux, fabsf(bx[i] - tx));                                     // admm.cpp:96
                sx[i] = tx;
                s = xn;
            }
            pN = lin_cost<EXACT, H16>(pterm, rho, sx[N - 1] - ax[N - 1]); // admm.cpp:83-84
            // ---------------- termination_condition (admm.cpp:91-109) ----------------
            const float pri_x = quad_max(prx), dua_x = quad_max(dux);
            itn = it + 1;
            bool conv = false;
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pru; r_di = duu * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv)
            {
                st = TINY_STATUS_SOLVED_;
                active = false;
            }
            else
            {
                // ---------------- backward sweep: v = vnew, z = znew, linear cost, backward_pass_grad ----------------
                float p = pN;
                bx[N - 1] = sx[N - 1];
                ran_bwd = true;
#pragma unroll
                for (int i = N - 2; i >= 0; i--)
                {
                    const float lin_x = lin_cost<EXACT, H16>(cq[i], rho, sx[i] - ax[i]);  // q_i (row j)   admm.cpp:81-82
                    const float lin_u = lin_cost<EXACT, H16>(-0.f, rho, sz[i] - ay[i]);    // r_i           admm.cpp:80
                    float dnew, pn;
                    if constexpr (EXACT)
                    {
                        float tB[4], tP[4];
                        quad_products2(tB, tP, p, Bcol, Am);
                        const float tmp = ((tB[0] + tB[2]) + (tB[1] + tB[3])) + lin_u;    // Bdyn^T p + r   admm.cpp:19
                        dnew = rnd<H16>(Quu * tmp);
                        pn = rnd<H16>((lin_x + (((tP[0] + tP[1]) + tP[2]) + tP[3])) - Kj * lin_u); // admm.cpp:20
                    }
                    else
                    {
                        float aB, aP;
                        quad_fma2(aB, aP, p, Bcol, Am);
                        dnew = rnd<H16>(Quu * (aB + lin_u));
                        pn = rnd<H16>(__builtin_fmaf(Kj, lin_u, lin_x + aP)); // Kj holds -Kinf here
                    }
                    if (!keep_d) dd[i] = dnew;
                    pp[i] = pn; dl[i] = dnew;                      // [p_i ; d_i] of this sweep
                    bx[i] = sx[i]; bz[i] = sz[i];                  // admm.cpp:141-142
                    p = pn;
                }
            }
        }
    }
    if (!MPC || ms + 1 >= P.mpc_steps) break;
    // ---------------- advance to the next MPC step on chip (examples/codegen_cartpole.cpp closed loop) ----------------
    {
        float u0v, x1;
        lqr(x0, dd[0], u0v, x1); // u_0 of the solve that just finished, in the solver's own arithmetic
        if (P.u0_traj && valid && lead) P.u0_traj[(long long)ms * P.batch + inst] = u0v;
        if constexpr (MPC)
        {
            // plant step x_1 = Adyn x0 + Bdyn u_0 in the plant kernel's (sequential, separately rounded) arithmetic
            float tK[4], tA[4];
            quad_products2(tK, tA, x0, Kneg, Arow);
            x0 = (((tA[0] + tA[1]) + tA[2]) + tA[3]) + Bj * u0v;
        }
        wstart += P.window_advance;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            if (P.xref_mode == 1)
            {
                int row = wstart + i;
                row = row < P.table_rows ? row : P.table_rows - 1;
                const float xr = ldw<H16>(P.xref_table, row * 16 + j);
                cq[i] = rnd<H16>(-(xr * qrow));
                if (i == N - 1) xrN = xr;
            }
            dd[i] = dl[i];           // d of the workspace = d of the last executed backward sweep
            ax[i] = 0.f; ay[i] = 0.f; // y = g = 0
        }
        if (P.xref_mode == 1) pterm = terminal(xrN);
    }
    }

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && lead)
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
            const int ox = rowx + i * 16, ou = rowu + i * 16;
            float un = 0.f, xn = 0.f;
            if (i < N - 1) lqr(s, dd[i], un, xn); // x,u regenerated from the d of the last executed forward sweep
            stw<H16>(P.xu, ox, s);
            const float lin_x = lin_cost<EXACT, H16>(cq[i], rho, sx[i] - ax[i]);
            stw<H16>(P.qr, ox, lin_x);
            stw<H16>(P.pd, ox, i == N - 1 ? pN : pp[i]);
            stw<H16>(P.vz, ox, bx[i]);
            stw<H16>(P.vzn, ox, sx[i]);
            stw<HD>(P.gy, ox, ax[i]);
            if (lead)
            {
                const bool inp = i < N - 1; // the input-type members have N-1 columns; column N-1 of the row layout is zero
                stw<H16>(P.xu, ou, inp ? un : 0.f);
                stw<H16>(P.qr, ou, inp ? lin_cost<EXACT, H16>(-0.f, rho, sz[i] - ay[i]) : 0.f);
                stw<H16>(P.pd, ou, inp ? dl[i] : 0.f);
                stw<H16>(P.vz, ou, inp ? bz[i] : 0.f);
                stw<H16>(P.vzn, ou, inp ? sz[i] : 0.f);
                stw<HD>(P.gy, ou, inp ? ay[i] : 0.f);
            }
            s = xn;
        }
        if (MPC) // the host's plant step continues from here
        {
            P.x0buf[inst * NX + j] = x0;
            if (lead && P.xref_mode == 1) P.xref_start[inst] = wstart;
        }
        if (lead)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

#define TINY_FOR_EACH_QUADLANE(X) X(10)

bool quadlane_supported(int nx, int nu, int N)
{
    if (nx != 4 || nu != 1) return false;
#define TINY_QUADLANE_CHECK(NN) \
    if (N == NN) return true;
    TINY_FOR_EACH_QUADLANE(TINY_QUADLANE_CHECK)
    return false;
}

hipError_t launch_admm_quadlane(int N, bool exact, bool h16, const RowParams &P, hipStream_t stream)
{
    const int nblocks = (P.batch + 15) / 16;
    if (P.mpc_steps > 1) // closed loop on chip: fp32 storage only
    {
        if (h16 || P.dual32) return hipErrorInvalidValue;
#define TINY_QUADLANE_MPC_DISPATCH(NN)                                                                                   \
    if (N == NN)                                                                                                         \
    {                                                                                                                    \
        if (exact) hipLaunchKernelGGL((admm_quadlane_kernel<NN, true, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_quadlane_kernel<NN, false, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);      \
        return hipGetLastError();                                                                                        \
    }
        TINY_FOR_EACH_QUADLANE(TINY_QUADLANE_MPC_DISPATCH)
        return hipErrorInvalidValue;
    }
    if (P.dual32) // fp16 storage with fp32 duals
    {
        if (!h16) return hipErrorInvalidValue;
#define TINY_QUADLANE_D32_DISPATCH(NN)                                                                                   \
    if (N == NN)                                                                                                         \
    {                                                                                                                    \
        if (exact) hipLaunchKernelGGL((admm_quadlane_kernel<NN, true, true, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);  \
        else hipLaunchKernelGGL((admm_quadlane_kernel<NN, false, true, false, true>), dim3(nblocks), dim3(WAVE), 0, stream, P);      \
        return hipGetLastError();                                                                                        \
    }
        TINY_FOR_EACH_QUADLANE(TINY_QUADLANE_D32_DISPATCH)
        return hipErrorInvalidValue;
    }
#define TINY_QUADLANE_LAUNCH(NN, EX, H) \
    hipLaunchKernelGGL((admm_quadlane_kernel<NN, EX, H>), dim3(nblocks), dim3(WAVE), 0, stream, P)
#define TINY_QUADLANE_DISPATCH(NN)                                \
    if (N == NN)                                                  \
    {                                                             \
        if (exact && !h16) TINY_QUADLANE_LAUNCH(NN, true, false);      \
        else if (exact) TINY_QUADLANE_LAUNCH(NN, true, true);          \
        else if (!h16) TINY_QUADLANE_LAUNCH(NN, false, false);         \
        else TINY_QUADLANE_LAUNCH(NN, false, true);                    \
        return hipGetLastError();                                 \
    }
    TINY_FOR_EACH_QUADLANE(TINY_QUADLANE_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
