// admm_tile16.hip — state-on-chip batched TinyMPC ADMM kernel with the gain x state products on the matrix cores
// (quadrotor class nx = 12, nu = 4; one instantiation per horizon).
//
// Restates tiny_solve() (src/tinympc/admm.cpp:111-152) like admm_rowlane.hip — whole loop-carried state of an instance on
// chip for the entire solve, HBM touched only for the live-in and the live-out — but with 16 instances per wavefront as the
// 16 columns of a 16x16 MFMA tile instead of 4 instances as DPP rows:
//
//   lane = 16*g + c holds, for instance 16*tile + c, the rows {4v + g : v = 0..3} of the stacked vector [x ; u] in four
//   registers (v = 0..2: x rows, v = 3: u row g).  That is the B-operand layout of the 16x16 f32 MFMAs (lane (g, c) supplies
//   B[k = g][col = c] of the K-slice v) and, with the gain rows taken in the order rho(i) = 4*(i&3) + (i>>2), also their D
//   layout (lane (g, c), register v = D row 4g + v), so a horizon sweep never moves data between lanes.  The gain matrices
//   are the row kernels' own packed table (RowParams::mats): its entry (k, r) is A-operand lane (k & 3, i) of K-slice k >> 2
//   with r = rho(i).
//
// Two arithmetic modes (template parameter EXACT), the row kernels' modes with the same results:
//   EXACT = false: v_mfma_f32_16x16x4_f32 — a k-ascending fp32 fma chain per output, bit for bit the v_fmac_f32_dpp chain of
//                  the row kernels' fma arithmetic.
//   EXACT = true : v_mfma_f32_16x16x1_4b_f32 with C = -0: K = 1, so every output is ONE product, fma(a, b, -0) = the
//                  separately rounded a*b with the sign of a zero product intact; one issue delivers the 4 x 16 x 16
//                  products of four gain columns for 16 instances (1024 exact products in 32 cycles, beside the vector
//                  pipe).  The sums are plain v_add_f32 in the reference's orders (RowPlans) over registers of the SAME
//                  lane: no cross-lane operand, no select between the x-row and u-row orders (a register is one or the other).
//                  Results are BITWISE identical to the compiled reference, like the row kernels'.
//
// Why: the row kernels are bound by the issue of their cross-lane (DPP) multiplies, 36 per instance-step pair at 5-6 cycles
// each with two waves per SIMD.  Here the products cost the vector pipe nothing and the per-instance
// vector work drops from 107 (exact) / 55 (fma) instructions per 4 instances to about 190 / 90 per 16.  The price is the
// state: 16 instances x 1068 loop-carried floats = 267 registers per lane, so one wave per SIMD with the whole 512-entry
// register file (duals, feed-forward and old slack in VGPRs/AGPRs, one v_accvgpr move per access of the latter; new slack in
// LDS, 30 KB per wave; the live-out [p;d] of a backward sweep is written through to its array, whole 64-byte rows that L2
// and the Infinity Cache absorb), and 16 instances run in lock step.
//
// Scope: shared box bounds, reference = window of a trajectory table or one shared reference (a per-instance reference
// array would have to stay resident: the row kernels serve that), fp32 storage, one solve per launch.
#include "rowlane_math.h"

namespace tinympc
{

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef TINY_T16_ABLATE
#define TINY_T16_ABLATE 0
#endif

#if TINY_T16_ABLATE == 1 // timing experiment: no matrix-core work (results are wrong)
__device__ __forceinline__ f32x4 t16_fake4(float a, float b, f32x4 c) { c[0] += a; c[1] += b; c[2] += a; c[3] += b; return c; }
#define TINY_MFMA4(a, b, c) t16_fake4((a), (b), (c))
#define TINY_MFMA1(a, b, c) __builtin_amdgcn_mfma_f32_16x16x1f32((a), (b), (c), 0, 0, 0)
#else
#define TINY_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define TINY_MFMA1(a, b, c) __builtin_amdgcn_mfma_f32_16x16x1f32((a), (b), (c), 0, 0, 0)
#endif

// products of K-slices 0..2 for output register V: t[k] = M[row 4V+g][k] * s[k], k = 0..11
template <int V>
__device__ __forceinline__ void gather12(float (&t)[12], const f32x16 &p0, const f32x16 &p1, const f32x16 &p2)
{
#pragma unroll
    for (int b = 0; b < 4; b++)
    {
        t[b] = p0[4 * b + V];
        t[4 + b] = p1[4 * b + V];
        t[8 + b] = p2[4 * b + V];
    }
}
template <int V>
__device__ __forceinline__ void gather4(float (&t)[4], const f32x16 &p)
{
#pragma unroll
    for (int b = 0; b < 4; b++) t[b] = p[4 * b + V];
}

template <bool EXACT>
struct TileMath
{
    static constexpr int NX = 12, NU = 4;
    using PL = RowPlans<NX, NU>;
    float A1[3], A2, A3[3], A45, AP[3]; // MFMA A operands (one VGPR each)
    f32x16 negz;                        // C input of the exact products

    __device__ __forceinline__ void load(const float *mats, int g, int c)
    {
        const float *m = mats + 4 * (c & 3) + (c >> 2); // rho(c)
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            A1[ch] = m[(4 * ch + g) * 16];
            A3[ch] = m[(NX + NU + 4 * ch + g) * 16];
            AP[ch] = m[(2 * NX + 2 * NU + 1 + 4 * ch + g) * 16];
        }
        A2 = m[(NX + g) * 16];
        A45 = m[(2 * NX + NU + g) * 16];
#pragma unroll
        for (int e = 0; e < 16; e++) negz[e] = -0.f;
    }

    // forward_pass step (admm.cpp:31,35): s = x_i (registers 0..2), di = d_i (u row)  ->  un = u_i, xn = x_{i+1}
    __device__ __forceinline__ void lqr(const float (&s)[3], float di, float &un, float (&xn)[3]) const
    {
        if constexpr (EXACT)
        {
            const f32x16 p0 = TINY_MFMA1(A1[0], s[0], negz), p1 = TINY_MFMA1(A1[1], s[1], negz), p2 = TINY_MFMA1(A1[2], s[2], negz);
            float t[12], acc[3];
            gather12<0>(t, p0, p1, p2); acc[0] = reduce<PL::FWD_XA>(t);
            gather12<1>(t, p0, p1, p2); acc[1] = reduce<PL::FWD_XA>(t);
            gather12<2>(t, p0, p1, p2); acc[2] = reduce<PL::FWD_XA>(t);
            gather12<3>(t, p0, p1, p2);
            un = -reduce<PL::FWD_U>(t) - di; // -(K x) - d: the SUM is negated, as in the reference
            const f32x16 pb = TINY_MFMA1(A2, un, negz);
            float t2[4];
            gather4<0>(t2, pb); xn[0] = acc[0] + reduce<PL::FWD_XB>(t2);
            gather4<1>(t2, pb); xn[1] = acc[1] + reduce<PL::FWD_XB>(t2);
            gather4<2>(t2, pb); xn[2] = acc[2] + reduce<PL::FWD_XB>(t2);
        }
        else
        {
            f32x4 acc = {-0.f, -0.f, -0.f, -0.f}; // fma(a, b, -0) = a*b: the chain starts with a plain product, like dpp_fma_dot
            acc = TINY_MFMA4(A1[0], s[0], acc);
            acc = TINY_MFMA4(A1[1], s[1], acc);
            acc = TINY_MFMA4(A1[2], s[2], acc);
            un = acc[3] - di; // u rows of M1 hold -Kinf in the fma table
            acc = TINY_MFMA4(A2, un, acc);
            xn[0] = acc[0]; xn[1] = acc[1]; xn[2] = acc[2];
        }
    }

    // backward_pass_grad step (admm.cpp:19-20): p = p_{i+1}, lin = [q_i ; r_i]  ->  pn = p_i, dd = d_i
    __device__ __forceinline__ void riccati(const float (&p)[3], const float (&lin)[4], float (&pn)[3], float &dd) const
    {
        if constexpr (EXACT)
        {
            const f32x16 p0 = TINY_MFMA1(A3[0], p[0], negz), p1 = TINY_MFMA1(A3[1], p[1], negz), p2 = TINY_MFMA1(A3[2], p[2], negz);
            const f32x16 pk = TINY_MFMA1(A45, lin[3], negz); // Kinf^T r (x rows)
            float t[12], wv[4];
            gather12<0>(t, p0, p1, p2); wv[0] = lin[0] + reduce<PL::BWD_PA>(t);
            gather12<1>(t, p0, p1, p2); wv[1] = lin[1] + reduce<PL::BWD_PA>(t);
            gather12<2>(t, p0, p1, p2); wv[2] = lin[2] + reduce<PL::BWD_PA>(t);
            gather12<3>(t, p0, p1, p2); wv[3] = lin[3] + reduce<PL::BWD_TMP>(t); // Bdyn^T p + r
            const f32x16 pq = TINY_MFMA1(A45, wv[3], negz); // Quu_inv (Bdyn^T p + r) (u row)
            float tk[4];
            gather4<0>(tk, pk); pn[0] = wv[0] - reduce<PL::BWD_PK>(tk);
            gather4<1>(tk, pk); pn[1] = wv[1] - reduce<PL::BWD_PK>(tk);
            gather4<2>(tk, pk); pn[2] = wv[2] - reduce<PL::BWD_PK>(tk);
            gather4<3>(tk, pq); dd = reduce<PL::BWD_D>(tk);
        }
        else
        {
            f32x4 acc = {lin[0], lin[1], lin[2], lin[3]};
            acc = TINY_MFMA4(A3[0], p[0], acc);
            acc = TINY_MFMA4(A3[1], p[1], acc);
            acc = TINY_MFMA4(A3[2], p[2], acc);
            const f32x4 nz = {-0.f, -0.f, -0.f, -0.f};
            const f32x4 dq = TINY_MFMA4(A45, acc[3], nz); // u row: Quu_inv
            acc = TINY_MFMA4(A45, lin[3], acc);           // x rows: -Kinf^T
            pn[0] = acc[0]; pn[1] = acc[1]; pn[2] = acc[2];
            dd = dq[3];
        }
    }

    // -(Xref_{N-1}^T Pinf) (admm.cpp:83), x rows
    __device__ __forceinline__ void terminal(const float (&xr)[3], float (&pt)[3]) const
    {
        if constexpr (EXACT)
        {
            const f32x16 p0 = TINY_MFMA1(AP[0], xr[0], negz), p1 = TINY_MFMA1(AP[1], xr[1], negz), p2 = TINY_MFMA1(AP[2], xr[2], negz);
            float t[12];
            gather12<0>(t, p0, p1, p2); pt[0] = -reduce<PL::TERM>(t);
            gather12<1>(t, p0, p1, p2); pt[1] = -reduce<PL::TERM>(t);
            gather12<2>(t, p0, p1, p2); pt[2] = -reduce<PL::TERM>(t);
        }
        else
        {
            f32x4 acc = {-0.f, -0.f, -0.f, -0.f};
            acc = TINY_MFMA4(AP[0], xr[0], acc);
            acc = TINY_MFMA4(AP[1], xr[1], acc);
            acc = TINY_MFMA4(AP[2], xr[2], acc);
            pt[0] = -acc[0]; pt[1] = -acc[1]; pt[2] = -acc[2];
        }
    }
};

// max over the four lanes (g = 0..3) that hold one instance's rows
__device__ __forceinline__ float tile_inst_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// State that is written far more often than read (the backup of the old slack, the live-out [p;d]) lives in the accumulator
// half of the register file for its whole life: the only instructions that touch it are these two, so the allocator keeps it
// there instead of shuttling it through VGPRs.
__device__ __forceinline__ void acc_put(float &dst, float val) { asm volatile("v_accvgpr_write_b32 %0, %1" : "+a"(dst) : "v"(val)); }
__device__ __forceinline__ float acc_get(const float &src)
{
    float r;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(src));
    return r;
}
__device__ __forceinline__ void acc_init(float &dst, float val) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(val)); }

// The state updates of one horizon step, applied to the lanes in `m` only (the instances still iterating).  Written as ONE
// asm block that narrows EXEC itself: expressed as `if (active) { ... }` the 60 small regions per iteration make hipcc keep
// the old and the new value of every state word alive side by side (twice the state: hundreds of spilled registers).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float4 lds_float4;
__device__ __forceinline__ void masked_forward_update(unsigned long long m, float (&a)[4], const float (&an)[4], float (&bo)[4],
                                                      const float (&old)[4], unsigned sn_addr, const float (&t)[4])
{
    unsigned long long sx;
    const f32x4v tv = {t[0], t[1], t[2], t[3]};
    asm volatile("s_and_saveexec_b64 %[sx], %[m]\n\t"
                 "v_mov_b32 %[a0], %[n0]\n\tv_mov_b32 %[a1], %[n1]\n\tv_mov_b32 %[a2], %[n2]\n\tv_mov_b32 %[a3], %[n3]\n\t"
                 "v_accvgpr_write_b32 %[b0], %[o0]\n\tv_accvgpr_write_b32 %[b1], %[o1]\n\tv_accvgpr_write_b32 %[b2], %[o2]\n\tv_accvgpr_write_b32 %[b3], %[o3]\n\t"
                 "ds_write_b128 %[ad], %[tv]\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [sx] "=&s"(sx), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [b0] "+a"(bo[0]), [b1] "+a"(bo[1]),
                   [b2] "+a"(bo[2]), [b3] "+a"(bo[3])
                 : [m] "s"(m), [n0] "v"(an[0]), [n1] "v"(an[1]), [n2] "v"(an[2]), [n3] "v"(an[3]), [o0] "v"(old[0]), [o1] "v"(old[1]),
                   [o2] "v"(old[2]), [o3] "v"(old[3]), [ad] "v"(sn_addr), [tv] "v"(tv)
                 : "memory", "scc"); // s_and_saveexec writes SCC
}
__device__ __forceinline__ void masked_backward_update(unsigned long long m, float (&pl)[3], float &dl, const float (&pn)[3], float dd)
{
    unsigned long long sx;
    asm volatile("s_and_saveexec_b64 %[sx], %[m]\n\t"
                 "v_accvgpr_write_b32 %[p0], %[n0]\n\tv_accvgpr_write_b32 %[p1], %[n1]\n\tv_accvgpr_write_b32 %[p2], %[n2]\n\tv_accvgpr_write_b32 %[d], %[dd]\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [sx] "=&s"(sx), [p0] "+a"(pl[0]), [p1] "+a"(pl[1]), [p2] "+a"(pl[2]), [d] "+a"(dl)
                 : [m] "s"(m), [n0] "v"(pn[0]), [n1] "v"(pn[1]), [n2] "v"(pn[2]), [dd] "v"(dd)
                 : "scc");
}

constexpr int TILE16_WAVES = 4;         // waves per workgroup = one per SIMD of a CU; they share the bounds and reference tables
constexpr int TILE16_MAX_TABLE_ROWS = 512;

template <int N, bool EXACT>
__global__ __launch_bounds__(WAVE * TILE16_WAVES, 1) void admm_tile16_kernel(const RowParams P)
{
    constexpr int NX = 12, NU = 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int inst = (blockIdx.x * TILE16_WAVES + wv) * 16 + c;
    const bool valid = inst < P.batch;
    const int inst_a = valid ? inst : P.batch - 1; // padding columns of the last tile load a valid instance and store nothing
    const float rho = P.rho;

    // ---- LDS (dynamic): per wave the slack [v|vnew ; z|znew] of every step (lane-linear float4), then the tables the four
    //      waves share: box bounds [step][g] -> registers v = 0..3 (row 4v + g), reference rows [row][g] -> x rows 4v + g
    extern __shared__ __attribute__((aligned(16))) float4 lds4[];
    float4 *sn = lds4 + wv * (N * WAVE) + lane; // sn[i * WAVE]
    float4 *blo = lds4 + TILE16_WAVES * N * WAVE, *bhi = blo + N * 4, *tab = bhi + N * 4;
    const int tab_rows = P.xref_mode == 1 ? P.table_rows : N;
    const float *tab_src = P.xref_mode == 1 ? P.xref_table : P.xref; // [rows][16]
    for (int e = threadIdx.x; e < N * 16; e += WAVE * TILE16_WAVES)
    {
        const float2 lh = reinterpret_cast<const float2 *>(P.bounds)[e];
        const int i = e >> 4, r = e & 15;
        reinterpret_cast<float *>(&blo[i * 4 + (r & 3)])[r >> 2] = lh.x;
        reinterpret_cast<float *>(&bhi[i * 4 + (r & 3)])[r >> 2] = lh.y;
    }
    for (int e = threadIdx.x; e < tab_rows * 16; e += WAVE * TILE16_WAVES)
        reinterpret_cast<float *>(&tab[(e >> 4) * 4 + (e & 3)])[(e & 15) >> 2] = tab_src[e];
    __syncthreads();

    TileMath<EXACT> M;
    M.load(P.mats, g, c);
    float qv[3];
#pragma unroll
    for (int v = 0; v < 3; v++) qv[v] = P.mats[(2 * NX + 2 * NU) * 16 + 4 * v + g];

    // ---- per-instance state, four words per horizon step (row 4v + g) ----
    //   a[i]   : g_i | y_i      duals                                                           (VGPR)
    //   dr[i]  : d_i            feed-forward the forward sweep uses                              (VGPR)
    //   sn[i]  : before forward step i of an iteration the OLD slack v_i | z_i (what the previous iteration's sweep left, = v
    //            after admm.cpp:141-142), afterwards the new one vnew_i | znew_i                (LDS)
    //   bo[i]  : backup of the old slack this iteration's forward sweep replaced — the live-out v, z of an instance that
    //            converges in this iteration (the reference returns before v = vnew)            (AGPR, write-mostly)
    //   pl[i], dl[i] : p_i, d_i of the last executed backward sweep, live-out only             (AGPR, write-only in the loop)
    float a[N][4], dr[N], bo[N][4], pl[N][3], dl[N];
    const int ebase = (inst_a * N) * 16 + g; // element 4v + g of step i: ebase + i*16 + 4v
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[inst_a];
    const bool cold = P.cold_start != 0;
    const bool zdual = cold || (P.duals_zero != 0);

#pragma unroll
    for (int i = 0; i < N; i++)
    {
        float vzl[4], pdv[4];
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
            const int o = ebase + i * 16 + 4 * v;
            vzl[v] = P.vz[o]; pdv[v] = P.pd[o];
            const float gyv = P.gy[o];
            vzl[v] = cold ? 0.f : vzl[v];
            pdv[v] = cold ? 0.f : pdv[v];
            a[i][v] = zdual ? 0.f : gyv;
            acc_init(bo[i][v], vzl[v]);
        }
        sn[i * WAVE] = make_float4(vzl[0], vzl[1], vzl[2], vzl[3]);
        acc_init(pl[i][0], pdv[0]); acc_init(pl[i][1], pdv[1]); acc_init(pl[i][2], pdv[2]); acc_init(dl[i], pdv[3]);
        dr[i] = pdv[3];
    }
    float x0[3];
#pragma unroll
    for (int v = 0; v < 3; v++) x0[v] = P.xu[ebase + 4 * v];

    // x rows of Xref_i for this lane (registers 0..2)
    auto load_xref = [&](const float4 *tb, int ws, int i, float(&xr)[3]) {
        int row = ws + i;
        row = row < tab_rows ? row : tab_rows - 1;
        const float4 t4 = tb[row * 4 + g];
        xr[0] = t4.x; xr[1] = t4.y; xr[2] = t4.z;
    };
    float pterm[3];
    {
        float xrN[3];
        load_xref(tab, wstart, N - 1, xrN);
        M.terminal(xrN, pterm);
    }

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !P.cold_start)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN[3] = {0.f, 0.f, 0.f}; // p_{N-1} of the last executed forward sweep

    bool active = valid && (P.max_iter > 0);
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        // the last permitted iteration must not overwrite d in dr[]: x,u of an instance that exhausts max_iter come from the
        // d its last forward sweep used (regenerated in the epilogue); the final d itself goes to dl[] only
        const bool keep_d = (it == P.max_iter - 1);
        // An opaque zero, re-made in every iteration, enters every LDS address of the sweeps: the addresses (30 slack slots, 30
        // bounds rows, 30 clamped reference rows per lane) are loop invariant, and hipcc would otherwise compute them all
        // ahead of the iteration loop and hold — in fact spill — them
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        float4 *const snI = sn + oz;
        const float4 *const bloI = blo + oz, *const bhiI = bhi + oz, *const tabI = tab + oz;
        const int wsI = wstart + oz;
        const unsigned long long amask = __ballot(active); // instances still iterating: wave-uniform during the forward sweep
        const unsigned sn_addr = (unsigned)(size_t)(lds_float4 *)snI;
        // The arithmetic of a sweep runs for all 16 columns (the MFMAs are wave-wide); only the state updates of a
        // converged instance are masked, which freezes it exactly where the reference returns.
        // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
        float s[3] = {x0[0], x0[1], x0[2]};
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            float sv[4], xn[3] = {0.f, 0.f, 0.f};
            sv[0] = s[0]; sv[1] = s[1]; sv[2] = s[2]; sv[3] = 0.f;
            if (i < N - 1) M.lqr(s, dr[i], sv[3], xn);
            const float4 lo = bloI[i * 4 + g], hi = bhiI[i * 4 + g], ol = snI[i * WAVE];
            const float lov[4] = {lo.x, lo.y, lo.z, lo.w}, hiv[4] = {hi.x, hi.y, hi.z, hi.w}, old[4] = {ol.x, ol.y, ol.z, ol.w};
            float t[4], an[4];
#if TINY_T16_ABLATE == 2 // timing experiment: no slack / dual / residual arithmetic (results are wrong)
#pragma unroll
            for (int v = 0; v < 4; v++) { t[v] = sv[v]; an[v] = a[i][v]; }
#else
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
                const float tp = sv[v] + a[i][v];                         // admm.cpp:47-48
                t[v] = __builtin_amdgcn_fmed3f(tp, lov[v], hiv[v]);       // :51-60 (lo := min(lo, hi) on the host)
                an[v] = tp - t[v];                                        // :69-70  (y + u) - znew
                const float dp = fabsf(sv[v] - t[v]), dd_ = fabsf(old[v] - t[v]); // :95-98
                if (v < 3) { pri_x = fmaxf(pri_x, dp); dua_x = fmaxf(dua_x, dd_); }
                else { pri_u = fmaxf(pri_u, dp); dua_u = fmaxf(dua_u, dd_); }
            }
#endif
            masked_forward_update(amask, a[i], an, bo[i], old, sn_addr + i * (WAVE * 16), t);
            if (i == N - 1)
            {
#pragma unroll
                for (int v = 0; v < 3; v++)
                {
                    const float pn_ = lin_cost<EXACT>(pterm[v], rho, t[v] - an[v]); // admm.cpp:83-84
                    pN[v] = active ? pn_ : pN[v];
                }
            }
            s[0] = xn[0]; s[1] = xn[1]; s[2] = xn[2];
        }
        // ---------------- termination_condition (admm.cpp:91-109) ----------------
        pri_x = tile_inst_max(pri_x); dua_x = tile_inst_max(dua_x);
        pri_u = tile_inst_max(pri_u); dua_u = tile_inst_max(dua_u);
        bool conv = false;
        if (active)
        {
            itn = it + 1;
            if ((it + 1) % P.check_termination == 0)
            {
                r_ps = pri_x; r_ds = dua_x * rho; r_pi = pri_u; r_di = dua_u * rho;
                conv = (r_ps < P.abs_pri_tol) && (r_pi < P.abs_pri_tol) && (r_ds < P.abs_dua_tol) && (r_di < P.abs_dua_tol);
            }
            if (conv) st = TINY_STATUS_SOLVED_;
        }
        active = active && !conv;
        if (!__any(active)) break;
        // ---------------- backward sweep: (v = vnew, z = znew are implicit: sn holds both) linear cost, backward_pass_grad ----------------
        {
            float p[3] = {pN[0], pN[1], pN[2]};
            const unsigned long long amask = __ballot(active);
            const bool upd_d = active && !keep_d;
#pragma unroll
            for (int i = N - 2; i >= 0; i--)
            {
                const float4 sl = snI[i * WAVE];
                const float sni[4] = {sl.x, sl.y, sl.z, sl.w};
                float xr[3], lin[4];
                load_xref(tabI, wsI, i, xr);
#pragma unroll
                for (int v = 0; v < 3; v++) lin[v] = lin_cost<EXACT>(-(xr[v] * qv[v]), rho, sni[v] - a[i][v]); // admm.cpp:81-82
                lin[3] = lin_cost<EXACT>(-0.f, rho, sni[3] - a[i][3]);                                         // :80
                float pn[3], dd;
                M.riccati(p, lin, pn, dd);
                dr[i] = upd_d ? dd : dr[i];
                masked_backward_update(amask, pl[i], dl[i], pn, dd);
                p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
            }
        }
    }

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && g == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }

    // ---------------- live-out: every work array written once ----------------
    {
        const bool solved = st == TINY_STATUS_SOLVED_;
        float s[3] = {x0[0], x0[1], x0[2]};
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            float sv[4], xn[3] = {0.f, 0.f, 0.f};
            sv[0] = s[0]; sv[1] = s[1]; sv[2] = s[2]; sv[3] = 0.f;
            if (i < N - 1) M.lqr(s, dr[i], sv[3], xn);
            const float4 sl = sn[i * WAVE];
            const float sni[4] = {sl.x, sl.y, sl.z, sl.w};
            float xr[3], lin[4];
            load_xref(tab, wstart, i, xr);
#pragma unroll
            for (int v = 0; v < 3; v++) lin[v] = lin_cost<EXACT>(-(xr[v] * qv[v]), rho, sni[v] - a[i][v]);
            lin[3] = (i < N - 1) ? lin_cost<EXACT>(-0.f, rho, sni[3] - a[i][3]) : 0.f;
            const float pdv[4] = {acc_get(pl[i][0]), acc_get(pl[i][1]), acc_get(pl[i][2]), acc_get(dl[i])};
            if (valid)
            {
#pragma unroll
                for (int v = 0; v < 4; v++)
                {
                    const int o = ebase + i * 16 + 4 * v;
                    P.xu[o] = sv[v];
                    P.qr[o] = lin[v];
                    // p.col(N-1) is rewritten by every forward sweep (admm.cpp:83-84); the other columns and d come from the
                    // last backward sweep this instance executed (an instance that never ran one keeps its live-in p, d)
                    P.pd[o] = (i == N - 1) ? (v < 3 ? pN[v] : 0.f) : pdv[v];
                    // a converged instance returned before v = vnew (admm.cpp:135-142): its v, z are the slack the last sweep replaced
                    P.vz[o] = solved ? acc_get(bo[i][v]) : sni[v];
                    P.vzn[o] = sni[v];
                    P.gy[o] = a[i][v];
                }
            }
            s[0] = xn[0]; s[1] = xn[1]; s[2] = xn[2];
        }
        if (valid && g == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
    }
}

#define TINY_FOR_EACH_TILE16(X) X(30)

bool tile16_supported(int nx, int nu, int N)
{
    if (nx != 12 || nu != 4) return false;
#define TINY_TILE16_CHECK(NN) \
    if (N == NN) return true;
    TINY_FOR_EACH_TILE16(TINY_TILE16_CHECK)
    return false;
}

int tile16_max_table_rows() { return TILE16_MAX_TABLE_ROWS; }

hipError_t launch_admm_tile16(int N, bool exact, const RowParams &P, hipStream_t stream)
{
    const int ntiles = (P.batch + 15) / 16, nblocks = (ntiles + TILE16_WAVES - 1) / TILE16_WAVES;
    const int rows = P.xref_mode == 1 ? P.table_rows : N;
    if (rows > TILE16_MAX_TABLE_ROWS) return hipErrorInvalidValue;
    const size_t lds = (size_t)(TILE16_WAVES * N * WAVE + 2 * N * 4 + rows * 4) * sizeof(float4);
#define TINY_TILE16_LAUNCH(NN, EX)                                                                                         \
    {                                                                                                                      \
        hipError_t e = hipFuncSetAttribute((const void *)admm_tile16_kernel<NN, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL((admm_tile16_kernel<NN, EX>), dim3(nblocks), dim3(WAVE * TILE16_WAVES), lds, stream, P);        \
        return hipGetLastError();                                                                                          \
    }
#define TINY_TILE16_DISPATCH(NN)                  \
    if (N == NN)                                  \
    {                                             \
        if (exact) TINY_TILE16_LAUNCH(NN, true)   \
        else TINY_TILE16_LAUNCH(NN, false)        \
    }
    TINY_FOR_EACH_TILE16(TINY_TILE16_DISPATCH)
    return hipErrorInvalidValue;
}

} // namespace tinympc
