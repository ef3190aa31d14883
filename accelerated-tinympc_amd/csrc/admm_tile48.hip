// admm_tile48.hip — the nx = 32, nu = 16 class (BASELINE.json configs[3]) with SIXTEEN INSTANCES PER WORKGROUP as the columns of
// 16 x 16 matrix-core tiles: the design of admm_tile16.hip for a stacked vector of 48 rows.
// Restates tiny_solve() (src/tinympc/admm.cpp:111-152) with its six step functions fused (forward_pass :27-37, update_slack :45-61,
// update_dual :67-71, update_linear_cost :77-85, termination_condition :91-109, backward_pass_grad :15-22), like the other kernels.
//
// admm_waveres.hip gives an instance one wave and is bound by vector-instruction issue: every product is a v_mul_f32 and every
// sum a v_add_f32 of one row of one instance per lane (296 vector instructions per instance and horizon step pair).  Here
//   * a workgroup of three waves solves a tile of 16 instances; wave 0 holds x rows 0..15, wave 1 x rows 16..31, wave 2 the 16 u
//     rows — lane (g, c) of a wave holds rows 4g .. 4g+3 of its 16 for instance c, the D layout of the 16x16 f32 MFMAs;
//   * the products are exact on the matrix cores: v_mfma_f32_16x16x1_4b_f32 with C = -0 returns, for four gain columns k, the
//     separately rounded products M[row][k] * s_c[k] of 16 rows x 16 instances (fma(a, b, -0) = a*b, signs of zeros intact);
//   * the sums are v_pk_add_f32 over 4-vectors of rows in the reference's orders (wave_math.h: sequential for the forward pass,
//     halving tree for AmBKt p, Eigen's four GEMV accumulators for Bdyn^T p, packet tree for Kinf^T r) — one instruction adds
//     eight products' worth: 34 instead of 296 vector instructions per instance and step pair;
//   * the duals g | y stay in LDS for the whole solve (150 KB per workgroup: one workgroup per CU); the slack lives in its own array
//     (vnew | znew: read one step ahead, off the dependent chain, written back in place), so no state is indexed in registers and the
//     horizon loops stay rolled;
//   * a stage's result crosses waves through LDS (4 ds_write_b32, one workgroup barrier, 8 ds_read_b32 that are directly the next
//     MFMAs' B operands): two barriers per forward step (u_i, x_{i+1}), one per backward step (p_i and r_{i-1}, double buffered);
//   * the replaced slack (the live-out v | z should this iteration converge), [x ; u] of a forward sweep and [p ; d] of a backward
//     sweep are written through to their arrays (write-only, off the dependent chain); the feed-forward d the next forward sweep needs is
//     read back from there one step ahead; -(Xref .* Q) is recomputed from the reference where it is used.  Every load of a step is
//     issued a step ahead and BEFORE that step's stores (vmcnt counts in order: a load issued behind a store could only be waited for
//     together with the store's acknowledgement), and no store sits in a divergent region: a finished column's write-through goes to
//     an array the live-out rewrites anyway.
// Sixteen instances run in lock step (state updates of a finished column are selects on its `active` flag); the workgroup ends when
// its last column has.  Arithmetic, orders and results are those of admm_waveres.hip: bitwise equal to the compiled reference.
// Two arithmetic modes like the wave kernels: EXACT (above) and fma (template parameter false): every stage ONE k-ascending fma chain
// of v_mfma_f32_16x16x4_f32 started from its additive term — no vector sums at all —, held to the bar of every other fma variant (the
// reference's own fp64/fp32 spread).
// Scope: nx = 32, nu = 16, N <= 50 (the LDS holds 50 steps of duals), fp32 storage.
#include "wave_math.h"
#include <atomic>

namespace tinympc
{

namespace
{
typedef float t48v4 __attribute__((ext_vector_type(4)));
typedef float t48v16 __attribute__((ext_vector_type(16)));
constexpr int T48_NX = 32, T48_NU = 16, T48_COLS = 16;
// LDS of a workgroup, in floats
constexpr int T48_XB = 0;                  // [2][32][16] x_i (forward sweep) / p_i (backward sweep), parity of the step
constexpr int T48_UB = 2 * 32 * 16;        // [16][16] u_i (forward) / Bdyn^T p + r (backward, u wave only)
constexpr int T48_RB = T48_UB + 16 * 16;   // [2][16][16] r_i
constexpr int T48_RES = T48_RB + 2 * 16 * 16; // [2][6][16] residual maxima {pri, dua} of the three waves
constexpr int T48_HEAD = 2048;
static_assert(T48_RES + 2 * 6 * 16 <= T48_HEAD, "LDS head");
constexpr size_t t48_lds_bytes(int N) { return (size_t)(T48_HEAD + 3 * N * WAVE * 4) * sizeof(float); }

__device__ __forceinline__ void t48_barrier() // this wave's LDS traffic done, then the workgroup barrier; global memory is not waited for
{
    __builtin_amdgcn_sched_barrier(0); // nothing is scheduled across: without the fences hipcc computes the addresses of all 50 steps of a sweep
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // ahead and spills them
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void t48_fence() { asm volatile("" ::: "memory"); }
// the barrier, with `dep` computed before it (pure arithmetic is otherwise free to sink below the barrier, towards its first use)
template <class T>
__device__ __forceinline__ void t48_barrier_after(T &dep)
{
    asm volatile("" : "+v"(dep));
    t48_barrier();
}

template <int B>
__device__ __forceinline__ t48v4 t48_blk(const t48v16 &p) { return __builtin_shufflevector(p, p, 4 * B, 4 * B + 1, 4 * B + 2, 4 * B + 3); }

// t[k] = M[rows 4g..4g+3][k] * s_c[k], k = 0 .. CNT-1: CNT / 4 matrix-core instructions, A[m] = gain columns 4m .. 4m+3 (lane (g, c)
// supplies column 4m + g of row c of the tile), B[m] = s_c[4m + g]
template <int CNT>
__device__ __forceinline__ void t48_products(t48v4 (&t)[CNT], const float (&A)[CNT / 4], const float (&B)[CNT / 4], const t48v16 &negz)
{
#pragma unroll
    for (int m = 0; m < CNT / 4; m++)
    {
        const t48v16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(A[m], B[m], negz, 0, 0, 0);
        t[4 * m + 0] = t48_blk<0>(d); t[4 * m + 1] = t48_blk<1>(d); t[4 * m + 2] = t48_blk<2>(d); t[4 * m + 3] = t48_blk<3>(d);
    }
}
// B operands of the next products: element 4m + g of column c of a vector stored [row][16]
template <int CNT>
__device__ __forceinline__ void t48_fetch(float (&B)[CNT / 4], const float *src, int g, int c)
{
#pragma unroll
    for (int m = 0; m < CNT / 4; m++) B[m] = src[(4 * m + g) * T48_COLS + c];
}
// rows row0 + 4g .. + 3 of column c
__device__ __forceinline__ void t48_put(float *dst, int row0, int g, int c, const t48v4 &v)
{
#pragma unroll
    for (int j = 0; j < 4; j++) dst[(row0 + 4 * g + j) * T48_COLS + c] = v[j];
}

// the reduction plans of rowlane_math.h / wave_math.h over 4-vectors of rows
template <int LO, int CNT, int NN>
__device__ __forceinline__ t48v4 t48_tree(const t48v4 (&t)[NN])
{
    if constexpr (CNT == 1) return t[LO];
    else
    {
        constexpr int H = CNT / 2;
        return t48_tree<LO, H>(t) + t48_tree<LO + H, CNT - H>(t);
    }
}
template <int PLO, int PCNT, int L, int NN>
__device__ __forceinline__ t48v4 t48_ptree(const t48v4 (&t)[NN])
{
    if constexpr (PCNT == 1) return t[4 * PLO + L];
    else
    {
        constexpr int H = PCNT / 2;
        return t48_ptree<PLO, H, L>(t) + t48_ptree<PLO + H, PCNT - H, L>(t);
    }
}
template <int PLAN, int NN>
__device__ __forceinline__ t48v4 t48_reduce(const t48v4 (&t)[NN])
{
    if constexpr (PLAN == PLAN_SEQ)
    {
        t48v4 acc = t[0];
#pragma unroll
        for (int k = 1; k < NN; k++) acc = acc + t[k];
        return acc;
    }
    else if constexpr (PLAN == PLAN_TREE) return t48_tree<0, NN>(t);
    else if constexpr (PLAN == PLAN_GEMV) // Eigen's row-major GEMV: four accumulators from zero, (c0+c2)+(c1+c3), 0 + 1*acc (wave_math.h)
    {
        static_assert(NN % 4 == 0, "whole packets");
        const t48v4 z = {0.f, 0.f, 0.f, 0.f};
        t48v4 c0 = z, c1 = z, c2 = z, c3 = z;
#pragma unroll
        for (int k = 0; k < NN / 4; k++)
        {
            c0 = c0 + t[4 * k + 0]; c1 = c1 + t[4 * k + 1]; c2 = c2 + t[4 * k + 2]; c3 = c3 + t[4 * k + 3];
        }
        return z + ((c0 + c2) + (c1 + c3));
    }
    else
    {
        static_assert(PLAN == PLAN_VEC && NN % 4 == 0, "packet tree: whole packets");
        constexpr int NPK = NN / 4;
        const t48v4 s0 = t48_ptree<0, NPK, 0>(t), s1 = t48_ptree<0, NPK, 1>(t), s2 = t48_ptree<0, NPK, 2>(t), s3 = t48_ptree<0, NPK, 3>(t);
        return (s0 + s2) + (s1 + s3);
    }
}
// sum_k M[rows][k] * s_c[k] in the order PLAN.  The products are consumed as the matrix cores deliver them, one instruction ahead:
// two results (32 registers) are live at a time.  Left to itself hipcc issues all eight first (128 registers) and the step spills; its
// scheduling fences do not bind pure arithmetic, so the order is tied by data: the B operand of instruction m + 1 passes through an
// empty asm that also names the running sum after instruction m - 1.
#ifndef T48_TIE
#define T48_TIE 0 // measured: tying the order of the matrix-core instructions to the sums costs 5 % once nothing spills
#endif
#ifndef T48_PROFILE
#define T48_PROFILE 0 // timing experiment only (the residual fields of instances 0..7 are overwritten): clocks of block 0 per phase
#endif
#if T48_PROFILE
#define T48_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof[k] += (float)(now_ - tprev); tprev = now_; } while (0)
#else
#define T48_STAMP(k) do { } while (0)
#endif
#ifndef T48_ABLATE
#define T48_ABLATE 0 // timing experiments only (results are wrong): 1 = no backward sweep, 2 = no write-through stores
#endif
__device__ __forceinline__ float t48_after(float x, const t48v4 &dep)
{
#if T48_TIE
    asm volatile("" : "+v"(x) : "v"(dep));
#endif
    return x;
}
#define T48_MFMA_B(m, b) __builtin_amdgcn_mfma_f32_16x16x1f32(A[m], (b), negz, 0, 0, 0)
template <int PLAN, int CNT>
__device__ __forceinline__ t48v4 t48_dot(const float (&A)[CNT / 4], const float (&B)[CNT / 4], const t48v16 &negz)
{
    constexpr int NM = CNT / 4;
    if constexpr (PLAN == PLAN_SEQ)
    {
        t48v4 acc;
        t48v16 d = T48_MFMA_B(0, B[0]);
#pragma unroll
        for (int m = 0; m < NM; m++)
        {
            t48v16 dn = d;
            if (m + 1 < NM)
            {
                const int mn = m + 1 < NM ? m + 1 : m;
                dn = T48_MFMA_B(mn, m > 0 ? t48_after(B[mn], acc) : B[mn]);
            }
            if (m == 0) acc = t48_blk<0>(d);
            else acc = acc + t48_blk<0>(d);
            acc = acc + t48_blk<1>(d); acc = acc + t48_blk<2>(d); acc = acc + t48_blk<3>(d);
            d = dn;
        }
        return acc;
    }
    else if constexpr (PLAN == PLAN_GEMV) // four accumulators from zero, (c0+c2)+(c1+c3), 0 + 1*acc
    {
        const t48v4 z = {0.f, 0.f, 0.f, 0.f};
        t48v4 c0 = z, c1 = z, c2 = z, c3 = z;
        t48v16 d = T48_MFMA_B(0, B[0]);
#pragma unroll
        for (int m = 0; m < NM; m++)
        {
            t48v16 dn = d;
            if (m + 1 < NM)
            {
                const int mn = m + 1 < NM ? m + 1 : m;
                dn = T48_MFMA_B(mn, m > 0 ? t48_after(B[mn], c3) : B[mn]);
            }
            c0 = c0 + t48_blk<0>(d); c1 = c1 + t48_blk<1>(d); c2 = c2 + t48_blk<2>(d); c3 = c3 + t48_blk<3>(d);
            d = dn;
        }
        return z + ((c0 + c2) + (c1 + c3));
    }
    else if constexpr (PLAN == PLAN_TREE) // halving tree over k: its leaves of four are the four columns of one instruction
    {
        static_assert(NM == 8 || NM == 4 || NM == 2, "power-of-two trees");
        t48v4 e[NM];
        t48v16 d = T48_MFMA_B(0, B[0]);
#pragma unroll
        for (int m = 0; m < NM; m++)
        {
            t48v16 dn = d;
            if (m + 1 < NM)
            {
                const int mn = m + 1 < NM ? m + 1 : m;
                dn = T48_MFMA_B(mn, m > 0 ? t48_after(B[mn], e[m > 0 ? m - 1 : 0]) : B[mn]);
            }
            e[m] = (t48_blk<0>(d) + t48_blk<1>(d)) + (t48_blk<2>(d) + t48_blk<3>(d));
            d = dn;
        }
        return t48_tree<0, NM>(e);
    }
    else // packet tree (PLAN_VEC): element j of the packets summed by a halving tree over the packets, then (s0+s2)+(s1+s3)
    {
        static_assert(PLAN == PLAN_VEC && (NM == 8 || NM == 4), "power-of-two packet trees");
        // pairs of packets first: u[p][j] = packet 2p [j] + packet 2p+1 [j]
        t48v4 u[NM / 2][4];
#pragma unroll
        for (int p2 = 0; p2 < NM / 2; p2++)
        {
            const float b0 = p2 > 0 ? t48_after(B[2 * p2], u[p2 > 0 ? p2 - 1 : 0][3]) : B[2 * p2];
            const t48v16 d = T48_MFMA_B(2 * p2, b0), d1 = T48_MFMA_B(2 * p2 + 1, B[2 * p2 + 1]);
            u[p2][0] = t48_blk<0>(d) + t48_blk<0>(d1); u[p2][1] = t48_blk<1>(d) + t48_blk<1>(d1);
            u[p2][2] = t48_blk<2>(d) + t48_blk<2>(d1); u[p2][3] = t48_blk<3>(d) + t48_blk<3>(d1);
        }
        t48v4 sj[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            if constexpr (NM == 4) sj[j] = u[0][j] + u[1][j];
            else sj[j] = (u[0][j] + u[1][j]) + (u[2][j] + u[3][j]);
        }
        return (sj[0] + sj[2]) + (sj[1] + sj[3]);
    }
}
// fma arithmetic: init + sum_k M[rows][k] * s_c[k] as ONE k-ascending fma chain on the matrix cores (v_mfma_f32_16x16x4_f32 accumulates its
// four columns in order); a chain that starts a sum is started from -0, so that its first link is a plain product (admm_tile16.hip)
template <int CNT>
__device__ __forceinline__ t48v4 t48_chain(t48v4 acc, const float (&A)[CNT / 4], const float (&B)[CNT / 4])
{
#pragma unroll
    for (int m = 0; m < CNT / 4; m++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m], B[m], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ float t48_colmax(float v) // max over the four lanes (g = 0..3) that hold one column's rows
{
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ t48v4 t48_ld4(const float *p) { return *reinterpret_cast<const t48v4 *>(p); }
__device__ __forceinline__ void t48_st4(float *p, const t48v4 &v) { *reinterpret_cast<t48v4 *>(p) = v; }
__device__ __forceinline__ void t48_wt4(float *p, const t48v4 &v) // a write-through store of the sweeps
{
    if (T48_ABLATE != 2) *reinterpret_cast<t48v4 *>(p) = v;
}

// what the roles share
struct T48Ctx
{
    int lane, wid, g, c, inst, N, rowoff, base; // base: float offset of (instance, step 0, this lane's first row) in the row arrays
    bool valid;
    float rho;
    float *lds;
    t48v4 *dual; // dual[i * WAVE]: g | y of the 4 rows of step i (LDS)
    __device__ __forceinline__ void init(const RowParams &P, float *lds_, int wid_)
    {
        lds = lds_;
        lane = threadIdx.x & (WAVE - 1);
        wid = wid_;
        g = lane >> 4; c = lane & 15;
        const int raw = blockIdx.x * T48_COLS + c;
        valid = raw < P.batch;
        inst = valid ? raw : P.batch - 1; // loads of a column past the batch address the last instance; its stores go where they do no harm
        N = P.N;
        rowoff = (wid < 2 ? 16 * wid : T48_NX) + 4 * g;
        base = (inst * N) * WAVE + rowoff;
        rho = P.rho;
        dual = reinterpret_cast<t48v4 *>(lds + T48_HEAD) + wid * (N * WAVE) + lane;
    }
    // {lo, hi} of this lane's four rows at step i
    __device__ __forceinline__ void bounds(const RowParams &P, int i, t48v4 &lo, t48v4 &hi) const
    {
        const float *b = P.bounds + 2 * ((size_t)inst * P.bounds_inst_stride + (size_t)i * WAVE + rowoff);
        const t48v4 q0 = t48_ld4(b), q1 = t48_ld4(b + 4);
        lo = t48v4{q0[0], q0[2], q1[0], q1[2]};
        hi = t48v4{q0[1], q0[3], q1[1], q1[3]};
    }
};

// residual maxima of the three waves -> per-column convergence, identical in every wave (admm.cpp:91-109)
__device__ __forceinline__ void t48_check(const RowParams &P, const T48Ctx &C, float pri, float dua, int &rp, bool &act, int &st, float &r_ps, float &r_pi,
                                          float &r_ds, float &r_di)
{
    float *const res = C.lds + T48_RES + rp * (6 * T48_COLS);
    pri = t48_colmax(pri); dua = t48_colmax(dua);
    if (C.g == 0)
    {
        res[(2 * C.wid + 0) * T48_COLS + C.c] = pri;
        res[(2 * C.wid + 1) * T48_COLS + C.c] = dua;
    }
    t48_barrier();
    const float ps = fmaxf(res[0 * T48_COLS + C.c], res[2 * T48_COLS + C.c]), ds = fmaxf(res[1 * T48_COLS + C.c], res[3 * T48_COLS + C.c]) * C.rho;
    const float pi = res[4 * T48_COLS + C.c], di = res[5 * T48_COLS + C.c] * C.rho;
    t48_fence();
    const bool conv = (ps < P.abs_pri_tol) && (pi < P.abs_pri_tol) && (ds < P.abs_dua_tol) && (di < P.abs_dua_tol);
    if (act)
    {
        r_ps = ps; r_pi = pi; r_ds = ds; r_di = di;
        if (conv) { st = TINY_STATUS_SOLVED_; act = false; }
    }
    rp ^= 1;
}

// update_slack + update_dual + residual maxima of one step (admm.cpp:47-60, 69-70, 95-98): sv = [x ; u] rows, aold = g | y, bold = v | z
__device__ __forceinline__ void t48_slack_dual(const t48v4 &sv, const t48v4 &aold, const t48v4 &bold, const t48v4 &lo, const t48v4 &hi, t48v4 &tn, t48v4 &an,
                                               float &pri, float &dua)
{
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const float t0 = sv[j] + aold[j];
        const float t = __builtin_amdgcn_fmed3f(t0, lo[j], hi[j]); // lo := min(lo, hi) on the host
        an[j] = t0 - t;
        tn[j] = t;
        pri = fmaxf(pri, fabsf(sv[j] - t));
        dua = fmaxf(dua, fabsf(bold[j] - t));
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// x waves (wid 0, 1): 16 x rows each
// ------------------------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ void t48_x_role(const RowParams &P, float *lds, int wid)
{
    using PL = WavePlans<T48_NX, T48_NU>;
    T48Ctx C;
    C.init(P, lds, wid);
    const int g = C.g, c = C.c, row0 = 16 * wid, N = C.N;
    const float rho = C.rho;
    const bool valid = C.valid;
    float *const XB = lds + T48_XB, *const UB = lds + T48_UB, *const RB = lds + T48_RB;
    t48v16 negz;
#pragma unroll
    for (int e = 0; e < 16; e++) negz[e] = -0.f;
    asm volatile("" : "+v"(negz)); // a value the compiler cannot rebuild: as a known constant it lets the last product of every stage overwrite
                                   // the tuple and re-materialises it (16 scalar + 8 vector moves per stage, five stages per step pair)
    // gains: A operands, lane (g, c) supplies column 4m + g of tile row c
    const float *const mrow = P.mats + row0 + c;
    float A1[8], A2[4], A3[8], A45[4];
#pragma unroll
    for (int m = 0; m < 8; m++)
    {
        A1[m] = mrow[(4 * m + g) * WAVE];
        A3[m] = mrow[(T48_NX + T48_NU + 4 * m + g) * WAVE];
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
        A2[m] = mrow[(T48_NX + 4 * m + g) * WAVE];
        A45[m] = mrow[(2 * T48_NX + T48_NU + 4 * m + g) * WAVE];
    }
    const t48v4 qrow = t48_ld4(P.mats + (2 * T48_NX + 2 * T48_NU) * WAVE + C.rowoff);
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[C.inst];
    auto xref_at = [&](int i) -> t48v4 {
        if (P.xref_mode == 1)
        {
            int rw = wstart + i;
            rw = rw < P.table_rows ? rw : P.table_rows - 1;
            return t48_ld4(P.xref_table + rw * WAVE + C.rowoff);
        }
        return t48_ld4(P.xref + (size_t)C.inst * P.xref_inst_stride + i * WAVE + C.rowoff);
    };
    const bool cold = P.cold_start != 0, zdual = cold || (P.duals_zero != 0);
    const t48v4 z4 = {0.f, 0.f, 0.f, 0.f};
    float *const S = P.vzn; // the slack's working storage: vnew (it is the live-out vnew when the solve ends)

    // ---- live-in: duals into LDS, the slack v into the working array ----
#pragma unroll 1
    for (int i = 0; i < N; i++)
    {
        C.dual[i * WAVE] = zdual ? z4 : t48_ld4(P.gy + C.base + i * WAVE);
        if (valid) t48_st4(S + C.base + i * WAVE, cold ? z4 : t48_ld4(P.vz + C.base + i * WAVE));
    }
    const t48v4 x0 = t48_ld4(P.xu + C.base);
    t48v4 pterm;
    {
        // -(Xref_{N-1}^T Pinf) (admm.cpp:83); AP[m]: entry (k, r) = Pinf(k, r)
        float AP[8], xB[8];
#pragma unroll
        for (int m = 0; m < 8; m++) AP[m] = mrow[(2 * T48_NX + 2 * T48_NU + 1 + 4 * m + g) * WAVE];
        t48_put(XB, row0, g, c, xref_at(N - 1));
        t48_barrier();
        t48_fetch<32>(xB, XB, g, c);
        if constexpr (EXACT) pterm = -t48_dot<PL::TERM, 32>(AP, xB, negz);
        else pterm = -t48_chain<32>(t48v4{-0.f, -0.f, -0.f, -0.f}, AP, xB);
        t48_barrier_after(pterm); // XB is reused by the sweep
    }
    int st = TINY_STATUS_UNSOLVED_, itn = 1;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (!P.cold_start)
    {
        r_ps = P.res[4 * C.inst + 0]; r_pi = P.res[4 * C.inst + 1];
        r_ds = P.res[4 * C.inst + 2]; r_di = P.res[4 * C.inst + 3];
    }
    t48v4 pN = z4;
    bool ran_bwd = false, act = valid;
    int rp = 0;
#if T48_PROFILE
    float prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif

    for (int it = 0; it < P.max_iter; ++it)
    {
        float pri = 0.f, dua = 0.f;
        t48v4 t1 = z4;
        // write-through targets: a finished (or absent) column writes to an array the live-out rewrites anyway
        float *const s_dst = act ? S : P.qr, *const vz_dst = act ? P.vz : P.qr, *const xu_dst = act ? P.xu : P.qr;
        // ---------------- forward sweep ----------------
        {
            t48v4 s = x0;
            float xB[8];
            int par = 0;
            t48_put(XB, row0, g, c, s);
            t48v4 lo, hi, bold = t48_ld4(S + C.base);
            C.bounds(P, 0, lo, hi);
            t48_barrier();
            t48_fetch<32>(xB, XB, g, c);
#pragma unroll 1
            for (int i = 0; i < N - 1; i++)
            {
                const int off = C.base + i * WAVE;
                // the next step's bounds and old slack: issued before this step's stores
                t48v4 lo_n, hi_n;
                C.bounds(P, i + 1, lo_n, hi_n);
                const t48v4 bold_n = t48_ld4(S + off + WAVE);
                t48v4 acc; // Adyn x_i (admm.cpp:35)
                if constexpr (EXACT) acc = t48_dot<PL::FWD_XA, 32>(A1, xB, negz);
                else acc = t48_chain<32>(t48v4{-0.f, -0.f, -0.f, -0.f}, A1, xB);
                // the slack / dual update of the x rows needs nothing of u_i: it runs while the u wave finishes u_i
                const t48v4 aold = C.dual[i * WAVE];
                t48v4 tn, an;
                t48_slack_dual(s, aold, bold, lo, hi, tn, an, pri, dua);
                C.dual[i * WAVE] = act ? an : aold;
                t48_wt4(s_dst + off, tn);    // vnew_i
                t48_wt4(vz_dst + off, bold); // v_i, should this iteration converge
                t48_wt4(xu_dst + off, s);    // x_i of this sweep (live-out only)
                asm volatile("" : "+v"(acc));
                T48_STAMP(0);
                t48_barrier_after(acc); // u_i is there
                T48_STAMP(1);
                float uB[4];
                t48_fetch<16>(uB, UB, g, c);
                t48v4 xn; // x_{i+1} = Adyn x_i + Bdyn u_i
                if constexpr (EXACT) xn = acc + t48_dot<PL::FWD_XB, 16>(A2, uB, negz);
                else xn = t48_chain<16>(acc, A2, uB);
                t48_put(XB + (par ^ 1) * 512, row0, g, c, xn);
                asm volatile("" : "+v"(xn));
                T48_STAMP(2);
                t48_barrier(); // x_{i+1} is there
                T48_STAMP(3);
                par ^= 1;
                t48_fetch<32>(xB, XB + par * 512, g, c);
                s = xn; lo = lo_n; hi = hi_n; bold = bold_n;
            }
            {
                const int i = N - 1, off = C.base + i * WAVE;
                const t48v4 aold = C.dual[i * WAVE];
                t48v4 tn, an;
                t48_slack_dual(s, aold, bold, lo, hi, tn, an, pri, dua);
                C.dual[i * WAVE] = act ? an : aold;
                t48_wt4(s_dst + off, tn);
                t48_wt4(vz_dst + off, bold);
                t48_wt4(xu_dst + off, s);
                t1 = tn - an;
            }
        }
        if (act)
        {
            if constexpr (EXACT) pN = pterm - rho * t1; // admm.cpp:83-84
            else
            {
#pragma unroll
                for (int j = 0; j < 4; j++) pN[j] = __builtin_fmaf(-rho, t1[j], pterm[j]);
            }
            itn = it + 1;
        }
        if ((it + 1) % P.check_termination == 0) t48_check(P, C, pri, dua, rp, act, st, r_ps, r_pi, r_ds, r_di);
        if (__ballot(act) == 0) break; // the same columns in every wave
        // ---------------- backward sweep: linear cost + backward_pass_grad ----------------
        if (act) ran_bwd = true;
        if (T48_ABLATE != 1)
        {
            const int top = N - 2;
            float *const pd_dst = act ? P.pd : P.qr; // (after the check: a column that has just converged keeps the p of its last sweep)
            float pB[8], rB[4];
            int q = 0;
            t48v4 lin, tks = z4;
            auto linear = [&](const t48v4 &ai, const t48v4 &xr, const t48v4 &sn) -> t48v4 { // q_i = -(Xref_i .* Q) - rho (vnew_i - g_i) (admm.cpp:81-82)
                t48v4 l;
#pragma unroll
                for (int j = 0; j < 4; j++) l[j] = EXACT ? -(xr[j] * qrow[j]) - rho * (sn[j] - ai[j]) : __builtin_fmaf(-rho, sn[j] - ai[j], -(xr[j] * qrow[j]));
                return l;
            };
            auto fetch_r = [&](const float *src) { // Kinf^T r_i (admm.cpp:20)
                t48_fetch<16>(rB, src, g, c);
                if constexpr (EXACT) tks = t48_dot<PL::BWD_PK, 16>(A45, rB, negz);
            };
            lin = linear(C.dual[top * WAVE], xref_at(top), t48_ld4(S + C.base + top * WAVE));
            // the reference row and the slack of the next step (i - 1), loaded a step ahead
            t48v4 xr_n = xref_at(top > 0 ? top - 1 : 0), sn_n = t48_ld4(S + C.base + (top > 0 ? top - 1 : 0) * WAVE);
            t48_put(XB, row0, g, c, pN);
            t48_barrier();
            t48_fetch<32>(pB, XB, g, c);
            fetch_r(RB);
#pragma unroll 1
            for (int i = top; i >= 0; i--)
            {
                const int off = C.base + i * WAVE;
                const t48v4 xr = xr_n, sn = sn_n;
                const int i2 = i > 1 ? i - 2 : 0;
                xr_n = xref_at(i2);
                sn_n = t48_ld4(S + C.base + i2 * WAVE);
                t48v4 wv; // q + AmBKt p
                if constexpr (EXACT) wv = lin + t48_dot<PL::BWD_PA, 32>(A3, pB, negz);
                else wv = t48_chain<32>(lin, A3, pB);
#if T48_PROFILE
                asm volatile("" : "+v"(wv));
                T48_STAMP(7);
#endif
                t48v4 pn; // admm.cpp:20
                if constexpr (EXACT) pn = wv - tks;
                else pn = t48_chain<16>(wv, A45, rB); // x rows of M45 hold -Kinf^T (pack_gains, fast)
                t48_put(XB + (q ^ 1) * 512, row0, g, c, pn);
                t48_wt4(pd_dst + off, pn); // p_i of this sweep (live-out only)
                if (i > 0) lin = linear(C.dual[(i - 1) * WAVE], xr, sn);
                asm volatile("" : "+v"(pn), "+v"(lin));
                T48_STAMP(4);
                t48_barrier_after(pn); // p_i and r_{i-1} are there
                T48_STAMP(5);
                q ^= 1;
                t48_fetch<32>(pB, XB + q * 512, g, c);
                if (i > 0) fetch_r(RB + q * 256);
                asm volatile("" : "+v"(tks));
                T48_STAMP(6);
            }
        }
    }
    // ---------------- live-out (x, vnew are in place: every forward sweep wrote them) ----------------
    {
        const bool solved = (st == TINY_STATUS_SOLVED_);
        if (valid)
        {
#pragma unroll 1
            for (int i = 0; i < N; i++)
            {
                const int off = C.base + i * WAVE;
                const t48v4 sn = t48_ld4(S + off), xr = xref_at(i), av = C.dual[i * WAVE];
                t48v4 lq;
#pragma unroll
                for (int j = 0; j < 4; j++) lq[j] = EXACT ? -(xr[j] * qrow[j]) - rho * (sn[j] - av[j]) : __builtin_fmaf(-rho, sn[j] - av[j], -(xr[j] * qrow[j]));
                t48_st4(P.qr + off, lq);
                if (i == N - 1) t48_st4(P.pd + off, pN);
                else if (cold && !ran_bwd) t48_st4(P.pd + off, z4);
                if (!solved) t48_st4(P.vz + off, sn); // v = vnew happened; a solved instance keeps the stash
                t48_st4(P.gy + off, av);
            }
            if (wid == 0 && g == 0)
            {
                P.res[4 * C.inst + 0] = r_ps; P.res[4 * C.inst + 1] = r_pi;
                P.res[4 * C.inst + 2] = r_ds; P.res[4 * C.inst + 3] = r_di;
                P.status[C.inst] = st;
                P.iter[C.inst] = itn;
                if (!solved) atomicAdd(P.n_unsolved, 1);
            }
#if T48_PROFILE
            if (blockIdx.x == 0 && wid == 0 && C.lane == 0)
                for (int k = 0; k < 8; k++) P.res[k] = prof[k];
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// u wave (wid 2): the 16 u rows; d comes back from the pd array one step ahead
// ------------------------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__device__ __forceinline__ void t48_u_role(const RowParams &P, float *lds)
{
    using PL = WavePlans<T48_NX, T48_NU>;
    T48Ctx C;
    C.init(P, lds, 2);
    const int g = C.g, c = C.c, N = C.N;
    const float rho = C.rho;
    const bool valid = C.valid;
    float *const XB = lds + T48_XB, *const UB = lds + T48_UB, *const RB = lds + T48_RB;
    t48v16 negz;
#pragma unroll
    for (int e = 0; e < 16; e++) negz[e] = -0.f;
    asm volatile("" : "+v"(negz)); // a value the compiler cannot rebuild: as a known constant it lets the last product of every stage overwrite
                                   // the tuple and re-materialises it (16 scalar + 8 vector moves per stage, five stages per step pair)
    const float *const mrow = P.mats + T48_NX + c;
    float A1[8], A3[8], A45[4]; // Kinf rows | Bdyn^T rows | Quu_inv rows
#pragma unroll
    for (int m = 0; m < 8; m++)
    {
        A1[m] = mrow[(4 * m + g) * WAVE];
        A3[m] = mrow[(T48_NX + T48_NU + 4 * m + g) * WAVE];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) A45[m] = mrow[(2 * T48_NX + T48_NU + 4 * m + g) * WAVE];
    const bool cold = P.cold_start != 0, zdual = cold || (P.duals_zero != 0);
    const t48v4 z4 = {0.f, 0.f, 0.f, 0.f};
    float *const S = P.vzn;

#pragma unroll 1
    for (int i = 0; i < N; i++)
    {
        C.dual[i * WAVE] = zdual ? z4 : t48_ld4(P.gy + C.base + i * WAVE);
        if (valid) t48_st4(S + C.base + i * WAVE, cold ? z4 : t48_ld4(P.vz + C.base + i * WAVE));
    }
    t48_barrier(); // the x waves' terminal-term exchange
    t48_barrier();
    int st = TINY_STATUS_UNSOLVED_;
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    bool ran_bwd = false, act = valid;
    int rp = 0;

    for (int it = 0; it < P.max_iter; ++it)
    {
        float pri = 0.f, dua = 0.f;
        const bool d_zero = cold && it == 0; // the first sweep of a cold start: d = 0, the pd array is not read
        float *const s_dst = act ? S : P.qr, *const vz_dst = act ? P.vz : P.qr, *const xu_dst = act ? P.xu : P.qr;
        // ---------------- forward sweep ----------------
        {
            float xB[8];
            int par = 0;
            t48v4 lo, hi, bold = t48_ld4(S + C.base);
            C.bounds(P, 0, lo, hi);
            t48v4 di = d_zero ? z4 : t48_ld4(P.pd + C.base); // d_0 (the last backward sweep's, or the live-in)
            t48_barrier();
            t48_fetch<32>(xB, XB, g, c);
#pragma unroll 1
            for (int i = 0; i < N - 1; i++)
            {
                const int off = C.base + i * WAVE;
                t48v4 lo_n, hi_n;
                C.bounds(P, i + 1, lo_n, hi_n);
                const t48v4 bold_n = t48_ld4(S + off + WAVE);
                const t48v4 d_n = (d_zero || i + 1 >= N - 1) ? z4 : t48_ld4(P.pd + off + WAVE);
                t48v4 un; // u_i = -Kinf x_i - d_i (admm.cpp:31)
                if constexpr (EXACT)
                {
                    const t48v4 acc = t48_dot<PL::FWD_U, 32>(A1, xB, negz);
#pragma unroll
                    for (int j = 0; j < 4; j++) un[j] = -acc[j] - di[j];
                }
                else un = t48_chain<32>(t48v4{-0.f, -0.f, -0.f, -0.f}, A1, xB) - di; // u rows of M1 hold -Kinf (pack_gains, fast)
                t48_put(UB, 0, g, c, un);
                t48_barrier_after(un); // u_i is there
                const t48v4 aold = C.dual[i * WAVE];
                t48v4 tn, an;
                t48_slack_dual(un, aold, bold, lo, hi, tn, an, pri, dua);
                C.dual[i * WAVE] = act ? an : aold;
                t48_wt4(s_dst + off, tn);    // znew_i
                t48_wt4(vz_dst + off, bold); // z_i, should this iteration converge
                t48_wt4(xu_dst + off, un);   // u_i of this sweep (live-out only)
                t48_barrier(); // x_{i+1} is there
                par ^= 1;
                t48_fetch<32>(xB, XB + par * 512, g, c);
                lo = lo_n; hi = hi_n; bold = bold_n; di = d_n;
            }
            {
                const int i = N - 1, off = C.base + i * WAVE; // the u rows have no step N - 1: sv = 0 there (admm_waveres.hip does the same)
                const t48v4 aold = C.dual[i * WAVE];
                t48v4 tn, an;
                t48_slack_dual(z4, aold, bold, lo, hi, tn, an, pri, dua);
                C.dual[i * WAVE] = act ? an : aold;
                t48_wt4(s_dst + off, tn);
                t48_wt4(vz_dst + off, bold);
                t48_wt4(xu_dst + off, z4);
            }
        }
        if ((it + 1) % P.check_termination == 0) t48_check(P, C, pri, dua, rp, act, st, r_ps, r_pi, r_ds, r_di);
        if (__ballot(act) == 0) break;
        // ---------------- backward sweep ----------------
        if (act) ran_bwd = true;
        if (T48_ABLATE != 1)
        {
            const int top = N - 2;
            float *const pd_dst = act ? P.pd : P.qr;
            float pB[8], wB[4];
            int q = 0;
            auto linear = [&](const t48v4 &ai, const t48v4 &sn) -> t48v4 { // r_i = -rho (znew_i - y_i): -0 + ... keeps the sign of a zero difference (admm.cpp:80)
                t48v4 l;
#pragma unroll
                for (int j = 0; j < 4; j++) l[j] = EXACT ? -0.f - rho * (sn[j] - ai[j]) : __builtin_fmaf(-rho, sn[j] - ai[j], -0.f);
                return l;
            };
            t48v4 lin = linear(C.dual[top * WAVE], t48_ld4(S + C.base + top * WAVE));
            t48v4 sn_n = t48_ld4(S + C.base + (top > 0 ? top - 1 : 0) * WAVE);
            t48_put(RB, 0, g, c, lin);
            t48_barrier();
            t48_fetch<32>(pB, XB, g, c);
#pragma unroll 1
            for (int i = top; i >= 0; i--)
            {
                const int off = C.base + i * WAVE;
                const t48v4 sn = sn_n;
                sn_n = t48_ld4(S + C.base + (i > 1 ? i - 2 : 0) * WAVE);
                t48v4 wv; // Bdyn^T p + r
                if constexpr (EXACT) wv = lin + t48_dot<PL::BWD_TMP, 32>(A3, pB, negz);
                else wv = t48_chain<32>(lin, A3, pB);
                t48_put(UB, 0, g, c, wv); // d_i = Quu_inv (Bdyn^T p + r) (admm.cpp:19): this exchange stays inside the wave
                if (i > 0)
                {
                    lin = linear(C.dual[(i - 1) * WAVE], sn);
                    t48_put(RB + (q ^ 1) * 256, 0, g, c, lin);
                }
                t48_barrier_after(lin); // p_i and r_{i-1} are there; d_i, which nobody waits for, is computed behind the barrier
                q ^= 1;
                t48_fetch<16>(wB, UB, g, c);
                t48_fetch<32>(pB, XB + q * 512, g, c);
                t48v4 dd;
                if constexpr (!EXACT) dd = t48_chain<16>(t48v4{-0.f, -0.f, -0.f, -0.f}, A45, wB);
                else if constexpr (PL::GEMV) dd = z4 + (z4 + t48_dot<PLAN_SEQ, 16>(A45, wB, negz)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
                else dd = t48_dot<PL::BWD_D, 16>(A45, wB, negz);
                t48_wt4(pd_dst + off, dd); // d_i of this sweep: the next forward sweep reads it back
            }
        }
    }
    // ---------------- live-out (u, znew are in place) ----------------
    {
        const bool solved = (st == TINY_STATUS_SOLVED_);
        if (valid)
        {
#pragma unroll 1
            for (int i = 0; i < N; i++)
            {
                const int off = C.base + i * WAVE;
                const t48v4 sn = t48_ld4(S + off), av = C.dual[i * WAVE];
                t48v4 lr;
#pragma unroll
                for (int j = 0; j < 4; j++) lr[j] = EXACT ? -0.f - rho * (sn[j] - av[j]) : __builtin_fmaf(-rho, sn[j] - av[j], -0.f);
                t48_st4(P.qr + off, i < N - 1 ? lr : z4);
                if (i == N - 1) t48_st4(P.pd + off, z4);
                else if (cold && !ran_bwd) t48_st4(P.pd + off, z4);
                if (!solved) t48_st4(P.vz + off, sn);
                t48_st4(P.gy + off, av);
            }
        }
    }
}

template <bool EXACT>
__global__ __launch_bounds__(3 * WAVE, 1) void admm_tile48_kernel(const RowParams P)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        const int inst = blockIdx.x * T48_COLS + threadIdx.x;
        if (threadIdx.x < T48_COLS && inst < P.batch)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        return;
    }
    if (wid < 2) t48_x_role<EXACT>(P, lds, wid);
    else t48_u_role<EXACT>(P, lds);
}
} // namespace

constexpr int T48_MAX_N = 50;
bool tile48_supported(int nx, int nu, int N) { return nx == T48_NX && nu == T48_NU && N >= 2 && N <= T48_MAX_N; }

template <bool EXACT>
static hipError_t t48_launch(int N, const RowParams &P, hipStream_t stream)
{
    const size_t ldsb = t48_lds_bytes(N);
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((done.load() >> dev) & 1ull))
    {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&admm_tile48_kernel<EXACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t48_lds_bytes(T48_MAX_N));
        if (e != hipSuccess) return e;
        if (dev < 64) done.fetch_or(1ull << dev);
    }
    hipLaunchKernelGGL(admm_tile48_kernel<EXACT>, dim3((P.batch + T48_COLS - 1) / T48_COLS), dim3(3 * WAVE), ldsb, stream, P);
    return hipGetLastError();
}

hipError_t launch_admm_tile48(int N, bool exact, const RowParams &P, hipStream_t stream)
{
    if (N < 2 || N > T48_MAX_N) return hipErrorInvalidValue;
    return exact ? t48_launch<true>(N, P, stream) : t48_launch<false>(N, P, stream);
}

} // namespace tinympc
