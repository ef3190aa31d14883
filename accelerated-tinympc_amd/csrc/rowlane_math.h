// rowlane_math.h — per-step arithmetic of the "row lane" mapping (one DPP row of 16 lanes = one instance; lane r owns
// row r of the stacked vector [x ; u]), shared by the register-resident solver (admm_rowlane.hip) and the single-function
// kernels (admm_steps.hip).  See admm_rowlane.hip for the mapping and the two arithmetic modes.
#pragma once
#include "tinympc_internal.h"
#include "dpp_ops_gen.h"

#ifndef TINY_OPT_SEL
#define TINY_OPT_SEL 0
#endif
#ifndef TINY_OPT_DDNOW
#define TINY_OPT_DDNOW 0
#endif

namespace tinympc
{

// DPP row-broadcast chains (dpp_products, dpp_fma_dot, dpp_fma_acc) come from dpp_ops_gen.h: hipcc neither folds
// v_mov_b32_dpp into v_fmac_f32 nor keeps consecutive asm statements adjacent, so every chain is ONE generated inline-asm
// statement (tools/gen_dpp_ops.py).  A VALU write of the broadcast source followed by a DPP read needs two wait states
// the compiler does not track inside asm: every chain starts with s_nop 1.

// ---------------------------------------------------------------------------------------------
// Reduction orders of the reference build.  Eigen 3.4.90 picks them at compile time from storage order, sizes and
// the SSE2 packet size (4 floats):
//   SEQ  ((t0+t1)+t2)+...            packet-evaluated lazy products (result rows a multiple of 4: etor_product_packet_impl)
//   TREE T(lo,n) = T(lo,n/2) + T(lo+n/2, n-n/2)   coefficient-evaluated, completely unrolled redux (redux_novec_unroller),
//        taken while 3n-1 <= EIGEN_UNROLLING_LIMIT = 110, else SEQ
//   VEC  products grouped in packets of 4, packets summed by the same halving tree (redux_vec_unroller), then
//        predux (s0+s2)+(s1+s3), then the n%4 leftover (TREE) added; TREE when n < 4
// The parity tests check the result bit for bit against vectors produced by the compiled reference.
// ---------------------------------------------------------------------------------------------
enum : int { PLAN_SEQ = 0, PLAN_TREE = 1, PLAN_VEC = 2 };
constexpr int plan_novec(int n) { return (3 * n - 1 <= 110) ? PLAN_TREE : PLAN_SEQ; }
constexpr int plan_vec(int n) { return n < 4 ? plan_novec(n) : PLAN_VEC; }

template <int LO, int CNT, int NN>
__device__ __forceinline__ float tree_sum(const float (&t)[NN])
{
    if constexpr (CNT == 1) return t[LO];
    else
    {
        constexpr int H = CNT / 2;
        return tree_sum<LO, H>(t) + tree_sum<LO + H, CNT - H>(t);
    }
}
template <int PLO, int PCNT, int L, int NN>
__device__ __forceinline__ float ptree_sum(const float (&t)[NN]) // lane L of the packets [PLO, PLO+PCNT)
{
    if constexpr (PCNT == 1) return t[4 * PLO + L];
    else
    {
        constexpr int H = PCNT / 2;
        return ptree_sum<PLO, H, L>(t) + ptree_sum<PLO + H, PCNT - H, L>(t);
    }
}
template <int PLAN, int NN>
__device__ __forceinline__ float reduce(const float (&t)[NN])
{
    if constexpr (NN == 1) return t[0];
    else if constexpr (PLAN == PLAN_SEQ)
    {
        float acc = t[0];
#pragma unroll
        for (int k = 1; k < NN; k++) acc = acc + t[k];
        return acc;
    }
    else if constexpr (PLAN == PLAN_TREE) return tree_sum<0, NN>(t);
    else
    {
        constexpr int NPK = NN / 4;
        const float s0 = ptree_sum<0, NPK, 0>(t), s1 = ptree_sum<0, NPK, 1>(t), s2 = ptree_sum<0, NPK, 2>(t),
                    s3 = ptree_sum<0, NPK, 3>(t);
        float res = (s0 + s2) + (s1 + s3); // SSE2 predux
        if constexpr (NN % 4 != 0) res = res + tree_sum<4 * NPK, NN - 4 * NPK>(t);
        return res;
    }
}

// max over the 16 lanes of a DPP row; every lane gets the result
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_max(float v)
{
    v = fmaxf(v, dpp_mov<0x128>(v)); // row_ror:8
    v = fmaxf(v, dpp_mov<0x124>(v)); // row_ror:4
    v = fmaxf(v, dpp_mov<0x122>(v)); // row_ror:2
    v = fmaxf(v, dpp_mov<0x121>(v)); // row_ror:1
    return v;
}

// ---------------------------------------------------------------------------------------------
// Storage precision (template parameter H16 of the row kernels).  H16 = false: the work arrays are fp32, as in the
// reference.  H16 = true ("fp16 storage, fp32 arithmetic", BASELINE.json configs[4]): every per-instance horizon array
// (the twelve work arrays, Xref, the bounds) is IEEE binary16 in HBM; every ASSIGNMENT to a work array rounds the stored
// value (round to nearest even, subnormals kept) and the fp32 registers carry that rounded value on; products, sums and
// the residual reductions stay fp32.  oracle/ restates exactly this (the _h16 instantiation) and exact arithmetic is
// bit-identical to it.
// ---------------------------------------------------------------------------------------------
template <bool H16>
__device__ __forceinline__ float rnd(float v)
{
    if constexpr (H16)
    {
        // the fp32 result is rounded first, THEN stored as binary16 (what "fp16 storage" means).  Without the barrier
        // hipcc folds fptrunc(fmul/fma) into v_fma_mixlo_f16, which rounds the exact product once, directly to binary16.
        asm("" : "+v"(v));
        return (float)(_Float16)v;
    }
    else return v;
}
template <bool H16>
__device__ __forceinline__ float ldw(const float *base, int o)
{
    if constexpr (H16) return (float)reinterpret_cast<const _Float16 *>(base)[o];
    else return base[o];
}
template <bool H16>
__device__ __forceinline__ void stw(float *base, int o, float v)
{
    if constexpr (H16)
    {
        asm("" : "+v"(v)); // see rnd()
        reinterpret_cast<_Float16 *>(base)[o] = (_Float16)v;
    }
    else base[o] = v;
}
// {lo, hi} entry e of the bounds table
template <bool H16>
__device__ __forceinline__ float2 ld_bounds(const float *base, int e)
{
    if constexpr (H16)
    {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 v = reinterpret_cast<const h2 *>(base)[e];
        return make_float2((float)v.x, (float)v.y);
    }
    else return reinterpret_cast<const float2 *>(base)[e];
}

// max(acc, |d|) as ONE instruction: inside a rolled loop hipcc cannot prove the loop-carried maximum canonical and puts a
// v_max_f32 acc, acc, acc in front of every fmaxf (acc is never a signalling NaN here: it starts at 0)
__device__ __forceinline__ float max_abs(float acc, float d)
{
    float r;
    asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(acc), "v"(d));
    return r;
}

template <int NX, int NU>
struct RowPlans
{
    static_assert(NX > 1 && NU >= 1 && NX + NU <= 16, "rowlane kernel needs 1 < nx, nx + nu <= 16");
    static_assert(!(NU >= 8 && NX >= 8), "Eigen switches to its GEMV kernel there; not restated");
    // For a result with rows > 4 and rows % 4 != 0 the reference's order depends on the 16-byte alignment of each
    // destination column (Eigen LinearVectorized assignment; measured for nu = 7): no bitwise claim is possible there.
    static_assert((NX <= 4 || NX % 4 == 0) && (NU <= 4 || NU % 4 == 0), "exact arithmetic is defined for nx, nu <= 4 or multiples of 4");
    // forward_pass (admm.cpp:31,35)
    static constexpr int FWD_U = (NU > 1 && NU % 4 == 0) ? PLAN_SEQ : (NU == 1 ? plan_vec(NX) : plan_novec(NX));
    static constexpr int FWD_XA = (NX % 4 == 0) ? PLAN_SEQ : plan_novec(NX);
    static constexpr int FWD_XB = (NX % 4 == 0) ? PLAN_SEQ : plan_novec(NU);
    // update_linear_cost terminal term (admm.cpp:83)
    static constexpr int TERM = plan_vec(NX);
    // backward_pass_grad (admm.cpp:19-20)
    static constexpr int BWD_TMP = plan_vec(NX);
    static constexpr int BWD_D = (NU > 1 && NU % 4 == 0) ? PLAN_SEQ : plan_novec(NU);
    static constexpr int BWD_PA = (NU == 1 && NX % 4 == 0) ? PLAN_SEQ : plan_novec(NX);
    static constexpr int BWD_PK = plan_vec(NU);
};

// Gain rows of one lane (packed by the host, tinympc_batch.hip: pack_gains):
//   M1[k]  x rows A(r,k)      | u rows K(m,k) (exact) / -K(m,k) (fast)     M2[m]  x rows B(r,m) | u rows 0
//   M3[k]  x rows AmBKt(r,k)  | u rows  B(k,m)            M45[m] x rows K(m,r) (exact) / -K(m,r) (fast) | u rows Quu_inv(mr,m)
template <int NX, int NU>
struct RowGains
{
    float M1[NX], M2[NU], M3[NX], M45[NU];
    __device__ __forceinline__ void load(const float *mats, int r16)
    {
        const float *m = mats + r16;
#pragma unroll
        for (int k = 0; k < NX; k++) M1[k] = m[(k) * 16];
#pragma unroll
        for (int k = 0; k < NU; k++) M2[k] = m[(NX + k) * 16];
#pragma unroll
        for (int k = 0; k < NX; k++) M3[k] = m[(NX + NU + k) * 16];
#pragma unroll
        for (int k = 0; k < NU; k++) M45[k] = m[(2 * NX + NU + k) * 16];
    }
};

// forward_pass step (admm.cpp:31,35): from s = x_i (x rows) and ci = d_i (u rows) compute
//   sv = [x_i ; u_i]  with u_i = -Kinf*x_i - d_i,   xn = x_{i+1} = Adyn*x_i + Bdyn*u_i (x rows)
template <int NX, int NU, bool EXACT, bool H16 = false>
__device__ __forceinline__ void lqr_step(const RowGains<NX, NU> &G, bool is_x, bool is_u, float s, float ci, float &sv, float &xn)
{
    using PL = RowPlans<NX, NU>;
    if constexpr (EXACT)
    {
        float t[NX];
        dpp_products<0, NX>(t, s, G.M1);
        float acc;
        if constexpr (PL::FWD_U == PL::FWD_XA) acc = reduce<PL::FWD_XA>(t);
        else acc = is_x ? reduce<PL::FWD_XA>(t) : reduce<PL::FWD_U>(t);
        // u rows of M1 hold +Kinf: the SUM is negated, as in the reference's -(K x) - d.  (Negated gains would give the same
        // value but the other sign of zero when products of mixed-sign zeros are summed.)
        const float un = rnd<H16>(-acc - ci);
        float t2[NU];
        dpp_products<NX, NU>(t2, un, G.M2);
        xn = rnd<H16>(acc + reduce<PL::FWD_XB>(t2));
        sv = is_u ? un : s;
    }
    else
    {
        float acc = dpp_fma_dot<0, NX>(s, G.M1);
        const float un = rnd<H16>(acc - ci);
        dpp_fma_acc<NX, NU>(acc, un, G.M2);
        xn = rnd<H16>(acc);
        sv = is_u ? un : s;
    }
}

// plant step of the closed loop (examples/quadrotor_hovering.cpp:110-111): xn = Adyn*x + Bdyn*u with sv = [x ; u], always in
// the reference's own order whatever the arithmetic mode of the solver, so that the on-chip closed loop and the
// step-by-step one (plant_step_kernel, tinympc_batch.hip) continue from identical states.  The order is Eigen's for
// "x1 = work.Adyn * x0 + work.Bdyn * work.u.col(0)" (pinned against that expression compiled from the reference, tests/test_oracle.py:
// test_plant_step_bit_exact_vs_compiled_reference): a product whose rows and depth are both >= 8 goes through the column-major GEMV kernel, whose row
// accumulator starts at +0 (visible in the sign of a zero result); otherwise the lazy product's plain sequential sum.
template <int NX, int NU>
__device__ __forceinline__ float plant_step(const RowGains<NX, NU> &G, float sv)
{
    using PL = RowPlans<NX, NU>;
    static_assert(PL::FWD_XA == PLAN_SEQ && PL::FWD_XB == PLAN_SEQ, "the plant kernel sums sequentially");
    static_assert(!(NX >= 8 && NU >= 8), "Bdyn*u would take the GEMV kernel too; not needed for nx + nu <= 16");
    float t[NX], t2[NU];
    dpp_products<0, NX>(t, sv, G.M1);
    dpp_products<NX, NU>(t2, sv, G.M2);
    float a = reduce<PLAN_SEQ>(t);
    if constexpr (NX >= 8)
    {
        float z = 0.f;
        asm("" : "+v"(z)); // keep the +0 start of the GEMV accumulator: (+0) + (-0) = +0
        a = z + t[0];
#pragma unroll
        for (int k = 1; k < NX; k++) a = a + t[k];
    }
    return a + reduce<PLAN_SEQ>(t2);
}

// backward_pass_grad step (admm.cpp:19-20): from p = p_{i+1} (x rows) and lin = [q_i ; r_i] compute
//   pn = p_i = q_i + AmBKt*p_{i+1} - Kinf^T*r_i (x rows),   dd = d_i = Quu_inv*(Bdyn^T*p_{i+1} + r_i) (u rows)
template <int NX, int NU, bool EXACT, bool H16 = false>
__device__ __forceinline__ void riccati_step(const RowGains<NX, NU> &G, bool is_x, float p, float lin, float &pn, float &dd)
{
    using PL = RowPlans<NX, NU>;
    if constexpr (EXACT)
    {
        float t[NX];
        dpp_products<0, NX>(t, p, G.M3);
        float dot;
        if constexpr (PL::BWD_PA == PL::BWD_TMP) dot = reduce<PL::BWD_PA>(t);
        else
        {
#if TINY_OPT_SEL
            // x rows and u rows sum the same products in different orders.  Both sums are computed by every lane and
            // selected: left to itself hipcc turns the select into two EXEC-masked branches (one dependent chain each,
            // nothing to overlap them with, 7 scalar instructions per step around them)
            float rx = reduce<PL::BWD_PA>(t), ru = reduce<PL::BWD_TMP>(t);
            asm volatile("" : "+v"(rx), "+v"(ru));
            dot = is_x ? rx : ru;
#else
            dot = is_x ? reduce<PL::BWD_PA>(t) : reduce<PL::BWD_TMP>(t);
#endif
        }
        const float wv = lin + dot;            // q + AmBKt*p  |  Bdyn^T*p + r
        float tk[NU], td[NU];
        dpp_products<NX, NU>(tk, lin, G.M45);  // Kinf^T * r
        dpp_products<NX, NU>(td, wv, G.M45);   // Quu_inv * (Bdyn^T p + r)
        pn = rnd<H16>(wv - reduce<PL::BWD_PK>(tk));
        dd = rnd<H16>(reduce<PL::BWD_D>(td));
#if TINY_OPT_DDNOW
        // d_i is off the critical chain of the sweep: without this hipcc postpones its three adds to the end of the sweep
        // and keeps (spills) the four products of every step until then
        asm volatile("" : "+v"(dd));
#endif
    }
    else
    {
        float acc = lin;
        dpp_fma_acc<0, NX>(acc, p, G.M3);
        dd = rnd<H16>(dpp_fma_dot<NX, NU>(acc, G.M45));  // u rows: Quu_inv
        dpp_fma_acc<NX, NU>(acc, lin, G.M45);             // x rows: -Kinf^T
        pn = rnd<H16>(acc);
    }
}

// The term the reference comments out at the end of admm.cpp:20, "+ coeff_d2p * d.col(i)": Eigen assigns the rest of the
// expression to p.col(i) first (one store, hence the rounding of pn by the caller) and then adds the product, which it
// evaluates into a temporary in sequential order (measured; the test suite pins it against Eigen).  CD[m] = coeff_d2p(r, m) on x rows.
template <int NX, int NU, bool EXACT, bool H16 = false>
__device__ __forceinline__ float d2p_term(const float (&CD)[NU], float pn, float dd)
{
    if constexpr (EXACT)
    {
        float t[NU];
        dpp_products<NX, NU>(t, dd, CD);
        return rnd<H16>(pn + reduce<PLAN_SEQ>(t));
    }
    else
    {
        float acc = pn;
        dpp_fma_acc<NX, NU>(acc, dd, CD);
        return rnd<H16>(acc);
    }
}

// cost term of a step as lin_cost() wants it: x rows keep c = -(Xref.*Q), u rows (where c holds d) get NEGATIVE zero, so that
// cq - rho*t1 equals the reference's r = -rho*(znew - y) also in the sign of a zero (one v_and_or_b32)
__device__ __forceinline__ float cost_term(float c, bool is_x)
{
    const unsigned keep = is_x ? 0xffffffffu : 0u, set = is_x ? 0u : 0x80000000u;
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, c) & keep) | set);
}

// [q_i ; r_i] of update_linear_cost (admm.cpp:80-82): cq = -(Xref_i .* Q) on x rows and 0 on u rows, t1 = snew - dual
template <bool EXACT, bool H16 = false>
__device__ __forceinline__ float lin_cost(float cq, float rho, float t1)
{
    if constexpr (EXACT) return rnd<H16>(cq - rho * t1);
    else return rnd<H16>(__builtin_fmaf(-rho, t1, cq));
}

// -(Xref_{N-1}^T Pinf) (admm.cpp:83), x rows; PT[k] = Pinf(k, r)
template <int NX, int NU, bool EXACT, bool H16 = false>
__device__ __forceinline__ float terminal_term(const float *mats, int r16, float xrN)
{
    float PT[NX];
#pragma unroll
    for (int k = 0; k < NX; k++) PT[k] = mats[(2 * NX + 2 * NU + 1 + k) * 16 + r16];
    if constexpr (EXACT)
    {
        float t[NX];
        dpp_products<0, NX>(t, xrN, PT);
        return rnd<H16>(-reduce<RowPlans<NX, NU>::TERM>(t));
    }
    else
        return rnd<H16>(-dpp_fma_dot<0, NX>(xrN, PT));
}

} // namespace tinympc
