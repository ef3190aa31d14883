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
#include <cstdlib>

namespace tinympc
{

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef TINY_T16_ABLATE
#define TINY_T16_ABLATE 0
#endif
#ifndef TINY_T16_CQ
#define TINY_T16_CQ 1 // the staged reference table holds -(Xref o Q) (round 4): no Q registers and no multiply in the sweeps
#endif
#ifndef TINY_T16_BADDR
#define TINY_T16_BADDR 1 // the lane's bounds-table addresses are remade per iteration instead of being reloaded from scratch (round 4)
#endif
#ifndef TINY_T16_PADTAB
#define TINY_T16_PADTAB 1 // the staged reference table carries N - 1 copies of its last row: no per-step clamp of the window's row (round 4)
#endif
#ifndef TINY_T16_XPOSE
#define TINY_T16_XPOSE 1 // live-out rows transposed across the four lane groups in registers (v_permlane32_swap / v_permlane16_swap) instead of through LDS (round 4)
#endif
#ifndef TINY_T16_SCHED
#define TINY_T16_SCHED 1 // the scheduling fences pay in exact arithmetic only (measured: exact 2.01 -> 1.88 ms, fma 1.02 -> 1.08)
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

// ---- packed sums (round 3).  An MFMA result holds, for gain column b, the products of output registers V = 0..3 in the four
// CONSECUTIVE registers 4b .. 4b+3: a natural 4-vector.  Where the four rows of a lane are summed in the same order the sums
// are written over such 4-vectors, and every 4-vector add is two v_pk_add_f32 (registers (0,1) and (2,3); each half is the IEEE
// fp32 add of v_add_f32, separately rounded: the results stay bitwise those of the reference).  With ONE wave per SIMD — all
// this kernel's register budget allows — a wave issues a vector instruction every 4 clocks whatever it is, so packed adds double
// the add rate (tools/micro/pk_rate.hip, profiles/r03_pk_rate.txt).  The two halves of a 4-vector add are independent, which
// also keeps a packed result one instruction away from its use: gfx950 needs one wait state there, and a lone dependent chain
// of v_pk_add_f32 would pay it as an s_nop per add.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int B> // products of gain column B of an MFMA result, rows V = 0..3
__device__ __forceinline__ f32x4 blk(const f32x16 &p) { return __builtin_shufflevector(p, p, 4 * B, 4 * B + 1, 4 * B + 2, 4 * B + 3); }
__device__ __forceinline__ void gather12v(f32x4 (&t)[12], const f32x16 &p0, const f32x16 &p1, const f32x16 &p2)
{
    t[0] = blk<0>(p0); t[1] = blk<1>(p0); t[2] = blk<2>(p0); t[3] = blk<3>(p0);
    t[4] = blk<0>(p1); t[5] = blk<1>(p1); t[6] = blk<2>(p1); t[7] = blk<3>(p1);
    t[8] = blk<0>(p2); t[9] = blk<1>(p2); t[10] = blk<2>(p2); t[11] = blk<3>(p2);
}
__device__ __forceinline__ void gather4v(f32x4 (&t)[4], const f32x16 &p)
{
    t[0] = blk<0>(p); t[1] = blk<1>(p); t[2] = blk<2>(p); t[3] = blk<3>(p);
}
// the reduction plans of rowlane_math.h over 4-vectors of rows
template <int LO, int CNT, int NN>
__device__ __forceinline__ f32x4 tree_sum4(const f32x4 (&t)[NN])
{
    if constexpr (CNT == 1) return t[LO];
    else
    {
        constexpr int H = CNT / 2;
        return tree_sum4<LO, H>(t) + tree_sum4<LO + H, CNT - H>(t);
    }
}
template <int PLO, int PCNT, int L, int NN>
__device__ __forceinline__ f32x4 ptree_sum4(const f32x4 (&t)[NN])
{
    if constexpr (PCNT == 1) return t[4 * PLO + L];
    else
    {
        constexpr int H = PCNT / 2;
        return ptree_sum4<PLO, H, L>(t) + ptree_sum4<PLO + H, PCNT - H, L>(t);
    }
}
template <int PLAN, int NN>
__device__ __forceinline__ f32x4 reduce4(const f32x4 (&t)[NN])
{
    if constexpr (NN == 1) return t[0];
    else if constexpr (PLAN == PLAN_SEQ)
    {
        f32x4 acc = t[0];
#pragma unroll
        for (int k = 1; k < NN; k++) acc = acc + t[k];
        return acc;
    }
    else if constexpr (PLAN == PLAN_TREE) return tree_sum4<0, NN>(t);
    else
    {
        static_assert(NN % 4 == 0, "packed VEC plan: whole packets only");
        constexpr int NPK = NN / 4;
        const f32x4 s0 = ptree_sum4<0, NPK, 0>(t), s1 = ptree_sum4<0, NPK, 1>(t), s2 = ptree_sum4<0, NPK, 2>(t), s3 = ptree_sum4<0, NPK, 3>(t);
        return (s0 + s2) + (s1 + s3);
    }
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

    // The matrix-core part of a step is issued separately from the sums that consume it (round 3): the sweeps put independent
    // vector work of the neighbouring step between the two, in the 40 .. 100 clocks the products take to arrive.
    struct InFlight
    {
        f32x16 p0, p1, p2, pk; // exact: products of the three state slices (and of Kinf^T r in the backward step)
        f32x4 acc;             // fma: the running chain
    };

    // forward_pass step (admm.cpp:31,35): s = x_i (registers 0..2), di = d_i (u row)  ->  un = u_i, xn = x_{i+1}
    __device__ __forceinline__ void lqr_issue(const float (&s)[3], InFlight &F) const
    {
        if constexpr (EXACT)
        {
            F.p0 = TINY_MFMA1(A1[0], s[0], negz); F.p1 = TINY_MFMA1(A1[1], s[1], negz); F.p2 = TINY_MFMA1(A1[2], s[2], negz);
        }
        else
        {
            f32x4 acc = {-0.f, -0.f, -0.f, -0.f}; // fma(a, b, -0) = a*b: the chain starts with a plain product, like dpp_fma_dot
            acc = TINY_MFMA4(A1[0], s[0], acc);
            acc = TINY_MFMA4(A1[1], s[1], acc);
            F.acc = TINY_MFMA4(A1[2], s[2], acc);
        }
    }
    __device__ __forceinline__ void lqr_finish(const InFlight &F, float di, float &un, float (&xn)[3]) const
    {
        if constexpr (EXACT)
        {
            static_assert(PL::FWD_XA == PL::FWD_U, "x rows and the u row of a lane are summed in one order (both SEQ for nx = 12, nu = 4)");
            f32x4 t[12];
            gather12v(t, F.p0, F.p1, F.p2);
            const f32x4 acc = reduce4<PL::FWD_XA>(t); // [0..2]: A x of the x rows, [3]: K x of the u row
            un = -acc[3] - di; // -(K x) - d: the SUM is negated, as in the reference
            const f32x16 pb = TINY_MFMA1(A2, un, negz);
            f32x4 t2[4];
            gather4v(t2, pb);
            const f32x4 xn4 = acc + reduce4<PL::FWD_XB>(t2); // register 3 carries no row here
            xn[0] = xn4[0]; xn[1] = xn4[1]; xn[2] = xn4[2];
        }
        else
        {
            un = F.acc[3] - di; // u rows of M1 hold -Kinf in the fma table
            const f32x4 acc = TINY_MFMA4(A2, un, F.acc);
            xn[0] = acc[0]; xn[1] = acc[1]; xn[2] = acc[2];
        }
    }
    __device__ __forceinline__ void lqr(const float (&s)[3], float di, float &un, float (&xn)[3]) const
    {
        InFlight F;
        lqr_issue(s, F);
        lqr_finish(F, di, un, xn);
    }

    // backward_pass_grad step (admm.cpp:19-20): p = p_{i+1}, lin = [q_i ; r_i]  ->  pn = p_i, dd = d_i
    __device__ __forceinline__ void riccati_issue(const float (&p)[3], const f32x4 &lin, InFlight &F) const
    {
        if constexpr (EXACT)
        {
            F.p0 = TINY_MFMA1(A3[0], p[0], negz); F.p1 = TINY_MFMA1(A3[1], p[1], negz); F.p2 = TINY_MFMA1(A3[2], p[2], negz);
            F.pk = TINY_MFMA1(A45, lin[3], negz); // Kinf^T r (x rows)
        }
        else
        {
            f32x4 acc = lin;
            acc = TINY_MFMA4(A3[0], p[0], acc);
            acc = TINY_MFMA4(A3[1], p[1], acc);
            F.acc = TINY_MFMA4(A3[2], p[2], acc);
        }
    }
    __device__ __forceinline__ void riccati_finish(const InFlight &F, const f32x4 &lin, float (&pn)[3], float &dd) const
    {
        if constexpr (EXACT)
        {
            f32x4 t4[12];
            gather12v(t4, F.p0, F.p1, F.p2);
            const f32x4 wv = lin + reduce4<PL::BWD_PA>(t4); // the three x rows in one packed tree; register 3 is not used:
            float t[12];                                    // the u row sums in its own order
            gather12<3>(t, F.p0, F.p1, F.p2);
            const float wv3 = lin[3] + reduce<PL::BWD_TMP>(t); // Bdyn^T p + r
            const f32x16 pq = TINY_MFMA1(A45, wv3, negz);       // Quu_inv (Bdyn^T p + r) (u row)
            f32x4 tk4[4];
            gather4v(tk4, F.pk);
            const f32x4 pn4 = wv - reduce4<PL::BWD_PK>(tk4);
            pn[0] = pn4[0]; pn[1] = pn4[1]; pn[2] = pn4[2];
            float tk[4];
            gather4<3>(tk, pq); dd = reduce<PL::BWD_D>(tk);
        }
        else
        {
            const f32x4 nz = {-0.f, -0.f, -0.f, -0.f};
            const f32x4 dq = TINY_MFMA4(A45, F.acc[3], nz); // u row: Quu_inv
            const f32x4 acc = TINY_MFMA4(A45, lin[3], F.acc); // x rows: -Kinf^T
            pn[0] = acc[0]; pn[1] = acc[1]; pn[2] = acc[2];
            dd = dq[3];
        }
    }
    __device__ __forceinline__ void riccati(const float (&p)[3], const f32x4 &lin, float (&pn)[3], float &dd) const
    {
        InFlight F;
        riccati_issue(p, lin, F);
        riccati_finish(F, lin, pn, dd);
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
// The new dual a = tp - t (admm.cpp:69-70, (y + u) - znew) is computed here, straight into its state registers.
typedef __attribute__((address_space(3))) float4 lds_float4;
__device__ __forceinline__ void masked_forward_update(unsigned long long m, float (&a)[4], const f32x4 &tp, float (&bo)[4], const f32x4 &old,
                                                      unsigned sn_addr, const f32x4 &t)
{
    unsigned long long sx;
    // the new dual a = tp - t (admm.cpp:69-70, (y + u) - znew) is computed here, under the narrowed EXEC, straight into its state
    // registers: four subtractions instead of four subtractions plus four masked moves
    asm volatile("s_and_saveexec_b64 %[sx], %[m]\n\t"
                 "v_sub_f32 %[a0], %[p0], %[t0]\n\tv_sub_f32 %[a1], %[p1], %[t1]\n\tv_sub_f32 %[a2], %[p2], %[t2]\n\tv_sub_f32 %[a3], %[p3], %[t3]\n\t"
                 "v_accvgpr_write_b32 %[b0], %[o0]\n\tv_accvgpr_write_b32 %[b1], %[o1]\n\tv_accvgpr_write_b32 %[b2], %[o2]\n\tv_accvgpr_write_b32 %[b3], %[o3]\n\t"
                 "ds_write_b128 %[ad], %[tv]\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [sx] "=&s"(sx), [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [b0] "+a"(bo[0]), [b1] "+a"(bo[1]),
                   [b2] "+a"(bo[2]), [b3] "+a"(bo[3])
                 : [m] "s"(m), [p0] "v"(tp[0]), [p1] "v"(tp[1]), [p2] "v"(tp[2]), [p3] "v"(tp[3]), [t0] "v"(t[0]), [t1] "v"(t[1]), [t2] "v"(t[2]),
                   [t3] "v"(t[3]), [o0] "v"(old[0]), [o1] "v"(old[1]), [o2] "v"(old[2]), [o3] "v"(old[3]), [ad] "v"(sn_addr), [tv] "v"(t)
                 : "memory", "scc"); // s_and_saveexec writes SCC
}
__device__ __forceinline__ void masked_backward_update(unsigned long long m, float (&pl)[3], float &dr, const float (&pn)[3], float dd)
{
    unsigned long long sx;
    asm volatile("s_and_saveexec_b64 %[sx], %[m]\n\t"
                 "v_accvgpr_write_b32 %[p0], %[n0]\n\tv_accvgpr_write_b32 %[p1], %[n1]\n\tv_accvgpr_write_b32 %[p2], %[n2]\n\tv_mov_b32 %[d], %[dd]\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [sx] "=&s"(sx), [p0] "+a"(pl[0]), [p1] "+a"(pl[1]), [p2] "+a"(pl[2]), [d] "+v"(dr)
                 : [m] "s"(m), [n0] "v"(pn[0]), [n1] "v"(pn[1]), [n2] "v"(pn[2]), [dd] "v"(dd)
                 : "scc");
}

// closed loop on chip: the lanes in `m` (instances that CONVERGED: the reference returned before v = vnew, admm.cpp:135-142) get their old
// slack back from the backup, so that the next solve's first sweep finds v | z where every sweep expects the previous slack
__device__ __forceinline__ void masked_restore_slack(unsigned long long m, const float (&bo)[4], unsigned sn_addr)
{
    const f32x4 tv = {acc_get(bo[0]), acc_get(bo[1]), acc_get(bo[2]), acc_get(bo[3])};
    unsigned long long sx;
    asm volatile("s_and_saveexec_b64 %[sx], %[m]\n\t"
                 "ds_write_b128 %[ad], %[tv]\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [sx] "=&s"(sx)
                 : [m] "s"(m), [ad] "v"(sn_addr), [tv] "v"(tv)
                 : "memory", "scc");
}

// [q_i ; r_i] of update_linear_cost (admm.cpp:80-82) for the four rows of a lane: cq - rho * (snew - dual)
template <bool EXACT>
__device__ __forceinline__ f32x4 lin_cost4(const f32x4 &cq, const f32x4 &rho4, const f32x4 &t1)
{
    if constexpr (EXACT) return cq - rho4 * t1;
    else return __builtin_elementwise_fma(-rho4, t1, cq);
}

// Live-out (round 4): lane (g, c) holds elements {4v + g : v = 0..3} of its instance's 64-byte step row in four registers; the row wants
// elements 4g .. 4g + 3 in ONE lane — the transpose of the 4 x 4 matrix (register v, lane group g).  Two butterfly stages do it in the
// registers: v_permlane32_swap exchanges lanes 32..63 of its first operand with lanes 0..31 of its second (bit 1 of v <-> bit 1 of g),
// v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second (bit 0 <-> bit 0).  Four instructions per 4-vector,
// against a ds_write_b128, four ds_read_b32 and two waits through the slack slot (round 3).
__device__ __forceinline__ f32x4 t16_transpose4(const f32x4 &r)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[0]), __float_as_uint(r[2]), false, false);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[1]), __float_as_uint(r[3]), false, false);
    const u32x2 lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const u32x2 hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    return f32x4{__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(hi[0]), __uint_as_float(hi[1])};
}

constexpr int TILE16_WAVES = 4;         // waves per workgroup = one per SIMD of a CU; they share the bounds and reference tables
constexpr int TILE16_MAX_TABLE_ROWS = 512;

// ---- BR / XR = true (round 4, the "pi" instantiations): box bounds (BR) and / or the reference (XR) PER INSTANCE (types.hpp:88-92: every
// reference workspace owns its u_min .. x_max and its Xref).  The kernel has no register to prefetch them into (512 of 512 in use) and the
// 61 KB + 30 KB a tile would need do not fit LDS beside the slack, so the rows travel HBM/L2 -> LDS by LDS-DMA (`global_load_lds_dwordx4`: no
// VGPR destination) into per-wave slots:
//   * lane (g, c) fetches 16-byte piece g of instance c's 64-byte reference row (for the 128-byte {lo, hi} row: pieces 2g and 2g + 1), so whole
//     rows are fetched, four lanes each; piece g of instance c lands lane-linear at slot + 16 (16 g + c);
//   * lane (g', c) then reads element 4 v + g' of ITS instance from slot + 256 v + 16 c + 4 g' (banks 4 c + g': conflict free) — the same
//     registers the shared tables deliver, so the arithmetic is untouched (tools/micro/glds_ring.hip checks the mapping, a destination above
//     64 KB in M0 and the counted waits in isolation);
//   * a table that does not change along the horizon (RowParams::pi_flags bit clear: the usual case — every robot its own limits, its own
//     set point) is ONE row per instance: fetched once per tile into a resident slot (bounds: two — the last step's input rows are unbounded), nothing
//     moves inside the iteration loop;
//   * a table that does (bit set) goes through a ring of 3 - 6 step slots (t16_ring_b / t16_ring_x), fetched depth - 1 steps ahead of its use and
//     retired with counted `s_waitcnt vmcnt`; a slot is refilled only after the values read from it have been consumed (the DMA statement takes one
//     of them as an operand).  A DMA costs a lone wave about 30 clocks of issue (M0, one wait state, the instruction): with the tables
//     L2-resident the reference ring measured + 1.6 %, the bounds ring + 8 %;
//   * ring tables come from TILE IMAGES (tile16_image_kernel, built when the inputs are set): [tile][step][piece][lane][16 B], i.e. exactly the LDS
//     image of a slot, so that a DMA reads 1 KB of consecutive bytes in whole 128-byte lines — from the [instance][step][16] arrays the same DMA is
//     sixteen half lines 1 920 B apart, and a CU's outstanding misses, not the memory, bound the ring (measured: reference ring 1.93 -> 1.81 ms).
//     RowParams::bounds / ::xref point to the images then.  (A window of the trajectory table is never read through a ring: with a table too
//     long for the LDS share beside the bounds' slots the handle keeps the 16-lane kernel.)
// The other table of a BR-only / XR-only instantiation is the batch-shared one staged in LDS, as in the base kernel.  hipcc does not count an
// asm DMA in its own s_waitcnt bookkeeping; the iteration loop issues no other vector memory instruction, and a foreign entry in the
// in-order counter can only make either side's counted wait longer, never shorter.  M0 is written by these statements only
// (tests/test_isa.py checks that nothing else in the pi kernels reads or writes it), so it is not saved around them.
// `on`: wave-uniform; a clear flag skips the statement (a branch INSIDE the asm: the unrolled sweep stays one basic block)
__device__ __forceinline__ void t16_dma_row(unsigned on, const float *sbase, unsigned voff, unsigned lds_dst, float after)
{
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n1:"
                 : : "s"(on), "v"(voff), "s"(sbase), "s"(lds_dst), "v"(after) : "memory", "scc");
}
__device__ __forceinline__ void t16_dma_row2(unsigned on, const float *sbase, unsigned voff, unsigned lds_dst, float after0, float after1)
{
    // a 128-byte {lo, hi} row: pieces 2g and 2g + 1 into two consecutive 1 KB images (the instruction's immediate offset is added to the
    // memory address AND to the LDS address: M0 advances by 1 KB - 16)
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:16\n1:"
                 : : "s"(on), "v"(voff), "s"(sbase), "s"(lds_dst), "v"(after0), "v"(after1) : "memory", "scc");
}
// the same row from its tile image (below): the two 1 KB pieces are 1 KB apart in memory AND in LDS — the immediate offset moves both, M0 stays
__device__ __forceinline__ void t16_dma_img2(unsigned on, const float *sbase, unsigned voff, unsigned lds_dst, float after0, float after1)
{
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n1:"
                 : : "s"(on), "v"(voff), "s"(sbase), "s"(lds_dst), "v"(after0), "v"(after1) : "memory", "scc");
}
__device__ __forceinline__ void t16_wait_vm(int k) // k is a constant once the sweeps are unrolled
{
    if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (k == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (k == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (k == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (k == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (k == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (k == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // (a smaller count than the true one only waits longer)
}
// ring depths (step slots): what the 40 KB behind the slack of four waves hold — both tables through rings: 3 x 2 KB + 4 x 1 KB per wave; one of
// them: 4 x 2 KB (beside a staged reference of <= 128 rows) / 6 x 1 KB (beside the staged bounds).  A row is fetched depth - 1 steps ahead.
constexpr int t16_ring_b(bool XR) { return XR ? 3 : 4; }
constexpr int t16_ring_x(bool BR) { return BR ? 4 : 6; }
constexpr int T16_RING_B = 2048, T16_RING_X = 1024;           // bytes of a slot: {lo, hi} rows / reference rows of sixteen instances

// COLD: the launch starts from reset_workspace() (RowParams::cold_start): no live-in array is read (round 3: a separate
// instantiation — as a run-time branch the two initialisations meet in 270 phi values and the allocator spills)
// MPC: the closed loop on chip (tiny_batch_mpc_run_async): P.mpc_steps solves of every tile inside the launch, the state staying in
// registers / LDS between them — u_0, the plant step x0 <- Adyn x0 + Bdyn u_0 in the plant kernel's arithmetic, the window slide, the dual
// reset and the terminal term happen on chip (quadrotor_tracking.cpp:93-118), like admm_rowlane.hip's MPC instantiation
template <int N, bool EXACT, bool COLD, bool MPC = false, bool BR = false, bool XR = false>
__global__ __launch_bounds__(WAVE * TILE16_WAVES, 1) void admm_tile16_kernel(const RowParams P)
{
    static_assert(!(MPC && (BR || XR)), "the closed loop on chip is instantiated for batch-shared bounds and a shared / windowed reference");
    constexpr int NX = 12, NU = 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int ntiles = (P.batch + 15) >> 4;
    const float rho = P.rho;
    const f32x4 rho4 = {rho, rho, rho, rho};

    // ---- LDS (dynamic): per wave the slack [v|vnew ; z|znew] of every step (lane-linear float4), then the tables the four
    //      waves share: box bounds [step][g] -> registers v = 0..3 (row 4v + g), reference rows [row][g] -> x rows 4v + g
    extern __shared__ __attribute__((aligned(16))) float4 lds4_base[];
    float4 *const lds4 = lds4_base;
    float4 *const sn0 = lds4 + wv * (N * WAVE) + lane; // sn[i * WAVE]
    // behind the slack: the bounds — the shared table (2 x N float4 rows) or, BR, the four waves' slots (1 or DB x 2 KB each) —, then the
    // reference — the shared table / trajectory table or, XR, the four waves' slots (1 or DX x 1 KB each)
    constexpr int DB = t16_ring_b(XR), DX = t16_ring_x(BR);
    const unsigned ringsB = (P.pi_flags & 1u) ? DB : 2, ringsX = (P.pi_flags & 2u) ? DX : 1; // resident bounds: the row of steps < N - 1 and the last step's
    const unsigned bnd_f4 = BR ? TILE16_WAVES * ringsB * (T16_RING_B / 16) : 2 * N * 4;
    float4 *const blo = lds4 + TILE16_WAVES * N * WAVE, *const bhi = blo + N * 4, *const tab0 = blo + bnd_f4;
    float4 *const tab = tab0;
    const int tab_rows = P.xref_mode == 1 ? P.table_rows : N;
    const float *tab_src = P.xref_mode == 1 ? P.xref_table : P.xref; // [rows][16]
    unsigned dmaB = 0, dmaX = 0; // wave-uniform LDS address of the wave's slot 0 (DMA destination)
    if constexpr (BR)
    {
        const unsigned ring = (unsigned)(size_t)(lds_float4 *)blo + wv * (ringsB * T16_RING_B);
        dmaB = __builtin_amdgcn_readfirstlane(ring);
    }
    else
    {
        for (int e = threadIdx.x; e < N * 16; e += WAVE * TILE16_WAVES)
        {
            const float2 lh = reinterpret_cast<const float2 *>(P.bounds)[e];
            const int i = e >> 4, r = e & 15;
            reinterpret_cast<float *>(&blo[i * 4 + (r & 3)])[r >> 2] = lh.x;
            reinterpret_cast<float *>(&bhi[i * 4 + (r & 3)])[r >> 2] = lh.y;
        }
    }
    if constexpr (XR)
    {
        const unsigned ring = (unsigned)(size_t)(lds_float4 *)tab0 + wv * (ringsX * T16_RING_X);
        dmaX = __builtin_amdgcn_readfirstlane(ring);
    }
    else
    {
        // Round 4: the table is staged as cq = -(Xref o Q), the only form the sweeps use (admm.cpp:80-82: the product is one IEEE multiply, the same
        // bits wherever it is computed; columns 12 .. 15 — the u rows' register — hold -(0 * 1) = -0: r = -rho (znew - y) keeps the sign of a zero
        // difference).  That takes the four Q registers and two packed multiplies per backward step out of the iteration loop; the raw last row the
        // terminal term needs is read from memory once per tile.
        // ... and carries N - 1 copies of its last row behind it: a window start clamped ONCE to the last row then reads rows ws + i without a per-step
        // clamp (three vector instructions per backward step: add, min, shift-add in front of the LDS read)
        for (int e = threadIdx.x; e < (tab_rows + N - 1) * 16; e += WAVE * TILE16_WAVES)
        {
            const int er = (e >> 4) < tab_rows ? (e >> 4) : tab_rows - 1;
            float val = tab_src[er * 16 + (e & 15)];
            if (TINY_T16_CQ) val = -(val * ((e & 15) < NX ? P.mats[(2 * NX + 2 * NU) * 16 + (e & 15)] : 1.f));
            reinterpret_cast<float *>(&tab[(e >> 4) * 4 + (e & 3)])[(e & 15) >> 2] = val;
        }
    }
    if constexpr (!BR || !XR) __syncthreads();

    TileMath<EXACT> M;
    M.load(P.mats, g, c);
    // Q of the lane's x rows; register 3 is the u row, whose cost term is -(0 * 1) = -0 (the table's columns 12..15 are zero):
    // r = -rho*(znew - y) keeps the sign of a zero difference
    f32x4 qv;
#pragma unroll
    for (int v = 0; v < 3; v++) qv[v] = P.mats[(2 * NX + 2 * NU) * 16 + 4 * v + g];
    qv[3] = 1.f;
    constexpr bool CQ = TINY_T16_CQ && !XR; // what load_xref returns is already -(Xref o Q)
    auto cost_of = [&](const f32x4 &xr) { return CQ ? xr : -(xr * qv); };

    // Persistent waves (round 3): the launch is one workgroup per CU and every wave draws tiles from a queue (an atomic counter
    // behind RowParams::n_unsolved, zeroed by the host with it) until it is empty.  A CU's LDS belongs to its workgroup, so with
    // one tile per wave the CU could not start new work before the slowest of its four tiles had ended; now a wave that finishes
    // takes the next tile at once, and the tables above are staged once per CU instead of once per four tiles.  Tiles are taken
    // in queue order = dispatch order: slot k is tile order[k] (longest predicted first) or tile k.
    // Two-ended queue (round 4, second session): under longest-first dispatch a launch of q tiles per wave slot ends in a partial round — the
    // slots all come back from their q-th tile at about the same time and the few tiles left keep a sixth of the chip busy for one more tile's
    // length (tests/fuzz/sim_tile_deque.py on the true iteration counts: makespan 132.5 iterations against 113.8 of work per slot).  Every
    // `tail_stride`-th wave therefore takes its tiles from the SHORT end of the order: it fits one tile more into the same time, the partial round
    // disappears (123).  Head and tail counts share one word (low / high 16 bits, one atomic claims from either end; a claim is good while
    // head + tail < ntiles); tail_stride = 0 is the plain counter.  The host (t16_tail_stride) decides; results do not depend on it.
    // (ONE scalar lives across the tile loop for this — the wave's increment; the pi instantiations sit at the allocator's cliff and two more
    //  sent their iteration loops to scratch.  The decode goes by the tile count, which the loop holds anyway: up to 32 768 tiles the word is
    //  read as two halves — with no wave at the short end the high half stays zero — beyond that it is the plain counter, and the host sets no stride.)
    constexpr bool DQ = COLD && !MPC; // longest-first dispatch is the default of launches from a reset workspace: the other instantiations keep the plain counter
    int qinc = 1;
    if constexpr (DQ)
    {
        const unsigned tail_stride = (P.pi_flags >> 8) & 0xffu;
        if (tail_stride && (((unsigned)blockIdx.x * TILE16_WAVES + (unsigned)wv) % tail_stride) == 0u) qinc = 0x10000;
        qinc = __builtin_amdgcn_readfirstlane(qinc);
    }
    for (;;)
    {
    int slot = 0;
    if (lane == 0) slot = atomicAdd(P.n_unsolved + 1, qinc);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (DQ && ntiles <= 32768)
    {
        const int qh = slot & 0xffff, qt = (int)((unsigned)slot >> 16);
        if (qh + qt >= ntiles) break; // every wave reaches this: the queue only grows
        slot = qinc == 1 ? qh : ntiles - 1 - qt;
    }
    else if (slot >= ntiles) break;
    // (an opaque zero per tile in every LDS base address: they are invariant across tiles and hipcc would otherwise hoist the
    // per-step addresses of prologue and epilogue out of the queue loop and spill them, as it would inside the iteration loop)
    int ozt;
    asm volatile("s_mov_b32 %0, 0" : "=s"(ozt));
    float4 *const sn = sn0 + ozt;
    float4 *const lds4 = lds4_base + ozt;
    int tile = slot;
    if (P.order) tile = P.order[slot];
    const bool tile_ok = tile >= 0 && tile < ntiles; // also rejects a bad entry of a caller-supplied order: such a wave stores nothing
    const int inst = tile * 16 + c;
    const bool valid = tile_ok && inst < P.batch;
    const int inst_a = valid ? inst : P.batch - 1; // padding columns of the last tile load a valid instance and store nothing

    // ---- per-instance state, four words per horizon step (row 4v + g) ----
    //   a[i]   : g_i | y_i      duals                                                           (VGPR; four SEPARATE registers: as
    //            register pairs or quads the 120 words fragment the file and the allocator spills 250 registers)
    //   dr[i]  : d_i            feed-forward the forward sweep uses                              (VGPR)
    //   sn[i]  : before forward step i of an iteration the OLD slack v_i | z_i (what the previous iteration's sweep left, = v
    //            after admm.cpp:141-142), afterwards the new one vnew_i | znew_i                (LDS)
    //   bo[i]  : backup of the old slack this iteration's forward sweep replaced — the live-out v, z of an instance that
    //            converges in this iteration (the reference returns before v = vnew)            (AGPR, write-mostly)
    //   pl[i]  : p_i of the last backward sweep executed inside the iteration loop, live-out only (AGPR, write-only in the loop);
    //            its d_i is dr[i].  The backward sweep of the LAST permitted iteration (an instance that exhausts max_iter) is
    //            deferred to the epilogue: x, u of such an instance come from the d its last forward sweep used, which that
    //            sweep must not overwrite — round 2 kept a second copy dl[] of d for this, 30 registers the allocator no
    //            longer has since the sums are packed
    float a[N][4];
    float dr[N], bo[N][4], pl[N][3];
    auto dual4 = [&](int i) { return f32x4{a[i][0], a[i][1], a[i][2], a[i][3]}; };
    const int ebase = (inst_a * N) * 16 + g; // element 4v + g of step i: ebase + i*16 + 4v
    int wstart = 0;
    if (P.xref_mode == 1) wstart = P.xref_start[inst_a];
    const bool zdual = COLD || (P.duals_zero != 0);

    if constexpr (COLD) // reset_workspace() folded into the launch: nothing is read (round 3: the loads used to be issued and discarded)
    {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            a[i][0] = a[i][1] = a[i][2] = a[i][3] = 0.f;
#pragma unroll
            for (int v = 0; v < 4; v++) acc_init(bo[i][v], 0.f);
            sn[i * WAVE] = z4;
            acc_init(pl[i][0], 0.f); acc_init(pl[i][1], 0.f); acc_init(pl[i][2], 0.f);
            dr[i] = 0.f;
        }
    }
    else
    {
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
                a[i][v] = zdual ? 0.f : gyv;
                acc_init(bo[i][v], vzl[v]);
            }
            sn[i * WAVE] = make_float4(vzl[0], vzl[1], vzl[2], vzl[3]);
            acc_init(pl[i][0], pdv[0]); acc_init(pl[i][1], pdv[1]); acc_init(pl[i][2], pdv[2]);
            dr[i] = pdv[3];
        }
    }
    float x0[3];
#pragma unroll
    for (int v = 0; v < 3; v++) x0[v] = P.xu[ebase + 4 * v];

    // BR / XR: byte offsets of the lane's DMA pieces from the arrays' bases (saddr + 32-bit offset addressing: the step's row offset is
    // added to the uniform base, on the scalar unit)
    const unsigned onB = BR ? (P.pi_flags & 1u) : 0u, onX = XR ? (P.pi_flags & 2u) : 0u; // the table changes along the horizon: ring of step slots
    // The lane's addresses are made from its lane number: per tile (A0: carried across the iteration loop) in the warm-start instantiations, per
    // ITERATION (AI: from the slack address that is live there anyway, 16 x lane = sn_addr - the wave's base) in the cold-start ones.  The register
    // allocator sits at a cliff in this kernel and the two kinds tip opposite ways (scratch accesses per iteration on the listing, pinned by
    // tests/test_isa.py): cold start 2 - 5 remade against 21 - 26 carried, warm start 33 - 36 carried against 170 - 300 remade.  A scratch access is
    // a vector-memory round trip that a lone wave cannot hide: twenty of them per iteration measured 10 %.
    struct PiAddr { unsigned rdB, rdX, voffB, voffX; };
    auto pi_addr = [&](unsigned lane16) { // lane16 = 16 x lane
        const unsigned c16 = lane16 & 0xf0u, gg = lane16 >> 8, cc = c16 >> 4;
        const int ins = tile * 16 + (int)cc;
        const unsigned ia = (unsigned)((tile_ok && ins < P.batch) ? ins : P.batch - 1);
        PiAddr A;
        A.rdB = dmaB + 1024u * (gg >> 1) + c16 + 8u * (gg & 1u); // pair {lo, hi} of row 4v + g: + 256 v
        A.rdX = dmaX + c16 + 4u * gg;                            // element 4v + g: + 256 v
        A.voffB = ia * P.bounds_inst_stride * 8u + 32u * gg;
        A.voffX = ia * P.xref_inst_stride * 4u + 16u * gg;
        return A;
    };
    const float *const xsrc = P.xref; // XR: the per-instance array (one resident row per instance) or its tile image (ring): row i of the tile at
                                      // ((tile N + i) 64 + lane) 16 bytes
    // Xref_i for this lane: x rows in registers 0..2, register 3 = 0 (column 12 + g of the 16-wide table row)
    auto load_xref = [&](const float4 *tb, int ws, int i, int gsel = -1) { // gsel: the lane's g where the caller has remade it (iteration loop)
        if constexpr (XR) // prologue / epilogue (outside the sweeps): straight from memory
        {
            // (the lane number passes through an opaque statement: the addresses below are then made where they are used — hoisted in front of the
            //  iteration loop and carried across it they push the warm-start instantiations over the allocator's cliff)
            unsigned ll = lane;
            asm volatile("" : "+v"(ll));
            const unsigned cc = ll & 15u, gg = ll >> 4;
            if (onX) // element 4v + g of instance c: piece v = lane 16 v + c of the image row, word g
            {
                const float *rp = xsrc + ((size_t)((tile_ok ? tile : 0) * N + i) * 64 + cc) * 4 + gg;
                return f32x4{rp[0], rp[64], rp[128], 0.f};
            }
            const int ins = tile * 16 + (int)cc;
            const float *rp = xsrc + (size_t)((tile_ok && ins < P.batch) ? ins : P.batch - 1) * P.xref_inst_stride + i * 16 + gg; // element g of the row
            return f32x4{rp[0], rp[4], rp[8], 0.f};
        }
        else
        {
        int row;
        // (same-box A/B, 65 536 tracking instances: warm start 1.78 -> 1.71 ms, fma 0.874 -> 0.860, closed loop on chip 1.040 -> 1.030 / 0.540 -> 0.502; the
        //  exact cold-start instantiation alone loses 0.8 % — two more scratch accesses per iteration at the allocator's cliff — and keeps the clamp)
        if constexpr (TINY_T16_PADTAB && !(EXACT && COLD && !MPC && !BR && !XR)) row = (ws < tab_rows - 1 ? ws : tab_rows - 1) + i; // rows wc .. wc + N - 1 exist
        else { row = ws + i; row = row < tab_rows ? row : tab_rows - 1; }
        const float4 t4 = tb[row * 4 + (gsel >= 0 ? gsel : g)];
        return f32x4{t4.x, t4.y, t4.z, t4.w};
        }
    };
    PiAddr A0 = {0u, 0u, 0u, 0u};
    if constexpr (BR || XR)
    {
        A0 = pi_addr(16u * lane);
        // a table that does not change along the horizon: the tile's sixteen rows, once (the slot's previous readers — the last sweeps of the wave's
        // previous tile — are long retired: its epilogue stored through the same LDS pipe since)
        // (two rows of bounds: z has N - 1 columns — the input rows of the last step are not bounded, the table holds -inf | +inf there)
        if constexpr (BR) { t16_dma_row2(onB ^ 1u, P.bounds, A0.voffB, dmaB, 0.f, 0.f); t16_dma_row2(onB ^ 1u, P.bounds, A0.voffB + (N - 1) * 128u, dmaB + T16_RING_B, 0.f, 0.f); }
        if constexpr (XR) t16_dma_row((onX >> 1) ^ 1u, xsrc, A0.voffX, dmaX, 0.f); // (a resident reference row is never an image)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // the RAW reference row (the terminal term -(Xref_{N-1}^T Pinf), admm.cpp:83): with the staged table holding -(Xref o Q) it is read from memory
    auto load_xref_raw = [&](const float4 *tb, int ws, int i) {
        if constexpr (CQ)
        {
            int row = ws + i;
            row = row < tab_rows ? row : tab_rows - 1;
            const float *rp = tab_src + row * 16 + g;
            return f32x4{rp[0], rp[4], rp[8], 0.f};
        }
        else return load_xref(tb, ws, i);
    };
    float pterm[3];
    {
        const f32x4 xrN4 = load_xref_raw(tab, wstart, N - 1);
        const float xrN[3] = {xrN4[0], xrN4[1], xrN4[2]};
        M.terminal(xrN, pterm);
    }

    int st = TINY_STATUS_UNSOLVED_, itn = 1; // admm.cpp:114-115
    float r_ps = 0.f, r_pi = 0.f, r_ds = 0.f, r_di = 0.f;
    if (valid && !COLD)
    {
        r_ps = P.res[4 * inst + 0]; r_pi = P.res[4 * inst + 1];
        r_ds = P.res[4 * inst + 2]; r_di = P.res[4 * inst + 3];
    }
    float pN[3] = {0.f, 0.f, 0.f}; // p_{N-1} of the last executed forward sweep

    // backward_pass_grad with the linear cost (admm.cpp:15-22, :80-82) for the lanes in `amask`, the text of the iteration loop's sweep: the closed loop on
    // chip runs it between two solves for the instances that exhausted max_iter (their last sweep is deferred out of the loop).  (The iteration loop keeps
    // its own copy: routed through this lambda the warm-start instantiation picks up ten more scratch accesses per iteration.)
    auto backward_sweep = [&](unsigned long long amask, const float4 *snI, const float4 *tabI, int wsI) {
        float p[3] = {pN[0], pN[1], pN[2]};
        // the linear cost of step i - 1 depends on the state only, not on p: it is computed while the products of step i
        // are in flight
        auto load_lin = [&](int i, float4 &sl, f32x4 &xr) { // LDS reads of step i's linear cost: issued a step ahead, in front of the MFMAs
            sl = snI[i * WAVE];
            xr = load_xref(tabI, wsI, i);
        };
        auto make_lin = [&](int i, const float4 &sl, const f32x4 &xr) {
            const f32x4 sni = {sl.x, sl.y, sl.z, sl.w};
            return lin_cost4<EXACT>(cost_of(xr), rho4, sni - dual4(i)); // admm.cpp:80-82
        };
        float4 sl0; f32x4 xr0;
        load_lin(N - 2, sl0, xr0);
        f32x4 lin = make_lin(N - 2, sl0, xr0);
#pragma unroll
        for (int i = N - 2; i >= 0; i--)
        {
            float4 sl_n = make_float4(0.f, 0.f, 0.f, 0.f); f32x4 xr_n = {0.f, 0.f, 0.f, 0.f};
            if (EXACT && i > 0) load_lin(i - 1, sl_n, xr_n); // exact: in front of the MFMAs (measured 1.72 -> 1.69 ms; fma: 0.91 -> 0.93 the other way)
            typename TileMath<EXACT>::InFlight F;
            M.riccati_issue(p, lin, F);
            if (!EXACT && i > 0) load_lin(i - 1, sl_n, xr_n);
#if TINY_T16_SCHED
            if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
            f32x4 lin_n = lin;
            if (i > 0) lin_n = make_lin(i - 1, sl_n, xr_n); // in the shadow of the four MFMAs, not behind the sums
#if TINY_T16_SCHED
            if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
            float pn[3], dd;
            M.riccati_finish(F, lin, pn, dd);
            masked_backward_update(amask, pl[i], dr[i], pn, dd);
            p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
            lin = lin_n;
        }
    };

    for (int ms = 0;; ++ms) // MPC steps of the closed loop on chip (one pass otherwise)
    {
    if (MPC && ms > 0) { st = TINY_STATUS_UNSOLVED_; itn = 1; }
    bool active = valid && (P.max_iter > 0);
    for (int it = 0; it < P.max_iter; ++it)
    {
        if (!__any(active)) break;
        // An opaque zero, re-made in every iteration, enters every LDS address of the sweeps: the addresses (30 slack slots, 30
        // bounds rows, 30 clamped reference rows per lane) are loop invariant, and hipcc would otherwise compute them all
        // ahead of the iteration loop and hold — in fact spill — them
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        float4 *const snI = sn + oz;
        const float4 *bloI = blo + oz, *bhiI = bhi + oz;
        const float4 *const tabI = tab + oz;
        const int wsI = wstart + oz;
        const unsigned long long amask = __ballot(active); // instances still iterating: wave-uniform during the forward sweep
        const unsigned sn_addr = (unsigned)(size_t)(lds_float4 *)snI;
        // the lane's rows of the staged bounds tables ([step][g] float4): g remade from the slack address (16 x lane = sn_addr - the wave's base, g = lane >> 4)
        // — carried across the iteration in registers of their own the two addresses are spilled, and reloaded from scratch at the head of every
        // iteration, two vector-memory round trips in front of the first step
        const unsigned gI = (sn_addr - (unsigned)__builtin_amdgcn_readfirstlane(sn_addr)) >> 8;
        const float4 *const bloG = bloI + gI, *const bhiG = bhiI + gI;
        // BR / XR rings.  dma_x(i): reference row of step i -> slot i % 3; dma_b(i): {lo, hi} row of step i -> slot i % 3 (two pieces); both skip
        // themselves when their table is one row per instance (fetched at the head of the tile).  `after` = a value that was read from the slot's
        // previous occupant: the refill cannot be issued before that read has returned.
        const float *const bsrcI = P.bounds + oz, *const xsrcI = xsrc + oz;
        unsigned rdBI = 0, rdXI = 0;
        if constexpr (BR || XR)
        {
            const PiAddr AI = pi_addr(sn_addr - __builtin_amdgcn_readfirstlane(sn_addr)); // lane 0's slack address is the wave's base
            rdBI = COLD ? AI.rdB : A0.rdB + oz; rdXI = COLD ? AI.rdX : A0.rdX + oz;
        }
        const unsigned sB1 = onB ? T16_RING_B : 0u, sX1 = onX ? T16_RING_X : 0u; // distance of the step slots (0: the one resident row)
        const unsigned sBlast = onB ? ((N - 1) % DB) * T16_RING_B : T16_RING_B;
        // (the step's row offset rides in the lane's 32-bit offset — one vector add per DMA —, not in the scalar base: thirty scalar base pairs per
        //  sweep, hoisted to the head of the iteration by the scheduler, spilled scalar registers into vector lanes)
        // ring sources: the bounds' tile image (2 KB per step), the reference's tile image (1 KB per step)
        const unsigned lane16I = sn_addr - __builtin_amdgcn_readfirstlane(sn_addr);
        const unsigned timg = (unsigned)(tile_ok ? tile : 0) * (unsigned)N; // image row of step 0
        unsigned xw = 0, bw = 0;
        if constexpr (XR) xw = timg * 1024u + (COLD ? lane16I : 16u * lane);
        if constexpr (BR) bw = timg * 2048u + (COLD ? lane16I : 16u * lane);
        auto dma_x = [&](int i, float after) { t16_dma_row(onX, xsrcI, xw + (unsigned)i * 1024u, dmaX + (i % DX) * T16_RING_X, after); };
        auto dma_b = [&](int i, float after0, float after1) { t16_dma_img2(onB, bsrcI, bw + (unsigned)i * 2048u, dmaB + (i % DB) * T16_RING_B, after0, after1); };
        typedef __attribute__((address_space(3))) const f32x2 lds_cfloat2;
        typedef __attribute__((address_space(3))) const float lds_cfloat;
        auto ring_bounds = [&](int i, float4 &lo, float4 &hi) { // the registers the shared tables deliver: rows 4v + g
            // slot of step i: ring slot i % DB; resident rows: slot 0, and slot 1 for the last step
            lds_cfloat2 *rp = reinterpret_cast<lds_cfloat2 *>(rdBI + (i == N - 1 ? sBlast : (i % DB) * sB1));
            const f32x2 b0 = rp[0], b1 = rp[32], b2 = rp[64], b3 = rp[96];
            lo = make_float4(b0.x, b1.x, b2.x, b3.x); hi = make_float4(b0.y, b1.y, b2.y, b3.y);
        };
        auto ring_xref = [&](int i) {
            lds_cfloat *rp = reinterpret_cast<lds_cfloat *>(rdXI + (i % DX) * sX1);
            return f32x4{rp[0], rp[64], rp[128], 0.f};
        };
        // the first DX reference rows of the backward sweep (every slot: the previous sweep consumed them), then the first DB - 1 steps' bounds:
        // oldest first, so that every later counted wait has retired them
        if constexpr (XR)
        {
#pragma unroll
            for (int k = 0; k < DX; k++) dma_x(N - 2 - k, 0.f);
        }
        if constexpr (BR)
        {
#pragma unroll
            for (int k = 0; k < DB - 1; k++) dma_b(k, 0.f, 0.f);
        }
        // The arithmetic of a sweep runs for all 16 columns (the MFMAs are wave-wide); only the state updates of a
        // converged instance are masked, which freezes it exactly where the reference returns.
        // ---------------- forward sweep: forward_pass + update_slack + update_dual + residual maxima ----------------
        // Software pipelined by one step (round 3): step i issues its LDS reads (bounds, old slack) and its three product MFMAs,
        // then runs the slack / dual / residual arithmetic of step i - 1 while those are in flight, then sums the products.
        float s[3] = {x0[0], x0[1], x0[2]};
        float pri_x = 0.f, dua_x = 0.f, pri_u = 0.f, dua_u = 0.f;
        auto elementwise = [&](int i, const f32x4 &sv, const float4 &lo, const float4 &hi, const float4 &ol) {
            const f32x4 old = {ol.x, ol.y, ol.z, ol.w};
            const f32x4 tp = sv + dual4(i);                                        // admm.cpp:47-48
            f32x4 t;
            t[0] = __builtin_amdgcn_fmed3f(tp[0], lo.x, hi.x);                  // :51-60 (lo := min(lo, hi) on the host)
            t[1] = __builtin_amdgcn_fmed3f(tp[1], lo.y, hi.y);
            t[2] = __builtin_amdgcn_fmed3f(tp[2], lo.z, hi.z);
            t[3] = __builtin_amdgcn_fmed3f(tp[3], lo.w, hi.w);
            const f32x4 dp = sv - t, dd_ = old - t;                             // :95-98
            pri_x = fmaxf(fmaxf(fmaxf(pri_x, fabsf(dp[0])), fabsf(dp[1])), fabsf(dp[2]));
            dua_x = fmaxf(fmaxf(fmaxf(dua_x, fabsf(dd_[0])), fabsf(dd_[1])), fabsf(dd_[2]));
            pri_u = fmaxf(pri_u, fabsf(dp[3]));
            dua_u = fmaxf(dua_u, fabsf(dd_[3]));
            if (i == N - 1)
            {
                const f32x4 an = tp - t; // what masked_forward_update stores below
#pragma unroll
                for (int v = 0; v < 3; v++)
                {
                    const float pn_ = lin_cost<EXACT>(pterm[v], rho, t[v] - an[v]); // admm.cpp:83-84
                    pN[v] = active ? pn_ : pN[v];
                }
            }
            masked_forward_update(amask, a[i], tp, bo[i], old, sn_addr + i * (WAVE * 16), t); // a = tp - t (:69-70), slack, backup
        };
        f32x4 sv_p = {0.f, 0.f, 0.f, 0.f};
        float4 lo_p = make_float4(0.f, 0.f, 0.f, 0.f), hi_p = lo_p, ol_p = lo_p;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            float4 lo, hi;
            if constexpr (BR)
            {
                if (i + DB - 1 < N) dma_b(i + DB - 1, lo_p.x, hi_p.w); // into the slot of step i - 1, whose rows are in lo_p / hi_p
                t16_wait_vm(2 * ((N - 1 - i) < (DB - 1) ? (N - 1 - i) : (DB - 1))); // the two pieces of step i have landed (newer: steps i + 1 .. i + DB - 1)
                ring_bounds(i, lo, hi);
            }
            else if constexpr (TINY_T16_BADDR && (COLD || !XR)) { lo = bloG[i * 4]; hi = bhiG[i * 4]; } // (warm start with the reference ring: 225 against 33)
            else { lo = bloI[i * 4 + g]; hi = bhiI[i * 4 + g]; }
            const float4 ol = snI[i * WAVE];
            typename TileMath<EXACT>::InFlight F;
            if (i < N - 1) M.lqr_issue(s, F);
#if TINY_T16_SCHED
            if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
            if (i > 0) elementwise(i - 1, sv_p, lo_p, hi_p, ol_p);
            float un = 0.f, xn[3] = {0.f, 0.f, 0.f};
            if (i < N - 1) M.lqr_finish(F, dr[i], un, xn);
            sv_p = f32x4{s[0], s[1], s[2], un};
            lo_p = lo; hi_p = hi; ol_p = ol;
            s[0] = xn[0]; s[1] = xn[1]; s[2] = xn[2];
        }
        elementwise(N - 1, sv_p, lo_p, hi_p, ol_p);
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
        if (it == P.max_iter - 1) break; // the backward sweep of the last permitted iteration runs in the epilogue
        // ---------------- backward sweep: (v = vnew, z = znew are implicit: sn holds both) linear cost, backward_pass_grad ----------------
        {
            float p[3] = {pN[0], pN[1], pN[2]};
            const unsigned long long amask = __ballot(active);
            // the linear cost of step i - 1 depends on the state only, not on p: it is computed while the products of step i
            // are in flight
            auto load_lin = [&](int i, float4 &sl, f32x4 &xr) { // LDS reads of step i's linear cost: issued a step ahead, in front of the MFMAs
                sl = snI[i * WAVE];
                if constexpr (XR) xr = ring_xref(i);
                else xr = load_xref(tabI, wsI, i, TINY_T16_BADDR ? (int)gI : -1);
            };
            auto make_lin = [&](int i, const float4 &sl, const f32x4 &xr) {
                const f32x4 sni = {sl.x, sl.y, sl.z, sl.w};
                return lin_cost4<EXACT>(cost_of(xr), rho4, sni - dual4(i)); // admm.cpp:80-82
            };
            float4 sl0; f32x4 xr0;
            if constexpr (XR) t16_wait_vm(0); // rows N-2 .. N-1-DX were fetched at the head of the iteration
            load_lin(N - 2, sl0, xr0);
            f32x4 lin = make_lin(N - 2, sl0, xr0);
#pragma unroll
            for (int i = N - 2; i >= 0; i--)
            {
                float4 sl_n = make_float4(0.f, 0.f, 0.f, 0.f); f32x4 xr_n = {0.f, 0.f, 0.f, 0.f};
                if constexpr (XR)
                {
                    if (i >= DX) dma_x(i - DX, lin[0]); // into the slot of row i, which `lin` was made from
                    // row i - 1 has landed; newer, still in flight: the rows j = i - 2 .. i - DX that this loop fetched (j <= N - 2 - DX; the DX rows above
                    // that were fetched at the head of the iteration and are retired)
                    if (i > 0)
                    {
                        const int hi_j = (i - 2) < (N - 2 - DX) ? (i - 2) : (N - 2 - DX), lo_j = (i - DX) > 0 ? (i - DX) : 0;
                        t16_wait_vm(hi_j >= lo_j ? hi_j - lo_j + 1 : 0);
                    }
                }
                if (EXACT && i > 0) load_lin(i - 1, sl_n, xr_n); // exact: in front of the MFMAs (measured 1.72 -> 1.69 ms; fma: 0.91 -> 0.93 the other way)
                typename TileMath<EXACT>::InFlight F;
                M.riccati_issue(p, lin, F);
                if (!EXACT && i > 0) load_lin(i - 1, sl_n, xr_n);
#if TINY_T16_SCHED
                if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
                f32x4 lin_n = lin;
                if (i > 0) lin_n = make_lin(i - 1, sl_n, xr_n); // in the shadow of the four MFMAs, not behind the sums
#if TINY_T16_SCHED
                if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
                float pn[3], dd;
                M.riccati_finish(F, lin, pn, dd);
                masked_backward_update(amask, pl[i], dr[i], pn, dd);
                p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
                lin = lin_n;
            }
        }
    }
    if (!MPC || ms + 1 >= P.mpc_steps) break;
    // ---------------- advance to the next MPC step (quadrotor_tracking.cpp:101-118): nothing leaves the chip ----------------
    if constexpr (MPC)
    {
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        float4 *const snA = sn + oz;
        const unsigned sn_addr = (unsigned)(size_t)(lds_float4 *)snA;
        const bool solved = st == TINY_STATUS_SOLVED_;
        // [x_0 ; u_0] of the solve that just finished, in the solver's own arithmetic — from the d its last FORWARD sweep used, i.e. before the
        // deferred backward sweep below replaces it (the live-out's order) —; then the plant step in the plant kernel's
        // (plant_step_kernel, tinympc_batch.hip: separately rounded products; Adyn x0 through Eigen's GEMV accumulator from +0,
        // Bdyn u_0 the lazy product's sequential sum — rowlane_math.h plant_step)
        float un, xn_[3];
        M.lqr(x0, dr[0], un, xn_);
        if (P.u0_traj && valid) P.u0_traj[((long long)ms * P.batch + inst) * NU + g] = un;
        {
            const f32x16 p0 = TINY_MFMA1(M.A1[0], x0[0], M.negz), p1 = TINY_MFMA1(M.A1[1], x0[1], M.negz), p2 = TINY_MFMA1(M.A1[2], x0[2], M.negz);
            const f32x16 pb = TINY_MFMA1(M.A2, un, M.negz);
            f32x4 t[12], t2[4];
            gather12v(t, p0, p1, p2);
            gather4v(t2, pb);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(z)); // keep the +0 start of the GEMV accumulator: (+0) + (-0) = +0
            f32x4 acc = z + t[0];
#pragma unroll
            for (int k = 1; k < 12; k++) acc = acc + t[k];
            const f32x4 x1 = acc + reduce4<PLAN_SEQ>(t2);
            x0[0] = x1[0]; x0[1] = x1[1]; x0[2] = x1[2];
        }
        // an instance that exhausted max_iter still owes the backward sweep of its last iteration (deferred out of the loop): its d, p
        const unsigned long long umask = __ballot(valid && !solved);
        if (umask != 0ull) backward_sweep(umask, snA, tab + oz, wstart + oz);
        // v | z of the workspace: the old slack for an instance that converged (the reference returned before v = vnew), the new one otherwise
        const unsigned long long smask = __ballot(valid && solved);
#pragma unroll
        for (int i = 0; i < N; i++) masked_restore_slack(smask, bo[i], sn_addr + i * (WAVE * 16));
        if (P.xref_mode == 1)
        {
            wstart += P.window_advance;
            const f32x4 xrN4 = load_xref_raw(tab + oz, wstart, N - 1);
            const float xrN[3] = {xrN4[0], xrN4[1], xrN4[2]};
            M.terminal(xrN, pterm);
        }
#pragma unroll
        for (int i = 0; i < N; i++) a[i][0] = a[i][1] = a[i][2] = a[i][3] = 0.f; // y = g = 0 (:106-107)
    }
    } // MPC steps

    if (P.max_iter <= 0) // tiny_solve only sets status and iter (admm.cpp:114-117,151)
    {
        if (valid && g == 0)
        {
            P.status[inst] = TINY_STATUS_UNSOLVED_;
            P.iter[inst] = 1;
            atomicAdd(P.n_unsolved, 1);
        }
        continue;
    }

    // ---------------- live-out: every work array written once ----------------
    // Round 3: a lane holds elements {4v + g} of its instance's 64-byte step row, so stored as they stand the four registers go
    // out as four dword stores that each cover 16 x 16 scattered bytes.  Instead every 4-vector passes through the step's own
    // slack slot in LDS (free once it has been read; LDS operations of a wave execute in order): written lane-linear, read back
    // transposed, so that lane l holds elements 4(l & 3) .. +3 of instance l >> 2 and ONE dwordx4 store per array and step puts
    // out sixteen whole 64-byte rows, four adjacent lanes each.
    {
        const bool solved = st == TINY_STATUS_SOLVED_;
        const int c2 = lane >> 2, q2 = lane & 3;
        const int inst2 = tile * 16 + c2;
        const bool valid2 = tile_ok && inst2 < P.batch;
        const int obase2 = ((valid2 ? inst2 : 0) * N) * 16 + 4 * q2;
        // (may_alias: the same LDS words are accessed as float4 slack, as 4-vectors and as single floats, in program order)
        typedef float float_ma __attribute__((may_alias));
        typedef f32x4 f32x4_ma __attribute__((may_alias));
        float_ma *const stage_w = reinterpret_cast<float_ma *>(lds4 + wv * (N * WAVE));     // wave's slack area, as floats
        const int rd_off = (c2 * 4 + q2);                                                    // + 64 * r floats, r = 0..3
        const int obase4 = ebase + 3 * g; // element 4g of the lane's own instance: ebase = (inst_a N) 16 + g
        // (the exact warm-start instantiations keep the LDS path: with the register transpose their iteration loop falls off the allocator's cliff,
        //  600 - 800 scratch accesses per iteration on the listing, where every other instantiation drops to 0 - 1)
        constexpr bool XPOSE = TINY_T16_XPOSE && !(EXACT && !COLD && !MPC);
        auto put = [&](float *dst, int i, const f32x4 &val) {
            if constexpr (XPOSE)
            {
                const f32x4 o = t16_transpose4(val); // elements 4g .. 4g + 3 of instance c
                if (valid) *reinterpret_cast<f32x4 *>(dst + obase4 + i * 16) = o;
                return;
            }
            float_ma *slot = stage_w + i * (WAVE * 4);
            reinterpret_cast<f32x4_ma *>(slot)[lane] = val;                                  // lane (g, c): elements 4v + g at [lane][v]
            asm volatile("" ::: "memory"); // the reads below fetch what OTHER lanes just wrote: nothing may move across (hipcc
            f32x4 o;                       // otherwise sinks them under the store's predicate, past the next array's write)
            o[0] = slot[rd_off]; o[1] = slot[rd_off + 64]; o[2] = slot[rd_off + 128]; o[3] = slot[rd_off + 192]; // element 4 q2 + r of instance c2
            asm volatile("" ::: "memory");
            if (valid2) *reinterpret_cast<f32x4 *>(dst + obase2 + i * 16) = o;
        };
        float s[3] = {x0[0], x0[1], x0[2]};
#pragma unroll
        for (int i = 0; i < N; i++)
        {
            // x,u: regenerated from the d of the last executed forward sweep by the same instruction sequence
            float un = 0.f, xn[3] = {0.f, 0.f, 0.f};
            if (i < N - 1) M.lqr(s, dr[i], un, xn);
            const f32x4 sv = {s[0], s[1], s[2], un};
            const f32x4 sni = reinterpret_cast<const f32x4_ma *>(stage_w + i * (WAVE * 4))[lane];
            const f32x4 xr = load_xref(tab, wstart, i);
            f32x4 lin = lin_cost4<EXACT>(cost_of(xr), rho4, sni - dual4(i));
            if (i == N - 1) lin[3] = 0.f;
            // p.col(N-1) is rewritten by every forward sweep (admm.cpp:83-84); the other columns and d come from the
            // last backward sweep this instance executed (an instance that never ran one keeps its live-in p, d)
            f32x4 pdv = {acc_get(pl[i][0]), acc_get(pl[i][1]), acc_get(pl[i][2]), dr[i]};
            if (i == N - 1) pdv = f32x4{pN[0], pN[1], pN[2], 0.f};
            // a converged instance returned before v = vnew (admm.cpp:135-142): its v, z are the slack the last sweep replaced
            f32x4 vzv;
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
                const float bov = acc_get(bo[i][v]);
                vzv[v] = solved ? bov : sni[v];
            }
            put(P.xu, i, sv);
            put(P.qr, i, lin);
            put(P.pd, i, pdv);
            put(P.vz, i, vzv);
            put(P.vzn, i, sni);
            put(P.gy, i, dual4(i));
            if constexpr (!XPOSE) reinterpret_cast<f32x4_ma *>(stage_w + i * (WAVE * 4))[lane] = sni; // the slot gets its slack back: the deferred sweep below reads it
            s[0] = xn[0]; s[1] = xn[1]; s[2] = xn[2];
        }
        // The deferred backward sweep of the last permitted iteration (admm.cpp:141-144 with iter = max_iter): instances that
        // exhausted max_iter get p, d from it; the others keep what pass 1 stored (the same lanes store again, in program order).
        const unsigned long long umask = __ballot(valid && !solved);
        if (umask != 0ull)
        {
            const bool unsolved2 = valid2 && ((umask >> c2) & 1ull); // storing lane l serves instance l >> 2 (g = 0 holds its flag at bit c)
            float p[3] = {pN[0], pN[1], pN[2]};
#pragma unroll
            for (int i = N - 2; i >= 0; i--)
            {
                const f32x4 sni = reinterpret_cast<const f32x4_ma *>(stage_w + i * (WAVE * 4))[lane];
                const f32x4 xr = load_xref(tab, wstart, i);
                const f32x4 lin = lin_cost4<EXACT>(cost_of(xr), rho4, sni - dual4(i));
                float pn[3], dd;
                M.riccati(p, lin, pn, dd);
                if constexpr (XPOSE)
                {
                    const f32x4 o = t16_transpose4(f32x4{pn[0], pn[1], pn[2], dd});
                    if (valid && !solved) *reinterpret_cast<f32x4 *>(P.pd + obase4 + i * 16) = o;
                }
                else
                {
                float_ma *slot = stage_w + i * (WAVE * 4);
                reinterpret_cast<f32x4_ma *>(slot)[lane] = f32x4{pn[0], pn[1], pn[2], dd};
                asm volatile("" ::: "memory");
                f32x4 o;
                o[0] = slot[rd_off]; o[1] = slot[rd_off + 64]; o[2] = slot[rd_off + 128]; o[3] = slot[rd_off + 192];
                asm volatile("" ::: "memory");
                if (unsolved2) *reinterpret_cast<f32x4 *>(P.pd + obase2 + i * 16) = o;
                }
                p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
            }
        }
        if (valid && g == 0)
        {
            P.res[4 * inst + 0] = r_ps; P.res[4 * inst + 1] = r_pi;
            P.res[4 * inst + 2] = r_ds; P.res[4 * inst + 3] = r_di;
            P.status[inst] = st;
            P.iter[inst] = itn;
            if (!solved) atomicAdd(P.n_unsolved, 1);
        }
        if (MPC && valid) // the host's plant step of the last solve continues from here
        {
#pragma unroll
            for (int v = 0; v < 3; v++) P.x0buf[inst * NX + 4 * v + g] = x0[v];
            if (g == 0 && P.xref_mode == 1) P.xref_start[inst] = wstart;
        }
    }
    } // tile queue
}

// horizons: the slack of a wave is N KB of LDS, four waves and the shared tables must fit 160 KB: N <= 30
#ifdef TINY_T16_ONLY30 // developer switch: one horizon, for quick looks at the listing
#define TINY_FOR_EACH_TILE16(X) X(30)
#else
#define TINY_FOR_EACH_TILE16(X) X(30) X(25) X(20) X(10)
#endif

// Which waves take their tiles from the short end of a longest-first order (the kernel's two-ended queue; 0 = none, k = every k-th wave).
// Measured by replaying the true iteration counts of tracking batches (tests/fuzz/sim_tile_deque.py): with every 8th wave at the short end the
// makespan of q = 3 / 4 / 6 / 8 tiles per slot drops 8.6 / 7.2 / 4.5 / 2.7 %; at q = 2.5 no slot can fit a tile more and the mid-sized tiles
// left for last cost 2 %, and in index order (no predictor: warm-started steps, the on-chip closed loop) both ends are alike.  So: a predictor's
// order, one solve per launch, at least three tiles per wave slot, and a tile count the two 16-bit halves of the queue word can hold.
// tiny_batch_set_tile_queue(tb, k) / TINYMPC_T16_TAIL=<k> override (0 = plain counter) for A/B runs and tests.
static inline unsigned t16_tail_stride(const RowParams &P, int ntiles, int nblocks, int asked)
{
    static const int env = [] { const char *e = getenv("TINYMPC_T16_TAIL"); return e ? atoi(e) : -1; }();
    const int forced = asked >= 0 ? asked : env; // tiny_batch_set_tile_queue, then the environment
    if (ntiles > 32768) return 0u; // head and tail counts overshoot by at most one failed claim per wave
    if (forced >= 0) return (P.cold_start && P.mpc_steps <= 1) ? (unsigned)(forced > 255 ? 255 : forced) : 0u;
    if (!P.order || P.mpc_steps > 1 || !P.cold_start) return 0u; // (the kernel's cold-start instantiations carry the two-ended queue)
    // how many waves: with q = tiles per slot, same-box A/B of strides 4 / 6 / 8 / 12 (tools/t16_queue_ab.py, kernel ms, two passes each): q = 3: 1.192 / 1.193 / 1.185 / 1.179,
    // q = 4: 1.488 / 1.494 / 1.505 / 1.520 (one counter 1.600), q = 5: 1.808 / 1.831 / 1.858 / 1.854 (1.918); the replay says the same (q = 5: 151.0 against 152.5
    // iterations; q = 8: 240 against 238): every 4th wave for 4 <= q < 8, every 8th otherwise
    const int slots = nblocks * TILE16_WAVES;
    if (ntiles < 3 * slots) return 0u;
    return (ntiles >= 4 * slots && ntiles < 8 * slots) ? 4u : 8u;
}

#ifdef TINY_T16_PI_UNIT
// the PI instantiations are a translation unit of their own (admm_tile16_pi.hip includes this file): they compile beside the others, and the
// device code of the shared-table instantiations — to which the recorded HBM traffic figures are bound (build.device_isa_sha) — does not move
size_t tile16_pi_lds_bytes(int N, bool bounds_ring, bool xref_ring, unsigned pi_flags, int table_rows)
{
    const size_t bnd = bounds_ring ? (size_t)TILE16_WAVES * ((pi_flags & 1u) ? t16_ring_b(xref_ring) : 2) * T16_RING_B : (size_t)2 * N * 4 * sizeof(float4);
    const size_t ref = xref_ring ? (size_t)TILE16_WAVES * ((pi_flags & 2u) ? t16_ring_x(bounds_ring) : 1) * T16_RING_X : (size_t)(table_rows + N - 1) * 4 * sizeof(float4);
    return (size_t)TILE16_WAVES * N * WAVE * sizeof(float4) + bnd + ref;
}

hipError_t launch_admm_tile16_pi(int N, bool exact, bool bounds_ring, bool xref_ring, const RowParams &P, hipStream_t stream, int n_cu, int tail)
{
    const int ntiles = (P.batch + 15) / 16;
    if (n_cu <= 0) n_cu = 256;
    const int want = (ntiles + TILE16_WAVES - 1) / TILE16_WAVES, nblocks = want < n_cu ? want : n_cu; // one persistent workgroup per CU
    if (P.mpc_steps > 1 || (!bounds_ring && !xref_ring)) return hipErrorInvalidValue;
    if (P.xref_mode == 1 && P.table_rows < 1) return hipErrorInvalidValue;
    if (!bounds_ring && P.bounds_inst_stride != 0) return hipErrorInvalidValue;           // a staged table is the batch's
    if (!xref_ring && P.xref_mode != 1 && P.xref_inst_stride != 0) return hipErrorInvalidValue;
    if (xref_ring && P.xref_mode == 1) return hipErrorInvalidValue; // a window of the trajectory table is staged, never read through the slots
    const size_t lds = tile16_pi_lds_bytes(N, bounds_ring, xref_ring, P.pi_flags, P.xref_mode == 1 ? P.table_rows : N);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    RowParams Q = P;
    Q.pi_flags = (Q.pi_flags & 0xffu) | (t16_tail_stride(P, ntiles, nblocks, tail) << 8);
#define TINY_TILE16_PI_LAUNCH3(NN, EX, BRR, XRR)                                                                           \
    {                                                                                                                      \
        auto kern = P.cold_start ? admm_tile16_kernel<NN, EX, true, false, BRR, XRR> : admm_tile16_kernel<NN, EX, false, false, BRR, XRR>; \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WAVE * TILE16_WAVES), lds, stream, Q);                                \
        return hipGetLastError();                                                                                          \
    }
#define TINY_TILE16_PI_LAUNCH(NN, EX)                                             \
    {                                                                             \
        if (bounds_ring && xref_ring) TINY_TILE16_PI_LAUNCH3(NN, EX, true, true)  \
        else if (bounds_ring) TINY_TILE16_PI_LAUNCH3(NN, EX, true, false)         \
        else TINY_TILE16_PI_LAUNCH3(NN, EX, false, true)                          \
    }
#define TINY_TILE16_PI_DISPATCH(NN)                  \
    if (N == NN)                                     \
    {                                                \
        if (exact) TINY_TILE16_PI_LAUNCH(NN, true)   \
        else TINY_TILE16_PI_LAUNCH(NN, false)        \
    }
    TINY_FOR_EACH_TILE16(TINY_TILE16_PI_DISPATCH)
    return hipErrorInvalidValue;
}
#else
bool tile16_supported(int nx, int nu, int N)
{
    if (nx != 12 || nu != 4) return false;
#define TINY_TILE16_CHECK(NN) \
    if (N == NN) return true;
    TINY_FOR_EACH_TILE16(TINY_TILE16_CHECK)
    return false;
}

int tile16_max_table_rows() { return TILE16_MAX_TABLE_ROWS; }

hipError_t launch_admm_tile16(int N, bool exact, const RowParams &P, hipStream_t stream, int n_cu, int tail)
{
    const int ntiles = (P.batch + 15) / 16;
    if (n_cu <= 0) n_cu = 256; // the caller passes the CU count of the handle's device
    const int want = (ntiles + TILE16_WAVES - 1) / TILE16_WAVES, nblocks = want < n_cu ? want : n_cu; // one persistent workgroup per CU
    const int rows = P.xref_mode == 1 ? P.table_rows : N;
    if (rows > TILE16_MAX_TABLE_ROWS) return hipErrorInvalidValue;
    const size_t lds = (size_t)(TILE16_WAVES * N * WAVE + 2 * N * 4 + (rows + N - 1) * 4) * sizeof(float4); // the staged table is padded with N - 1 copies of its last row
    RowParams Q = P;
    Q.pi_flags = (Q.pi_flags & 0xffu) | (t16_tail_stride(P, ntiles, nblocks, tail) << 8);
#define TINY_TILE16_LAUNCH(NN, EX)                                                                                         \
    {                                                                                                                      \
        auto kern = P.mpc_steps > 1 ? (P.cold_start ? admm_tile16_kernel<NN, EX, true, true> : admm_tile16_kernel<NN, EX, false, true>) \
                                    : (P.cold_start ? admm_tile16_kernel<NN, EX, true> : admm_tile16_kernel<NN, EX, false>); \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WAVE * TILE16_WAVES), lds, stream, Q);                                \
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
#endif // TINY_T16_PI_UNIT

} // namespace tinympc
