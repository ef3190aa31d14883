// wave_math.h — per-step arithmetic of the "one wavefront = one instance" mapping (lane r owns row r of the stacked vector
// [x ; u], 16 < nx + nu <= 64), shared by the streaming wave kernel (admm_wave.hip) and the state-on-chip one
// (admm_waveres.hip).  See admm_wave.hip for the mapping and the reduction orders.
#pragma once
#include "rowlane_math.h"

namespace tinympc
{

enum : int { PLAN_GEMV = 3 };

// Eigen's row-major GEMV inner product (general_matrix_vector_product, RowMajor lhs): four packet lanes accumulated
// sequentially FROM ZERO, predux (c0+c2)+(c1+c3), scalar leftover, then res = 0 + 1*acc
template <int NN>
__device__ __forceinline__ float reduce_gemv(const float (&t)[NN])
{
    constexpr int NPK = NN / 4;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
    for (int k = 0; k < NPK; k++)
    {
        c0 = c0 + t[4 * k + 0]; c1 = c1 + t[4 * k + 1]; c2 = c2 + t[4 * k + 2]; c3 = c3 + t[4 * k + 3];
    }
    float res = (c0 + c2) + (c1 + c3);
#pragma unroll
    for (int k = 4 * NPK; k < NN; k++) res = res + t[k];
    return 0.f + res;
}
template <int PLAN, int NN>
__device__ __forceinline__ float wreduce(const float (&t)[NN])
{
    if constexpr (PLAN == PLAN_GEMV) return reduce_gemv(t);
    else return reduce<PLAN>(t);
}

template <int NX, int NU>
struct WavePlans
{
    static_assert(NX > 1 && NU >= 1 && NX + NU <= 64, "wave kernel needs 1 < nx, nx + nu <= 64");
    static_assert((NX <= 4 || NX % 4 == 0) && (NU <= 4 || NU % 4 == 0), "exact arithmetic is defined for nx, nu <= 4 or multiples of 4 (rowlane_math.h)");
    static constexpr bool GEMV = (NU >= 8 && NX >= 8); // product_type_selector<Large,1,Large> = GemvProduct
    // forward_pass (admm.cpp:31,35)
    static constexpr int FWD_U = (NU > 1 && NU % 4 == 0) ? PLAN_SEQ : (NU == 1 ? plan_vec(NX) : plan_novec(NX));
    static constexpr int FWD_XA = (NX % 4 == 0) ? PLAN_SEQ : plan_novec(NX);
    static constexpr int FWD_XB = (NX % 4 == 0) ? PLAN_SEQ : plan_novec(NU);
    static constexpr int TERM = plan_vec(NX);                                   // admm.cpp:83
    // backward_pass_grad (admm.cpp:19-20)
    static constexpr int BWD_TMP = GEMV ? PLAN_GEMV : plan_vec(NX);
    static constexpr int BWD_D = GEMV ? PLAN_SEQ : ((NU > 1 && NU % 4 == 0) ? PLAN_SEQ : plan_novec(NU));
    static constexpr int BWD_PA = (NU == 1 && NX % 4 == 0) ? PLAN_SEQ : plan_novec(NX);
    static constexpr int BWD_PK = plan_vec(NU);
};

// t[k] = M[k] * s[K0 + k].  The broadcast goes through LDS: every lane stores its element, then all lanes read the same
// 16-byte groups (ds_read_b128, broadcast reads are conflict free) and multiply with plain VGPR operands.  LDS
// operations of one wave execute in order, so no barrier is needed between the store and the loads.  (The alternative,
// v_readlane_b32 + v_mul_f32 with an SGPR operand, costs two 4-cycle VALU instructions per product: measured 1.2-1.4x slower.)
template <int K0, int CNT>
__device__ __forceinline__ void lane_products(float (&t)[CNT], float s, const float (&M)[CNT], float *vec, int lane)
{
    vec[lane] = s;
    if constexpr (K0 % 4 == 0 && CNT % 4 == 0)
    {
#pragma unroll
        for (int k4 = 0; k4 < CNT / 4; k4++)
        {
            const float4 v = reinterpret_cast<const float4 *>(vec + K0)[k4];
            t[4 * k4 + 0] = M[4 * k4 + 0] * v.x; t[4 * k4 + 1] = M[4 * k4 + 1] * v.y;
            t[4 * k4 + 2] = M[4 * k4 + 2] * v.z; t[4 * k4 + 3] = M[4 * k4 + 3] * v.w;
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < CNT; k++) t[k] = M[k] * vec[K0 + k];
    }
}

// fma arithmetic: acc += sum_k M[k] * s[K0 + k], four interleaved fma chains (no order to keep), same LDS broadcast
template <int K0, int CNT>
__device__ __forceinline__ float lane_fma_dot(float acc, float s, const float (&M)[CNT], float *vec, int lane)
{
    vec[lane] = s;
    if constexpr (K0 % 4 == 0 && CNT % 4 == 0)
    {
        float a0 = acc, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < CNT / 4; k4++)
        {
            const float4 v = reinterpret_cast<const float4 *>(vec + K0)[k4];
            a0 = __builtin_fmaf(M[4 * k4 + 0], v.x, a0); a1 = __builtin_fmaf(M[4 * k4 + 1], v.y, a1);
            a2 = __builtin_fmaf(M[4 * k4 + 2], v.z, a2); a3 = __builtin_fmaf(M[4 * k4 + 3], v.w, a3);
        }
        return (a0 + a1) + (a2 + a3);
    }
    else
    {
#pragma unroll
        for (int k = 0; k < CNT; k++) acc = __builtin_fmaf(M[k], vec[K0 + k], acc);
        return acc;
    }
}

template <int NX, int NU>
struct WaveGains
{
    float M1[NX], M2[NU], M3[NX], M45[NU]; // same packing as RowGains (tinympc_batch.hip: pack_gains), row width 64
    __device__ __forceinline__ void load(const float *mats, int lane)
    {
        const float *m = mats + lane;
#pragma unroll
        for (int k = 0; k < NX; k++) M1[k] = m[(k) * WAVE];
#pragma unroll
        for (int k = 0; k < NU; k++) M2[k] = m[(NX + k) * WAVE];
#pragma unroll
        for (int k = 0; k < NX; k++) M3[k] = m[(NX + NU + k) * WAVE];
#pragma unroll
        for (int k = 0; k < NU; k++) M45[k] = m[(2 * NX + NU + k) * WAVE];
    }
};

// ---- the same arithmetic cut at its broadcasts (round 3, admm_waveres.hip): a broadcast is "every lane stores its element, then every
// lane reads the 16-byte groups it needs"; issued as early as its input exists and consumed as late as possible, independent work
// of the step sits between the two halves instead of the wave waiting out the LDS round trip.
template <int K0, int CNT>
__device__ __forceinline__ void bcast_fetch(float (&v)[CNT], const float *vec)
{
    static_assert(K0 % 4 == 0 && CNT % 4 == 0, "whole 16-byte groups");
#pragma unroll
    for (int k4 = 0; k4 < CNT / 4; k4++)
    {
        const float4 q = reinterpret_cast<const float4 *>(vec + K0)[k4];
        v[4 * k4 + 0] = q.x; v[4 * k4 + 1] = q.y; v[4 * k4 + 2] = q.z; v[4 * k4 + 3] = q.w;
    }
}
template <int CNT>
__device__ __forceinline__ void products_of(float (&t)[CNT], const float (&M)[CNT], const float (&v)[CNT])
{
#pragma unroll
    for (int k = 0; k < CNT; k++) t[k] = M[k] * v[k];
}
// acc + sum_k M[k] * v[k] as four interleaved fma chains: exactly lane_fma_dot's arithmetic
template <int CNT>
__device__ __forceinline__ float fma_dot_of(float acc, const float (&M)[CNT], const float (&v)[CNT])
{
    static_assert(CNT % 4 == 0, "whole groups of four");
    float a0 = acc, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < CNT / 4; k4++)
    {
        a0 = __builtin_fmaf(M[4 * k4 + 0], v[4 * k4 + 0], a0); a1 = __builtin_fmaf(M[4 * k4 + 1], v[4 * k4 + 1], a1);
        a2 = __builtin_fmaf(M[4 * k4 + 2], v[4 * k4 + 2], a2); a3 = __builtin_fmaf(M[4 * k4 + 3], v[4 * k4 + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}
// the gains of ONE sweep (same packing as WaveGains): loaded at the head of the sweep, so that the other sweep's 48 registers are free
template <int NX, int NU>
struct WaveGainsF
{
    float M1[NX], M2[NU];
    __device__ __forceinline__ void load(const float *mats, int lane)
    {
        const float *m = mats + lane;
#pragma unroll
        for (int k = 0; k < NX; k++) M1[k] = m[(k) * WAVE];
#pragma unroll
        for (int k = 0; k < NU; k++) M2[k] = m[(NX + k) * WAVE];
    }
};
template <int NX, int NU>
struct WaveGainsB
{
    float M3[NX], M45[NU];
    __device__ __forceinline__ void load(const float *mats, int lane)
    {
        const float *m = mats + lane;
#pragma unroll
        for (int k = 0; k < NX; k++) M3[k] = m[(NX + NU + k) * WAVE];
#pragma unroll
        for (int k = 0; k < NU; k++) M45[k] = m[(2 * NX + NU + k) * WAVE];
    }
};

// forward_pass step (admm.cpp:31,35)
template <int NX, int NU, bool EXACT = true>
__device__ __forceinline__ void wave_lqr_step(const WaveGains<NX, NU> &G, float *vec, int lane, bool is_x, bool is_u, float s, float ci, float &sv, float &xn)
{
    using PL = WavePlans<NX, NU>;
    if constexpr (!EXACT)
    {
        const float acc = lane_fma_dot<0, NX>(0.f, s, G.M1, vec, lane); // u rows of M1 hold -Kinf (pack_gains, fast)
        const float un = acc - ci;
        xn = lane_fma_dot<NX, NU>(acc, un, G.M2, vec, lane);
        sv = is_u ? un : s;
        return;
    }
    float t[NX];
    lane_products<0, NX>(t, s, G.M1, vec, lane);
    float acc;
    if constexpr (PL::FWD_U == PL::FWD_XA) acc = wreduce<PL::FWD_XA>(t);
    else acc = is_x ? wreduce<PL::FWD_XA>(t) : wreduce<PL::FWD_U>(t);
    const float un = -acc - ci; // u rows of M1 hold +Kinf: -(K x) - d with the sum negated, like the reference
    float t2[NU];
    lane_products<NX, NU>(t2, un, G.M2, vec, lane);
    xn = acc + wreduce<PL::FWD_XB>(t2);
    sv = is_u ? un : s;
}

// backward_pass_grad step (admm.cpp:19-20)
template <int NX, int NU, bool EXACT = true>
__device__ __forceinline__ void wave_riccati_step(const WaveGains<NX, NU> &G, float *vec, int lane, bool is_x, float p, float lin, float &pn, float &dd)
{
    using PL = WavePlans<NX, NU>;
    if constexpr (!EXACT)
    {
        const float wv = lane_fma_dot<0, NX>(lin, p, G.M3, vec, lane);  // q + AmBKt*p  |  Bdyn^T*p + r
        dd = lane_fma_dot<NX, NU>(0.f, wv, G.M45, vec, lane);           // u rows: Quu_inv
        pn = lane_fma_dot<NX, NU>(wv, lin, G.M45, vec, lane);           // x rows hold -Kinf^T (pack_gains, fast)
        return;
    }
    float t[NX];
    lane_products<0, NX>(t, p, G.M3, vec, lane);
    float dot;
    if constexpr (PL::BWD_PA == PL::BWD_TMP) dot = wreduce<PL::BWD_PA>(t);
    else dot = is_x ? wreduce<PL::BWD_PA>(t) : wreduce<PL::BWD_TMP>(t);
    const float wv = lin + dot; // q + AmBKt*p  |  Bdyn^T*p + r
    float tk[NU], td[NU];
    lane_products<NX, NU>(tk, lin, G.M45, vec, lane); // Kinf^T * r
    lane_products<NX, NU>(td, wv, G.M45, vec, lane);  // Quu_inv * (Bdyn^T p + r)
    pn = wv - wreduce<PL::BWD_PK>(tk);
    if constexpr (PL::GEMV) dd = 0.f + (0.f + wreduce<PLAN_SEQ>(td)); // 0 + 1*(0 + dot_seq), as the GEMV path leaves it
    else dd = wreduce<PL::BWD_D>(td);
}

} // namespace tinympc
