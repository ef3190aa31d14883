// tile_math.h — per-step arithmetic of the "tile" mapping (16 instances = the 16 columns of v_mfma_f32_16x16x4_f32; lane
// 16*g + c holds rows {4*ch + g} of instance c), shared by the streaming kernel (admm_stream.hip) and the register-resident
// kernel (admm_tile.hip).  The gain x state products are MFMA issues (fp32 fma chains); see admm_stream.hip.
#pragma once
#include "tinympc_internal.h"

namespace tinympc
{

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NC>
__device__ __forceinline__ void ldv(const float *p, float (&o)[NC])
{
#pragma unroll
    for (int k = 0; k < NC; k++) o[k] = p[k];
}
template <int NC>
__device__ __forceinline__ void stv(float *p, const float (&v)[NC], bool pred)
{
    if (pred)
    {
#pragma unroll
        for (int k = 0; k < NC; k++) p[k] = v[k];
    }
}

// MFMA A operands of one problem class, one VGPR each.
template <class D>
struct Operands
{
    float A1[D::NT][D::NXC], A2[D::NTX][D::NUC], A3[D::NT][D::NXC], A4[D::NTX][D::NUC], A5[D::NTU][D::NUC],
        AP[D::NTX][D::NXC];
    __device__ __forceinline__ void load(const float *opnd, int lane)
    {
        int idx = 0;
#pragma unroll
        for (int t = 0; t < D::NT; t++)
#pragma unroll
            for (int k = 0; k < D::NXC; k++) A1[t][k] = opnd[(idx++) * WAVE + lane];
#pragma unroll
        for (int t = 0; t < D::NTX; t++)
#pragma unroll
            for (int m = 0; m < D::NUC; m++) A2[t][m] = opnd[(idx++) * WAVE + lane];
#pragma unroll
        for (int t = 0; t < D::NT; t++)
#pragma unroll
            for (int k = 0; k < D::NXC; k++) A3[t][k] = opnd[(idx++) * WAVE + lane];
#pragma unroll
        for (int t = 0; t < D::NTX; t++)
#pragma unroll
            for (int m = 0; m < D::NUC; m++) A4[t][m] = opnd[(idx++) * WAVE + lane];
#pragma unroll
        for (int t = 0; t < D::NTU; t++)
#pragma unroll
            for (int m = 0; m < D::NUC; m++) A5[t][m] = opnd[(idx++) * WAVE + lane];
#pragma unroll
        for (int t = 0; t < D::NTX; t++)
#pragma unroll
            for (int k = 0; k < D::NXC; k++) AP[t][k] = opnd[(idx++) * WAVE + lane];
    }
};

// u_i = -Kinf x_i - d_i ; x_{i+1} = Adyn x_i + Bdyn u_i      (admm.cpp:31,35)
template <class D>
__device__ __forceinline__ void lqr_step(const Operands<D> &op, const float (&xs)[D::NXC], const float (&d)[D::NUC],
                                         float (&us)[D::NUC], float (&xn)[D::NXC])
{
    f32x4 acc[D::NT];
#pragma unroll
    for (int t = 0; t < D::NT; t++)
    {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D::NXC; k++) acc[t] = MFMA(op.A1[t][k], xs[k], acc[t]);
    }
#pragma unroll
    for (int m = 0; m < D::NUC; m++) us[m] = acc[(D::NXC + m) / 4][(D::NXC + m) % 4] - d[m];
#pragma unroll
    for (int t = 0; t < D::NTX; t++)
#pragma unroll
        for (int m = 0; m < D::NUC; m++) acc[t] = MFMA(op.A2[t][m], us[m], acc[t]);
#pragma unroll
    for (int k = 0; k < D::NXC; k++) xn[k] = acc[k / 4][k % 4];
}

// d_i = Quu_inv (Bdyn^T p_{i+1} + r_i) ; p_i = q_i + AmBKt p_{i+1} - Kinf^T r_i     (admm.cpp:19-20)
template <class D>
__device__ __forceinline__ void riccati_step(const Operands<D> &op, float (&p)[D::NXC], const float (&q)[D::NXC],
                                             const float (&r)[D::NUC], float (&d)[D::NUC])
{
    f32x4 acc[D::NT];
#pragma unroll
    for (int t = 0; t < D::NT; t++)
    {
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            // C input = the stacked linear-cost vector [q_i ; r_i]
            const int ch = 4 * t + e;
            float val = 0.f;
            if (ch < D::NXC) val = q[ch < D::NXC ? ch : 0];
            else if (ch < D::NCH) val = r[(ch >= D::NXC && ch < D::NCH) ? ch - D::NXC : 0];
            acc[t][e] = val;
        }
#pragma unroll
        for (int k = 0; k < D::NXC; k++) acc[t] = MFMA(op.A3[t][k], p[k], acc[t]);
    }
    float tu[D::NUC];
#pragma unroll
    for (int m = 0; m < D::NUC; m++) tu[m] = acc[(D::NXC + m) / 4][(D::NXC + m) % 4];
#pragma unroll
    for (int t = 0; t < D::NTX; t++)
#pragma unroll
        for (int m = 0; m < D::NUC; m++) acc[t] = MFMA(op.A4[t][m], r[m], acc[t]);
    f32x4 dacc[D::NTU];
#pragma unroll
    for (int t = 0; t < D::NTU; t++)
    {
        dacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < D::NUC; m++) dacc[t] = MFMA(op.A5[t][m], tu[m], dacc[t]);
    }
#pragma unroll
    for (int k = 0; k < D::NXC; k++) p[k] = acc[k / 4][k % 4];
#pragma unroll
    for (int m = 0; m < D::NUC; m++) d[m] = dacc[(D::NXC + m) / 4 - D::TU0][(D::NXC + m) % 4];
}

// max over the four lanes (gq = 0..3) that hold one instance's rows
__device__ __forceinline__ float inst_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

} // namespace tinympc
