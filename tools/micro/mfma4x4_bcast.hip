// Probe for a "broadcast-multiply on the matrix cores" formulation of the one-wave-per-instance class (16 < nx + nu <= 64):
//   v_mfma_f32_4x4x1_16b_f32 computes sixteen 4x4 outer products D_b[i][j] = A_b[i] * B_b[j] (K = 1: with C = -0 every output is ONE
//   separately rounded product).  With the A-broadcast controls (cbsz = 4, abid = b'), block b' of A feeds ALL sixteen blocks:
//   D_b[i][j] = A_b'[i] * B_b[j].  Lane 4b + j then holds, in register i, gain(row 4b + j) * value_i — the product of ONE element
//   of up to four instances' state vectors (supplied by the four lanes of block b') with every row's gain: a mat-vec column
//   without any LDS round trip or DPP broadcast.
// (1) layout + broadcast semantics + bit-exactness of the products against host arithmetic (signs of zeros included);
// (2) issue rate of independent 4x4x1 MFMAs, alone and with two packed adds per MFMA, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mfma4x4_bcast.hip -o build/mfma4x4_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ABID>
__global__ void probe(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f32x4 c = {-0.f, -0.f, -0.f, -0.f};
    f32x4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, ABID, 0); // cbsz = 4: one A block for all 16
    for (int e = 0; e < 4; e++) d[e * 64 + l] = r[e];
}

template <int ADDS>
__global__ __launch_bounds__(256, 1) void rate(float *out, const float *in, int iters, long long *clk)
{
    const int l = threadIdx.x & 63;
    float g[8];
    for (int k = 0; k < 8; k++) g[k] = in[k * 64 + l];
    float s = in[8 * 64 + l];
    const f32x4 negz = {-0.f, -0.f, -0.f, -0.f};
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
    {
        // eight independent MFMAs in flight, then their sums (the shape of a mat-vec column group: issue, then consume in order)
        const f32x4 p0 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[0], negz, 4, 0, 0), p1 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[1], negz, 4, 1, 0),
                    p2 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[2], negz, 4, 2, 0), p3 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[3], negz, 4, 3, 0),
                    p4 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[4], negz, 4, 4, 0), p5 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[5], negz, 4, 5, 0),
                    p6 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[6], negz, 4, 6, 0), p7 = __builtin_amdgcn_mfma_f32_4x4x1f32(s, g[7], negz, 4, 7, 0);
        __builtin_amdgcn_sched_barrier(0);
#define USE(p)                                                 \
        if constexpr (ADDS)                                    \
        {                                                      \
            acc0 = acc0 + __builtin_shufflevector(p, p, 0, 1); \
            acc1 = acc1 + __builtin_shufflevector(p, p, 2, 3); \
        }                                                      \
        else acc0[0] += p[0];
        USE(p0) USE(p1) USE(p2) USE(p3) USE(p4) USE(p5) USE(p6) USE(p7)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc0[1] + acc1[0] + acc1[1];
}

int main()
{
    std::vector<float> a(64), b(64), d(256);
    unsigned lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return ((lcg >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.0f; };
    for (int i = 0; i < 64; i++) { a[i] = rnd(); b[i] = rnd(); }
    a[4 * 5 + 1] = 0.f; a[4 * 5 + 2] = -0.f; b[7] = -0.f; b[9] = 0.f; // signed zeros
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    int bad = 0;
    auto check = [&](int abid) {
        hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++)
            for (int i = 0; i < 4; i++)
            {
                const float want = a[4 * abid + i] * b[l]; // D_b[i][j] with lane = 4b + j: A of block abid, lane i within it; B of this lane
                float got = d[i * 64 + l];
                if (std::memcmp(&want, &got, 4) != 0) { if (bad < 8) std::printf("abid %d lane %d i %d: got %g want %g\n", abid, l, i, got, want); bad++; }
            }
    };
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, da, db, dd); check(0);
    hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, da, db, dd); check(5);
    hipLaunchKernelGGL(probe<15>, dim3(1), dim3(64), 0, 0, da, db, dd); check(15);
    std::printf("broadcast semantics D[lane 4b+j][reg i] = A[lane 4*abid + i] * B[lane 4b+j], bitwise incl. zero signs: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);

    float *o, *in; long long *clk;
    hipMalloc(&o, 256 * 256 * 4); hipMalloc(&in, 9 * 64 * 4); hipMalloc(&clk, 8);
    std::vector<float> hin(9 * 64);
    for (auto &v : hin) v = rnd();
    hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    for (int adds = 0; adds < 2; adds++)
    {
        const int iters = 20000;
        for (int w = 0; w < 5; w++)
        {
            if (adds) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, o, in, iters, clk);
            else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, o, in, iters, clk);
        }
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        std::printf("%s: %.2f shader clocks per v_mfma_f32_4x4x1_16b_f32 (one wave per SIMD, 8 independent MFMAs per loop)\n",
                    adds ? "MFMA + 2 packed adds each" : "MFMA alone              ", (double)c / ((double)iters * 8));
    }
    return bad != 0;
}
