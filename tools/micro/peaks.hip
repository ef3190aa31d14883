// Measured peaks of the box next to the datasheet figures: HBM stream copy and fp32 FMA loops.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/peaks.hip -o build/peaks && build/peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_kernel(const float4 *__restrict__ a, float *__restrict__ out, size_t n)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

// MODE 0: v_fma_f32 (VGPR operands), 1: v_pk_fma_f32 (two fp32 fmas per lane and instruction), 2: v_mul_f32 + v_add_f32 pairs
// (what separately rounded arithmetic issues), 3: v_fmac_f32_dpp row_newbcast (the cross-lane MAC of the row kernels)
template <int MODE>
__global__ __launch_bounds__(256) void fma_kernel(float *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-3f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int r = 0; r < 8; r++)
        {
            if constexpr (MODE == 0)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            else if constexpr (MODE == 1)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc));
            else if constexpr (MODE == 2)
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_mul_f32 %1, %1, %8\n v_add_f32 %1, %1, %9\n"
                             "v_mul_f32 %2, %2, %8\n v_add_f32 %2, %2, %9\n v_mul_f32 %3, %3, %8\n v_add_f32 %3, %3, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            else
                asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
int run_fma(const char *name, float *out, double flops_per_instr_lane, int waves_per_simd)
{
    const int iters = 4096, blocks = 256 * waves_per_simd; // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)iters * 8 * 8, fl = instr * flops_per_instr_lane * 256.0 * blocks;
    printf("%-46s %d waves/SIMD: %7.1f TFLOP/s  (%.3f ms)\n", name, waves_per_simd, fl / (ms * 1e-3) / 1e12, ms);
    return 0;
}

int main()
{
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    float4 *a, *b; float *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64 << 20));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {2048, 8192, 32768})
    {
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n4);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream copy 2 GiB -> 2 GiB, %5d blocks: %7.1f GB/s read+write\n", blocks, 5.0 * 2 * bytes / (ms * 1e-3) / 1e9);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, a, out, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream read 2 GiB,          %5d blocks: %7.1f GB/s\n", blocks, 5.0 * bytes / (ms * 1e-3) / 1e9);
    }
    for (int w : {2, 4, 8})
    {
        if (run_fma<0>("v_fma_f32", out, 2, w)) return 1;
        if (run_fma<1>("v_pk_fma_f32 (2 fmas per lane)", out, 4, w)) return 1;
        if (run_fma<2>("v_mul_f32 + v_add_f32 (separately rounded)", out, 1, w)) return 1;
        if (run_fma<3>("v_fmac_f32_dpp row_newbcast", out, 2, w)) return 1;
    }
    return 0;
}
