// Microbenchmark: does issue rate depend on encoding size (VOP2 e32 = 4 B vs VOP3/DPP = 8 B) or on operand source?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float sval)
{
    float acc0 = threadIdx.x, acc1 = 1.f, x = 0.5f + threadIdx.x, m = 1.0001f;
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MODE == 0) asm volatile(REP16("v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3\n v_fmac_f32_e32 %0, %2, %3\n v_fmac_f32_e32 %1, %2, %3\n") : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 1) asm volatile(REP16("v_fmac_f32_e64 %0, %2, %3\n v_fmac_f32_e64 %1, %2, %3\n v_fmac_f32_e64 %0, %2, %3\n v_fmac_f32_e64 %1, %2, %3\n") : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 2) asm volatile(REP16("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %2, %3, %1\n") : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 3) asm volatile(REP16("v_fmac_f32_e32 %0, %3, %2\n v_fmac_f32_e32 %1, %3, %2\n v_fmac_f32_e32 %0, %3, %2\n v_fmac_f32_e32 %1, %3, %2\n") : "+v"(acc0), "+v"(acc1) : "v"(x), "s"(sval));
        else if constexpr (MODE == 4) asm volatile(REP16("v_add_f32_e32 %0, %2, %0\n v_add_f32_e32 %1, %2, %1\n v_add_f32_e32 %0, %2, %0\n v_add_f32_e32 %1, %2, %1\n") : "+v"(acc0), "+v"(acc1) : "v"(x));
        else if constexpr (MODE == 5) asm volatile(REP16("v_mul_f32_e32 %0, %2, %0\n v_mul_f32_e32 %1, %2, %1\n v_mul_f32_e32 %0, %2, %0\n v_mul_f32_e32 %1, %2, %1\n") : "+v"(acc0), "+v"(acc1) : "v"(m));
        else if constexpr (MODE == 6) // alternating 4-byte add and 8-byte dpp mul (exact-mode mix)
            asm volatile(REP16("v_mul_f32_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_e32 %1, %2, %1\n v_mul_f32_dpp %0, %2, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_e32 %1, %2, %1\n") : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 7) asm volatile(REP16("v_max_f32_e32 %0, %2, %0\n v_max_f32_e32 %1, %2, %1\n v_max_f32_e32 %0, %2, %0\n v_max_f32_e32 %1, %2, %1\n") : "+v"(acc0), "+v"(acc1) : "v"(x));
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc0 + acc1;
}
template <int MODE>
void run(const char *name, float *d, int w)
{
    const int blocks = 256 * 4 * w, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-30s waves/SIMD %d: %.2f nominal cycles per wave-instruction per SIMD\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {2, 4})
    {
        run<0>("fmac e32 (4B)", d, w); run<1>("fmac e64 (8B VOP3)", d, w); run<2>("fma VOP3 (8B)", d, w); run<3>("fmac e32 sgpr src0", d, w);
        run<4>("add e32", d, w); run<5>("mul e32", d, w); run<7>("max e32", d, w); run<6>("mul_dpp + add alternating", d, w);
    }
}
