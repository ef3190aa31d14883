// Microbenchmark: issue rate of v_fmac_f32 / v_fmac_f32_dpp(row_newbcast) / v_mul_f32_dpp+v_add_f32 chains on gfx950
// at 1, 2, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/dpp_rate.hip -o build/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    float acc0 = threadIdx.x, acc1 = 1.f, acc2 = 2.f, acc3 = 3.f, x = 0.5f + threadIdx.x, m = 1.0001f, t0, t1, t2, t3;
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MODE == 0) // independent plain fmac x4
        {
            asm volatile(REP16("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(x), "v"(m));
        }
        else if constexpr (MODE == 1) // dependent plain fmac chain
        {
            asm volatile(REP16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n")
                         : "+v"(acc0) : "v"(x), "v"(m));
        }
        else if constexpr (MODE == 2) // dependent dpp fmac chain (what the fast kernel does)
        {
            asm volatile(REP16("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(acc0) : "v"(x), "v"(m));
        }
        else if constexpr (MODE == 3) // independent dpp mul x4 (what the exact kernel's product groups do)
        {
            asm volatile(REP16("v_mul_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mul_f32_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mul_f32_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mul_f32_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x), "v"(m));
            acc0 += t0 + t1 + t2 + t3;
        }
        else if constexpr (MODE == 4) // dependent plain add chain
        {
            asm volatile(REP16("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n")
                         : "+v"(acc0) : "v"(x));
        }
        else if constexpr (MODE == 5) // two independent dpp fmac chains interleaved
        {
            asm volatile(REP16("v_fmac_f32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %1, %2, %3 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_fmac_f32_dpp %1, %2, %3 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
template <int MODE>
void run(const char *name, float *d, int waves_per_simd)
{
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_wave = (double)iters * 64;
    double cyc = ms * 1e-3 * 2.4e9; // at nominal 2.4 GHz
    printf("%-34s waves/SIMD %d: %.2f ms  -> %.2f nominal cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           cyc / (insts_per_wave * waves_per_simd));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {1, 2, 4, 8})
    {
        run<0>("fmac x4 independent", d, w);
        run<1>("fmac dependent chain", d, w);
        run<2>("fmac_dpp dependent chain", d, w);
        run<5>("fmac_dpp 2 chains interleaved", d, w);
        run<3>("mul_dpp x4 independent", d, w);
        run<4>("add dependent chain", d, w);
    }
    return 0;
}
