// Microbenchmark: issue rate of v_fmac_f32_dpp by DPP control, and of alternatives, at 2 and 4 waves/SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define CHAIN4(CTRL) "v_fmac_f32_dpp %0, %2, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_fmac_f32_dpp %1, %2, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_fmac_f32_dpp %0, %2, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                     "v_fmac_f32_dpp %1, %2, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters)
{
    float acc0 = threadIdx.x, acc1 = 1.f, x = 0.5f + threadIdx.x, m = 1.0001f;
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MODE == 0) asm volatile(REP16(CHAIN4("quad_perm:[1,1,1,1]")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 1) asm volatile(REP16(CHAIN4("row_ror:3")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 2) asm volatile(REP16(CHAIN4("row_shr:1")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 3) asm volatile(REP16(CHAIN4("row_mirror")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 4) asm volatile(REP16(CHAIN4("row_newbcast:5")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 5) asm volatile(REP16(CHAIN4("row_half_mirror")) : "+v"(acc0), "+v"(acc1) : "v"(x), "v"(m));
        else if constexpr (MODE == 6) // plain with SGPR operand
            asm volatile(REP16("v_fmac_f32 %0, s4, %2\n v_fmac_f32 %1, s5, %2\n v_fmac_f32 %0, s6, %2\n v_fmac_f32 %1, s7, %2\n")
                         : "+v"(acc0), "+v"(acc1) : "v"(x) : "s4", "s5", "s6", "s7");
        else if constexpr (MODE == 7) // v_pk_fma_f32 (2 fmas per lane per instruction)
        {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {acc0, acc1}, xx = {x, x}, mm = {m, m};
            asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n")
                         : "+v"(a) : "v"(xx), "v"(mm));
            acc0 = a[0]; acc1 = a[1];
        }
        else if constexpr (MODE == 8) // SDWA? not for f32 lanes. v_mov_b32_dpp only
        {
            float t0, t1;
            asm volatile(REP16("v_mov_b32_dpp %0, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %0, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "=&v"(t0), "=&v"(t1) : "v"(x));
            acc0 += t0; acc1 += t1;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc0 + acc1;
}
template <int MODE>
void run(const char *name, float *d, int w)
{
    const int blocks = 256 * 4 * w, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s waves/SIMD %d: %.2f nominal cycles per wave-instruction per SIMD\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * w));
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    for (int w : {2, 4})
    {
        run<0>("fmac_dpp quad_perm", d, w); run<1>("fmac_dpp row_ror:3", d, w); run<2>("fmac_dpp row_shr:1", d, w);
        run<3>("fmac_dpp row_mirror", d, w); run<5>("fmac_dpp row_half_mirror", d, w); run<4>("fmac_dpp row_newbcast", d, w);
        run<6>("fmac sgpr operand", d, w); run<7>("pk_fma_f32", d, w); run<8>("mov_dpp row_newbcast", d, w);
    }
}
