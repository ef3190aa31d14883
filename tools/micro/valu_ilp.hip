// VALU issue rate of ONE wave per SIMD (and two) as a function of instruction-level parallelism: K independent chains of
// dependent v_add_f32, interleaved round-robin in one asm block.  Also the same next to back-to-back MFMAs (16x16x1_4b).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_ilp.hip -o build/valu_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int K, bool MFMA>
__global__ __launch_bounds__(64) void k(float *out, const float *in, int iters)
{
    extern __shared__ float pad[];
    float s0 = in[threadIdx.x], s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7, c = in[64 + threadIdx.x];
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MFMA) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(s0, c, acc, 0, 0, 0);
        if constexpr (K == 1) asm volatile(REP16("v_add_f32 %0, %0, %1\n" "v_add_f32 %0, %0, %1\n" "v_add_f32 %0, %0, %1\n" "v_add_f32 %0, %0, %1\n") : "+v"(s1) : "v"(c));
        if constexpr (K == 2) asm volatile(REP16("v_add_f32 %0, %0, %2\n" "v_add_f32 %1, %1, %2\n" "v_add_f32 %0, %0, %2\n" "v_add_f32 %1, %1, %2\n") : "+v"(s1), "+v"(s2) : "v"(c));
        if constexpr (K == 4) asm volatile(REP16("v_add_f32 %0, %0, %4\n" "v_add_f32 %1, %1, %4\n" "v_add_f32 %2, %2, %4\n" "v_add_f32 %3, %3, %4\n") : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4) : "v"(c));
        if constexpr (K == 8) asm volatile(REP16("v_add_f32 %0, %0, %8\n" "v_add_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %8\n" "v_add_f32 %3, %3, %8\n") REP16("v_add_f32 %4, %4, %8\n" "v_add_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %8\n" "v_add_f32 %7, %7, %8\n")
                                           : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(c));
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + acc[0] + acc[5];
}
template <int K, bool MFMA>
void run(int wps, float *o, float *in)
{
    const int iters = 4000, n = (K == 8 ? 128 : 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = (160 * 1024 / (4 * wps)) - 64; // workgroups of one wave: exactly wps per SIMD fit
    hipFuncSetAttribute((const void *)k<K, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<K, MFMA>), dim3(256 * 4 * wps), dim3(64), lds, 0, o, in, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, MFMA>), dim3(256 * 4 * wps), dim3(64), lds, 0, o, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chains %d  %s  waves/SIMD %d : %7.3f ms  %6.2f nominal cycles per v_add per SIMD\n", K, MFMA ? "+1 mfma/64 adds" : "               ", wps, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * n * wps));
}
int main()
{
    float *o, *in; hipMalloc(&o, 256 * 4 * 8 * 256); hipMalloc(&in, 512); hipMemset(in, 0, 512);
    for (int wps = 1; wps <= 2; wps++)
    {
        run<1, false>(wps, o, in); run<2, false>(wps, o, in); run<4, false>(wps, o, in); run<8, false>(wps, o, in);
        run<4, true>(wps, o, in); run<8, true>(wps, o, in);
    }
    return 0;
}
