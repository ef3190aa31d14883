// pk_rate.hip — how fast can ONE wave per SIMD (and two) add: scalar v_add_f32 chains against v_pk_add_f32 chains, alone and
// beside the exact-product MFMA (v_mfma_f32_16x16x1_4b_f32) at the density the tile kernels issue it (one MFMA per 16
// scalar adds = 8 packed adds).  Reported per variant: wall time -> nominal (2.4 GHz) clocks per ADD (a packed add counts as
// two) per SIMD, and the true shader clocks per add from s_memtime of one wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/pk_rate.hip -o build/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

// MODE 0: scalar, ILP 4 (4 chains x 4 dependent adds per block of 16)
// MODE 1: packed, ILP 2 (2 chains x 4 dependent pk_adds per block of 8 = 16 adds)
// MODE 2: packed, ILP 4 (4 chains x 2)
// MODE 3: scalar, ILP 1     MODE 4: packed, ILP 1
template <int MODE, bool MFMA>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters, long long *clk)
{
    extern __shared__ float pad[];
    float s0 = in[threadIdx.x & 63], s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, c = in[64 + (threadIdx.x & 63)];
    f32x2 p0 = {s0, s1}, p1 = {s2, s3}, p2 = {s1, s2}, p3 = {s3, s0}, cc = {c, c};
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MFMA) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(s0, c, acc, 0, 0, 0);
        if constexpr (MODE == 0)
            asm volatile(REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(c));
        if constexpr (MODE == 1)
            asm volatile(REP4("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n") : "+v"(p0), "+v"(p1) : "v"(cc));
        if constexpr (MODE == 2)
            asm volatile(REP4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n") REP4("v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
        if constexpr (MODE == 3)
            asm volatile(REP4(REP4("v_add_f32 %0, %0, %1\n")) : "+v"(s0) : "v"(c));
        if constexpr (MODE == 4)
            asm volatile(REP8("v_pk_add_f32 %0, %0, %1\n") : "+v"(p0) : "v"(cc));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + acc[0] + acc[5];
}
template <int MODE, bool MFMA>
void run(int wps, float *o, float *in, long long *clk, const char *name)
{
    const int iters = 20000;
    const int adds = (MODE == 2) ? 32 : 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = 100 * 1024; // one workgroup per CU; its 4 * wps waves are dealt round-robin over the 4 SIMDs
    hipFuncSetAttribute((const void *)k<MODE, MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL((k<MODE, MFMA>), dim3(256), dim3(256 * wps), lds, 0, o, in, iters, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MFMA>), dim3(256), dim3(256 * wps), lds, 0, o, in, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cr[2]; hipMemcpy(cr, clk, 16, hipMemcpyDeviceToHost); const long long c = cr[0];
    printf("%-28s %s waves/SIMD %d : %7.3f ms  %5.2f nominal clocks per add per SIMD  %5.2f memtime ticks per add per wave (%.0f adds per MFMA; shader clock %.2f GHz)\n", name, MFMA ? "+mfma" : "     ", wps, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * adds * wps), (double)c / ((double)iters * adds), (double)adds, (double)cr[0] / (double)cr[1] * 0.1);
}
int main()
{
    float *o, *in; long long *clk;
    hipMalloc(&o, 256 * 4 * 8 * 256 * 4); hipMalloc(&in, 512); hipMemset(in, 0, 512); hipMalloc(&clk, 16);
    for (int wps = 1; wps <= 2; wps++)
    {
        run<3, false>(wps, o, in, clk, "v_add_f32 ILP1");
        run<0, false>(wps, o, in, clk, "v_add_f32 ILP4");
        run<4, false>(wps, o, in, clk, "v_pk_add_f32 ILP1");
        run<1, false>(wps, o, in, clk, "v_pk_add_f32 ILP2");
        run<2, false>(wps, o, in, clk, "v_pk_add_f32 ILP4");
        run<0, true>(wps, o, in, clk, "v_add_f32 ILP4");
        run<1, true>(wps, o, in, clk, "v_pk_add_f32 ILP2");
        run<2, true>(wps, o, in, clk, "v_pk_add_f32 ILP4");
    }
    return 0;
}
