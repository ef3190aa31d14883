// Microbenchmark: cost of ONE gain x state product in the shapes the rowlane kernels use, with the step-to-step
// dependency of the horizon sweeps (the result of a block is the broadcast source of the next one), at 1-4 waves/SIMD.
//   E0  exact, as built today: 12 independent v_mul_f32_dpp, then the 11 dependent v_add_f32 of the SEQ order
//   E1  exact, products and SEQ adds interleaved in one block (each add issues >= 2 instructions after its producer)
//   F0  fast, as built today: 12 dependent v_fmac_f32_dpp
//   F1  fast, two accumulators (6+6) and one add        F2  fast, four accumulators (3+3+3+3) and three adds
//   L0/L1 the same products with the broadcast going through LDS (ds_write_b32 + 3 ds_read_b128) and plain VALU instructions
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/chain_rate.hip -o build/chain_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define D(k) " row_newbcast:" #k " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// 256 threads = 4 waves = one per SIMD; dynamic LDS sized by the host so that exactly `waves/SIMD` workgroups fit a CU
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, unsigned long long *clk)
{
    extern __shared__ float pad[];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float m[12];
    float x = in[threadIdx.x & 15], acc = 0.f, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;
    pad[threadIdx.x] = x;
    for (int j = 0; j < 12; j++) m[j] = in[j * 64 + (threadIdx.x & 63)];
    for (int i = 0; i < iters; i++)
    {
        if constexpr (MODE == 0)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %12, %13" D(0) "v_mul_f32_dpp %1, %12, %14" D(1) "v_mul_f32_dpp %2, %12, %15" D(2)
                         "v_mul_f32_dpp %3, %12, %16" D(3) "v_mul_f32_dpp %4, %12, %17" D(4) "v_mul_f32_dpp %5, %12, %18" D(5)
                         "v_mul_f32_dpp %6, %12, %19" D(6) "v_mul_f32_dpp %7, %12, %20" D(7) "v_mul_f32_dpp %8, %12, %21" D(8)
                         "v_mul_f32_dpp %9, %12, %22" D(9) "v_mul_f32_dpp %10, %12, %23" D(10) "v_mul_f32_dpp %11, %12, %24" D(11)
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9), "=&v"(t10), "=&v"(t11)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
            asm volatile("v_add_f32 %0, %1, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %6\n"
                         "v_add_f32 %0, %0, %7\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %0, %0, %11\n v_add_f32 %0, %0, %12\n"
                         : "=&v"(acc) : "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(t8), "v"(t9), "v"(t10), "v"(t11));
        }
        else if constexpr (MODE == 1)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %12, %13" D(0) "v_mul_f32_dpp %1, %12, %14" D(1) "v_mul_f32_dpp %2, %12, %15" D(2)
                         "v_add_f32 %0, %0, %1\n"
                         "v_mul_f32_dpp %3, %12, %16" D(3) "v_add_f32 %0, %0, %2\n" "v_mul_f32_dpp %4, %12, %17" D(4) "v_add_f32 %0, %0, %3\n"
                         "v_mul_f32_dpp %5, %12, %18" D(5) "v_add_f32 %0, %0, %4\n" "v_mul_f32_dpp %6, %12, %19" D(6) "v_add_f32 %0, %0, %5\n"
                         "v_mul_f32_dpp %7, %12, %20" D(7) "v_add_f32 %0, %0, %6\n" "v_mul_f32_dpp %8, %12, %21" D(8) "v_add_f32 %0, %0, %7\n"
                         "v_mul_f32_dpp %9, %12, %22" D(9) "v_add_f32 %0, %0, %8\n" "v_mul_f32_dpp %10, %12, %23" D(10) "v_add_f32 %0, %0, %9\n"
                         "v_mul_f32_dpp %11, %12, %24" D(11) "v_add_f32 %0, %0, %10\n" "v_add_f32 %0, %0, %11\n"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9), "=&v"(t10), "=&v"(t11)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
            acc = t0;
        }
        else if constexpr (MODE == 2)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %1, %2" D(0) "v_fmac_f32_dpp %0, %1, %3" D(1) "v_fmac_f32_dpp %0, %1, %4" D(2) "v_fmac_f32_dpp %0, %1, %5" D(3)
                         "v_fmac_f32_dpp %0, %1, %6" D(4) "v_fmac_f32_dpp %0, %1, %7" D(5) "v_fmac_f32_dpp %0, %1, %8" D(6) "v_fmac_f32_dpp %0, %1, %9" D(7)
                         "v_fmac_f32_dpp %0, %1, %10" D(8) "v_fmac_f32_dpp %0, %1, %11" D(9) "v_fmac_f32_dpp %0, %1, %12" D(10) "v_fmac_f32_dpp %0, %1, %13" D(11)
                         : "=&v"(acc)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
        }
        else if constexpr (MODE == 3)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %2, %3" D(0) "v_mul_f32_dpp %1, %2, %4" D(1) "v_fmac_f32_dpp %0, %2, %5" D(2) "v_fmac_f32_dpp %1, %2, %6" D(3)
                         "v_fmac_f32_dpp %0, %2, %7" D(4) "v_fmac_f32_dpp %1, %2, %8" D(5) "v_fmac_f32_dpp %0, %2, %9" D(6) "v_fmac_f32_dpp %1, %2, %10" D(7)
                         "v_fmac_f32_dpp %0, %2, %11" D(8) "v_fmac_f32_dpp %1, %2, %12" D(9) "v_fmac_f32_dpp %0, %2, %13" D(10) "v_fmac_f32_dpp %1, %2, %14" D(11)
                         "v_add_f32 %0, %0, %1\n"
                         : "=&v"(acc), "=&v"(t1)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
        }
        else if constexpr (MODE == 4)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %4, %5" D(0) "v_mul_f32_dpp %1, %4, %6" D(1) "v_mul_f32_dpp %2, %4, %7" D(2) "v_mul_f32_dpp %3, %4, %8" D(3)
                         "v_fmac_f32_dpp %0, %4, %9" D(4) "v_fmac_f32_dpp %1, %4, %10" D(5) "v_fmac_f32_dpp %2, %4, %11" D(6) "v_fmac_f32_dpp %3, %4, %12" D(7)
                         "v_fmac_f32_dpp %0, %4, %13" D(8) "v_fmac_f32_dpp %1, %4, %14" D(9) "v_fmac_f32_dpp %2, %4, %15" D(10) "v_fmac_f32_dpp %3, %4, %16" D(11)
                         "v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %0, %0, %2\n"
                         : "=&v"(acc), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
        }
        else if constexpr (MODE == 5) // 11 dependent plain adds
        {
            asm volatile("v_add_f32 %0, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n"
                         "v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2\n"
                         : "=&v"(acc) : "v"(x), "v"(m[0]));
        }
        else if constexpr (MODE == 6) // 12 independent plain muls (no DPP)
        {
            asm volatile("v_mul_f32 %0, %12, %13\n v_mul_f32 %1, %12, %14\n v_mul_f32 %2, %12, %15\n v_mul_f32 %3, %12, %16\n v_mul_f32 %4, %12, %17\n v_mul_f32 %5, %12, %18\n"
                         "v_mul_f32 %6, %12, %19\n v_mul_f32 %7, %12, %20\n v_mul_f32 %8, %12, %21\n v_mul_f32 %9, %12, %22\n v_mul_f32 %10, %12, %23\n v_mul_f32 %11, %12, %24\n"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9), "=&v"(t10), "=&v"(t11)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
            acc = t11;
        }
        else if constexpr (MODE == 7) // 12 independent DPP muls only
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %12, %13" D(0) "v_mul_f32_dpp %1, %12, %14" D(1) "v_mul_f32_dpp %2, %12, %15" D(2)
                         "v_mul_f32_dpp %3, %12, %16" D(3) "v_mul_f32_dpp %4, %12, %17" D(4) "v_mul_f32_dpp %5, %12, %18" D(5)
                         "v_mul_f32_dpp %6, %12, %19" D(6) "v_mul_f32_dpp %7, %12, %20" D(7) "v_mul_f32_dpp %8, %12, %21" D(8)
                         "v_mul_f32_dpp %9, %12, %22" D(9) "v_mul_f32_dpp %10, %12, %23" D(10) "v_mul_f32_dpp %11, %12, %24" D(11)
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9), "=&v"(t10), "=&v"(t11)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
            acc = t11;
        }
        else if constexpr (MODE == 8) // exact, halving tree instead of SEQ (same instruction count, depth 4)
        {
            asm volatile("s_nop 1\n"
                         "v_mul_f32_dpp %0, %12, %13" D(0) "v_mul_f32_dpp %1, %12, %14" D(1) "v_mul_f32_dpp %2, %12, %15" D(2)
                         "v_mul_f32_dpp %3, %12, %16" D(3) "v_mul_f32_dpp %4, %12, %17" D(4) "v_mul_f32_dpp %5, %12, %18" D(5)
                         "v_mul_f32_dpp %6, %12, %19" D(6) "v_mul_f32_dpp %7, %12, %20" D(7) "v_mul_f32_dpp %8, %12, %21" D(8)
                         "v_mul_f32_dpp %9, %12, %22" D(9) "v_mul_f32_dpp %10, %12, %23" D(10) "v_mul_f32_dpp %11, %12, %24" D(11)
                         "v_add_f32 %1, %1, %2\n v_add_f32 %4, %4, %5\n v_add_f32 %7, %7, %8\n v_add_f32 %10, %10, %11\n"
                         "v_add_f32 %0, %0, %1\n v_add_f32 %3, %3, %4\n v_add_f32 %6, %6, %7\n v_add_f32 %9, %9, %10\n"
                         "v_add_f32 %0, %0, %3\n v_add_f32 %6, %6, %9\n v_add_f32 %0, %0, %6\n"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(t8), "=&v"(t9), "=&v"(t10), "=&v"(t11)
                         : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]));
            acc = t0;
        }
        else if constexpr (MODE == 9 || MODE == 10) // broadcast through LDS instead of DPP: the row writes its 16 values, every lane reads the 12 it needs
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const unsigned base = (unsigned)(size_t)pad + (threadIdx.x >> 6) * 256u;
            const unsigned waddr = base + (threadIdx.x & 63) * 4u, raddr = base + ((threadIdx.x & 63) >> 4) * 64u;
            f4 q0, q1, q2;
            asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n" ::"v"(x), "v"(waddr) : "memory");
            asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b128 %2, %3 offset:32\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(raddr) : "memory");
            if constexpr (MODE == 9) // 12 plain muls + 11 SEQ adds, separately rounded (-ffp-contract=off)
            {
                acc = q0.x * m[0];
                acc = acc + q0.y * m[1]; acc = acc + q0.z * m[2]; acc = acc + q0.w * m[3];
                acc = acc + q1.x * m[4]; acc = acc + q1.y * m[5]; acc = acc + q1.z * m[6]; acc = acc + q1.w * m[7];
                acc = acc + q2.x * m[8]; acc = acc + q2.y * m[9]; acc = acc + q2.z * m[10]; acc = acc + q2.w * m[11];
            }
            else // 12 dependent plain fmas
            {
                acc = q0.x * m[0];
                acc = __builtin_fmaf(q0.y, m[1], acc); acc = __builtin_fmaf(q0.z, m[2], acc); acc = __builtin_fmaf(q0.w, m[3], acc);
                acc = __builtin_fmaf(q1.x, m[4], acc); acc = __builtin_fmaf(q1.y, m[5], acc); acc = __builtin_fmaf(q1.z, m[6], acc);
                acc = __builtin_fmaf(q1.w, m[7], acc); acc = __builtin_fmaf(q2.x, m[8], acc); acc = __builtin_fmaf(q2.y, m[9], acc);
                acc = __builtin_fmaf(q2.z, m[10], acc); acc = __builtin_fmaf(q2.w, m[11], acc);
            }
        }
        x = acc; // the next block broadcasts this result
    }
    out[blockIdx.x * 256 + threadIdx.x] = x + pad[(threadIdx.x + 1) & 63];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        clk[0] = __builtin_readcyclecounter() - c0; // s_memtime
        clk[1] = wall_clock64() - r0;               // s_memrealtime, constant 100 MHz
    }
}
template <int MODE>
void run(const char *name, float *d, const float *in, int waves_per_simd, int ninstr)
{
    const int blocks = 256 * waves_per_simd, iters = 100000;
    const size_t lds = (160 * 1024) / waves_per_simd - 1024; // exactly waves_per_simd workgroups fit one CU
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) // the last repetition runs at settled clocks
    {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, in, iters, clk);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    }
    const double cyc = ms * 1e-3 * 2.4e9 / iters; // nominal 2.4 GHz: wall cycles per block per wave
    printf("%-44s waves/SIMD %d: %6.2f ms %7.1f nominal cycles per block (wall) %6.1f per SIMD [%d instr]; wave0: memtime %.1f /block, realtime %.3f ms -> memtime ticks at %.0f MHz\n",
           name, waves_per_simd, ms, cyc, cyc / waves_per_simd, ninstr, (double)h[0] / iters, h[1] / 1e5, h[0] / (h[1] / 100.0));
}
int main()
{
    float *d, *in;
    hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    hipMalloc(&in, 12 * 64 * 4);
    float h[12 * 64];
    for (int i = 0; i < 12 * 64; i++) h[i] = 0.05f + 0.001f * (i % 7);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3, 4})
    {
        run<0>("E0 exact: 12 mul_dpp then 11 dependent adds", d, in, w, 23);
        run<1>("E1 exact: products and adds interleaved", d, in, w, 23);
        run<2>("F0 fast: 12 dependent fmac_dpp", d, in, w, 12);
        run<3>("F1 fast: two accumulators", d, in, w, 13);
        run<4>("F2 fast: four accumulators", d, in, w, 15);
        run<5>("A  11 dependent plain adds", d, in, w, 11);
        run<6>("M  12 independent plain muls", d, in, w, 12);
        run<7>("P  12 independent mul_dpp", d, in, w, 12);
        run<8>("E2 exact shape with a halving tree", d, in, w, 23);
        run<9>("L0 exact, broadcast through LDS (1 write, 3 b128 reads)", d, in, w, 27);
        run<10>("L1 fast, broadcast through LDS, 12 plain fmas", d, in, w, 16);
    }
    return 0;
}
