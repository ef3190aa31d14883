// glds_ring.hip — the LDS-DMA ring admm_tile16.hip uses for per-instance bounds / references (round 4), checked in isolation:
//   * `global_load_lds_dwordx4` with a destination ABOVE 64 KB of a 156 KB dynamic LDS allocation (M0 carries the byte address),
//   * lane (g, c) fetches bytes [16 g, 16 g + 16) of instance c's 64-byte step row (whole rows: four adjacent-in-memory 16-byte
//     pieces per instance), the piece lands lane-linear at slot + 16 (16 g + c),
//   * lane (g', c) then reads the words {4 v + g'} of ITS instance's row with ds_read_b32 at slot + 256 v + 16 c + 4 g'
//     (banks 4 c + g': conflict free),
//   * a ring of three slots, DMAs issued 1.5 steps ahead and retired with counted s_waitcnt vmcnt(N),
//   * the cost: shader clocks per step of a loop that does nothing else, against the same loop reading one shared LDS table.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/glds_ring.hip -o build/glds_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <bool DMA>
__global__ __launch_bounds__(256, 1) void k(const float *rows, float *out, int steps, int ring_byte_off, long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int inst = (blockIdx.x * 4 + wv) * 16 + c;
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_float *)lds + ring_byte_off + wv * 3 * 1024);
    const float *src = rows + ((size_t)inst * steps) * 16 + 4 * g; // + i * 16 floats
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const unsigned rd = ring + 16 * c + 4 * g;
    if (!DMA) // the comparison: the rows of the wave's first instance staged once, every lane reads them like a shared table
    {
        for (int e = threadIdx.x; e < 3 * 256 * 4; e += 256) lds[ring_byte_off / 4 + e] = 1.f;
        __syncthreads();
    }
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (DMA)
    {
        glds16(src, ring);
        glds16(src + 16, ring + 1024);
    }
    for (int i = 0; i < steps; i++)
    {
        const unsigned slot = rd + (i % 3) * 1024;
        if (DMA)
        {
            if (i + 1 < steps) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        float2 a, b;
        asm volatile("ds_read2st64_b32 %0, %2 offset1:1\n\tds_read2st64_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(slot) : "memory");
        acc0 += a.x * (i + 1); acc1 += a.y * (i + 2); acc2 += b.x * (i + 3); acc3 += b.y * (i + 4);
        if (DMA && i + 2 < steps) glds16(src + (size_t)(i + 2) * 16, ring + ((i + 2) % 3) * 1024);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    float *o = out + (size_t)inst * 16 + g; // element 4 v + g
    o[0] = acc0; o[4] = acc1; o[8] = acc2; o[12] = acc3;
}

int main()
{
    const int wgs = 256, insts = wgs * 64, steps = 58;
    std::vector<float> h((size_t)insts * steps * 16);
    for (size_t e = 0; e < h.size(); e++) h[e] = (float)((e * 2654435761u) % 1021) - 510.f;
    float *d_rows, *d_out; long long *d_clk;
    hipMalloc(&d_rows, h.size() * 4); hipMalloc(&d_out, (size_t)insts * 16 * 4); hipMalloc(&d_clk, 16);
    hipMemcpy(d_rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int lds = 156 * 1024, ring_off = 144 * 1024; // the ring of the four waves: bytes 144 K .. 156 K
    hipFuncSetAttribute((const void *)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; rep++)
    {
        hipLaunchKernelGGL(k<true>, dim3(wgs), dim3(256), lds, 0, d_rows, d_out, steps, ring_off, d_clk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    }
    long long c_dma, c_tab;
    hipMemcpy(&c_dma, d_clk, 8, hipMemcpyDeviceToHost);
    std::vector<float> o((size_t)insts * 16);
    hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int inst = 0; inst < insts; inst++)
        for (int r = 0; r < 16; r++)
        {
            float acc = 0.f;
            const int v = r >> 2;
            for (int i = 0; i < steps; i++) acc += h[((size_t)inst * steps + i) * 16 + r] * (i + 1 + v);
            bad += acc != o[(size_t)inst * 16 + r];
        }
    hipLaunchKernelGGL(k<false>, dim3(wgs), dim3(256), lds, 0, d_rows, d_out, steps, ring_off, d_clk);
    hipDeviceSynchronize();
    hipMemcpy(&c_tab, d_clk, 8, hipMemcpyDeviceToHost);
    printf("glds ring above 64 KB: %ld of %ld words wrong; %.1f shader clocks (s_memtime) per step with the DMA ring, %.1f reading a resident table\n",
           bad, (long)insts * 16, (double)c_dma / steps, (double)c_tab / steps);
    return bad != 0;
}
