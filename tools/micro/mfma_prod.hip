// Microbenchmark / probe for the exact-arithmetic MFMA formulation:
//   v_mfma_f32_16x16x1_4b_f32 (K = 1, four 16x16 blocks) with C = -0 delivers 1024 SEPARATELY ROUNDED products
//   D_b[i][j] = A_b[i] * B_b[j] per issue (fma(a, b, -0) == round(a*b), the sign of a zero product included) — the products
//   of a gain x state mat-vec for 16 instances and four columns k at a time, with no cross-lane traffic.
// (1) semantics + D layout against host arithmetic, (2) issue rate alone and next to dependent v_add chains at one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mfma_prod.hip -o build/mfma_prod
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f32x16 c;
    for (int e = 0; e < 16; e++) c[e] = -0.f;
    f32x16 r = __builtin_amdgcn_mfma_f32_16x16x1f32(a[l], b[l], c, 0, 0, 0);
    for (int e = 0; e < 16; e++) d[e * 64 + l] = r[e];
}

// MODE 0: 3 MFMA (16x16x1_4b) per block, results consumed by 44 adds (4 rows x 11 sequential), next block's B operand
//         depends on the sums (the horizon chain).  MODE 1: the adds only.  MODE 2: the MFMAs only (dependent through one add).
// MODE 3: 3 x mfma_16x16x4 (fma chain) + 4 adds, the fast-arithmetic shape.
template <int MODE>
__global__ __launch_bounds__(64, 1) void rate(float *out, const float *in, int iters)
{
    const int l = threadIdx.x;
    float ga[3], x[3];
    for (int k = 0; k < 3; k++) { ga[k] = in[k * 64 + l]; x[k] = in[(3 + k) * 64 + l]; }
    f32x16 negz;
    for (int e = 0; e < 16; e++) negz[e] = -0.f;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int it = 0; it < iters; it++)
    {
        if constexpr (MODE == 0 || MODE == 2)
        {
            f32x16 p0 = __builtin_amdgcn_mfma_f32_16x16x1f32(ga[0], x[0], negz, 0, 0, 0);
            f32x16 p1 = __builtin_amdgcn_mfma_f32_16x16x1f32(ga[1], x[1], negz, 0, 0, 0);
            f32x16 p2 = __builtin_amdgcn_mfma_f32_16x16x1f32(ga[2], x[2], negz, 0, 0, 0);
            if constexpr (MODE == 0)
            {
                float s[4];
                for (int v = 0; v < 4; v++)
                {
                    float t = p0[v] + p0[4 + v];
                    t = t + p0[8 + v]; t = t + p0[12 + v];
                    t = t + p1[v]; t = t + p1[4 + v]; t = t + p1[8 + v]; t = t + p1[12 + v];
                    t = t + p2[v]; t = t + p2[4 + v]; t = t + p2[8 + v]; t = t + p2[12 + v];
                    s[v] = t;
                }
                x[0] = s[0]; x[1] = s[1]; x[2] = s[2]; acc3 += s[3];
            }
            else { x[0] = p0[0] + p1[1]; x[1] = p1[2]; x[2] = p2[3]; }
        }
        else if constexpr (MODE == 1)
        {
            float s[4] = {x[0], x[1], x[2], acc3};
            for (int v = 0; v < 4; v++)
                for (int k = 0; k < 11; k++) { s[v] = s[v] + ga[k % 3]; asm volatile("" : "+v"(s[v])); }
            x[0] = s[0]; x[1] = s[1]; x[2] = s[2]; acc3 = s[3];
        }
        else
        {
            f32x4 c = {acc0, acc1, acc2, acc3};
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[0], x[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[1], x[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[2], x[2], c, 0, 0, 0);
            x[0] = c[0] * 0.5f; x[1] = c[1] * 0.5f; x[2] = c[2] * 0.5f; acc3 = c[3] * 0.25f;
        }
    }
    out[blockIdx.x * 64 + l] = x[0] + x[1] + x[2] + acc0 + acc1 + acc2 + acc3;
}

template <int MODE>
void run_rate(const char *name, float *d_out, float *d_in, int iters, int instr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nblocks = 256 * 4; // one wave per SIMD
    hipLaunchKernelGGL(rate<MODE>, dim3(nblocks), dim3(64), 0, 0, d_out, d_in, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(nblocks), dim3(64), 0, 0, d_out, d_in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.3f ms  %7.1f nominal cycles (2.4 GHz) per block of %d instructions per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / iters, instr);
}

int main()
{
    // ---- (1) semantics and layout ----
    std::vector<float> a(64), b(64), d(16 * 64);
    srand(5);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int l = 0; l < 64; l++) { a[l] = rnd() * 3.7f; b[l] = rnd() * 0.013f; }
    a[3] = 0.f; a[17] = -0.f; b[5] = -0.f; b[18] = 0.f; a[40] = 1e-30f; b[41] = 1e-12f; b[42] = -1e-12f; a[20] = -2.5f;
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 16 * 256);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(d.data(), dd, 16 * 256, hipMemcpyDeviceToHost);
    // hypothesis: A lane l = (block l>>4, row i = l&15); B lane l = (block l>>4, col j = l&15);
    //             D reg e = 4*blk + v of lane (g = l>>4, c = l&15) = A_blk[i = 4g + v] * B_blk[j = c]
    int bad = 0, negzero = 0, denorm = 0;
    for (int l = 0; l < 64; l++)
        for (int e = 0; e < 16; e++)
        {
            const int blk = e >> 2, v = e & 3, g = l >> 4, c = l & 15;
            volatile float pa = a[blk * 16 + 4 * g + v], pb = b[blk * 16 + c];
            volatile float want = pa * pb;
            float got = d[e * 64 + l];
            unsigned uw, ug;
            float w2 = want;
            memcpy(&uw, &w2, 4); memcpy(&ug, &got, 4);
            if (uw != ug) { if (bad < 8) printf("  mismatch lane %d reg %d: want %a got %a\n", l, e, w2, got); bad++; }
            if (uw == 0x80000000u) negzero++;
            if (w2 != 0.f && fabsf(w2) < 1.2e-38f) denorm++;
        }
    printf("16x16x1_4b with C=-0: %d mismatching products of 1024 (layout hypothesis D[4*blk+v](g,c) = A_blk[4g+v]*B_blk[c]); %d negative-zero and %d subnormal products among them\n",
           bad, negzero, denorm);
    // ---- (2) rates ----
    float *d_in, *d_out;
    hipMalloc(&d_in, 6 * 256); hipMalloc(&d_out, 1024 * 256);
    std::vector<float> in(6 * 64);
    for (auto &v : in) v = rnd() * 0.3f;
    hipMemcpy(d_in, in.data(), 6 * 256, hipMemcpyHostToDevice);
    const int iters = 20000;
    run_rate<2>("3 x mfma_16x16x1_4b only", d_out, d_in, iters, 3);
    run_rate<1>("44 v_add (4 chains of 11) only", d_out, d_in, iters, 44);
    run_rate<0>("3 x mfma_16x16x1_4b + 44 v_add consuming them (exact shape)", d_out, d_in, iters, 47);
    run_rate<3>("3 x mfma_16x16x4 chained + 4 v_mul (fast shape)", d_out, d_in, iters, 7);
    return 0;
}
