// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access shape: one dword per lane, 256 contiguous
// bytes per wave-instruction (the row layout of admm_rowlane.hip).  MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for
// 16 B/lane streams (it reads 1/2 there); other widths must be calibrated on a known byte count.
// Reads 1 GiB once (kernel calib_read) and writes 1 GiB once (kernel calib_write); working set >> 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read(const float *__restrict__ src, float *__restrict__ out, size_t n)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) out[0] = acc; // never true: keeps the loads alive without a store stream
}
__global__ void calib_write(float *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)i;
}
int main()
{
    const size_t n = (size_t)1 << 28; // 1 GiB of floats
    float *a, *o;
    if (hipMalloc(&a, n * 4) != hipSuccess || hipMalloc(&o, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, n * 4);
    hipDeviceSynchronize();
    for (int r = 0; r < 3; r++)
    {
        hipLaunchKernelGGL(calib_write, dim3(4096), dim3(64), 0, 0, a, n);
        hipLaunchKernelGGL(calib_read, dim3(4096), dim3(64), 0, 0, a, o, n);
    }
    hipDeviceSynchronize();
    printf("calib: each launch moves %zu bytes\n", n * 4);
    return 0;
}
