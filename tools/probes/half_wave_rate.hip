// Probe: does a wave64 whose upper 32 lanes are inactive issue v_fma_f64 faster?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(double* out, int iters, int active)
{
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    if ((threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fma(v[i], 1.0000001, 0.5);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    double* out; hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int wg : {256, 512})
        for (int active : {64, 32, 16}) {
            k<<<wg, 256>>>(out, iters, active); hipDeviceSynchronize();
            hipEventRecord(a); k<<<wg, 256>>>(out, iters, active); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("grid %d x256, %2d active lanes per wave: %.3f ms  (%.2f clk per wave-FMA at 2.4 GHz, %d waves/SIMD)\n", wg, active, ms,
                   ms * 1e-3 * 2.4e9 / ((double)wg * 4 * iters * 128 / 1024.0), wg / 256);
        }
    return 0;
}
