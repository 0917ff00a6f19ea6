// Probe: throughput of v_mfma_f64_16x16x4_f64 and of v_fma_f64 (and both together) on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool WITH_VALU>
__global__ void __launch_bounds__(256) kern(double* out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        if (WITH_VALU) {
#pragma unroll
            for (int r = 0; r < 2 * NACC; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fma(v[i], 1.0000001, 0.5);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH>
__global__ void __launch_bounds__(256) valu(double* out, int iters)
{
    double v[NCH];
    for (int i = 0; i < NCH; ++i) v[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[i] = fma(v[i], 1.0000001, 0.5);
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
float timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    double* out; hipMalloc(&out, 8 * 256 * 4096);
    const int iters = 20000;
    for (int wg : {256, 512, 1024}) {
        float ms = timeit([&] { kern<4, false><<<wg, 256>>>(out, iters, 1.0, 2.0); });
        double n = (double)wg * 4 * iters * 4;  // waves * iters * NACC
        printf("mfma only  grid %4d x256 (4 waves/WG): %.3f ms, %.1f TFLOP/s, %.1f clk/MFMA/SIMD@2.4GHz (waves/SIMD=%d)\n", wg, ms,
               n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024.0), wg / 256);
        ms = timeit([&] { kern<4, true><<<wg, 256>>>(out, iters, 1.0, 2.0); });
        printf("mfma+valu  grid %4d: %.3f ms, mfma %.1f TFLOP/s + valu %.1f TFLOP/s\n", wg, ms,
               n * 2048 / ms / 1e9, (double)wg * 256 * iters * 64.0 * 2 / ms / 1e9);
        ms = timeit([&] { valu<8><<<wg, 256>>>(out, iters); });
        printf("valu only  grid %4d: %.3f ms, %.1f TFLOP/s, %.2f clk per wave64 v_fma_f64@2.4GHz\n", wg, ms,
               (double)wg * 256 * iters * 128.0 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wg * 4 * iters * 128 / 1024.0));
    }
    return 0;
}
