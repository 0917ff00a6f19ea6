// Probe: is v_mfma_f64_16x16x4_f64 bit-identical to a k-ordered fma chain (from C), and what
// are its layouts?  A: lane l holds A[i = l&15][k = l>>4]; B: lane l holds B[k = l>>4][j = l&15];
// C/D: 4 f64 per lane, D[row = (l>>4) + 4*reg][col = l&15]   (cdna_hip_programming.md §3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D, long long* cyc)
{
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];      // A[i][k], row-major 16x4
    double b = B[(l >> 4) * 16 + (l & 15)];     // B[k][j], row-major 4x16
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    long long t0 = clock64();
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    // dependent chain of 64 more to time issue rate
    d4 e = d;
    for (int it = 0; it < 64; ++it) e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0);
    long long t1 = clock64();
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
    if (l == 0) { cyc[0] = t1 - t0; D[256] = e[0]; }
}
int main()
{
    double hA[64], hB[64], hC[256], hD[257], ref[256], ref2[256];
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(10.0, (rand() % 7) - 3); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (auto& v : hC) v = rnd();
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = hC[i * 16 + j];
            for (int kk = 0; kk < 4; ++kk) s = fma(hA[i * 4 + kk], hB[kk * 16 + j], s);
            ref[i * 16 + j] = s;
            double s2 = hC[i * 16 + j];
            for (int kk = 3; kk >= 0; --kk) s2 = fma(hA[i * 4 + kk], hB[kk * 16 + j], s2);
            ref2[i * 16 + j] = s2;
        }
    double *dA, *dB, *dC, *dD; long long* dcyc;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMalloc(&dD, sizeof hD); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD, dcyc);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    long long cyc; hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    int same = 0, same2 = 0; double maxrel = 0;
    for (int i = 0; i < 256; ++i) {
        same += memcmp(&hD[i], &ref[i], 8) == 0;
        same2 += memcmp(&hD[i], &ref2[i], 8) == 0;
        maxrel = fmax(maxrel, fabs(hD[i] - ref[i]) / fabs(ref[i]));
    }
    printf("mfma_f64_16x16x4: bit-equal to ascending-k fma chain: %d/256; descending: %d/256; max rel diff %.3g; 65 dependent MFMAs took %lld clocks\n",
           same, same2, maxrel, cyc);
    return 0;
}
