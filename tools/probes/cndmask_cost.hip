// Probe: what does v_cndmask_b32 cost on gfx950?  Variants: implicit vcc (VOP2), an SGPR-pair
// mask (VOP3), the mask rewritten by a v_cmp every 8 selects, and the same selects done with
// integer logic (v_and / v_bfi with a 0 / -1 lane word).
//   hipcc --offload-arch=gfx950 -O2 -o cndmask_cost cndmask_cost.hip && ./cndmask_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = 0x9E3779B9u * (threadIdx.x + 17 * i + 1);
    const unsigned m = 0xD2511F53u;
    unsigned long long mask = 0x5555aaaa3333ccccull + blockIdx.x;
    unsigned sel = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[0]), "v"(m) : "vcc");
    asm volatile("s_mov_b64 %0, %0" : "+s"(mask));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define CND2(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m));
#define CND3(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "s"(mask));
#define CNDX(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(m));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(sel));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(sel), "v"(m));
#define CND3V(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m));
#define ADDC(i) asm volatile("v_addc_co_u32_e64 %0, %2, %0, %1, %2" : "+v"(u[i]) : "v"(m), "s"(mask));
#define ADDCV(i) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(m) : "vcc");
#define CMPW(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(m) : "vcc");
            if (OP == 0) { REP8(CND2) }
            if (OP == 1) { REP8(CND3) }
            if (OP == 2) { CMPW(0) REP8(CND2) }
            if (OP == 3) { REP8(AND) }
            if (OP == 4) { REP8(BFI) }
            if (OP == 5) { REP8(CNDX) }
            if (OP == 6) { REP8(CND3V) }
            if (OP == 7) { REP8(ADDC) }
            if (OP == 8) { REP8(ADDCV) }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += u[i];
    if (s == 12345u) out[0] = s;
}

template <int OP>
void run(const char* name, double* d, int per_iter)
{
    const int iters = 20000;
    for (int blocks : {256, 1024}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * per_iter * (blocks / 256);
        printf("%-44s %d waves/SIMD: %8.3f ms  %6.2f clk per wave-instruction (2.4 GHz)\n", name,
               blocks / 256, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
}

int main()
{
    double* d;
    (void)hipMalloc(&d, 64);
    run<0>("v_cndmask_b32 (vcc, dst = src0)", d, 32);
    run<5>("v_cndmask_b32 (vcc, dst != src0)", d, 32);
    run<1>("v_cndmask_b32_e64 (SGPR-pair mask)", d, 32);
    run<6>("v_cndmask_b32_e64 (mask operand = vcc)", d, 32);
    run<2>("v_cmp + 8 v_cndmask (counted: 9)", d, 36);
    run<7>("v_addc_co_u32_e64 (SGPR-pair carry in/out)", d, 32);
    run<8>("v_addc_co_u32_e32 (vcc carry in/out)", d, 32);
    run<3>("v_and_b32 with a 0 / -1 lane word", d, 32);
    run<4>("v_bfi_b32 with a 0 / -1 lane word", d, 32);
    return 0;
}
