// Probe: issue cost of single VALU instructions on gfx950, in SIMD clocks per wave64 instruction,
// with 4 (and 1) waves per SIMD.  Every op runs as 8 independent chains inside an unrolled loop of
// inline asm, so that neither dependences nor the compiler get in the way.
//   hipcc --offload-arch=gfx950 -O2 -o valu_op_cost valu_op_cost.hip && ./valu_op_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters)
{
    double a[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
        u[i] = 0x9E3779B9u * (threadIdx.x + 17 * i + 1);
    }
    const double c = 1.0000001, e = 1e-12;
    const unsigned m = 0xD2511F53u;
    if (OP == 12) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[0]), "v"(m) : "vcc");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
#define ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(e));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(m));
#define XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
#define RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define RSQ(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#define SQRT(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
#define CVT(i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(a[i]) : "v"(u[i]));
#define CMP(i) asm volatile("v_cmp_le_f64 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
// (vcc is set once before the loop and only read here: declaring it clobbered makes the
// compiler put an s_nop between the selects)
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m));
#define FREXPM(i) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[i]));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(u[i]));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(m) : "vcc");
#define MULF32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
#define FMAF32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(m));
#define LOGF32(i) asm volatile("v_log_f32 %0, %0" : "+v"(u[i]));
#define MAXF64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(m));
            if (OP == 0) { REP8(FMA) }
            if (OP == 1) { REP8(ADD) }
            if (OP == 2) { REP8(MULLO) }
            if (OP == 3) { REP8(MULHI) }
            if (OP == 4) { REP8(MUL24) }
            if (OP == 5) { REP8(XOR) }
            if (OP == 6) { REP8(RCP) }
            if (OP == 7) { REP8(RSQ) }
            if (OP == 8) { REP8(SQRT) }
            if (OP == 9) { REP8(CVT) }
            if (OP == 10) { REP8(CMP) }
            if (OP == 11) { REP8(DPP) }
            if (OP == 12) { REP8(CND) }
            if (OP == 13) { REP8(FREXPM) }
            if (OP == 14) { REP8(LDEXP) }
            if (OP == 15) { REP8(MAD64) }
            if (OP == 16) { REP8(MULF32) }
            if (OP == 17) { REP8(FMAF32) }
            if (OP == 18) { REP8(LOGF32) }
            if (OP == 19) { REP8(MAXF64) }
            if (OP == 20) { REP8(MULF64) }
            if (OP == 21) { REP8(LSHL64) }
            if (OP == 22) { REP8(ADD3) }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (double)u[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char* name, double* d)
{
    const int iters = 20000;
    for (int blocks : {256, 1024}) {   // 1 and 4 waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)iters * 32.0 * (blocks / 256);   // waves/SIMD * 32 per iter
        printf("%-16s %d waves/SIMD: %8.3f ms  %6.2f clk per wave-instruction (2.4 GHz)\n", name,
               blocks / 256, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
}

int main()
{
    double* d;
    hipMalloc(&d, 64);
    run<0>("v_fma_f64", d);
    run<1>("v_add_f64", d);
    run<20>("v_mul_f64", d);
    run<19>("v_max_f64", d);
    run<10>("v_cmp_le_f64", d);
    run<2>("v_mul_lo_u32", d);
    run<3>("v_mul_hi_u32", d);
    run<15>("v_mad_u64_u32", d);
    run<4>("v_mul_u32_u24", d);
    run<5>("v_xor_b32", d);
    run<22>("v_add3_u32", d);
    run<21>("v_lshlrev_b64", d);
    run<6>("v_rcp_f64", d);
    run<7>("v_rsq_f64", d);
    run<8>("v_sqrt_f64", d);
    run<9>("v_cvt_f64_u32", d);
    run<13>("v_frexp_mant_f64", d);
    run<14>("v_ldexp_f64", d);
    run<11>("v_mov_b32_dpp", d);
    run<12>("v_cndmask_b32", d);
    run<16>("v_mul_f32", d);
    run<17>("v_fma_f32", d);
    run<18>("v_log_f32", d);
    return 0;
}
