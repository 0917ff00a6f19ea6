// Fixed-operation-order device math for the mcmc_hip kernels (gfx950).
//
// Everything the walker kernels need beyond IEEE + - * / sqrt fma: the Philox4x32-10
// counter-based generator (the algorithm rocRAND ships as ROCRAND_RNG_PSEUDO_PHILOX4_32_10),
// exact 52-bit uniforms, and log / exp / sincos(2 pi u) with every operation spelled out so
// that the result is a pure function of the input bits (compiled with -ffp-contract=off;
// fused operations are written as fma()).  DESIGN.md "Ensemble specification" is the
// normative text; tests/test_gpu_parity.py checks these bit for bit against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "short_log_table.h"
#include <stdint.h>

namespace mcmc {

constexpr uint32_t kStreamStep = 0u;
constexpr uint32_t kStreamBasis = 1u;
constexpr uint32_t kBranchExp24 = 5536481u;  // floor(0.33 * 2^24), proposal.py:79

// The wave's lane mask of a condition (a v_cmp writes it to a scalar register pair), and
// selects that take such a mask.  The selects are written as the VOP3 encoding
// (v_cndmask_b32_e64) in inline asm on purpose: on gfx950 the VOP2 encoding the compiler prefers
// (v_cndmask_b32_e32 ... vcc) issues at 23.6 clocks per wave-instruction, the VOP3 one -- with
// vcc or any scalar pair as the mask -- at 4.4 (tools/probes/cndmask_cost.hip,
// profiles/r02_probe_cndmask_cost.log).
__device__ __forceinline__ unsigned long long lanes(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ int sel(unsigned long long m, int a, int b)   // lane in m ? a : b
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ double sel(unsigned long long m, double a, double b)
{
    return __hiloint2double(sel(m, __double2hiint(a), __double2hiint(b)),
                            sel(m, __double2loint(a), __double2loint(b)));
}

struct u32x4 {
    uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0,
                                               uint32_t c1, uint32_t c2, uint32_t c3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // v_mul_hi_u32 + v_mul_lo_u32 (full rate each) beat one v_mad_u64_u32 (half rate, measured)
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// u = (2k+1) 2^-53 for k < 2^52, built without an int->fp conversion:
// [1 + k 2^-52] - 1 is exact, adding 2^-53 lands on an odd multiple of 2^-53 below 1.
__device__ __forceinline__ double u52(uint64_t k)
{
    return (__longlong_as_double((long long)(0x3FF0000000000000ull | k)) - 1.0) + 0x1p-53;
}

// ln(x), x a positive normal double (fdlibm e_log.c reduction + minimax polynomial)
__device__ __forceinline__ double dlog(double x)
{
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                     Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                     Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    uint32_t hx = (uint32_t)(b >> 32);
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    const uint64_t nb = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (b & 0xffffffffull);
    k += (int)(i >> 20);
    const double f = __longlong_as_double((long long)nb) - 1.0;
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

// exp(x) for x <= 0; x < -708 -> 0 (fdlibm e_exp.c reduction + polynomial)
__device__ __forceinline__ double dexp(double x)
{
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                     invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                     P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                     P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const unsigned long long in_range = lanes(x >= -708.0);   // else 0 (also for NaN)
    const double kf = rint(x * invln2);
    const double hi = fma(-kf, ln2_hi, x);
    const double lo = kf * ln2_lo;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k = (int)kf;
    return sel(in_range, __longlong_as_double(__double_as_longlong(y) + ((long long)k << 52)), 0.0);
}

__device__ __forceinline__ double ksin(double x)
{
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                     S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                     S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    return fma(v, fma(z, r, S1), x);
}

__device__ __forceinline__ double kcos(double x)
{
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                     C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                     C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    return 1.0 - (0.5 * z - z * r);
}

// (sin, cos)(2 pi u), u = (2k+1) 2^-53
__device__ __forceinline__ void sincos2pi(uint64_t k, double& sn, double& cs)
{
    constexpr double PIO4 = 7.85398163397448278999e-01;
    const unsigned o = (unsigned)(k >> 49);
    const uint64_t rem = k & ((1ull << 49) - 1);
    // (2 rem + 1) 2^-50 = [1 + (2 rem + 1) 2^-52 ... ] built exactly like u52
    double phi = (__longlong_as_double((long long)(0x3FF0000000000000ull | (rem << 3) | 4ull)) -
                  1.0);
    if (o & 1u) phi = 1.0 - phi;
    const double a = phi * PIO4;
    const double s = ksin(a), c = kcos(a);
    const bool swap = ((o + 1u) & 2u) != 0u;           // octants 1,2,5,6
    const double ss = swap ? c : s, cc = swap ? s : c;
    sn = (o & 4u) ? -ss : ss;                           // octants 4..7
    cs = ((o + 2u) & 4u) ? -cc : cc;                    // octants 2..5
}

// ---------------------------------------------------------------------------------------
// The random variates of one Metropolis step -- r (radial proposal distance) and Ea (the
// Exp(1) variate of the accept test) -- as a STAGED computation: the same operations as
// philox4x32_10 + u52 + dlog + sqrt above, cut into kStages pieces so that the hot kernel can
// spread the next step's RNG arithmetic over the operand-stream chunks of the current step
// (filling its scalar-load waits).  run_all() == the un-staged sequence, bit for bit.
struct StepRng {
    static constexpr int kStages = 18;
    uint32_t c0, c1, c2, c3, k0, k1;
    uint64_t kr, ka;
    bool expo;
    double f, dk, s, Er, Ea, r;

    // sub: 0 = the step itself, 1..n = the interpolation steps of a dragging step
    __device__ __forceinline__ void begin(uint32_t key0, uint32_t key1, uint32_t gid,
                                          unsigned long long step, uint32_t sub = 0)
    {
        // the keys go through an empty asm: the nine bumped round keys are then recomputed on
        // the scalar ALU every step instead of being hoisted out of the step loop (where they
        // would sit in 18 SGPRs, i.e. get spilled to VGPR lanes and read back every step)
        asm volatile("; step keys" : "+s"(key0), "+s"(key1));
        k0 = key0; k1 = key1;
        c0 = gid; c1 = kStreamStep | (sub << 16); c2 = (uint32_t)step; c3 = (uint32_t)(step >> 32);
    }
    __device__ __forceinline__ void round()
    {
        // v_mul_hi_u32 + v_mul_lo_u32 (full rate each) beat one v_mad_u64_u32 (half rate, measured)
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // dlog split in two: (a) reduction + the division s = f / (2 + f); (b) polynomial
    __device__ __forceinline__ void log_a(double x)
    {
        const uint64_t b = (uint64_t)__double_as_longlong(x);
        uint32_t hx = (uint32_t)(b >> 32);
        int k = (int)(hx >> 20) - 1023;
        hx &= 0x000fffffu;
        const uint32_t i = (hx + 0x95f64u) & 0x100000u;
        const uint64_t nb = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (b & 0xffffffffull);
        k += (int)(i >> 20);
        f = __longlong_as_double((long long)nb) - 1.0;
        dk = (double)k;
        s = f / (2.0 + f);
    }
    __device__ __forceinline__ double log_b() const
    {
        constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                         Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                         Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                         Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                         Lg7 = 1.479819860511658591e-01;
        const double z = s * s;
        const double w = z * z;
        const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
        const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
        const double R = t2 + t1;
        const double hfsq = 0.5 * f * f;
        return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
    }
    __device__ __forceinline__ void stage(int k)
    {
        if (k >= 0 && k < 10) round();
        else if (k == 10) {
            kr = ((uint64_t)c1 << 20) | (c2 >> 12);
            ka = ((uint64_t)c3 << 20) | ((uint64_t)(c2 & 0xFFFu) << 8) | (c0 & 0xFFu);
            expo = (c0 >> 8) < kBranchExp24;
        } else if (k == 11) log_a(u52(kr));
        else if (k == 12) Er = -log_b();
        else if (k == 13) log_a(u52(ka));
        else if (k == 14) Ea = -log_b();
        else if (k == 15) r = sqrt(2.0 * Er);
        else if (k == 16) r = sel(lanes(expo), Er, r);
        // the walker's PRIVATE sign (bit 7 of w0; set = positive, as for one-parameter blocks):
        // the basis column is shared by the group, and x + r v with r > 0 is a symmetric
        // proposal only on average over v and -v.  With its own sign every walker's kernel is
        // symmetric for every fixed basis, so the walkers of a group are independent chains
        // given the bases ("Why a shared basis")
        else if (k == 17)
            r = __longlong_as_double(__double_as_longlong(r) ^
                                     ((long long)((c0 & 0x80u) ^ 0x80u) << 56));
    }
    __device__ __forceinline__ void run_all()
    {
#pragma unroll
        for (int k = 0; k < kStages; ++k) stage(k);
    }
};

// (r, E_a) of (walker, step, sub) on the un-paired stream (oracle: walker_variates); oned: the
// column belongs to a one-parameter block -- RandProposer1D (proposal.py:85-93): chi(1) as
// sqrt(2 E) |cos| of a Box-Muller pair, and E_a from a second block (| 0x100).
__device__ __forceinline__ void step_variates(uint32_t key0, uint32_t key1, uint32_t gid,
                                              unsigned long long step, uint32_t sub, bool oned,
                                              double& r, double& Ea)
{
    StepRng rng;
    rng.begin(key0, key1, gid, step, sub);
    rng.run_all();
    r = rng.r;
    Ea = rng.Ea;
    if (oned) {
        double sn, cs;
        sincos2pi(rng.ka, sn, cs);
        const double rr = rng.expo ? rng.Er : sqrt(2.0 * rng.Er) * fabs(cs);
        r = (rng.c0 & 0x80u) ? rr : -rr;
        const u32x4 q4 = philox4x32_10(key0, key1, gid, kStreamStep | (sub << 16) | 0x100u,
                                       (uint32_t)step, (uint32_t)(step >> 32));
        Ea = -dlog(u52(((uint64_t)q4.w0 << 20) | (q4.w1 >> 12)));
    }
}

// -log(n 2^-b) for an ODD integer n < 2^29: the logarithm of the paired variates, whose arguments
// are short (25 / 29 significant bits), so that a table step is exact (oracle:
// orc_neg_log_short; table: short_log_table.h).  n = m 2^e, m in [1/2, 1); j = top seven
// fraction bits of m; f = fma(m, RC_j, -1) exactly, |f| <= 2^-8; log1p(f) by its Taylor polynomial
// to f^6; result = (b - e) ln 2 + log RC_j - log1p(f).  18 instructions and one LDS read, against
// the 45 (a division among them) of dlog.  The table sits in LDS: short_log_load fills it.
typedef double __attribute__((ext_vector_type(2))) dpair_t;
typedef const dpair_t __attribute__((address_space(3))) * short_log_tab;
static __device__ const double kShortLog[SHORT_LOG_TABLE_SIZE][2] = SHORT_LOG_TABLE;

__device__ __forceinline__ short_log_tab short_log_load(dpair_t* lds_table)
{
    for (int i = threadIdx.x; i < SHORT_LOG_TABLE_SIZE; i += blockDim.x) {
        dpair_t t;
        t.x = kShortLog[i][0];
        t.y = kShortLog[i][1];
        lds_table[i] = t;
    }
    return (short_log_tab)(unsigned long long)(unsigned)(unsigned long long)lds_table;
}

__device__ __forceinline__ double neg_log_short(uint32_t n, int b, short_log_tab tab)
{
    constexpr double LN2 = 6.93147180559945286227e-01, C2 = -0.5, C3 = 3.33333333333333314830e-01,
                     C4 = -0.25, C5 = 2.00000000000000011102e-01, C6 = -1.66666666666666657415e-01;
    const double xd = (double)n;
    const double m = __builtin_amdgcn_frexp_mant(xd);
    const int e = __builtin_amdgcn_frexp_exp(xd);
    const unsigned j = ((unsigned)__double2hiint(m) >> 13) & 0x7Fu;
    const dpair_t t = tab[j];
    const double f = fma(m, t.x, -1.0);
    const double p = f * fma(f, fma(f, fma(f, fma(f, fma(f, C6, C5), C4), C3), C2), 1.0);
    return fma((double)(b - e), LN2, t.y) - p;
}

// Table-driven log and exp of the incremental mixtures' log-sum-exp (round 5; oracle: orc_dlog_tab,
// orc_dexp_tab, where the error bounds are stated).  No division; the logarithm shares the table of
// the short-argument one, the exponential has 2^(j / 64) in 512 bytes of LDS (exp_tab_load).
typedef const double __attribute__((address_space(3))) * exp_tab;
static __device__ const double kExp64[64] = EXP64_TABLE;

__device__ __forceinline__ exp_tab exp_tab_load(double* lds_table)
{
    for (int i = threadIdx.x; i < 64; i += blockDim.x) lds_table[i] = kExp64[i];
    return (exp_tab)(unsigned long long)(unsigned)(unsigned long long)lds_table;
}

__device__ __forceinline__ double dlog_tab(double x, short_log_tab tab)
{
    constexpr double LN2 = 6.93147180559945286227e-01, C2 = -0.5, C3 = 3.33333333333333314830e-01,
                     C4 = -0.25, C5 = 2.00000000000000011102e-01, C6 = -1.66666666666666657415e-01;
    const double m = __builtin_amdgcn_frexp_mant(x);
    const int e = __builtin_amdgcn_frexp_exp(x);
    const unsigned j = ((unsigned)__double2hiint(m) >> 13) & 0x7Fu;
    const dpair_t t = tab[j];
    const double f = fma(m, t.x, -1.0);
    const double p = f * fma(f, fma(f, fma(f, fma(f, fma(f, C6, C5), C4), C3), C2), 1.0);
    return p - fma((double)(-e), LN2, t.y);
}

__device__ __forceinline__ double dexp_tab(double x, exp_tab tab)
{
    constexpr double C2 = 0.5, C3 = 1.66666666666666657415e-01, C4 = 4.16666666666666643537e-02,
                     C5 = 8.33333333333333321769e-03;
    const unsigned long long in_range = lanes(x >= -708.0);   // else 0 (also for NaN)
    const double kf = rint(x * EXP64_INV_LN2);
    double r = fma(-kf, EXP64_LN2_HI, x);
    r = fma(-kf, EXP64_LN2_LO, r);
    const double p = r * fma(r, fma(r, fma(r, fma(r, C5, C4), C3), C2), 1.0);
    // (outside the range kf is not an int: the index is masked, the result discarded)
    const int k = (int)kf;
    const double T = tab[k & 63];
    const double y = fma(T, p, T);
    return sel(in_range, __longlong_as_double(__double_as_longlong(y) + ((long long)(k >> 6) << 52)), 0.0);
}

// The variates of TWO consecutive steps from one Philox block (incremental kernels, plain steps;
// oracle: walker_variates_pair): block (walker, kStreamStep | 0x4000, P), P = step >> 1; half
// h = step & 1 uses the words a = w[2h], b = w[2h+1]: sign = bit 31 of a (set = positive),
// exponential branch iff bits 30..20 of a < 676, k_r = (a & 0xFFFFF) << 4 | b >> 28 (24 bits),
// u_r = (2 k_r + 1) 2^-25, k_a = b & 0xFFFFFFF (28 bits), u_a = (2 k_a + 1) 2^-29; the two
// logarithms are short-argument ones (neg_log_short above).
// sqrt of a double in [2^-100, 2^100]: the correctly rounded result (what sqrt() returns) without
// the range scaling and the zero / infinity fix-up of the general expansion -- the hardware
// reciprocal-square-root estimate, one Goldschmidt step and two residual corrections.
__device__ __forceinline__ double sqrt_midrange(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    g = fma(fma(-g, g, a), h, g);
    return g;
}

// The far tail of a short uniform (oracle: pair_tail): a variate whose 24- / 28-bit uniform fell
// into the LOWEST bin stands for u in (0, 2^-b); it is redrawn there at full width -- u = 2^-b u',
// u' = u52 of a Philox block of its own (walker, kStreamStep | 0x8000 | which << 13, step) --, so
// -log u = b ln 2 - log u': the exponential laws of |r| and of the accept variate have their exact
// tails instead of ending at 17.3 / 20.1.  Rare (2^-24 / 2^-28 per step and
// walker): kept out of line, behind a wave-uniform branch.
__device__ __attribute__((noinline)) double pair_tail(uint32_t key0, uint32_t key1, uint32_t gid,
                                                       unsigned long long step, uint32_t which, double bits)
{
    constexpr double LN2 = 6.93147180559945286227e-01;
    const u32x4 q = philox4x32_10(key0, key1, gid, kStreamStep | 0x8000u | (which << 13),
                                  (uint32_t)step, (uint32_t)(step >> 32));
    return fma(bits, LN2, -dlog(u52(((uint64_t)q.w0 << 20) | (q.w1 >> 12))));
}

struct PairRng {
    double r[2], Ea[2];
    __device__ __forceinline__ void run(uint32_t key0, uint32_t key1, uint32_t gid,
                                        unsigned long long pair, short_log_tab tab)
    {
        StepRng g;
        g.begin(key0, key1, gid, pair);
        g.c1 = kStreamStep | 0x4000u;
#pragma unroll
        for (int k = 0; k < 10; ++k) g.round();
        const uint32_t w[4] = {g.c0, g.c1, g.c2, g.c3};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t a = w[2 * h], b = w[2 * h + 1];
            const uint32_t kr = ((a & 0xFFFFFu) << 4) | (b >> 28);
            const uint32_t ka = b & 0x0FFFFFFFu;
            double Er = neg_log_short(2u * kr + 1u, 25, tab);
            double Eah = neg_log_short(2u * ka + 1u, 29, tab);
            if (lanes((kr == 0u) | (ka == 0u)) != 0ull) {   // wave-uniform, rare: the lowest bins
                const unsigned long long step = 2ull * pair + (unsigned long long)h;
                const double tr = pair_tail(key0, key1, gid, step, 0u, 24.0);
                const double ta = pair_tail(key0, key1, gid, step, 1u, 28.0);
                Er = kr == 0u ? tr : Er;
                Eah = ka == 0u ? ta : Eah;
            }
            // (2 E_r lies in [2^-24, 35], or up to ~110 after a redraw of the tail)
            const double rr = sel(lanes(((a >> 20) & 0x7FFu) < 676u), Er, sqrt_midrange(2.0 * Er));
            // sign: bit 31 of a set = positive
            r[h] = __longlong_as_double(__double_as_longlong(rr) ^
                                        ((long long)(~a & 0x80000000u) << 32));
            Ea[h] = Eah;
        }
    }
};

// a / w given R = RN(1 / w): the correctly rounded quotient without a division: q0 = a R,
// q1 = fma(fma(-q0, w, a), R, q0), q2 = fma(fma(-q1, w, a), R, q1) (q1 is faithful -- its exact
// argument is within 2^-52 ulp of a / w -- and a faithful quotient corrected once with the
// correctly rounded reciprocal is the IEEE quotient: Markstein 1990) -- bit for bit the oracle's
// `/`, 5 instructions instead of 30
// (checked against the IEEE division on 2 10^9 random operand pairs of the ranges it is used on:
// no difference)
__device__ __forceinline__ double div_by(double a, double w, double R)
{
    double q = a * R;
    q = fma(fma(-q, w, a), R, q);
    return fma(fma(-q, w, a), R, q);
}

// accepted steps of this launch: summed over the wave, one atomic per wave
__device__ __forceinline__ void wave_add_accepts(unsigned long long* total, long long mine)
{
    unsigned v = (unsigned)mine;  // < 2^32 per launch
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, (unsigned long long)v);
}

}  // namespace mcmc
