// mcmc_hip -- INCREMENTAL EVALUATION kernels (gfx950 only), 2 <= d <= 128.
//
// The target is one Gaussian mode (gaussian_mixture.py:138-163 with K = 1, gaussian.py:96-112):
// logpdf(t) = -1/2 (c + |L^-1 (t - mu)|^2).  The walkers of a group move along SHARED directions
// v, so the whitened direction u = L^-1 v is shared as well, and the
// whitened residual of a trial is the carried one moved along it:
//
//     t = x + r v      =>      L^-1 (t - mu) = y + r u,      y = L^-1 (x - mu).
//
// A step therefore costs O(d) per walker instead of the O(d^2) triangular product -- the product
// is done once per (group, step) for the direction (whiten_directions_kernel) and once per
// walker every `refresh_every` steps to stop rounding drift (whiten_state_kernel).  Same
// posterior, same proposal stream, same accept rule; the arithmetic is specified in
// oracle/mcmc_oracle.c (step_core_inc, orc_whiten, orc_whiten_directions) and matched bit for
// bit (tests/test_gpu_parity.py::test_incremental_*).
//
// Layout: FOUR lanes serve one walker; lane class c = lane & 3 holds the dimensions i = 4 kk + c
// of x and y in registers.  The sums over dimensions (chi2, normal-prior terms) are four
// interleaved chains, one per lane class, combined (p0 + p1) + (p2 + p3) through two DPP quad
// permutes -- no LDS, no barrier inside a step.  The random variates of eight consecutive steps
// are drawn at once, one pair of steps per lane class, and fetched by quad broadcasts; the
// prior-support test and the accept decision live in scalar registers as lane masks.  Per step a
// lane reads its (v_i, u_i) pairs from LDS, where the columns of the launch are staged in chunks
// (double-buffered, one workgroup barrier per chunk).
#include <string>

#include "incremental_common.h"

namespace mcmc {
namespace {

// ---------------------------------------------------------------- the step kernel
// 256 threads = 64 walkers of ONE group (group_size is a multiple of 64).
//   MODE 0 "box":    every prior uniform on the SAME interval [lo, hi] (kernel arguments; the
//                    padded dimensions sit at its midpoint) -- BASELINE configs 2-4
//   MODE 1 general bounds: per-dimension [lo_i, hi_i] in registers (DQ <= 12) or LDS
//   MODE 2 ... and normal priors, whose log-density is CARRIED along the direction like the
//          log-likelihood (round 5; oracle: carries_prior): a third stream w_i = (v_i / s_i) / s_i
//          of the columns beside the (v, u) pairs, one more fma per dimension and trial
// Registers: x and y only (+ the bounds for MODE > 0, DQ <= 12); the (v_i, u_i) pairs of a step
// are read from LDS twice (trial, commit) -- ds_read_b128 costs 4 LDS cycles per wave, far below
// what the step's VALU work takes.  The columns of the launch reach LDS by global->LDS DMA,
// one chunk ahead (double-buffered; one s_waitcnt + workgroup barrier per chunk).
// waves per SIMD the register allocation is held to: MEASURED, per DQ and MODE, with builds held to
// 1, 2, 3 and 4 waves (tools/occupancy_sweep.py over tools/exp_inc_variants.sh builds; 65 536
// walkers = 4 waves per SIMD at most).  Not "the largest occupancy that does not spill": four
// waves with a few spilled registers beat three without up to DQ = 12 (d = 40: +40 %), three are
// almost never the best choice, and above that two waves -- which also get the read-ahead of
// the LDS pairs (PIPE) -- beat one even where they spill (d = 128, MODE 0: +12 % in round 2;
// round 3, with the pairs kept in registers, one wave wins from dq = 31: see inc_keep_pairs).
// The odd entries (13, 15) are where the per-dimension constants move from registers to LDS.
// General bounds at the top of the dimension range (round 5 late): ONE wave per SIMD with
// everything in its 512 registers -- x, y, the bounds AND the step's pairs (KEEP: one LDS read per
// dimension and step instead of three).  Measured against the two-wave / LDS-bounds form (same box,
// 65 536 walkers, ms per 40 d steps, MODE 1): d = 128: 36.6 -> 27.2; d = 124: 26.1 -> 25.0; below that
// two waves win (d = 116: 18.4 against 21.0; d = 108: 16.0 / 18.8; d = 100: 14.0 / 16.4).
__host__ __device__ constexpr bool inc_one_wave_regs(int dq, int mode, bool per)
{
    return !per && ((mode == 1 && dq >= 31) || (mode == 2 && dq >= 32));
}

__host__ __device__ constexpr int inc_min_waves(int dq, int mode, bool per = false)
{
    // (periodic parameters: the LDS of a workgroup is laid out for four waves per SIMD up to
    // dq = 8 and for two above, inc_chunk)
    if (per) return MCMC_EXP_WAVES(STEP, dq <= 8 ? 4 : dq <= 31 ? 2 : 1);
    return MCMC_EXP_WAVES(STEP,
        mode == 0 ? (dq <= 12 ? 4 : dq <= 30 ? 2 : 1)
        : mode == 1 ? ((dq <= 8 || dq == 13) ? 4 : dq <= 30 ? 2 : 1)
        // (MODE 2, round 5: MODE 1's registers + the stream of the carried log-prior in LDS --
        // 24 bytes per dimension and column --, which four workgroups per CU hold up to dq = 8)
        : (dq <= 8 ? 4 : dq <= 31 ? 2 : 1));
}

// per-dimension bounds [lo_i, hi_i] (MODE 1, 2): in registers up to dq = 12, in LDS above.  With
// periodic parameters the kernels at four waves per SIMD keep them in registers up to dq = 5 only:
// at dq = 7, 8 the compiler spilled two of them INSIDE the step loop (two scratch loads and their
// s_waitcnt vmcnt(0) per step: 2.47 ms per 1200 steps at d = 30 against 1.15 without the flag)
__host__ __device__ constexpr bool inc_bounds_in_lds(int dq, int mode, bool per)
{
    return MCMC_EXP_BOUNDS_LDS(mode > 0 && ((dq > 12 && !inc_one_wave_regs(dq, mode, per)) ||
                                            (per && dq >= 6 && dq <= 8)));
}

// The step's (v, u) pairs kept in registers from the trial to the commit (no second LDS read):
// MEASURED (round 3, tools/exp_inc_variants.sh with -DEXP_KEEP=1 / -DEXP_STEP_WAVES=1, 65 536
// walkers, step kernel ms per 40 d steps, default -> kept): d = 52: 3.15 -> 3.01, 64: 4.25 -> 3.81,
// 80: 5.98 -> 5.36, 96: 8.25 -> 7.28, 100: 9.83 -> 8.96, 112: 11.99 -> 11.42, 116: 14.5 -> 13.9;
// d = 124 / 128 (one wave per SIMD now, inc_min_waves): 24.9 -> 17.7 / 35.4 -> 19.5.  The kernels
// at four waves per SIMD (dq <= 12, 128 registers) cannot afford the 4 dq registers.
#ifndef MCMC_EXP_FLOAT_LDS
#define MCMC_EXP_FLOAT_LDS(tuned) (tuned)
#endif
#ifndef MCMC_EXP_MIX_XLDS
#define MCMC_EXP_MIX_XLDS(tuned) (tuned)   // step_inc_mix_kernel: x in LDS where that buys a second wave per SIMD
#endif
#ifndef MCMC_EXP_FLOAT_VEC
#define MCMC_EXP_FLOAT_VEC(tuned) (tuned)
#endif
#ifndef MCMC_EXP_FLOAT_LDS_KEEP
#define MCMC_EXP_FLOAT_LDS_KEEP(tuned) (tuned)
#endif
// Round 6: general bounds (MODE 1, 2) at dq >= 13 -- the two-wave kernels, where the per-dimension
// bounds sit in LDS -- test the support on SINGLE-PRECISION copies of the bounds read from LDS,
// two rows per ds_read_b128 (inc_float_lds below: a quarter of the bytes of the (lo, hi) double
// pairs), and that frees the registers to keep the pairs like MODE 0 does.
__host__ __device__ constexpr bool inc_float_lds(int dq, int mode, bool per);
__host__ __device__ constexpr bool inc_keep_pairs(int dq, int mode, bool per = false)
{
    return (mode == 0 && dq >= 13) || (inc_float_lds(dq, mode, per) && MCMC_EXP_FLOAT_LDS_KEEP(true)) ||
           inc_one_wave_regs(dq, mode, per);
}
// Measured (round 6, same box, 65 536 walkers, step kernel ms per 40 d steps, round-5 form -> this one;
// tools/cliff_bench.py d:1:-1, tools/config5_bench.py; profiles/r06_float_lds.txt).  MODE 1:
//   d = 52: 3.45 -> 3.07 | 56: 4.12 -> 3.80 | 60: 4.48 -> 4.33 | 64: 5.03 -> 4.73 | 68: 5.66 -> 5.32 | 76: 6.28 -> 6.07
//   84: 7.50 -> 7.03 | 88: 8.38 -> 7.88 | 92: 10.55 -> 9.27 | 96: 11.35 -> 9.95 | 100: 13.79 -> 11.35 | 112: 16.44 -> 13.99
//   120: 19.7 -> 19.8.  MODE 2 gains up to dq = 18 (d = 52: 5.27 -> 4.37, 64: 5.98 -> 5.67, 72: 7.23 -> 6.80) and
//   spills above (d = 80: 10.1 -> 13.1, d = 96: 14.7 -> 49): it keeps the round-5 form there.
// (The per-dimension test from LDS doubles was half of the kernel's time at d = 100: 13.9 ms against 6.1
// with the test compiled out and the pairs kept, profiles/r06_margin_test.txt.)
__host__ __device__ constexpr bool inc_float_lds(int dq, int mode, bool per)
{
    return MCMC_EXP_FLOAT_LDS(!per && !inc_one_wave_regs(dq, mode, per) &&
                              ((mode == 1 && dq >= 13) || (mode == 2 && dq >= 13 && dq <= 18)));
}
// ... the verdict of that test gathered in a VECTOR register (v_med3_f32 + or, one compare per step)
// where its three more registers fit -- MODE 1 at dq = 14 .. 22: the two waves of a SIMD no longer
// wait on a chain of 2 dq scalar ANDs (d = 88: 8.58 -> 7.88) --, on the scalar unit elsewhere (dq = 13
// runs four waves on 128 registers: 3.07 against 17.2 spilled; dq >= 23: d = 96: 9.95 against 17.6)
__host__ __device__ constexpr bool inc_float_vec(int dq, int mode)
{
    return MCMC_EXP_FLOAT_VEC(mode == 1 && dq >= 14 && dq <= 22);
}

//   ONED: some parameter block has ONE parameter; the steps on its columns (a.colflag) draw the
//   RandProposer1D variates of the un-paired stream (step_variates), every lane for itself
//   EMIT: every accepted step past the burn-in stores the point it LEAVES with its weight
//   (mcmc.py:691-707: rows[w][n_rows[w]++] = (weight, logpost, logprior, loglike, x), a.s.rows /
//   n_rows / row_cap as in the from-scratch kernels) -- the reference's own product, `emit: chains`
//   PER: some parameter is PERIODIC (prior.py:658-676, Prior.reduce_periodic; up to kMaxPeriodic of
//   them, MODE 1 or 2; specification: step_core_inc with `carry_periodic`).  The bounds of a
//   periodic dimension are [lo, pred(hi)]: a step on which every lane of the wave is inside all
//   its bounds (most steps) is the plain step.  Only when some lane is outside (wave-uniform
//   branch) the rows are looked at again, branch-free: a periodic coordinate that LEFT [lo, hi)
//   is wrapped, t' = ((t - lo) / w - floor(.)) w + lo (inside the interval the reference's
//   expression returns t up to its own rounding; here it returns t), and when the winding number
//   changes (floor != 0) the move sh = t' - t is carried into the whitened residual,
//   y'_j += sh L^-1[j][i] for j >= i (ascending i), and the walker's chi2 is summed from that
//   residual instead of moved.  (Rounds 2-4 wrapped every periodic coordinate at every step;
//   round 5 first had a kernel of its own for this, incremental_periodic.hip, whose fast path
//   read the bounds from LDS four rows at a time and waited for them: 1.78 ms per 1200 steps at
//   d = 30 against 1.15 for the same bounds without the flag -- now it IS the plain step.)
template <int DQ, int MODE, bool UNIT_T, bool ONED, bool EMIT = false, bool PER = false>
__global__ void __launch_bounds__(256, inc_min_waves(DQ, MODE, PER)) step_inc_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double2 smem2[];
    constexpr int COLB = 4 * DQ;                 // (v, u) pairs per column
    constexpr bool NORMP = MODE == 2;
    static_assert(!PER || (MODE >= 1 && !EMIT), "periodic parameters: general bounds, no emitted rows");
    constexpr int C = inc_chunk(DQ, NORMP, PER);
    constexpr int CHUNK = C * COLB;              // pairs per chunk
    constexpr bool kBoundsInLds = inc_bounds_in_lds(DQ, MODE, PER);
    constexpr bool kBoundsInRegs = MODE > 0 && !kBoundsInLds;
    // (round 5 late) kernels at four waves per SIMD whose bounds sit in LDS -- periodic parameters
    // at dq = 6..8 -- keep single-precision copies of them in registers, rounded inward and moved
    // in by one more ulp (2 dq registers): the support test of a step is taken on those, and the
    // double-precision bounds are read from LDS only when some coordinate is not inside for
    // certain.  (Read from LDS at every step the compiler, short of registers, fetched them one
    // row at a time -- eight dependent round trips per trial: SQ_WAIT_ANY 49 % of the wave-cycles.)
    // ... and the two-wave kernels of general bounds at dq = 14..22 (MODE 2: ..18; at dq = 21 it
    // spilled: 19.0 ms per 3360 steps at d = 84 against 11.9 per 3520 at d = 88), where 2 dq more
    // registers still fit: one LDS read per dimension and step less -- measured, same box, MODE 1,
    // ms per 40 d steps of 65 536 walkers: d = 56: 4.65 -> 4.18, 60: 5.37 -> 4.49, 64: 5.94 -> 5.15,
    // 68: 6.68 -> 5.66, 80: 8.34 -> 6.83, 88: 9.83 -> 8.22; d = 96 and 100 (dq = 24, 25) spill: 11.4 ->
    // 16.2, 13.9 -> 25.1; dq = 13 runs at four waves (128 registers): 3.33 -> 12.9
    constexpr bool kFloatLds = kBoundsInLds && inc_float_lds(DQ, MODE, PER);
    constexpr bool kFloatVec = kFloatLds && inc_float_vec(DQ, MODE);
    constexpr bool kFloatBounds =
        kBoundsInLds && !kFloatLds &&
        // (periodic parameters at dq = 14..20 as well: d = 56 with one 5.93 -> 5.07, d = 64 with two
        // 8.25 -> 6.75, d = 80 with two 11.74 -> 9.72; dq = 23 without: no difference)
        MCMC_EXP_FLOAT_BOUNDS(PER ? (DQ <= 8 || (DQ >= 14 && DQ <= 20))
                                  : (DQ >= 14 && DQ <= (MODE == 1 ? 22 : 18)));
    // (MODE 0 = ONE box [0, hi] for every dimension, BASELINE configs 2-4: its support test works
    // on the high words of the trial coordinates, see `trial` below.  Round 2 took it on lane masks
    // at four waves per SIMD and on v_max / v_min_f64 at two: 2 DQ FP64 instructions per step
    // either way; now DQ 32-bit ones.)
    // KEEP: the step's DQ (v, u) pairs stay in registers from the trial to the commit instead of
    // being read from LDS twice (4 DQ VGPRs: only where the kernel runs two waves per SIMD on
    // 256 registers and the LDS pipe, not the register file, is what binds)
    constexpr bool KEEP = MCMC_EXP_KEEP(inc_keep_pairs(DQ, MODE, PER));
    // pairs fetched ahead of the trial arithmetic, PIPE at a time.  The kernels at three / four waves
    // per SIMD got it in round 4, once the staged variates had freed 8 registers: measured over
    // d = 4 ... 52 (round-4 sweep, same box): dq 5 ... 12 gain 0.2 - 2 % (MODE 1 at d = 30: 3.9 %),
    // dq <= 4 nothing, dq = 13 at four waves (MODE 1 / 2) LOSES 5 % to spills
    constexpr int PIPE = KEEP ? 0
        : MCMC_EXP_PIPE(inc_min_waves(DQ, MODE, PER) <= 2 ? 4 : (DQ >= 5 && DQ <= 12) ? 4 : 0);
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, c = tid & 3, wave = tid >> 6, lane = tid & 63;
    const int W = s.W, d = a.d;
    const int w = blockIdx.x * 64 + (tid >> 2);
    const int g = __builtin_amdgcn_readfirstlane(w / s.group_size);
    const int ncols = s.n_steps;
    // (the launch's columns inside its direction set: see IncStepArgs::vu_cols)
    const int set_cols = a.vu_cols > 0 ? a.vu_cols : ncols;
    const double2* __restrict__ gVU =
        (const double2*)a.VU + ((size_t)g * set_cols + (size_t)a.col0) * COLB;
    constexpr int dpad = 4 * DQ;
    double2* const sVU = smem2;                          // [2][CHUNK]
    double* const sW = (double*)(smem2 + 2 * CHUNK);     // [2][CHUNK] doubles, NORMP only
    // [dpad] (lo, hi), kBoundsInLds only.  A STATIC array (round 5 late): behind the chunks in the
    // dynamic block the compiler could not tell its reads from the chunk DMA's writes and put an
    // s_waitcnt vmcnt(0) in front of the first one of every step -- the next chunk's transfer
    // was waited for at once instead of at the end of the chunk
    __shared__ double2 sLHs[kBoundsInLds ? dpad : 1];
    double2* const sLH = sLHs;
    // kFloatLds: single-precision copies of the bounds, rounded INWARD and moved in by one more ulp
    // (a trial rounded to single precision that lies within them is inside for certain), two rows
    // per 16 bytes: sFB[(kk / 2) * 4 + c] = (lo, hi of row kk & ~1; lo, hi of row kk | 1)
    __shared__ float4 sFB[kFloatLds ? 4 * ((DQ + 1) / 2) : 1];
    // periodic parameters, behind that: the wrap moves of a step [walker of the workgroup][periodic
    // parameter], written by the lane that owns the dimension and read by its quad, and
    // L^-1[j][i_q] for j >= i_q, the q-th periodic dimension -- what a wrap moves y by
    int np = 0;
    if (PER)
        for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
    double* const sShift = (double*)(smem2 + 2 * CHUNK + (NORMP ? CHUNK : 0));   // [64][np]
    double* const sLc = sShift + 64 * np;                                // [np][dpad]
    __shared__ double4 sPer[PER ? dpad : 1];     // periodic dimensions: (lo, hi, w, RN(1 / w)), w = hi - lo
    __shared__ int sPdim[PER ? kMaxPeriodic : 1];   // the periodic dimensions, ascending
    auto is_periodic = [&](int i) { return PER && ((a.periodic_mask4[i >> 5] >> (i & 31)) & 1u); };
    const double* __restrict__ gW =
        NORMP ? a.VW + ((size_t)g * set_cols + (size_t)a.col0) * COLB : nullptr;

    // chunk k of the launch -> buffer k & 1, by DMA: every wave moves every fourth KiB
    auto stage = [&](int k) {
        const int first = k * C;
        if (first >= ncols) return;
        const int cols = ncols - first < C ? ncols - first : C;
        const int bytes = cols * COLB * 16;
        const char* src = (const char*)(gVU + (size_t)first * COLB);
        char* dst = (char*)(sVU + (k & 1) * CHUNK);
        for (int kb = wave; kb * 1024 < bytes; kb += 4) {
            if (kb * 1024 + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
        }
        if (NORMP) {   // the chunk's w (8 bytes per dimension and column), dealt from the last wave down
            const int wbytes = cols * COLB * 8;
            const char* wsrc = (const char*)(gW + (size_t)first * COLB);
            char* wdst = (char*)(sW + (k & 1) * CHUNK);
            for (int kb = 3 - wave; kb * 1024 < wbytes; kb += 4) {
                if (kb * 1024 + lane * 16 < wbytes)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(wsrc + kb * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void*)(wdst + kb * 1024), 16, 0, 0);
            }
        }
    };
    // (a launch that refreshes y itself uses the chunk buffers as scratch first: its first
    // chunk is staged behind that, below)
#ifdef MCMC_EXP_NO_FOLD
    const bool refresh_y = false;   // (timing experiment: the in-kernel refresh compiled out)
#else
    const bool refresh_y = !PER && !(NORMP && DQ >= 29) && (a.anchor & 2) != 0;   // wave-uniform
#endif
    if (!refresh_y) stage(0);
    MCMC_EXP_BLOCK_BEGIN();

    const double blo = a.box_lo, bhi = a.box_hi;
    const unsigned bhi_word = (unsigned)__double2hiint(bhi);   // (MODE 0: blo == +0, 0 < bhi < inf)
    double x[DQ], y[DQ], lo[kBoundsInRegs ? DQ : 1], hi[kBoundsInRegs ? DQ : 1];
    unsigned mine = 0;     // (PER) bit kk: dimension 4 kk + c of this lane is periodic
    float flo[kFloatBounds ? DQ : 1], fhi[kFloatBounds ? DQ : 1];
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        // (MODE 0: a padded dimension rests at the middle of the box, inside for every step)
        x[kk] = in ? s.x[(size_t)i * W + w] : (MODE == 0 ? 0.5 * (blo + bhi) : 0.0);
        y[kk] = in ? a.y[(size_t)i * W + w] : 0.0;
        if (kBoundsInRegs) {
            lo[kk] = a.prior[i];               // padded: -inf / +inf beyond d
            hi[kk] = a.prior[dpad + i];
            // (a periodic dimension: [lo, pred(hi)] -- "t <= pred(hi)" is "t < hi": inside means
            // that nothing has to be wrapped)
            if (PER && in && is_periodic(i)) hi[kk] = pred_double(hi[kk]);
        }
        if (PER && in && is_periodic(i)) mine |= 1u << kk;
        if (kFloatBounds) {
            const double blo_d = a.prior[i];
            double bhi_d = a.prior[dpad + i];
            if (in && is_periodic(i)) bhi_d = pred_double(bhi_d);
            // lo: rounded up, then one ulp further in; hi: rounded down, then one further in
            // (infinite bounds -- the padding, a dimension without a bound -- stay infinite)
            float fl = __double2float_ru(blo_d), fh = __double2float_rd(bhi_d);
            if (fl > -INFINITY) fl = nextafterf(fl, INFINITY);
            if (fh < INFINITY) fh = nextafterf(fh, -INFINITY);
            flo[kk] = fl;
            fhi[kk] = fh;
        }
    }
    if (kBoundsInLds)
        for (int i = tid; i < dpad; i += 256) {
            const double bh = a.prior[dpad + i];
            sLH[i] = make_double2(a.prior[i], (i < d && is_periodic(i)) ? pred_double(bh) : bh);
        }
    if (kFloatLds)
        for (int e = tid; e < 4 * ((DQ + 1) / 2); e += 256) {
            const int kk0 = 2 * (e >> 2), cc = e & 3;
            float v[4];
            for (int h = 0; h < 2; ++h) {
                const int i = 4 * (kk0 + h) + cc;   // (beyond the padded rows: no bound)
                float fl = -INFINITY, fh = INFINITY;
                if (i < dpad) {
                    // lo: rounded up, then one ulp further in; hi: rounded down, then one further in
                    // (infinite bounds -- the padding, a dimension without a bound -- stay infinite)
                    fl = __double2float_ru(a.prior[i]);
                    fh = __double2float_rd(a.prior[dpad + i]);
                    if (fl > -INFINITY) fl = nextafterf(fl, INFINITY);
                    if (fh < INFINITY) fh = nextafterf(fh, -INFINITY);
                }
                v[2 * h] = fl;
                v[2 * h + 1] = fh;
            }
            sFB[e] = make_float4(v[0], v[1], v[2], v[3]);
        }
    if (PER) {
        for (int i = tid; i < dpad; i += 256) {
            const double plo = a.prior[i], phi = a.prior[dpad + i];
            sPer[i] = (i < d && is_periodic(i)) ? make_double4(plo, phi, phi - plo, 1.0 / (phi - plo))
                                                : make_double4(0.0, 1.0, 1.0, 1.0);
        }
        if (tid == 0) {
            int n = 0;
            for (int i = 0; i < d; ++i)
                if (is_periodic(i) && n < kMaxPeriodic) sPdim[n++] = i;
        }
        __syncthreads();   // (sPdim)
        for (int e = tid; e < np * dpad; e += 256) {
            const int j = e % dpad, q = e / dpad, i = sPdim[q];
            sLc[e] = (j >= i && j < d) ? a.Lrow[(size_t)j * d + i] : 0.0;
        }
    }
    // the slot of the periodic dimension 4 kk + c in sPdim: the periodic dimensions of the rows
    // below (scalar) plus those of this row in the lane classes below c
    const unsigned below_c = (1u << c) - 1u;
    auto slot_of = [&](int kk) {
        int n = 0;
        for (int q = 0; q < ((4 * kk) >> 5); ++q) n += __builtin_popcount(a.periodic_mask4[q]);
        const unsigned word = a.periodic_mask4[(4 * kk) >> 5];
        n += __builtin_popcount(word & ((1u << ((4 * kk) & 31)) - 1u));
        return n + __builtin_popcount((word >> ((4 * kk) & 31)) & below_c);
    };
    double* const myShift = sShift + (tid >> 2) * np;
    // (wave-uniform) some lane class of row kk holds a periodic dimension
    auto row_periodic = [&](int kk) {
        return PER && ((a.periodic_mask4[(4 * kk) >> 5] >> ((4 * kk) & 31)) & 15u) != 0u;
    };
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    __shared__ pair_t sRE[kStagedPairs];   // the (r, Ea) pairs of the current octet (StagedVariates)
    if (refresh_y) {
        // y = L^-1 (x - mu) from the walker's x (round 5: whiten_state_kernel folded into the
        // launch that needs it -- one kernel and ~17 us less on the main stream between two step
        // kernels).  The deviations of the wave's 16 walkers go to LDS ([walker][dimension], in the
        // chunk buffers, which nothing has been staged into yet: every wave its own 16 x 4 DQ
        // doubles, no barrier), then ONE rolled loop over the dimensions i ascending: every row
        // j = 4 kk + c of the lane takes fma(L^-1[j][i], dev_i, y_j) -- for i > j the factor is
        // the exact +0.0 above the diagonal (tri_inverse_lower), which leaves the chain where it
        // ended at i = j: bit for bit orc_whiten's ascending chain over i <= j.
        // L^-1 comes through LDS eight columns at a time (all 256 threads fill a [4 DQ][8] tile,
        // every lane then reads its rows there): read per lane straight from memory, the rolled
        // loop waited for one memory round trip per dimension -- +40 us per launch at d = 30, +260
        // at d = 100, more than the kernel it replaced (same-box A/B, profiles/r05_launch_gap.txt).
        static_assert((NORMP ? 3 : 2) * C >= 32 || (NORMP && DQ >= 29) || PER,
                      "the chunk buffers hold the deviations of the workgroup's 64 walkers");
        // (MODE 2 from d = 113 on: the chunks are too small; the host refreshes y with
        // whiten_state_kernel before every such launch -- capi.hip: IncPlan::fold)
        constexpr int TW = 8;
        static_assert(sizeof(pair_t) * kStagedPairs >= sizeof(double) * dpad * TW, "the tile fits sRE");
        double* const sdev = (double*)sVU + (size_t)(tid >> 2) * dpad;   // this walker's deviations
        double* const tile = (double*)sRE;   // [dpad][TW] (the staged variates are not in use yet)
#pragma unroll
        for (int kk = 0; kk < DQ; ++kk) {
            const int i = 4 * kk + c;
            sdev[i] = i < d ? x[kk] - a.mean[i] : 0.0;
            y[kk] = 0.0;
        }
        for (int i0 = 0; i0 < d; i0 += TW) {
            __syncthreads();   // (the previous tile has been used; the first: sdev is written)
            for (int e = tid; e < dpad * TW; e += 256) {
                const int j = e / TW, i = i0 + e % TW;
                tile[e] = (j < d && i < d) ? a.Lrow[(size_t)j * d + i] : 0.0;
            }
            __syncthreads();
            const lds_doubles pt = relaunder(tile + c * TW), pd = relaunder(sdev + i0);
#pragma unroll
            for (int ii = 0; ii < TW; ++ii) {
                const double dv = pd[ii];   // (beyond d: 0, and the tile's column is 0)
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) y[kk] = fma(pt[4 * kk * TW + ii], dv, y[kk]);
            }
        }
        __syncthreads();   // every wave is done with its scratch: the first chunk may land
        stage(0);
    }
    if (a.anchor) {   // (wave-uniform) y has just been refreshed from x: the carried log-likelihood
        double pa = 0.0;   // is re-anchored on it (orc_anchor_loglike)
#pragma unroll
        for (int kk = 0; kk < DQ; ++kk) pa = fma(y[kk], y[kk], pa);
        llik = -0.5 * (s.cnorm0 + quad_sum(pa));
        if (NORMP) {
            // ... and the carried log-prior on x: the normal terms, branch-free -- a dimension
            // without one has 1/scale = 0 and mls = 0, so its term is fma(-0, 0, 0) = +0 and
            // leaves the chain untouched (four chains over i mod 4, as every sum here)
            double sc = 0.0;
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                const int i = 4 * kk + c;
                const double qq = (x[kk] - a.prior[2 * dpad + i]) * a.prior[3 * dpad + i];
                sc = sc + fma(-0.5 * qq, qq, a.prior[4 * dpad + i]);
            }
            lpri = s.uniform_logp + quad_sum(sc);
        }
        lpost = lpri + llik;
    }
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;     // accepted steps of this launch (< 2^31)
    int nrow = EMIT ? s.n_rows[w] : 0;
    const bool thinning = EMIT && s.thin > 1;   // (wave-uniform)
    int tacc = thinning ? s.thin_acc[w] : 0;
    const double inv_thin = thinning ? 1.0 / (double)s.thin : 1.0;
    // |u|^2 of the launch's columns, through the constant address space: the address is
    // wave-uniform, the load a scalar one (s_load_dwordx2 on the scalar cache's counter -- a vector
    // load would queue behind the DMA of the next chunk on vmcnt)
    const cdoubles gUU = (cdoubles)(unsigned long long)(a.UU + (size_t)g * set_cols + (size_t)a.col0);
    // (v.w, loc.w) of the columns (carried log-prior), the same way
    const cdoubles gNL =
        (cdoubles)(unsigned long long)(NORMP ? a.NL + 2 * ((size_t)g * set_cols + (size_t)a.col0) : a.UU);
    const uint32_t gid = s.walker0 + (uint32_t)w;
    // stuck test (mcmc.py:717-743) on integers: (double)n > m  <=>  n > floor(m) for n integer
    const double mt10 = s.max_tries * 10.0;
    const int lim1 = s.max_tries < 2.0e9 ? (int)floor(s.max_tries) : 0x7fffffff;
    const int lim10 = mt10 < 2.0e9 ? (int)floor(mt10) : 0x7fffffff;

    __shared__ dpair_t short_log_lds[SHORT_LOG_TABLE_SIZE];
    const short_log_tab slog = short_log_load(short_log_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool burning = lanes(burn > 0) != 0ull;   // wave-uniform
    unsigned long long cur_oct = ~0ull;
    StagedVariates sv;
    sv.init(sRE, wave, lane);
    const int hw_slot = hw_wave_slot();

    for (int base = 0, k = 0; base < ncols; base += C, ++k) {
        const double2* __restrict__ cur = sVU + (k & 1) * CHUNK;
        stage(k + 1);     // travels while this chunk is consumed
        const int cols = __builtin_amdgcn_readfirstlane(ncols - base < C ? ncols - base : C);
        unsigned long long oned_cols = 0;   // bit sl: column sl of the chunk is a 1-D one
        if (ONED)
            oned_cols = lanes(lane < cols &&
                              a.colflag[(size_t)g * set_cols + (size_t)a.col0 + base + lane] != 0);
        // (the step loop is rolled: an unrolled body lets the scheduler hoist the LDS reads of
        // several steps and costs the registers that decide the occupancy)
        // (the LDS address of the step's column is CARRIED and passes through the empty asms in
        // place: a laundered copy of a pointer that stays live costs a v_mov per step)
        unsigned coff = lds_offset(cur + c);
        unsigned woff = NORMP ? lds_offset(sW + (k & 1) * CHUNK + c) : 0u;
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl, coff += COLB * 16, woff += NORMP ? COLB * 8 : 0) {
            {
                {
                    // Variates: EIGHT consecutive steps (an aligned octet of the global step index)
                    // at once -- lane class c draws the Philox block of the step pair 4 * octet + c
                    // (PairRng: both halves, four logarithms, two square roots) and stages its two
                    // (r, Ea) pairs in LDS; a step reads its pair there (StagedVariates).
                    const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
                    if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step
                        cur_oct = S >> 3;
                        rotate_priority<inc_min_waves(DQ, MODE, PER)>(hw_slot);
                        PairRng pr;
                        pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)c, slog);
                        sv.fill(sRE, wave, lane, c, pr, S);
                    }
                    double r, Ea;
                    if (ONED && ((oned_cols >> sl) & 1ull)) {   // wave-uniform
                        step_variates(s.key0, s.key1, gid, S, 0, true, r, Ea);
                    } else {
                        sv.fetch(r, Ea);
                    }
                    sv.next();
                    const double uu = gUU[base + sl];   // (wave-uniform address: a scalar load)
                    const lds_pairs col = (lds_pairs)(unsigned long long)coff;
                    const lds_doubles wcol = (lds_doubles)(unsigned long long)woff;
                    double pc = 0.0, sc = 0.0;
                    // (the support test is kept as the wave's lane mask: every comparison lands
                    // in a scalar register pair and the ANDs run on the scalar unit)
                    unsigned long long inb = ~0ull;
                    // MODE 0 (one box [0, bhi] for every dimension): non-negative doubles order
                    // like their bit patterns, so a trial coordinate whose HIGH WORD is below
                    // bhi's is inside for certain -- one 32-bit max per dimension instead of two
                    // FP64 compares (or min / max); whatever is not certain (within 2^-20 of bhi,
                    // beyond it, negative, -0) is decided by the exact comparisons below, a
                    // wave-uniform branch that a posterior away from the walls never takes
                    unsigned hmx = 0u;
                    float4 fb2 = make_float4(0.f, 0.f, 0.f, 0.f);   // (kFloatLds) the bounds of two rows
                    unsigned facc = 0u;   // (kFloatLds) 0: every trial coordinate so far is inside its float bounds
                    auto trial = [&](int kk, const pair_t p) {
                        const double t = fma(r, p.x, x[kk]);
                        if (MODE == 0) {
                            const unsigned h = (unsigned)__double2hiint(t);
                            hmx = hmx > h ? hmx : h;
                        }
                        else if (kBoundsInRegs) inb &= lanes(t <= hi[kk]) & lanes(t >= lo[kk]);
                        else if (kFloatBounds) {
                            // (inside for certain: the trial rounded to single precision lies
                            // within bounds rounded INWARD and moved in by one more ulp)
                            const float tf = __double2float_rn(t);
                            inb &= lanes(tf <= fhi[kk]) & lanes(tf >= flo[kk]);
                        }
                        else if (kFloatLds) {
                            // ... the same on copies read from LDS: the rows kk, kk + 1 share a read
                            // (the verdict is gathered in a VECTOR register: the median of (trial, lo,
                            // hi) IS the trial exactly when it lies within them -- one v_med3_f32 and
                            // an or of the differing bits per dimension, one compare per step.  With
                            // two compares and two scalar ANDs per dimension the two waves of a SIMD
                            // waited on the chain through the scalar unit: d = 100, 11.4 ms per
                            // 4 000 steps against 13.8 with the double-precision bounds)
                            if ((kk & 1) == 0) fb2 = sFB[(kk >> 1) * 4 + c];
                            const float tf = __double2float_rn(t);
                            const float bl = (kk & 1) ? fb2.z : fb2.x, bh = (kk & 1) ? fb2.w : fb2.y;
                            if (kFloatVec)
                                facc |= __float_as_uint(__builtin_amdgcn_fmed3f(tf, bl, bh)) ^ __float_as_uint(tf);
                            else   // (inc_float_vec: where the vector form's registers do not fit)
                                inb &= lanes(tf <= bh) & lanes(tf >= bl);
                        }
                        else {
                            const double2 lh = sLH[4 * kk + c];
                            inb &= lanes(t <= lh.y) & lanes(t >= lh.x);
                        }
                        pc = fma(y[kk], p.y, pc);   // (y . u: the log-likelihood is carried)
                        if (NORMP) sc = fma(x[kk], wcol[4 * kk], sc);   // (x . w: the log-prior is carried)
                    };
                    pair_t pk[KEEP ? DQ : 1];
                    if constexpr (KEEP) {
                        // (all DQ reads up front.  Fetching them in batches, one batch ahead of
                        // the arithmetic, was measured at d = 100: batches of 4 / 6 / 9 / 13 pairs
                        // 7.96 / 7.78 / 7.61 / 7.84 ms against 7.74 -- no gain worth the code)
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) pk[kk] = col[4 * kk];
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) trial(kk, pk[kk]);
                    } else if constexpr (PIPE == 0) {
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) trial(kk, col[4 * kk]);
                    } else {
                        // (kernels held to one or two waves per SIMD have registers to spare
                        // and little else to cover the LDS latency: the pairs are fetched PIPE
                        // at a time, one batch ahead of the arithmetic)
                        constexpr int NB = (DQ + PIPE - 1) / PIPE;
                        pair_t buf[2][PIPE];
#pragma unroll
                        for (int j = 0; j < PIPE; ++j)
                            if (j < DQ) buf[0][j] = col[4 * j];
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
#pragma unroll
                            for (int j = 0; j < PIPE; ++j)
                                if (b + 1 < NB && (b + 1) * PIPE + j < DQ)
                                    buf[(b + 1) & 1][j] = col[4 * ((b + 1) * PIPE + j)];
#pragma unroll
                            for (int j = 0; j < PIPE; ++j)
                                if (b * PIPE + j < DQ) trial(b * PIPE + j, buf[b & 1][j]);
                        }
                    }
                    // inside the prior support = all four lanes of the walker are: the AND over
                    // the quad is taken on the wave's lane mask (scalar unit, no vector work)
                    unsigned long long inside_m;   // lane mask: the walker's four lanes are all inside
                    unsigned long long wound = 0ull;   // (PER) lanes whose coordinate wrapped with a move
                    bool slow = false;                 // (PER, wave-uniform) some lane left some bound
                    // (PER) the residual with the wrap moves of this step (a wrap in the wave only):
                    // on top of every row's fma(r, u, y), the moves in ascending dimension -- the
                    // columns of L^-1 of the periodic dimensions sit in LDS (sLc) -- for the lanes `on`
                    auto shift_rows = [&](double (&yr)[DQ], unsigned long long on_m) {
                        const bool on_l = __builtin_amdgcn_inverse_ballot_w64(on_m);
#pragma unroll 1
                        for (int q = 0; q < np; ++q) {
                            const double sv_ = myShift[q];               // the same in the walker's quad
                            if (lanes(sv_ != 0.0) == 0ull) continue;     // wave-uniform
                            const int i = sPdim[q];                      // the dimension that wrapped
                            const double* __restrict__ lc = sLc + q * dpad + c;
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) {
                                const int j = 4 * kk + c;
                                const bool on = on_l & (sv_ != 0.0) & (j >= i) & (j < d);
                                yr[kk] = on ? fma(sv_, lc[4 * kk], yr[kk]) : yr[kk];
                            }
                        }
                    };
                    if (MODE == 0) {
                        // (ONE vector compare: "some lane is not certainly inside" is read off
                        // the mask on the scalar unit)
                        inside_m = lanes(quad_max_u32(hmx) < bhi_word);
                        if (inside_m != lanes(true)) {   // (wave-uniform, rare) the exact test
                            asm volatile("" : "+v"(coff));
                            const lds_pairs colx = (lds_pairs)(unsigned long long)coff;
                            // (fully unrolled: a rolled loop would index x[] at run time and
                            // move the walker's state from registers to scratch memory)
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) {
                                const double t = fma(r, colx[4 * kk].x, x[kk]);
                                inb &= lanes(t <= bhi) & lanes(t >= blo);
                            }
                            inside_m = quad_all_mask(inb);
                        }
                    } else {
                        if constexpr (kFloatVec) inb = lanes(facc == 0u);
                        if constexpr ((kFloatBounds || kFloatLds) && !PER) {
                            // (wave-uniform, rare: some trial coordinate is not inside for certain
                            // -- within 2^-23 of a bound, or outside: the exact comparisons, rows
                            // from LDS; with periodic parameters the pass below does them)
                            if (__builtin_expect(inb != ~0ull, 0)) {
                                inb = ~0ull;
                                asm volatile("" : "+v"(coff));
                                const lds_pairs colx = (lds_pairs)(unsigned long long)coff;
#pragma unroll
                                for (int kk = 0; kk < DQ; ++kk) {
                                    const double t = fma(r, colx[4 * kk].x, x[kk]);
                                    const double2 lh = sLH[4 * kk + c];
                                    inb &= lanes(t <= lh.y) & lanes(t >= lh.x);
                                }
                            }
                        }
                        if constexpr (PER) {
                            // ---- some lane of the wave is outside some bound (wave-uniform; rare
                            // away from the walls except for walkers at the seam of a periodic
                            // parameter): row by row, branch-free -- a periodic coordinate that
                            // left [lo, hi) is wrapped (prior.py:675, the division by the period
                            // as div_by), its move goes to the walker's slot in LDS; the support
                            // test is taken on the wrapped coordinates
                            slow = __builtin_expect(inb != ~0ull, 0);
                            if (slow) {
                                inb = ~0ull;
                                asm volatile("" : "+v"(coff));
                                const lds_pairs colw = (lds_pairs)(unsigned long long)coff;
#pragma unroll
                                for (int kk = 0; kk < DQ; ++kk) {
                                    const bool per = (mine >> kk) & 1u;
                                    const double tk = fma(r, colw[4 * kk].x, x[kk]);
                                    double blo_k, bhi_k;
                                    if (kBoundsInRegs) { blo_k = lo[kk]; bhi_k = hi[kk]; }
                                    else { const double2 lh = sLH[4 * kk + c]; blo_k = lh.x; bhi_k = lh.y; }
                                    const bool out = !((tk <= bhi_k) & (tk >= blo_k));
                                    // (wave-uniform: a row without a periodic dimension has
                                    // nothing to wrap -- its part of the support test is all)
                                    if (!row_periodic(kk)) {
                                        inb &= lanes(!out);
                                        continue;
                                    }
                                    const double4 pw = sPer[4 * kk + c];   // (lo, hi, w, RN(1 / w)); no period: (0, 1, 1, 1)
                                    const double yv = div_by(tk - pw.x, pw.z, pw.w);
                                    const double fl = floor(yv);
                                    const double tw = (yv - fl) * pw.z + pw.x;
                                    const bool wr = per & out;
                                    const double shk = (wr & (fl != 0.0)) ? tw - tk : 0.0;
                                    wound |= lanes(shk != 0.0);
                                    if (per) myShift[slot_of(kk)] = shk;
                                    const bool ins = wr ? ((tw <= pw.y) & (tw >= pw.x)) : !out;
                                    inb &= lanes(ins);
                                }
                            }
                        }
                        inside_m = quad_all_mask(inb);
                    }
                    // chi2(y + r u) - chi2(y) = r (2 y.u + r |u|^2): the trial's log-likelihood from
                    // the carried one (step_core_inc, `carry`)
                    const double yu = quad_sum(pc);
                    const double mhr = -0.5 * r;
                    double lp = s.uniform_logp;
                    if (NORMP) {
                        // lp(x + r v) - lp(x) = -r/2 (r v.w + 2 (x.w - loc.w)) (step_core_inc, `carry_p`)
                        const double xw = quad_sum(sc) - gNL[2 * (base + sl) + 1];
                        lp = fma(mhr, fma(r, gNL[2 * (base + sl)], xw + xw), lpri);
                    }
                    double ll = fma(mhr, fma(r, uu, yu + yu), llik);
                    if constexpr (PER) {
                        if (__builtin_expect(wound != 0ull, 0)) {   // wave-uniform, rarer still: chi2 of the walkers whose
                            // residual took a move (any lane of the quad wrote a non-zero one) is summed anew
                            const unsigned long long wq = ~quad_all_mask(~wound);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes
                            double ytw[DQ];
                            asm volatile("" : "+v"(coff));
                            const lds_pairs colw = (lds_pairs)(unsigned long long)coff;
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) ytw[kk] = fma(r, colw[4 * kk].y, y[kk]);
                            shift_rows(ytw, ~0ull);
                            double ps = 0.0;
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) ps = fma(ytw[kk], ytw[kk], ps);
                            const double ll_w = -0.5 * (s.cnorm0 + quad_sum(ps));
                            ll = sel(wq, ll_w, ll);
                        }
                    }
                    // (outside the support lt is not used; an overflow gives lt = -inf or NaN,
                    // which fail both comparisons like the specification's explicit lt != -inf)
                    const double lt = lp + ll;
                    const double delta = UNIT_T ? (lpost - lt) : (lpost - lt) / s.temperature;
                    const unsigned long long acc_m = inside_m & (lanes(lt > lpost) | lanes(Ea > delta));
                    const bool accept = __builtin_amdgcn_inverse_ballot_w64(acc_m);
                    if (EMIT) {
                        // the point the walker leaves, with its weight (mcmc.py:691-707); each of
                        // the walker's four lanes stores its quarter: header word c, then the
                        // dimensions 4 kk + c -- 32 contiguous bytes per walker and instruction
                        bool em = accept & (burn <= 0);
                        if (lanes(em) != 0ull) {   // (wave-uniform: some walker emits)
                            int ew = wt;   // the weight the row is written with
                            if (thinning) {
                                // thinned output (collection.py:1373-1383): the weights add up; a
                                // row goes out when the sum reaches `thin`, with weight sum / thin
                                // (the quotient by a reciprocal, set right by the remainder)
                                const int tot = tacc + wt;
                                int q = (int)((double)tot * inv_thin);
                                int rem = tot - q * s.thin;
                                q += rem >= s.thin ? 1 : 0;
                                rem -= rem >= s.thin ? s.thin : 0;
                                q -= rem < 0 ? 1 : 0;
                                rem += rem < 0 ? s.thin : 0;
                                tacc = em ? rem : tacc;
                                ew = q;
                                em = em & (q > 0);
                            }
                            if (em & (nrow < s.row_cap)) {
                                double* __restrict__ row =
                                    s.rows + ((size_t)w * s.row_cap + nrow) * (size_t)(d + 4);
                                row[c] = c == 0 ? (double)ew : c == 1 ? lpost : c == 2 ? lpri : llik;
#pragma unroll
                                for (int kk = 0; kk < DQ; ++kk)
                                    if (4 * kk + c < d) row[4 + 4 * kk + c] = x[kk];
                            }
                            nrow += em ? 1 : 0;   // rows beyond the capacity are counted as dropped
                        }
                    }
                    llik = accept ? ll : llik;
                    if (NORMP) lpri = accept ? lp : lpri;   // (else: uniform_logp, always)
                    // (burn-in, mcmc.py:685-690, ends early in a run: its bookkeeping sits
                    // behind a wave-uniform test of "some lane is still burning in")
                    const bool was_burning = burning;   // (wave-uniform)
                    int lim_b;
                    asm("" : "=v"(lim_b));   // (set and used while burning in only)
                    if (burning) {
                        lim_b = burn > 0 ? lim10 : lim1;
                        burn -= (accept & (burn > 0)) ? 1 : 0;
                        burning = lanes(burn > 0) != 0ull;
                    }
                    const double ra = accept ? r : 0.0;
                    // (the pairs are read AGAIN from LDS: the pointer passes through an empty
                    // asm so that the compiler cannot keep the first reads alive in 4 DQ registers)
                    asm volatile("" : "+v"(coff));
                    lds_pairs col2 = (lds_pairs)(unsigned long long)coff;
                    if constexpr (KEEP) {
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            x[kk] = fma(ra, pk[kk].x, x[kk]);
                            y[kk] = fma(ra, pk[kk].y, y[kk]);
                        }
                    } else if constexpr (PIPE == 0) {
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            // (four pairs at a time: the pointer of the next four depends, through
                            // an empty asm, on the last result of these four -- else all DQ reads
                            // are issued up front into 4 DQ registers)
                            if (kk % 4 == 0 && kk) {
                                asm volatile("" : "+v"(coff) : "v"(y[kk - 1]));
                                col2 = (lds_pairs)(unsigned long long)coff;
                            }
                            const pair_t p = col2[4 * kk];
                            x[kk] = fma(ra, p.x, x[kk]);
                            y[kk] = fma(ra, p.y, y[kk]);
                        }
                    } else {   // one batch ahead, as in the trial loop
                        constexpr int NB = (DQ + PIPE - 1) / PIPE;
                        pair_t buf[2][PIPE];
#pragma unroll
                        for (int j = 0; j < PIPE; ++j)
                            if (j < DQ) buf[0][j] = col2[4 * j];
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            // (the batch after next must not start before this one is used)
                            if (b + 1 < NB && b > 0) {
                                asm volatile("" : "+v"(coff) : "v"(y[b * PIPE - 1]));
                                col2 = (lds_pairs)(unsigned long long)coff;
                            }
#pragma unroll
                            for (int j = 0; j < PIPE; ++j)
                                if (b + 1 < NB && (b + 1) * PIPE + j < DQ)
                                    buf[(b + 1) & 1][j] = col2[4 * ((b + 1) * PIPE + j)];
#pragma unroll
                            for (int j = 0; j < PIPE; ++j) {
                                const int kk = b * PIPE + j;
                                if (kk < DQ) {
                                    x[kk] = fma(ra, buf[b & 1][j].x, x[kk]);
                                    y[kk] = fma(ra, buf[b & 1][j].y, y[kk]);
                                }
                            }
                        }
                    }
                    if constexpr (PER) {
                        if (slow) {   // wave-uniform: an accepted coordinate that left [lo, hi) is the wrapped one
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) {
                                if (!row_periodic(kk)) continue;   // (wave-uniform)
                                const bool per = (mine >> kk) & 1u;
                                double blo_k, bhi_k;
                                if (kBoundsInRegs) { blo_k = lo[kk]; bhi_k = hi[kk]; }
                                else { const double2 lh = sLH[4 * kk + c]; blo_k = lh.x; bhi_k = lh.y; }
                                const bool out = !((x[kk] <= bhi_k) & (x[kk] >= blo_k));
                                const double4 pw = sPer[4 * kk + c];
                                const double yv = div_by(x[kk] - pw.x, pw.z, pw.w);
                                const double fl = floor(yv);
                                const double tw = (yv - fl) * pw.z + pw.x;
                                x[kk] = (per & out & accept) ? tw : x[kk];   // (a walker that stays keeps its x)
                            }
                            // ... and the residual of a walker that accepted a wrapping trial takes
                            // the moves (on top of fma(r, u, y): the same operations in the same
                            // order as for the trial)
                            if (wound != 0ull) shift_rows(y, acc_m);
                        }
                    }
                    lpost = accept ? lt : lpost;
                    {
                        const bool inside = __builtin_amdgcn_inverse_ballot_w64(inside_m);
                        prej = accept ? 0 : (prej + (inside ? 0 : 1));
                        wt = accept ? 1 : wt + 1;
                        nacc += accept ? 1 : 0;
                    }
                    // (after the burn-in the limit is the wave-uniform lim1)
                    unsigned long long over_m;
                    if (was_burning) over_m = lanes(wt - prej > lim_b);
                    else over_m = lanes(wt - prej > lim1);
                    if (__builtin_amdgcn_inverse_ballot_w64(over_m) && c == 0)
                        atomicCAS(s.stuck, 0, 1 + (int)gid);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk has landed
        __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        if (i < d) {
            s.x[(size_t)i * W + w] = x[kk];
            a.y[(size_t)i * W + w] = y[kk];
        }
    }
    if (c == 0) {
        s.logpost[w] = lpost; s.logprior[w] = lpri; s.loglike[w] = llik;
        s.weight[w] = wt; s.prior_rej[w] = prej; s.burn_left[w] = burn;
        s.n_accept[w] = nacc0 + nacc;
        if (EMIT) s.n_rows[w] = nrow;
        if (thinning) s.thin_acc[w] = tacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc : 0);
    MCMC_EXP_BLOCK_END();
}

// ---------------------------------------------------------------- the dragging step
// mcmc.py:564-668 in incremental mode (oracle: drag_core_inc): a dragging step is 1 + n_drag
// consecutive columns of VU -- the slow direction, then the fast directions of its interpolation
// steps -- and 1 + 2 n_drag evaluations, every one of them O(d): the start and the end point
// carry their whitened residuals (ys, ye), a move by r v moves them by fma(r, u, .).  The variates
// (r_i, E_i) of the sub-steps i = 0 .. n_drag are drawn four at a time, one per lane class.
__host__ __device__ constexpr int inc_drag_min_waves(int dq, int mode)
{
    // (measured like inc_min_waves, tools/drag_bench.py over builds held to 1..4 waves: two waves
    // with some spilled registers beat one up to d = 68 -- d = 44: 3.3e10 against 1.9e10 --, one
    // wave wins from d = 80 on)
    return MCMC_EXP_WAVES(DRAG, mode == 0 ? (dq <= 3 ? 4 : dq <= 5 ? 3 : dq <= 17 ? 2 : 1)
                                          : (dq <= 3 ? 3 : dq <= 17 ? 2 : 1));
}

template <int DQ, int MODE, bool UNIT_T, bool ONED>
__global__ void __launch_bounds__(256, inc_drag_min_waves(DQ, MODE))
drag_inc_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double2 smem2[];
    constexpr int COLB = 4 * DQ;
    constexpr bool kBoundsInRegs = MODE > 0 && DQ <= 12;
    constexpr bool kBoundsInLds = MODE > 0 && DQ > 12;   // (six DQ-long arrays live already)
    constexpr bool NORMP = MODE == 2;
    constexpr int dpad = 4 * DQ;
    __shared__ double2 sLH[kBoundsInLds ? 4 * DQ : 1];   // (lo, hi) per dimension
    __shared__ double2 sNA[NORMP ? 4 * DQ : 1];          // normal priors: (loc, 1/scale)
    __shared__ double sNM[NORMP ? 4 * DQ : 1];           //                -log(scale sqrt(2 pi))
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, c = tid & 3, wave = tid >> 6, lane = tid & 63;
    const int W = s.W, d = a.d, nd = a.n_drag, cps = 1 + nd;
    const int w = blockIdx.x * 64 + (tid >> 2);
    const int g = __builtin_amdgcn_readfirstlane(w / s.group_size);
    const int nsteps = s.n_steps;
    const int Sc = a.chunk_steps;                // dragging steps per LDS chunk
    const int CHUNK = Sc * cps * COLB;           // pairs per chunk
    const double2* __restrict__ gVU = (const double2*)a.VU + (size_t)g * nsteps * cps * COLB;
    double2* const sVU = smem2;
    auto stage = [&](int k) {
        const int first = k * Sc;
        if (first >= nsteps) return;
        const int n = nsteps - first < Sc ? nsteps - first : Sc;
        const int bytes = n * cps * COLB * 16;
        const char* src = (const char*)(gVU + (size_t)first * cps * COLB);
        char* dst = (char*)(sVU + (k & 1) * CHUNK);
        for (int kb = wave; kb * 1024 < bytes; kb += 4) {
            if (kb * 1024 + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
        }
    };
    stage(0);
    const double blo = a.box_lo, bhi = a.box_hi;
    // x0 / y0: the walker's point; cs / ys and ce / ye: start and end point of the dragging step
    double x0[DQ], y0[DQ], cs[DQ], ce[DQ], ys[DQ], ye[DQ];
    double lo[kBoundsInRegs ? DQ : 1], hi[kBoundsInRegs ? DQ : 1];
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        x0[kk] = in ? s.x[(size_t)i * W + w] : (MODE == 0 ? 0.5 * (blo + bhi) : 0.0);
        y0[kk] = in ? a.y[(size_t)i * W + w] : 0.0;
        if (kBoundsInRegs) {
            lo[kk] = a.prior[i];
            hi[kk] = a.prior[dpad + i];
        }
    }
    if (kBoundsInLds)   // (read after the barrier that precedes the step loop)
        for (int i = tid; i < dpad; i += 256) sLH[i] = make_double2(a.prior[i], a.prior[dpad + i]);
    if (NORMP)
        for (int i = tid; i < dpad; i += 256) {
            sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
            sNM[i] = a.prior[4 * dpad + i];
        }
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    long long nacc = s.n_accept[w];
    const long long nacc0 = nacc;
    const uint32_t gid = s.walker0 + (uint32_t)w;
    const double mt10 = s.max_tries * 10.0;
    const int lim1 = s.max_tries < 2.0e9 ? (int)floor(s.max_tries) : 0x7fffffff;
    const int lim10 = mt10 < 2.0e9 ? (int)floor(mt10) : 0x7fffffff;
    const double navg = (double)cps;
    const int hw_slot = hw_wave_slot();

    // log-posterior of the point t (already formed) with residual yt: lp, ll, lt (-inf outside)
    // (every select below takes its condition as a lane mask and is a VOP3 v_cndmask: sel(),
    // det_math.h)
    auto finish = [&](bool inb, double pc, double sc, double& lp, double& ll) -> double {
        const double chi2 = quad_sum(sel(lanes(inb), pc, INFINITY));
        lp = s.uniform_logp + (NORMP ? quad_sum(sc) : 0.0);
        ll = -0.5 * (s.cnorm0 + chi2);
        return sel(lanes(chi2 < INFINITY), lp + ll, -INFINITY);
    };
    // One box for all dimensions and at most two waves per SIMD: the support test is taken on
    // the largest and smallest trial coordinate, as in step_inc_kernel (kBoxMinMax there).
    constexpr bool kBoxMinMax = MODE == 0 && inc_drag_min_waves(DQ, MODE) <= 2;
    struct Support {
        double mx = -INFINITY, mn = INFINITY;
        bool ok = true;
    };
    auto inside = [&](double t, int kk) -> bool {
        if (MODE == 0) return (t <= bhi) & (t >= blo);
        if (kBoundsInLds) {
            const double2 lh = sLH[4 * kk + c];
            return (t <= lh.y) & (t >= lh.x);
        }
        return (t <= hi[kk]) & (t >= lo[kk]);
    };
    auto test = [&](Support& sp, double t, int kk) {
        if (kBoxMinMax) {
            sp.mx = __builtin_fmax(sp.mx, t);
            sp.mn = __builtin_fmin(sp.mn, t);
        } else sp.ok = sp.ok & inside(t, kk);
    };
    auto supported = [&](const Support& sp) -> bool {
        return kBoxMinMax ? ((sp.mx <= bhi) & (sp.mn >= blo)) : sp.ok;
    };
    auto prior_term = [&](double t, int kk, double sc) -> double {
        if (!NORMP) return sc;
        const int i = 4 * kk + c;
        const double2 li = sNA[i];
        const double qq = (t - li.x) * li.y;
        return sc + fma(-0.5 * qq, qq, sNM[i]);
    };
    auto metropolis = [&](double trial, double current, double Ea) -> unsigned long long {
        const double delta = UNIT_T ? (current - trial) : (current - trial) / s.temperature;
        return lanes(trial != -INFINITY) & (lanes(trial > current) | lanes(Ea > delta));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int base = 0, kc = 0; base < nsteps; base += Sc, ++kc) {
        const double2* __restrict__ cur = sVU + (kc & 1) * CHUNK;
        stage(kc + 1);
        const int nhere = nsteps - base < Sc ? nsteps - base : Sc;
#pragma unroll 1
        for (int sl = 0; sl < nhere; ++sl) {
            const unsigned long long step = s.step0 + (unsigned long long)(base + sl);
            rotate_priority<inc_drag_min_waves(DQ, MODE)>(hw_slot);
            const double2* __restrict__ col0 = cur + (size_t)sl * cps * COLB + c;
            double cs_lt = lpost, ce_lt = -INFINITY, ce_lp = 0.0, ce_ll = 0.0;
            double start_acc = 0.0, end_acc = 0.0, Ea0 = 0.0;
            bool dead = false;
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) { cs[kk] = x0[kk]; ys[kk] = y0[kk]; }
#pragma unroll 1
            for (int i4 = 0; i4 < cps; i4 += 4) {
                // lane class c draws the variates of sub-step i4 + c (ONED: those of
                // RandProposer1D where the sub-step's column belongs to a one-parameter block)
                double r4, E4;
                const bool od = ONED && i4 + c < cps &&
                                a.colflag[((size_t)g * nsteps + base + sl) * cps + i4 + c] != 0;
                step_variates(s.key0, s.key1, gid, step, (uint32_t)(i4 + c), od, r4, E4);
                const int nq = cps - i4 < 4 ? cps - i4 : 4;
#pragma unroll 1
                for (int q = 0; q < nq; ++q) {
                    double r, Ea;
                    switch (q) {   // wave-uniform
                    case 0: r = quad_perm<0x00>(r4); Ea = quad_perm<0x00>(E4); break;
                    case 1: r = quad_perm<0x55>(r4); Ea = quad_perm<0x55>(E4); break;
                    case 2: r = quad_perm<0xAA>(r4); Ea = quad_perm<0xAA>(E4); break;
                    default: r = quad_perm<0xFF>(r4); Ea = quad_perm<0xFF>(E4); break;
                    }
                    const int i = i4 + q;
                    const double2* __restrict__ col = col0 + (size_t)i * COLB;
                    if (i == 0) {
                        // the slow proposal: end = x + r0 v_slow (fused, as drag_core)
                        Ea0 = Ea;
                        Support inb;
                        double pc = 0.0, sc = 0.0;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            const double2 p = col[4 * kk];
                            ce[kk] = fma(r, p.x, cs[kk]);
                            ye[kk] = fma(r, p.y, ys[kk]);
                            test(inb, ce[kk], kk);
                            sc = prior_term(ce[kk], kk, sc);
                            pc = fma(ye[kk], ye[kk], pc);
                        }
                        ce_lt = finish(supported(inb), pc, sc, ce_lp, ce_ll);
                        dead = ce_lt == -INFINITY;      // mcmc.py:590-592: only the weight grows
                        start_acc = cs_lt;
                        end_acc = ce_lt;
                    } else {
                        // interpolation step i: both points move by delta = r v_fast (a product,
                        // then a sum -- not fused, as drag_core)
                        Support in_s, in_e;
                        double pcs = 0.0, pce = 0.0, scs = 0.0, sce = 0.0;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            const double2 p = col[4 * kk];
                            const double delta = r * p.x;
                            const double t = cs[kk] + delta, te = ce[kk] + delta;
                            const double yst = fma(r, p.y, ys[kk]), yet = fma(r, p.y, ye[kk]);
                            test(in_s, t, kk);
                            test(in_e, te, kk);
                            scs = prior_term(t, kk, scs);
                            sce = prior_term(te, kk, sce);
                            pcs = fma(yst, yst, pcs);
                            pce = fma(yet, yet, pce);
                        }
                        double ps_lp, ps_ll, pe_lp, pe_ll;
                        const double ps_lt = finish(supported(in_s), pcs, scs, ps_lp, ps_ll);
                        const double pe_lt = finish(supported(in_e), pce, sce, pe_lp, pe_ll);
                        const double frac = (double)i / navg;
                        const double pi = (1.0 - frac) * ps_lt + frac * pe_lt;
                        const double ci = (1.0 - frac) * cs_lt + frac * ce_lt;
                        const unsigned long long acc_m = lanes(ps_lt != -INFINITY) &
                                                         lanes(pe_lt != -INFINITY) &
                                                         metropolis(pi, ci, Ea);
                        const double ra = sel(acc_m, r, 0.0);
                        const lds_pairs col2 = relaunder(col);
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            const pair_t p = col2[4 * kk];
                            const double delta = ra * p.x;     // +-0 when not accepted
                            cs[kk] = cs[kk] + delta;
                            ce[kk] = ce[kk] + delta;
                            ys[kk] = fma(ra, p.y, ys[kk]);
                            ye[kk] = fma(ra, p.y, ye[kk]);
                        }
                        cs_lt = sel(acc_m, ps_lt, cs_lt);
                        ce_lp = sel(acc_m, pe_lp, ce_lp);
                        ce_ll = sel(acc_m, pe_ll, ce_ll);
                        ce_lt = sel(acc_m, pe_lt, ce_lt);
                        start_acc += cs_lt;
                        end_acc += ce_lt;
                    }
                }
            }
            const unsigned long long accept_m =
                ~lanes(dead) & metropolis(end_acc / navg, start_acc / navg, Ea0);
            const bool accept = __builtin_amdgcn_inverse_ballot_w64(accept_m);
            const int lim = burn > 0 ? lim10 : lim1;
            burn -= (accept & (burn > 0)) ? 1 : 0;
            lpri = sel(accept_m, ce_lp, lpri);
            llik = sel(accept_m, ce_ll, llik);
            lpost = sel(accept_m, ce_lt, lpost);
            prej = sel(accept_m, 0, prej);
            wt = sel(accept_m, 1, wt + 1);
            nacc += sel(accept_m, 1, 0);
            if (!accept & !dead & (wt - prej > lim) && c == 0) atomicCAS(s.stuck, 0, 1 + (int)gid);
            // the walker's point: the dragged end point on accept (cs / ys were dragged along and
            // are re-seeded from it at the next step)
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                x0[kk] = sel(accept_m, ce[kk], x0[kk]);
                y0[kk] = sel(accept_m, ye[kk], y0[kk]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        if (i < d) {
            s.x[(size_t)i * W + w] = x0[kk];
            a.y[(size_t)i * W + w] = y0[kk];
        }
    }
    if (c == 0) {
        s.logpost[w] = lpost; s.logprior[w] = lpri; s.loglike[w] = llik;
        s.weight[w] = wt; s.prior_rej[w] = prej; s.burn_left[w] = burn;
        s.n_accept[w] = nacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc - nacc0 : 0);
}

// ---------------------------------------------------------------- y = L^-1 (x - mu)
// 64 walkers per workgroup of four waves; the deviations of the workgroup sit in LDS
// ([i][walker]) and the rows of L^-1 are read at wave-uniform addresses (scalar loads).  One
// ascending fma chain per row from +0.0 (orc_whiten); four rows advance together on every
// deviation read (four independent chains per lane), and the blocks of four rows are dealt to
// the four waves in turn.  Runs once per `refresh_every` steps.
__global__ void __launch_bounds__(256) whiten_state_kernel(const double* __restrict__ x,
                                                           double* __restrict__ y,
                                                           const double* __restrict__ mean,
                                                           const double* __restrict__ Lrow, int d,
                                                           int W, int K)
{
    extern __shared__ __attribute__((aligned(16))) double sdev[];
    // (part is wave-uniform, and said so: the rows of L^-1 it selects are then read by scalar
    // loads and enter the FMAs as scalar operands instead of one vector load per lane)
    const int l = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6),
              w = blockIdx.x * 64 + l;
    const bool live = w < W;
    const int nblk = (d + 3) / 4;
    for (int k = 0; k < K; ++k) {   // y is [K][d][W]
        __syncthreads();            // (the previous mode's deviations have been used)
        if (live)
            for (int i = part; i < d; i += 4) sdev[i * 64 + l] = x[(size_t)i * W + w] - mean[k * d + i];
        __syncthreads();
        if (!live) continue;
        for (int rb = part; rb < nblk; rb += 4) {
            const int j0 = 4 * rb;
            if (j0 + 4 <= d) {
                const double* __restrict__ r0 = Lrow + ((size_t)k * d + j0) * d;
                const double* __restrict__ r1 = r0 + d;
                const double* __restrict__ r2 = r1 + d;
                const double* __restrict__ r3 = r2 + d;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                for (int i = 0; i <= j0; ++i) {
                    const double dv = sdev[i * 64 + l];
                    a0 = fma(r0[i], dv, a0);
                    a1 = fma(r1[i], dv, a1);
                    a2 = fma(r2[i], dv, a2);
                    a3 = fma(r3[i], dv, a3);
                }
                const double d1 = sdev[(j0 + 1) * 64 + l], d2 = sdev[(j0 + 2) * 64 + l],
                             d3 = sdev[(j0 + 3) * 64 + l];
                a1 = fma(r1[j0 + 1], d1, a1);
                a2 = fma(r2[j0 + 1], d1, a2);
                a3 = fma(r3[j0 + 1], d1, a3);
                a2 = fma(r2[j0 + 2], d2, a2);
                a3 = fma(r3[j0 + 2], d2, a3);
                a3 = fma(r3[j0 + 3], d3, a3);
                double* __restrict__ out = y + ((size_t)k * d + j0) * W + w;
                out[0] = a0; out[(size_t)W] = a1; out[2 * (size_t)W] = a2; out[3 * (size_t)W] = a3;
            } else {
                for (int j = j0; j < d; ++j) {
                    const double* __restrict__ row = Lrow + ((size_t)k * d + j) * d;
                    double acc = 0.0;
                    for (int i = 0; i <= j; ++i) acc = fma(row[i], sdev[i * 64 + l], acc);
                    y[((size_t)k * d + j) * W + w] = acc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- (v, u = L^-1 v) per step
// One thread per (group, step of the launch): reads the step's direction column from the basis
// kernels' buffer V, forms u_j = sum_{i<=j} L^-1[j][i] v_i (ascending chain from +0.0,
// orc_whiten_directions) and writes the column in the step kernel's layout
// VU[g][step][kk][c] = (v_{4kk+c}, u_{4kk+c}), zero beyond d.
__global__ void __launch_bounds__(256) whiten_directions_kernel(const IncDirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sv[];   // [d][64]
    // 64 columns x 4 row parts (part is wave-uniform: scalar loads of the rows of L^-1)
    const int l = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sr = blockIdx.x * 64 + l;          // step of the launch
    const int g = blockIdx.y;
    const int d = a.d;
    const bool live = sr < a.n_steps;
    int flag1d = 0;
    if (live) {
        const unsigned long long step = a.step0 + (unsigned long long)sr;
        const int cyc = (int)(step / (unsigned long long)a.cps - a.cycle0);
        const int col = (int)(step % (unsigned long long)a.cps);
        const double* __restrict__ v = a.V + ((size_t)g * a.ncyc + cyc) * a.slab + (size_t)col * a.ld;
        for (int i = part; i < d; i += 4) sv[i * 64 + l] = v[i];
        if (a.vflag) flag1d = a.vflag[((size_t)g * a.ncyc + cyc) * a.cps + col];
    }
    __syncthreads();
    if (!live) return;
    // output column: sr for a plain launch; a dragging launch interleaves the slow column of a
    // step (slot 0) with the n_drag fast columns of its interpolation steps (slots 1 ..)
    const size_t ocol = a.out_div ? (size_t)(sr / a.out_div) * a.out_cols + a.out_slot0 + sr % a.out_div
                                  : (size_t)sr;
    double2* __restrict__ out = (double2*)a.VU + ((size_t)g * a.out_total + ocol) * (4 * a.dq);
    if (a.colflag && part == 0) a.colflag[(size_t)g * a.out_total + ocol] = flag1d;
    // Four rows at a time: four independent chains share every v_i read from LDS (each chain is
    // still one ascending fma chain from +0.0 -- the order of orc_whiten_directions); the row
    // blocks of a column are dealt to the four waves of the workgroup (the long ones last).
    const int nblk = (d + 3) / 4;
    for (int rb = part; rb < nblk; rb += 4) {
        const int j = 4 * rb;
        if (j + 4 <= d) {
            // (the rows of L^-1 through the constant address space: scalar loads, scalar operands)
            const cdoubles r0 = (cdoubles)(unsigned long long)(a.Lrow + (size_t)j * d);
            const cdoubles r1 = r0 + d;
            const cdoubles r2 = r1 + d;
            const cdoubles r3 = r2 + d;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int i = 0; i <= j; ++i) {
                const double v = sv[i * 64 + l];
                a0 = fma(r0[i], v, a0);
                a1 = fma(r1[i], v, a1);
                a2 = fma(r2[i], v, a2);
                a3 = fma(r3[i], v, a3);
            }
            const double v1 = sv[(j + 1) * 64 + l], v2 = sv[(j + 2) * 64 + l], v3 = sv[(j + 3) * 64 + l];
            a1 = fma(r1[j + 1], v1, a1);
            a2 = fma(r2[j + 1], v1, a2);
            a3 = fma(r3[j + 1], v1, a3);
            a2 = fma(r2[j + 2], v2, a2);
            a3 = fma(r3[j + 2], v2, a3);
            a3 = fma(r3[j + 3], v3, a3);
            out[j] = make_double2(sv[j * 64 + l], a0);
            out[j + 1] = make_double2(v1, a1);
            out[j + 2] = make_double2(v2, a2);
            out[j + 3] = make_double2(v3, a3);
        } else {
            for (int jj = j; jj < d; ++jj) {
                const cdoubles row = (cdoubles)(unsigned long long)(a.Lrow + (size_t)jj * d);
                double acc = 0.0;
                for (int i = 0; i <= jj; ++i) acc = fma(row[i], sv[i * 64 + l], acc);
                out[jj] = make_double2(sv[jj * 64 + l], acc);
            }
        }
    }
    if (part == 3)
        for (int j = d; j < 4 * a.dq; ++j) out[j] = make_double2(0.0, 0.0);
    if (a.UU) {
        // |u|^2 in the pattern of every chi2 here (orc_direction_norms): chain p over the rows
        // j = p (mod 4) ascending -- one wave each, the u's read back from the column this
        // workgroup has just written --, then (s0 + s1) + (s2 + s3)
        __shared__ double sq[4][64];
        __syncthreads();   // (only reached when every thread of the workgroup is live, see below)
        double sp = 0.0;
        for (int j = part; j < d; j += 4) {
            const double u = out[j].y;
            sp = fma(u, u, sp);
        }
        sq[part][l] = sp;
        __syncthreads();
        if (part == 0)
            a.UU[(size_t)g * a.out_total + ocol] = (sq[0][l] + sq[1][l]) + (sq[2][l] + sq[3][l]);
    }
    if (a.VW) {
        // the carried log-prior's stream (step_inc_kernel MODE 2; orc_direction_prior):
        // w_j = (v_j / s_j) / s_j -- 1/s_j = 0 where no normal prior is --, and v.w, loc.w as chain
        // p over the rows j = p (mod 4) ascending, one wave each, then (s0 + s1) + (s2 + s3)
        __shared__ double sn[2][4][64];
        const int dpad = 4 * a.dq;
        const cdoubles pr = (cdoubles)(unsigned long long)a.prior;
        double* __restrict__ wout = a.VW + ((size_t)g * a.out_total + ocol) * dpad;
        double nn = 0.0, lw = 0.0;
        for (int j = part; j < dpad; j += 4) {
            double w = 0.0;
            if (j < d) {
                const double inv = pr[3 * dpad + j], v = sv[j * 64 + l];
                w = (v * inv) * inv;
                nn = fma(v, w, nn);
                lw = fma(pr[2 * dpad + j], w, lw);
            }
            wout[j] = w;
        }
        sn[0][part][l] = nn;
        sn[1][part][l] = lw;
        __syncthreads();
        if (part == 0) {
            double2* __restrict__ nl = (double2*)a.NL + ((size_t)g * a.out_total + ocol);
            *nl = make_double2((sn[0][0][l] + sn[0][1][l]) + (sn[0][2][l] + sn[0][3][l]),
                               (sn[1][0][l] + sn[1][1][l]) + (sn[1][2][l] + sn[1][3][l]));
        }
    }
}

// Mixtures: the same per mode, written as PLANES -- VU[g][step][0] = v, [1 + k] = u_k = L_k^-1 v,
// each 4 dq doubles (zero beyond d).
__global__ void __launch_bounds__(64) whiten_directions_mix_kernel(const IncDirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sv[];   // [d][64]
    const int l = threadIdx.x;
    const int sr = blockIdx.x * 64 + l;
    const int g = blockIdx.y;
    const int d = a.d, K = a.n_modes, dpad = 4 * a.dq;
    const bool live = sr < a.n_steps;
    if (live) {
        const unsigned long long step = a.step0 + (unsigned long long)sr;
        const int cyc = (int)(step / (unsigned long long)a.cps - a.cycle0);
        const int col = (int)(step % (unsigned long long)a.cps);
        const double* __restrict__ v = a.V + ((size_t)g * a.ncyc + cyc) * a.slab + (size_t)col * a.ld;
        for (int i = 0; i < d; ++i) sv[i * 64 + l] = v[i];
        // (columns of one-parameter blocks: the step kernel draws RandProposer1D variates there)
        if (a.colflag)
            a.colflag[(size_t)g * a.n_steps + sr] =
                a.vflag ? a.vflag[((size_t)g * a.ncyc + cyc) * a.cps + col] : 0;
    }
    if (!live) return;
    double* __restrict__ out = a.VU + ((size_t)g * a.n_steps + sr) * (size_t)((1 + K) * dpad);
    for (int j = 0; j < dpad; ++j) out[j] = j < d ? sv[j * 64 + l] : 0.0;
    for (int k = 0; k < K; ++k) {
        double* __restrict__ uk = out + (size_t)(1 + k) * dpad;
        for (int j = 0; j < d; ++j) {
            double acc = 0.0;
            if (d <= 32) {   // (wave-uniform)
                // the row through the constant address space, eight terms per batch: wide scalar
                // loads instead of a vector load per element and lane (round 5: 0.150 -> 0.116 ms
                // per launch at d = 30, K = 2).  Small d only: at d = 128 the rows of four modes
                // (0.5 MB) thrash the scalar cache and the same change cost 4x (measured)
                const cdoubles row = (cdoubles)(unsigned long long)(a.Lrow + ((size_t)k * d + j) * d);
                int i = 0;
                for (; i + 8 <= j + 1; i += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fma(row[i + u], sv[(i + u) * 64 + l], acc);
                }
                for (; i <= j; ++i) acc = fma(row[i], sv[i * 64 + l], acc);
            } else {
                const double* __restrict__ row = a.Lrow + ((size_t)k * d + j) * d;
                for (int i = 0; i <= j; ++i) acc = fma(row[i], sv[i * 64 + l], acc);
            }
            uk[j] = acc;
        }
        for (int j = d; j < dpad; ++j) uk[j] = 0.0;
        if (a.UU) {   // |u_k|^2 in the four-chain pattern (orc_direction_norms), rows ascending
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int j = 0; j < d; j += 4) {
                s0 = fma(uk[j], uk[j], s0);
                if (j + 1 < d) s1 = fma(uk[j + 1], uk[j + 1], s1);
                if (j + 2 < d) s2 = fma(uk[j + 2], uk[j + 2], s2);
                if (j + 3 < d) s3 = fma(uk[j + 3], uk[j + 3], s3);
            }
            a.UU[((size_t)g * a.n_steps + sr) * K + k] = (s0 + s1) + (s2 + s3);
        }
    }
}

#if MCMC_DQ_LO <= 16
// ---------------------------------------------------------------- the step kernel, mixtures
// KM = 2..4 modes, DQ <= 16 (d <= 64): one carried residual y_k per mode, the direction planes
// (v, u_1 .. u_KM) of a step read with ds_read_b64, then the log-sum-exp of eval_point
// (gaussian_mixture.py:158-163) -- the exponential of mode k evaluated by lane class k, the
// logarithm by every lane.  Round 5: the log-density a_k = -(c_k + chi2_k) / 2 of every mode is
// CARRIED like step_inc_kernel's log-likelihood (oracle: carries_modes, step_core_inc):
// a_k' = fma(-r / 2, fma(r, |u_k|^2, y_k.u_k + y_k.u_k), a_k) -- one chain over the dimensions per
// mode and trial (y_k.u_k) instead of two (the trial residual and its square), |u_k|^2 formed once
// per (group, step, mode) by whiten_directions_mix_kernel and read with a scalar load; the a_k live
// in a.amode between launches and are re-anchored on y where it is refreshed (a.anchor).
__host__ __device__ constexpr int inc_chunk_mix(int dq, int km)
{
    // (14 KB of planes per chunk: beside the 8.5 KB of staged variates and the logarithm table a
    // workgroup at four waves per SIMD stays within its 40 KB of LDS)
    int c = (1792 / ((1 + km) * 4 * dq)) & ~3;
    return c < 4 ? 4 : (c > 64 ? 64 : c);
}

// five and six modes: up to dq = kIncMixWideDq (d <= 32) -- 7 dq doubles of state per lane
constexpr int kMixWideDq = kIncMixWideDq;
// Round 6 (late): where the state -- dq (km + 1) doubles per lane -- would hold the kernel to ONE wave
// per SIMD but the residuals alone (dq km) leave it two, x lives in LDS ([kk][lane] doubles, read for
// the trial, read and written at the commit; its offset passes through an empty asm at every use, as
// in step_duo_mix_kernel): three modes at d = 49 .. 64, four at d = 41 .. 48
__host__ __device__ constexpr bool inc_mix_x_in_lds(int dq, int km)
{
    return MCMC_EXP_MIX_XLDS(km <= 4 && dq * (km + 1) > 50 && dq * km <= 50);
}
__host__ __device__ constexpr int inc_mix_min_waves(int dq, int km)
{
    // measured (round 5, tools/mix_bench.py d:K over builds held to 2..4 waves, 65 536 walkers; ms per
    // 40 d steps at 2 / 3 / 4 waves): d = 30, K = 2: 3.33 / 3.66 / 5.32; K = 3: 3.93 / 7.36 / 17.3;
    // K = 4: 5.07 / 24.3 / 31.3; d = 16, K = 4: 1.87 / 1.91 / 2.00.  The walker's state is
    // dq (km + 1) + km doubles per lane and the step body (Philox block, two logarithms, the
    // log-sum-exp) wants ~100 registers beside it: above two waves per SIMD it spills, and the
    // kernel is bound by its instruction count (~270 VALU per wave-step at K = 2), not by latency,
    // so occupancy buys nothing once two waves overlap.
    // (round 5 late, after the box test moved to the high words: two modes at three waves 2.81 against
    // 2.91 ms per 1200 steps at d = 30; three and more modes still spill there)
    return MCMC_EXP_WAVES(MIX, dq * (km + 1) <= 12 ? 4 : (km == 2 && dq * (km + 1) <= 24) ? 3
                                                      : (dq * (km + 1) <= 50 || inc_mix_x_in_lds(dq, km)) ? 2 : 1);
}

#ifndef MCMC_MIX_FRESH_EPILOGUE
#define MCMC_MIX_FRESH_EPILOGUE 0
#endif
#ifndef MCMC_MIX_BOX0_LIMIT
#define MCMC_MIX_BOX0_LIMIT 1000   // state doubles per lane below which the box test runs on high words (measured: always)
#endif
#ifndef MCMC_MIX_ORDERED_READS
#define MCMC_MIX_ORDERED_READS 0   // 1: the reads of a step are issued plane by plane (fewer registers)
#endif
template <int DQ, int KM, bool UNIT_T, bool ONED>
__global__ void __launch_bounds__(256, inc_mix_min_waves(DQ, KM))
step_inc_mix_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int dpad = 4 * DQ;
    constexpr int COL = (1 + KM) * dpad;          // doubles per column
    constexpr int C = inc_chunk_mix(DQ, KM);
    constexpr int CHUNK = C * COL;
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, c = tid & 3, wave = tid >> 6, lane = tid & 63;
    const int W = s.W, d = a.d;
    const int w = blockIdx.x * 64 + (tid >> 2);
    const int g = __builtin_amdgcn_readfirstlane(w / s.group_size);
    const int ncols = s.n_steps;
    const double* __restrict__ gVU = a.VU + (size_t)g * ncols * COL;
    auto stage = [&](int k) {
        const int first = k * C;
        if (first >= ncols) return;
        const int cols = ncols - first < C ? ncols - first : C;
        const int bytes = cols * COL * 8;
        const char* src = (const char*)(gVU + (size_t)first * COL);
        char* dst = (char*)(smem + (k & 1) * CHUNK);
        for (int kb = wave; kb * 1024 < bytes; kb += 4) {
            if (kb * 1024 + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
        }
    };
    stage(0);
    // (the bounds sit in LDS as (lo, hi) pairs: registers are for x and the KM residuals)
    __shared__ double2 sLH[4 * DQ];
    __shared__ double2 sNA[4 * DQ];     // normal priors: (loc, 1/scale) and -log(scale sqrt(2 pi))
    __shared__ double sNM[4 * DQ];
    for (int i = tid; i < dpad; i += 256) {
        sLH[i] = make_double2(a.prior[i], a.prior[dpad + i]);
        sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
        sNM[i] = a.prior[4 * dpad + i];
    }
    constexpr bool XLDS = inc_mix_x_in_lds(DQ, KM);
    double x[XLDS ? 1 : DQ], y[KM][DQ];
    __shared__ double sXq[XLDS ? DQ * 256 : 1];
    typedef double __attribute__((address_space(3))) * lds_doubles_rw;
    const unsigned xoff0 = lds_offset(sXq + tid);
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        // (one box for all dimensions: a padded dimension rests at its middle, inside for every step)
        const double xv = in ? s.x[(size_t)i * W + w] : (a.box ? 0.5 * (a.box_lo + a.box_hi) : 0.0);
        if (XLDS) sXq[kk * 256 + tid] = xv;
        else x[XLDS ? 0 : kk] = xv;
#pragma unroll
        for (int k = 0; k < KM; ++k) y[k][kk] = in ? a.y[((size_t)k * d + i) * W + w] : 0.0;
    }
    double cn[KM], wk[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) { cn[k] = s.cblock[a.cnorm_off + k]; wk[k] = s.cblock[a.weight_off + k]; }
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;     // accepted steps of this launch
    const uint32_t gid = s.walker0 + (uint32_t)w;
    const double mt10 = s.max_tries * 10.0;
    const int lim1 = s.max_tries < 2.0e9 ? (int)floor(s.max_tries) : 0x7fffffff;
    const int lim10 = mt10 < 2.0e9 ? (int)floor(mt10) : 0x7fffffff;
    __shared__ dpair_t short_log_lds[SHORT_LOG_TABLE_SIZE];
    const short_log_tab slog = short_log_load(short_log_lds);
    __shared__ double exp64_lds[64];   // 2^(j / 64): the log-sum-exp's table-driven exponential
    const exp_tab etab = exp_tab_load(exp64_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long class1 = lanes(c == 1), class2 = lanes(c == 2), class3 = lanes(c == 3);
    // log-sum-exp of the mode log-densities (mixture_lse of the oracle): one exponential per lane
    // -- lane class k (< KM) takes the one of mode k -- and the weighted sum gathers them by quad
    // broadcasts in the order of the specification
    auto lse = [&](const double (&ak)[KM]) {
        double amax = ak[0];
#pragma unroll
        for (int k = 1; k < KM; ++k) amax = fmax(ak[k], amax);
        double mine = ak[0];
        if (KM > 1) mine = sel(class1, ak[1], mine);
        if (KM > 2) mine = sel(class2, ak[2 < KM ? 2 : 0], mine);
        if (KM > 3) mine = sel(class3, ak[3 < KM ? 3 : 0], mine);
        const double e_mine = dexp_tab(mine - amax, etab);
        double Ssum = fma(wk[0], quad_perm<0x00>(e_mine), 0.0);
        if (KM > 1) Ssum = fma(wk[1], quad_perm<0x55>(e_mine), Ssum);
        if (KM > 2) Ssum = fma(wk[2 < KM ? 2 : 0], quad_perm<0xAA>(e_mine), Ssum);
        if (KM > 3) Ssum = fma(wk[3 < KM ? 3 : 0], quad_perm<0xFF>(e_mine), Ssum);
        if (KM > 4) {   // modes 4 .. 7: a second exponential per lane, lane class k - 4 takes mode k
            double mine2 = ak[4 < KM ? 4 : 0];
            if (KM > 5) mine2 = sel(class1, ak[5 < KM ? 5 : 0], mine2);
            if (KM > 6) mine2 = sel(class2, ak[6 < KM ? 6 : 0], mine2);
            if (KM > 7) mine2 = sel(class3, ak[7 < KM ? 7 : 0], mine2);
            const double e2 = dexp_tab(mine2 - amax, etab);
            Ssum = fma(wk[4 < KM ? 4 : 0], quad_perm<0x00>(e2), Ssum);
            if (KM > 5) Ssum = fma(wk[5 < KM ? 5 : 0], quad_perm<0x55>(e2), Ssum);
            if (KM > 6) Ssum = fma(wk[6 < KM ? 6 : 0], quad_perm<0xAA>(e2), Ssum);
            if (KM > 7) Ssum = fma(wk[7 < KM ? 7 : 0], quad_perm<0xFF>(e2), Ssum);
        }
        return dlog_tab(Ssum, slog) + amax;
    };
    // the carried log-density of every mode (the same value in the four lanes of a walker)
    double am[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) am[k] = a.amode[(size_t)k * W + w];
    if (a.anchor) {   // (wave-uniform) y has just been refreshed from x: orc_anchor_modes
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            double pa = 0.0;
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) pa = fma(y[k][kk], y[k][kk], pa);
            am[k] = -0.5 * (cn[k] + quad_sum(pa));
        }
        llik = lse(am);
        lpost = lpri + llik;
    }
    const cdoubles gUU = (cdoubles)(unsigned long long)(a.UU + (size_t)g * ncols * KM);
    // (the box starts at +0 and ends at a positive finite number: its support test works on the
    // high words of the trial coordinates; the padded dimensions rest at its middle)
    // (measured, 65 536 walkers, d = 30, ms per 1200 steps, extremes -> high words: K = 2: 3.02 -> 2.83,
    // K = 3: 3.72 -> 3.43; K = 4, whose state fills the 256 registers of two waves per SIMD:
    // 4.21 -> 4.6 -- the instantiations with 40 or more state doubles per lane keep the extremes)
    const bool box0 = DQ * (1 + KM) < MCMC_MIX_BOX0_LIMIT && a.box && a.box_lo == 0.0 && a.box_hi > 0.0 &&
                      a.box_hi < INFINITY;
    const unsigned bhi_word = (unsigned)__double2hiint(a.box_hi);
    const int hw_slot = hw_wave_slot();
    bool burning = lanes(burn > 0) != 0ull;   // wave-uniform
    unsigned long long cur_oct = ~0ull;
    __shared__ pair_t sRE[kStagedPairs];   // the (r, Ea) pairs of the current octet (StagedVariates)
    StagedVariates sv;
    sv.init(sRE, wave, lane);

    for (int base = 0, kc = 0; base < ncols; base += C, ++kc) {
        const double* __restrict__ cur = smem + (kc & 1) * CHUNK;
        stage(kc + 1);
        const int cols = ncols - base < C ? ncols - base : C;
        // bit sl: column sl of the chunk belongs to a one-parameter block (the ONED
        // instantiations, chosen when the blocking has such a block: a.colflag)
        unsigned long long oned_cols = 0;
        if (ONED)
            oned_cols = lanes(lane < cols && a.colflag[(size_t)g * ncols + base + lane] != 0);
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl) {
            {
                const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
                if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step (see step_inc_kernel)
                    cur_oct = S >> 3;
                    rotate_priority<inc_mix_min_waves(DQ, KM)>(hw_slot);
                    PairRng pr;
                    pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)c, slog);
                    sv.fill(sRE, wave, lane, c, pr, S);
                }
                double r, Ea;
                if (ONED && ((oned_cols >> sl) & 1ull)) {   // wave-uniform: the un-paired 1-D variates
                    step_variates(s.key0, s.key1, gid, S, 0, true, r, Ea);
                } else
                {
                    sv.fetch(r, Ea);
                }
                sv.next();
                // (round 5: the LDS address of the step's column is carried as a 32-bit offset
                // and passes through empty asms -- every group of reads below then starts where the
                // previous group's results exist, instead of all (1 + KM) DQ operands of a step
                // being fetched up front: the old body wanted 213 registers at KM = 2 and more
                // than 256 at KM = 4, and spilled at the occupancy it was held to)
                unsigned coff = lds_offset(cur + sl * COL + c);
                asm volatile("" : "+v"(coff));
                const lds_doubles col = (lds_doubles)(unsigned long long)coff;
                // (XLDS: a new address for the compiler at every use -- nothing is promoted back to
                // registers, and the reads stay behind the writes of the step before)
                unsigned xo = xoff0;
                if (XLDS) asm volatile("" : "+v"(xo));
                const lds_doubles xs = (lds_doubles)(unsigned long long)xo;
                auto xat = [&](int kk) {
                    if constexpr (XLDS) return xs[kk * 256];
                    else return x[kk];
                };
                unsigned long long inb = ~0ull;   // the support test as a lane mask
                double sc = 0.0;
                double dep;                       // what the next group of reads is ordered behind
                if (box0) {   // wave-uniform: the box [0, hi] for every dimension (step_inc_kernel's
                    // MODE 0): a trial coordinate whose HIGH WORD is below hi's is inside for
                    // certain -- one 32-bit max per dimension; whatever is not certain (within
                    // 2^-20 of hi, beyond it, negative, -0) is decided by the exact comparisons,
                    // a wave-uniform branch that a posterior away from the walls never takes
                    unsigned hmx = 0u;
                    double t = 0.0;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        t = fma(r, col[4 * kk], xat(kk));
                        const unsigned h = (unsigned)__double2hiint(t);
                        hmx = hmx > h ? hmx : h;
                    }
                    dep = t;
                    inb = lanes(hmx < bhi_word);
                    if (inb != lanes(true)) {
                        unsigned xoff = coff;
                        asm volatile("" : "+v"(xoff));
                        const lds_doubles colx = (lds_doubles)(unsigned long long)xoff;
                        inb = ~0ull;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) {
                            const double tx = fma(r, colx[4 * kk], xat(kk));
                            inb &= lanes(tx <= a.box_hi) & lanes(tx >= a.box_lo);
                        }
                    }
                } else if (a.box) {   // wave-uniform: one box for every dimension, no normal priors --
                    // the test is taken on the extremes of the trial (no bounds read from LDS,
                    // no mask arithmetic per dimension; a trial coordinate is never NaN)
                    double tmx = -INFINITY, tmn = INFINITY;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double t = fma(r, col[4 * kk], xat(kk));
                        tmx = __builtin_fmax(tmx, t);
                        tmn = __builtin_fmin(tmn, t);
                    }
                    inb = lanes(tmx <= a.box_hi) & lanes(tmn >= a.box_lo);
                    dep = tmx;
                } else {
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double t = fma(r, col[4 * kk], xat(kk));
                        const double2 lh = sLH[4 * kk + c];
                        inb &= lanes(t <= lh.y) & lanes(t >= lh.x);
                        if (a.has_norm) {   // wave-uniform; branch-free inside (1/scale = 0: no term)
                            const int i = 4 * kk + c;
                            const double2 li = sNA[i];
                            const double qq = (t - li.x) * li.y;
                            sc = sc + fma(-0.5 * qq, qq, sNM[i]);
                        }
                        dep = t;
                    }
                }
                double ak[KM];
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    unsigned uoff = coff + (unsigned)((1 + k) * dpad * 8);
#if MCMC_MIX_ORDERED_READS
                    asm volatile("" : "+v"(uoff) : "v"(dep));
#endif
                    const lds_doubles uk = (lds_doubles)(unsigned long long)uoff;
                    double pc = 0.0;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) pc = fma(y[k][kk], uk[4 * kk], pc);   // y_k . u_k
                    dep = pc;
                    const double yu = quad_sum(pc);
                    const double uu = gUU[(size_t)(base + sl) * KM + k];   // (a scalar load)
                    ak[k] = fma(-0.5 * r, fma(r, uu, yu + yu), am[k]);
                }
                // inside the support = all four lanes of the walker are (outside, what follows is
                // computed and not used)
                const unsigned long long inside_m = quad_all_mask(inb);
                const double lp = s.uniform_logp + (a.has_norm ? quad_sum(sc) : 0.0);
                const double ll = lse(ak);
                const double lt = lp + ll;   // (finite: the sum of the weights' terms is >= w_max)
                const double delta = UNIT_T ? (lpost - lt) : (lpost - lt) / s.temperature;
                const unsigned long long acc_m = inside_m & (lanes(lt > lpost) | lanes(Ea > delta));
                const bool accept = __builtin_amdgcn_inverse_ballot_w64(acc_m);
                int lim = lim1;
                if (burning) {   // wave-uniform (see step_inc_kernel)
                    lim = burn > 0 ? lim10 : lim1;
                    burn -= (accept & (burn > 0)) ? 1 : 0;
                    burning = lanes(burn > 0) != 0ull;
                }
                const double ra = sel(acc_m, r, 0.0);
                // the commit reads the planes AGAIN, one plane at a time (x, then y_1 .. y_KM)
                {
                    unsigned poff = coff;
                    asm volatile("" : "+v"(poff) : "v"(ra));   // (re-read: not kept from the trial)
                    const lds_doubles pv = (lds_doubles)(unsigned long long)poff;
                    if constexpr (XLDS) {
                        unsigned xw = xoff0;
                        asm volatile("" : "+v"(xw) : "v"(ra));
                        const lds_doubles_rw px = (lds_doubles_rw)(unsigned long long)xw;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) px[kk * 256] = fma(ra, pv[4 * kk], px[kk * 256]);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) x[XLDS ? 0 : kk] = fma(ra, pv[4 * kk], x[XLDS ? 0 : kk]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    unsigned poff = coff + (unsigned)((1 + k) * dpad * 8);
#if MCMC_MIX_ORDERED_READS
                    asm volatile("" : "+v"(poff) : "v"(k == 0 ? (XLDS ? ra : x[XLDS ? 0 : DQ - 1]) : y[k > 0 ? k - 1 : 0][DQ - 1]));
#else
                    asm volatile("" : "+v"(poff) : "v"(ra));
#endif
                    const lds_doubles pu = (lds_doubles)(unsigned long long)poff;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) y[k][kk] = fma(ra, pu[4 * kk], y[k][kk]);
                }
#pragma unroll
                for (int k = 0; k < KM; ++k) am[k] = sel(acc_m, ak[k], am[k]);
                lpri = sel(acc_m, lp, lpri);
                llik = sel(acc_m, ll, llik);
                lpost = sel(acc_m, lt, lpost);
                prej = sel(acc_m, 0, prej + sel(inside_m, 0, 1));
                wt = sel(acc_m, 1, wt + 1);
                nacc += sel(acc_m, 1, 0);
                if (wt - prej > lim && c == 0) atomicCAS(s.stuck, 0, 1 + (int)gid);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // (the walker index passes through an empty asm: the addresses of the stores below are then
    // formed HERE -- else the compiler keeps the (1 + KM) DQ + KM + 7 addresses of the prologue's
    // loads alive through the whole step loop, ~60 registers at KM = 2, to reuse them)
#if MCMC_MIX_FRESH_EPILOGUE
    int we = w;
    asm volatile("" : "+v"(we));
#else
    const int we = w;
#endif
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        if (i < d) {
            s.x[(size_t)i * W + we] = XLDS ? sXq[kk * 256 + tid] : x[XLDS ? 0 : kk];
#pragma unroll
            for (int k = 0; k < KM; ++k) a.y[((size_t)k * d + i) * W + we] = y[k][kk];
        }
    }
    if (c == 0) {
#pragma unroll
        for (int k = 0; k < KM; ++k) a.amode[(size_t)k * W + we] = am[k];
        s.logpost[we] = lpost; s.logprior[we] = lpri; s.loglike[we] = llik;
        s.weight[we] = wt; s.prior_rej[we] = prej; s.burn_left[we] = burn;
        s.n_accept[we] = nacc0 + nacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc : 0);
}

template <int DQ, int KM>
hipError_t launch_inc_mix(const IncStepArgs& a, hipStream_t st)
{
    constexpr int C = inc_chunk_mix(DQ, KM);
    const size_t lds = sizeof(double) * 2 * C * (1 + KM) * 4 * DQ;
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    static const kern_t kerns[4] = {
        step_inc_mix_kernel<DQ, KM, false, false>, step_inc_mix_kernel<DQ, KM, true, false>,
        step_inc_mix_kernel<DQ, KM, false, true>, step_inc_mix_kernel<DQ, KM, true, true>};
    const std::string stem = "mcmc::step_inc_mix_kernel<" + std::to_string(DQ) + ", " + std::to_string(KM);
    static const std::string names[4] = {stem + ", false>", stem + ", true>",
                                         stem + ", false, 1-D blocks>", stem + ", true, 1-D blocks>"};
    const int v = (unit_t ? 1 : 0) + (a.colflag ? 2 : 0);
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int DQ>
hipError_t dispatch_inc_mix(const IncStepArgs& a, hipStream_t st)
{
    if constexpr (DQ > MCMC_DQ_HI) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) {
            if constexpr (DQ <= kMixWideDq) {
                if (a.n_modes == 5 || a.n_modes == 6)
                    return a.n_modes == 5 ? launch_inc_mix<DQ, 5>(a, st) : launch_inc_mix<DQ, 6>(a, st);
            }
            return a.n_modes == 2 ? launch_inc_mix<DQ, 2>(a, st)
                 : a.n_modes == 3 ? launch_inc_mix<DQ, 3>(a, st)
                 : a.n_modes == 4 ? launch_inc_mix<DQ, 4>(a, st) : hipErrorInvalidValue;
        }
        return dispatch_inc_mix<DQ + 1>(a, st);
    }
}
#endif  // MCMC_DQ_LO <= 16

#ifdef MCMC_INC_EMIT_TU
// (this translation unit holds the EMIT instantiations alone: incremental_emit.hip)
template <int DQ>
hipError_t launch_inc_dq(const IncStepArgs& a, hipStream_t st)
{
    const int mode = a.has_norm ? 2 : ((a.box && a.box_lo == 0.0) ? 0 : 1);   // MODE 0: [0, hi]
    // (MODE 2: the chunks hold the carried log-prior's stream as well, 24 bytes per dimension)
    const int C = mode == 2 ? inc_chunk(DQ, true) : inc_chunk(DQ);
    const size_t lds = sizeof(double2) * ((mode == 2 ? 3 : 2) * C * 4 * DQ);   // (the bounds: a static array)
    if (mode == 2 && (!a.VW || !a.NL)) return hipErrorInvalidValue;
    if (mode == 2 && DQ >= 29 && (a.anchor & 2)) return hipErrorInvalidValue;   // (no room for the refresh)
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    static const kern_t kerns[6] = {
        step_inc_kernel<DQ, 0, false, false, true>, step_inc_kernel<DQ, 0, true, false, true>,
        step_inc_kernel<DQ, 1, false, false, true>, step_inc_kernel<DQ, 1, true, false, true>,
        step_inc_kernel<DQ, 2, false, false, true>, step_inc_kernel<DQ, 2, true, false, true>};
    static const std::string names[6] = {
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, false, emit>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, true, emit>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, false, emit>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, true, emit>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, false, emit>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, true, emit>"};
    if (a.colflag || !a.s.rows) return hipErrorInvalidValue;
    const int v = 2 * mode + (unit_t ? 1 : 0);
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}
#else
// periodic parameters (step_inc_kernel<.., PER>): MODE 1 or 2
template <int DQ>
hipError_t launch_inc_periodic_dq(const IncStepArgs& a, int np, hipStream_t st)
{
    const int mode = a.has_norm ? 2 : 1;
    const int C = inc_chunk(DQ, mode == 2, true);
    const size_t lds = sizeof(double2) * ((mode == 2 ? 3 : 2) * C * 4 * DQ) + inc_periodic_lds(DQ, np);
    if (np < 1 || np > kMaxPeriodic || a.s.rows || (a.anchor & 2)) return hipErrorInvalidValue;
    if (mode == 2 && (!a.VW || !a.NL)) return hipErrorInvalidValue;
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    static const kern_t kerns[8] = {
        step_inc_kernel<DQ, 1, false, false, false, true>, step_inc_kernel<DQ, 1, true, false, false, true>,
        step_inc_kernel<DQ, 2, false, false, false, true>, step_inc_kernel<DQ, 2, true, false, false, true>,
        step_inc_kernel<DQ, 1, false, true, false, true>, step_inc_kernel<DQ, 1, true, true, false, true>,
        step_inc_kernel<DQ, 2, false, true, false, true>, step_inc_kernel<DQ, 2, true, true, false, true>};
    static const std::string names[8] = {
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, false, periodic>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, true, periodic>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, false, periodic>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, true, periodic>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, false, periodic, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, true, periodic, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, false, periodic, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, true, periodic, 1-D blocks>"};
    const int v = 2 * (mode - 1) + (unit_t ? 1 : 0) + (a.colflag ? 4 : 0);
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int DQ>
hipError_t launch_inc_dq(const IncStepArgs& a, hipStream_t st)
{
    const int mode = a.has_norm ? 2 : ((a.box && a.box_lo == 0.0) ? 0 : 1);   // MODE 0: [0, hi]
    // (MODE 2: the chunks hold the carried log-prior's stream as well, 24 bytes per dimension)
    const int C = mode == 2 ? inc_chunk(DQ, true) : inc_chunk(DQ);
    const size_t lds = sizeof(double2) * ((mode == 2 ? 3 : 2) * C * 4 * DQ);   // (the bounds: a static array)
    if (mode == 2 && (!a.VW || !a.NL)) return hipErrorInvalidValue;
    if (mode == 2 && DQ >= 29 && (a.anchor & 2)) return hipErrorInvalidValue;   // (no room for the refresh)
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    static const kern_t kerns[12] = {
        step_inc_kernel<DQ, 0, false, false>, step_inc_kernel<DQ, 0, true, false>,
        step_inc_kernel<DQ, 1, false, false>, step_inc_kernel<DQ, 1, true, false>,
        step_inc_kernel<DQ, 2, false, false>, step_inc_kernel<DQ, 2, true, false>,
        step_inc_kernel<DQ, 0, false, true>, step_inc_kernel<DQ, 0, true, true>,
        step_inc_kernel<DQ, 1, false, true>, step_inc_kernel<DQ, 1, true, true>,
        step_inc_kernel<DQ, 2, false, true>, step_inc_kernel<DQ, 2, true, true>};
    static const std::string names[12] = {
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, false>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, true>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, false>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, true>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, false>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, true>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, false, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 0, true, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, false, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 1, true, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, false, 1-D blocks>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", 2, true, 1-D blocks>"};
    const int v = 2 * mode + (unit_t ? 1 : 0) + (a.colflag ? 6 : 0);
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}
#endif  // MCMC_INC_EMIT_TU

template <int DQ>
hipError_t launch_drag_dq(const IncStepArgs& a, hipStream_t st)
{
    const int mode = a.has_norm ? 2 : (a.box ? 0 : 1);
    const size_t lds = sizeof(double2) * 2 * (size_t)a.chunk_steps * (1 + a.n_drag) * 4 * DQ;
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    static const kern_t kerns[12] = {
        drag_inc_kernel<DQ, 0, false, false>, drag_inc_kernel<DQ, 0, true, false>,
        drag_inc_kernel<DQ, 1, false, false>, drag_inc_kernel<DQ, 1, true, false>,
        drag_inc_kernel<DQ, 2, false, false>, drag_inc_kernel<DQ, 2, true, false>,
        drag_inc_kernel<DQ, 0, false, true>, drag_inc_kernel<DQ, 0, true, true>,
        drag_inc_kernel<DQ, 1, false, true>, drag_inc_kernel<DQ, 1, true, true>,
        drag_inc_kernel<DQ, 2, false, true>, drag_inc_kernel<DQ, 2, true, true>};
    static const std::string names[12] = {
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 0, false>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 0, true>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 1, false>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 1, true>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 2, false>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 2, true>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 0, false, 1-D blocks>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 0, true, 1-D blocks>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 1, false, 1-D blocks>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 1, true, 1-D blocks>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 2, false, 1-D blocks>",
        "mcmc::drag_inc_kernel<" + std::to_string(DQ) + ", 2, true, 1-D blocks>"};
    const int v = 2 * mode + (unit_t ? 1 : 0) + (a.colflag ? 6 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}

#ifdef MCMC_INC_EMIT_TU
template <int DQ>
hipError_t dispatch_inc(const IncStepArgs& a, hipStream_t st)
{
    if constexpr (DQ > MCMC_DQ_HI) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) return launch_inc_dq<DQ>(a, st);
        return dispatch_inc<DQ + 1>(a, st);
    }
}
#else
template <int DQ>
hipError_t dispatch_inc(const IncStepArgs& a, hipStream_t st)
{
    if constexpr (DQ > MCMC_DQ_HI) {
        return hipErrorInvalidValue;
    } else {
        int np = 0;
        for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
        if (a.dq == DQ && np > 0)
            return a.n_drag > 0 ? hipErrorInvalidValue : launch_inc_periodic_dq<DQ>(a, np, st);
        if (a.dq == DQ) return a.n_drag > 0 ? launch_drag_dq<DQ>(a, st) : launch_inc_dq<DQ>(a, st);
        return dispatch_inc<DQ + 1>(a, st);
    }
}
#endif  // MCMC_INC_EMIT_TU

}  // namespace
}  // namespace mcmc

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)
#ifdef MCMC_INC_EMIT_TU
// the EMIT instantiations of step_inc_kernel, one translation unit per range of DQ
extern "C" hipError_t MCMC_CAT(mcmc_hip_launch_inc_emit_, MCMC_DQ_LO)(const mcmc::IncStepArgs* a,
                                                                    hipStream_t st)
{
    if (a->dq < MCMC_DQ_LO || a->dq > MCMC_DQ_HI || a->n_modes != 1 || a->n_drag > 0)
        return hipErrorInvalidValue;
    return mcmc::dispatch_inc<MCMC_DQ_LO>(*a, st);
}
#else
// one translation unit per range of DQ = ceil(d / 4) (build.py: -DMCMC_DQ_LO=.. -DMCMC_DQ_HI=..)
extern "C" hipError_t MCMC_CAT(mcmc_hip_launch_inc_step_, MCMC_DQ_LO)(const mcmc::IncStepArgs* a,
                                                                    hipStream_t st)
{
    if (a->dq < MCMC_DQ_LO || a->dq > MCMC_DQ_HI) return hipErrorInvalidValue;
#if MCMC_DQ_LO <= 16
    if (a->n_modes > 1) return mcmc::dispatch_inc_mix<MCMC_DQ_LO>(*a, st);
#else
    if (a->n_modes > 1) return hipErrorInvalidValue;
#endif
    return mcmc::dispatch_inc<MCMC_DQ_LO>(*a, st);
}
#endif  // MCMC_INC_EMIT_TU

#if MCMC_DQ_LO == 1 && !defined(MCMC_INC_EMIT_TU)
extern "C" hipError_t mcmc_hip_launch_whiten_state(const double* x, double* y, const double* mean,
                                                   const double* Lrow, int d, int W, int K,
                                                   hipStream_t st)
{
    const size_t lds = sizeof(double) * 64 * (size_t)d;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_state_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mcmc::whiten_state_kernel, dim3((W + 63) / 64), dim3(256), lds, st, x, y,
                       mean, Lrow, d, W, K);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_whiten_directions(const mcmc::IncDirArgs* a, int n_groups,
                                                        hipStream_t st)
{
    const size_t lds = sizeof(double) * 64 * (size_t)a->d;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_directions_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (a->n_modes > 1) {
        if (lds > 40 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_directions_mix_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(mcmc::whiten_directions_mix_kernel,
                           dim3((a->n_steps + 63) / 64, n_groups), dim3(64), lds, st, *a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(mcmc::whiten_directions_kernel, dim3((a->n_steps + 63) / 64, n_groups),
                       dim3(256), lds, st, *a);
    return hipGetLastError();
}
#endif
