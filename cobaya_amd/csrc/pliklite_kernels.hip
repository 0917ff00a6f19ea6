// The binned-bandpower Gaussian likelihood of cobaya/likelihoods/base_classes/planck_pliklite.py
// (PlanckPlikLite.get_chi_squared, 143-155, + functions.chi_squared, functions.py:64-78) for an
// ensemble of walkers on gfx950, and the Metropolis step around it (mcmc.py:545-562, 670-748).
//
//     chi2(x) = delta^T Sigma^-1 delta,   delta_b = X_b - cl_b / A_planck^2,
//     cl_b    = sum_{l in bin b} D_l weights_l           (613 bins at plik_lite_v22's layout)
//
// The 2 n_bins^2 flops of the quadratic form are the evaluation (751 kflop at 613 bins, against
// 15 kflop of binning), and the one place on Cobaya's mcmc path where a batched dense FP64 GEMM is
// the natural kernel: with cov = L L^T,  chi2 = |L^-1 delta|^2  is a TRIANGULAR product
// Y[n_bins x walkers] = L^-1 Delta followed by a column sum of squares -- half the flops of
// Sigma^-1 delta, on v_mfma_f64_16x16x4_f64, which accumulates k in ascending order with one
// rounding per product-sum (tools/probes/mfma_f64_order.hip), i.e. y_j is bit for bit the fma
// chain of the oracle (oracle/mcmc_oracle.c: orc_binned_chi2_of_delta).
//
// A Metropolis step is four launches -- the calibration parameter A_planck makes the target
// non-Gaussian in x, so there is no incremental form and every trial is evaluated from scratch:
//     pl_walker_kernel    accept/reject of the previous trial, variates and trial of this step,
//                         prior (lane per walker)
//     pl_residual_kernel  delta of every trial from the binned response of the linear Cl(theta)
//                         stand-in (lane per walker, wave-uniform operands through the scalar cache)
//     pl_chi2_kernel      the triangular GEMM on the matrix cores (below)
// (the accept of step s and the proposal of step s + 1 share a launch).  The intermediate delta
// crosses HBM once each way (9.8 KB per evaluation against 376 kflop: 38 flop/B, MFMA-bound).
//
// pl_chi2_kernel: a workgroup of 8 waves owns 64 walkers (4 walker tiles of 16 = the N of the
// MFMA); wave q owns the 16-row tiles R of L^-1 with class(R) = q (a snake deal: the cost of a
// tile grows with R) and keeps ALL their accumulators -- 5 tiles x 4 walker tiles x 4 doubles --
// in registers while the k loop runs OUTSIDE: per PAIR of k-steps (2 x 4 columns) it loads, with
// one 16-byte load per lane each, two 16 x 4 tiles of L^-1 per active row tile (global memory,
// 1 KB in A-operand lane order, used for 8 MFMAs) and the two 4 x 16 slices of delta of each of
// its four walker tiles (B operands, shared by the active tiles: used for up to 10 MFMAs each),
// one pair ahead of the MFMAs (one 8-byte load per k-step and operand: 0.448 ms per launch; in
// pairs: 0.391 -- what the loads cost is their NUMBER); the loop is cut into one phase per
// set of active tiles (tile R is finished after k-step 4 R + 3), so an iteration has no branch.  Every tile of L^-1 is read once
// per 64 walkers (1.5 MB from L2), delta once per wave; nothing passes through LDS and there is NO
// barrier: a workgroup takes several sets of 64 walkers in turn (one workgroup per CU for the whole
// launch), its waves run free and leave their partial sums p[q][c] per walker in memory.
#include <type_traits>

#include "det_math.h"
#include "pliklite_args.h"

namespace mcmc {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
#ifndef MCMC_PL_PREFETCH
#define MCMC_PL_PREFETCH 1
#endif
constexpr int kPlPrefetch = MCMC_PL_PREFETCH;   // PAIRS of k-steps the operands of pl_chi2_kernel are fetched ahead
typedef const double __attribute__((address_space(4))) * cptr;
__device__ __forceinline__ cptr as_const(const double* p) { return (cptr)(unsigned long long)p; }

// ------------------------------------------------------------------------------ walkers
// ACCEPT: the Metropolis test of the trial a previous launch proposed and evaluated
// (mcmc.py:670-683) and the bookkeeping of mcmc.py:685-748 -- step_general_kernel's, without
// emitted rows.  PROPOSE: variates of step a.step (un-paired stream), trial t = fma(r, v, x)
// along the group's direction (proposal.py:69, 224), prior support and normal priors
// (prior.py:733-763; one ascending chain, d <= 32).
__device__ __forceinline__ double pl_combine(const double* __restrict__ psum, size_t W, size_t w);

template <bool ACCEPT, bool PROPOSE>
__global__ void __launch_bounds__(64) pl_walker_kernel(const PlWalkerArgs a)
{
    const StepArgs& s = a.s;
    const int d = a.d, W = s.W;
    const int w = blockIdx.x * 64 + threadIdx.x;
    const ConstLayout cl{d, 0};
    const double* __restrict__ C = s.cblock;
    const uint32_t gid = s.walker0 + (uint32_t)w;
    bool accept = false;
    if (ACCEPT) {
        double lpost = s.logpost[w];
        int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
        const double lp = a.lp_t[w], Ea = a.Ea[w];
        const bool inb = lp != -INFINITY;
        const double ll = -0.5 * pl_combine(a.psum_t, (size_t)W, (size_t)w);   // planck_pliklite.py:171
        const double lt = inb ? lp + ll : -INFINITY;
        accept = inb && lt != -INFINITY && (lt > lpost || Ea > (lpost - lt) / s.temperature);
        if (accept) {
            if (burn > 0) --burn;
            s.logprior[w] = lp; s.loglike[w] = ll; s.logpost[w] = lt;
            s.n_accept[w] += 1;
        }
        prej = accept ? 0 : (prej + (inb ? 0 : 1));
        wt = accept ? 1 : wt + 1;
        if (!accept) {
            const double max_now = s.max_tries * (burn > 0 ? 10.0 : 1.0);
            if ((double)(wt - prej) > max_now) atomicCAS(s.stuck, 0, 1 + (int)gid);
        }
        s.weight[w] = wt; s.prior_rej[w] = prej; s.burn_left[w] = burn;
        wave_add_accepts(s.accept_total, accept ? 1 : 0);
    }
    double r = 0.0, Ea = 0.0;
    const double* __restrict__ v = nullptr;
    if (PROPOSE) {
        StepRng rng;
        rng.begin(s.key0, s.key1, gid, s.step0);
        rng.run_all();
        r = rng.r; Ea = rng.Ea;
        const int group = __builtin_amdgcn_readfirstlane(w / s.group_size);
        v = s.V + ((size_t)group * s.ncyc + a.cyc) * (size_t)s.slab + (size_t)a.col * d;
    }
    bool inb = true;
    double sc = 0.0;
    for (int i = 0; i < d; ++i) {
        double xi = s.x[(size_t)i * W + w];
        if (ACCEPT && accept) {
            xi = a.trial[(size_t)i * W + w];
            s.x[(size_t)i * W + w] = xi;
        }
        if (PROPOSE) {
            const double t = fma(r, v[i], xi);
            a.trial[(size_t)i * W + w] = t;
            inb = inb && t <= C[cl.hi() + i] && t >= C[cl.lo() + i];
            if ((s.norm_mask >> i) & 1u) {
                const double q = (t - C[cl.loc() + i]) / C[cl.scale() + i];
                sc = sc + fma(-0.5 * q, q, C[cl.mls() + i]);
            }
        }
    }
    if (PROPOSE) {
        a.lp_t[w] = inb ? s.uniform_logp + sc : -INFINITY;
        a.Ea[w] = Ea;
    }
}

// log-prior of given points t[d][n] (mcmc_hip_evaluate; prior.py:733-763)
__global__ void __launch_bounds__(64) pl_prior_kernel(const double* __restrict__ t, int n, int d,
                                                     const double* __restrict__ C,
                                                     uint32_t norm_mask, double uniform_logp,
                                                     double* __restrict__ lp)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= n) return;
    const ConstLayout cl{d, 0};
    bool inb = true;
    double sc = 0.0;
    for (int i = 0; i < d; ++i) {
        const double ti = t[(size_t)i * n + w];
        inb = inb && ti <= C[cl.hi() + i] && ti >= C[cl.lo() + i];
        if ((norm_mask >> i) & 1u) {
            const double q = (ti - C[cl.loc() + i]) / C[cl.scale() + i];
            sc = sc + fma(-0.5 * q, q, C[cl.mls() + i]);
        }
    }
    lp[w] = inb ? uniform_logp + sc : -INFINITY;
}

// ------------------------------------------------------------------------------ residuals
// delta_b = fma(-cl_b, 1 / A^2, X_b), cl_b = Bc0_b then fma(BJ_bp, theta_p - theta0_p, .) for p
// ascending (oracle: orc_binned_delta).  One lane per walker, 4 waves per 64 walkers, each taking
// a quarter of the k-steps (4 bins); the records (Bc0_b, BJ_b0 .. BJ_b,NLP-1, X_b) are the same
// for every lane: constant address space -> s_load -> SGPR operands of v_fma_f64.  Output in the
// B-operand order of pl_chi2_kernel, two k-steps side by side: delta[wg][kk / 2][wt][16 c + n][kk & 1]
// = residual of bin 4 kk + c for
// walker 64 wg + 16 wt + n (zero for the padding bins).
template <int NLP>   // emulator parameters padded to a multiple of 4 (BJ and theta0 zero beyond n_lin)
__global__ void __launch_bounds__(256) pl_residual_kernel(const PlResidualArgs a)
{
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (wave-uniform: scalar)
    const int wg = blockIdx.x, w = wg * 64 + lane;
    const int W = a.W, calib = a.calib;
    double dth[NLP];
#pragma unroll
    for (int p = 0; p < NLP; ++p) {
        const int i = p + (p >= calib ? 1 : 0);
        dth[p] = p < a.n_lin ? a.trial[(size_t)i * W + w] - a.theta0[p] : 0.0;
    }
    const double A = a.trial[(size_t)calib * W + w];
    const double iA2 = 1.0 / (A * A);
    const int KT2 = a.KT / 2;      // (KT is even: pairs of k-steps, one 16-byte store per pair)
    const int k0 = (KT2 * part) / 4, k1 = (KT2 * (part + 1)) / 4;
    const cptr rec0 = as_const(a.resp);
    double2* __restrict__ out = (double2*)a.delta + (size_t)wg * KT2 * 256 + (lane >> 4) * 64 + (lane & 15);
    constexpr int RL = NLP + 2;
    auto residual = [&](int b) {
        double dl = 0.0;
        if (b < a.n_bins) {            // (wave-uniform)
            const cptr rec = rec0 + (size_t)b * RL;
            double cl = rec[0];
#pragma unroll
            for (int p = 0; p < NLP; ++p) cl = fma(rec[1 + p], dth[p], cl);
            dl = fma(-cl, iA2, rec[1 + NLP]);
        }
        return dl;
    };
    for (int kk2 = k0; kk2 < k1; ++kk2) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            out[(size_t)kk2 * 256 + c * 16] = make_double2(residual(8 * kk2 + c), residual(8 * kk2 + 4 + c));
    }
}

// The same residuals on the matrix cores (the step path): cl = Bc0 + BJ . dtheta is a
// [bins x params] . [params x walkers] product, and v_mfma_f64_16x16x4_f64 accumulates k ascending
// with one rounding per product-sum -- started from Bc0_b, the accumulator IS the fma chain over p
// ascending of the specification (padding parameters contribute fma(0, 0, cl) = cl).  Its layout
// -- register r of lane 16 c + n = bin 16 T + 4 r + c of walker n -- is the B-operand layout of
// k-step 4 T + r of pl_chi2_kernel, so a lane stores its four residuals as the two 16-byte pairs
// (k-steps 4 T, 4 T + 1), (4 T + 2, 4 T + 3).  A workgroup of 4 waves owns 64 walkers; wave q takes
// the bin tiles T = q (mod 4) for all four walker tiles (operands of a tile loaded once, used for
// 4 x 2 NP MFMAs); the scalar-cache walk of pl_residual_kernel (one record per bin, 30 dependent
// s_loads) is gone: 151 -> [measured in profiles/r04_pl_*] us per launch at 613 bins.
template <int NP>   // pairs of k-steps over the emulator parameters: ceil(n_lin / 8)
__global__ void __launch_bounds__(256) pl_residual_mfma_kernel(const PlResidualMfmaArgs a)
{
    const int lane = threadIdx.x & 63, c = lane >> 4, n = lane & 15;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x, W = a.W, calib = a.calib;
    double dth[4][2 * NP], iA2[4];
#pragma unroll
    for (int wt = 0; wt < 4; ++wt) {
        const size_t w = (size_t)wg * 64 + wt * 16 + n;
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) {
            const int p = 4 * j + c, i = p + (p >= calib ? 1 : 0);
            dth[wt][j] = p < a.n_lin ? a.trial[(size_t)i * W + w] - a.theta0[p] : 0.0;
        }
        const double A = a.trial[(size_t)calib * W + w];
        iA2[wt] = 1.0 / (A * A);
    }
    const int KT2 = a.KT / 2;
    const double2* __restrict__ bj = (const double2*)a.bjs + lane;
    const double2* __restrict__ es = (const double2*)a.es + lane;
    double2* __restrict__ out = (double2*)a.delta + (size_t)wg * KT2 * 256 + lane;
    for (int T = q; T < a.n_tiles; T += 4) {
        double2 av[NP], e[4];
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) av[jp] = bj[((size_t)T * NP + jp) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = es[((size_t)T * 4 + i) * 64];
        d4 acc[4];
#pragma unroll
        for (int wt = 0; wt < 4; ++wt) acc[wt] = d4{e[0].x, e[0].y, e[1].x, e[1].y};
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
#pragma unroll
            for (int wt = 0; wt < 4; ++wt)
                acc[wt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[jp].x, dth[wt][2 * jp], acc[wt], 0, 0, 0);
#pragma unroll
            for (int wt = 0; wt < 4; ++wt)
                acc[wt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[jp].y, dth[wt][2 * jp + 1], acc[wt], 0, 0, 0);
        }
        const bool second = 2 * T + 1 < KT2;     // (KT is even, not a multiple of 4: the last tile may end early)
#pragma unroll
        for (int wt = 0; wt < 4; ++wt) {
            const double d0 = fma(-acc[wt][0], iA2[wt], e[2].x), d1 = fma(-acc[wt][1], iA2[wt], e[2].y);
            const double d2 = fma(-acc[wt][2], iA2[wt], e[3].x), d3 = fma(-acc[wt][3], iA2[wt], e[3].y);
            out[((size_t)(2 * T) * 4 + wt) * 64] = make_double2(d0, d1);
            if (second) out[((size_t)(2 * T + 1) * 4 + wt) * 64] = make_double2(d2, d3);
        }
    }
}

// explicit spectra (mcmc_hip_evaluate_binned = get_chi_squared's own arguments): cl_b = fma chain
// over l ascending of D_l weights_l (np.dot, planck_pliklite.py:148-151).  One thread per
// (point, bin); n_pts is small (tests, checks).
__global__ void __launch_bounds__(64) pl_bin_kernel(const PlBinArgs a)
{
    const int pt = blockIdx.y, b = blockIdx.x * 64 + threadIdx.x;
    if (b >= 4 * a.KT) return;
    double dl = 0.0;
    if (b < a.n_bins && pt < a.n_pts) {
        const int tp = a.bins[3 * b], l0 = a.bins[3 * b + 1], l1 = a.bins[3 * b + 2];
        const double* __restrict__ cell = a.cl + ((size_t)pt * 3 + tp) * a.stride;
        double acc = 0.0;
        for (int l = l0; l <= l1; ++l) acc = fma(cell[l - a.L0], a.weights[l], acc);
        const double A = a.A[pt];
        dl = fma(-acc, 1.0 / (A * A), a.X[b]);
    }
    const int wg = pt >> 6, wt = (pt >> 4) & 3, n = pt & 15;
    a.delta[((((size_t)wg * (a.KT / 2) + (b >> 3)) * 4 + wt) * 64 + (b & 3) * 16 + n) * 2 + ((b >> 2) & 1)] = dl;
}

// ------------------------------------------------------------------------------ chi2
// See the header.  NTW = row tiles per wave (ceil(ceil(n_bins / 16) / 8), <= 5).
template <int NTW>
__global__ void __launch_bounds__(512, 2) pl_chi2_kernel(const PlChi2Args a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nk[NTW];
    const double2* __restrict__ ap[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        nk[t] = a.nk[wave][t];          // ascending in t; absent tiles first (0)
        ap[t] = (const double2*)(a.Astream + a.tile_off[wave][t]) + lane;
    }
    // a workgroup takes `batches` consecutive sets of 64 walkers; its eight waves run FREE -- no
    // barrier anywhere: every wave leaves the 4 partial sums p[q][c] of its rows per walker in
    // psum[q][c][walker] and the walker kernel (or the host) combines the 32 of a walker in the
    // specified order.  With one workgroup per CU for the whole launch the only idle time left is
    // the difference between the waves' totals (1.4 %), not a ramp and a tail per 64 walkers.
    for (int bt = 0; bt < a.batches; ++bt) {
    const int wg = blockIdx.x * a.batches + bt;
    if (wg >= a.n_sets) break;
    // (operands of TWO k-steps per 16-byte load: delta and the tiles of L^-1 hold the k-steps 2 m
    // and 2 m + 1 of a lane side by side -- half the load instructions of one per k-step)
    const double2* __restrict__ dl = (const double2*)a.delta + (size_t)wg * (a.KT / 2) * 256 + lane;
    d4 acc[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int wt = 0; wt < 4; ++wt) acc[t][wt] = d4{0.0, 0.0, 0.0, 0.0};
    // Operands are fetched PF pairs of k-steps ahead of their MFMAs.  The streams and delta are
    // padded, so the fetches past the end are harmless.
    constexpr int PF = kPlPrefetch;
    double2 av[PF][NTW], bv[PF][4];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
#pragma unroll
        for (int wt = 0; wt < 4; ++wt) bv[j][wt] = dl[((size_t)j * 4 + wt) * 64];
#pragma unroll
        for (int t = 0; t < NTW; ++t) av[j][t] = ap[t][(size_t)j * 64];
    }
    int kk = 0;   // pair of k-steps
    // phase P: the k-steps on which the tiles t >= P are active (tile t ends at k-step nk[t], even,
    // and nk ascends): a fixed set of loads and MFMAs per iteration, no branch inside
#pragma unroll
    for (int P = 0; P < NTW; ++P) {
        for (; 2 * kk < nk[P]; ++kk) {
            double2 an[NTW], bn[4];
#pragma unroll
            for (int wt = 0; wt < 4; ++wt) bn[wt] = dl[((size_t)(kk + PF) * 4 + wt) * 64];
#pragma unroll
            for (int t = P; t < NTW; ++t) an[t] = ap[t][(size_t)(kk + PF) * 64];
#pragma unroll
            for (int t = P; t < NTW; ++t)
#pragma unroll
                for (int wt = 0; wt < 4; ++wt)
                    acc[t][wt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][t].x, bv[0][wt].x, acc[t][wt], 0, 0, 0);
#pragma unroll
            for (int t = P; t < NTW; ++t)
#pragma unroll
                for (int wt = 0; wt < 4; ++wt)
                    acc[t][wt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][t].y, bv[0][wt].y, acc[t][wt], 0, 0, 0);
#pragma unroll
            for (int j = 0; j + 1 < PF; ++j) {
#pragma unroll
                for (int t = P; t < NTW; ++t) av[j][t] = av[j + 1][t];
#pragma unroll
                for (int wt = 0; wt < 4; ++wt) bv[j][wt] = bv[j + 1][wt];
            }
#pragma unroll
            for (int t = P; t < NTW; ++t) av[PF - 1][t] = an[t];
#pragma unroll
            for (int wt = 0; wt < 4; ++wt) bv[PF - 1][wt] = bn[wt];
        }
    }
    // lane 16 c + n holds the rows 16 R + 4 r + c of walker n (tile R, register r): the chain
    // p[q][c] runs over the wave's tiles in ascending R, r = 0..3
    const int c = lane >> 4, n = lane & 15;
#pragma unroll
    for (int wt = 0; wt < 4; ++wt) {
        double p = 0.0;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (nk[t] > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p = fma(acc[t][wt][r], acc[t][wt][r], p);
            }
        a.psum[(size_t)(wave * 4 + c) * a.n_walkers + (size_t)wg * 64 + wt * 16 + n] = p;
    }
    }   // batches
}

// ------------------------------------------------------------------------------ fused
// pl_fused_kernel: pl_residual_mfma_kernel and pl_chi2_kernel in one launch -- delta is produced
// chunk by chunk into LDS (128 bins x 64 walkers = 64 KB, double-buffered) in the B-operand order
// and consumed from there; it never crosses HBM (round 3: 643 MB of round trip per launch and a
// kernel of its own), and every wave reads it from LDS instead of fetching it from L2 on its own.
//
// The barrier per chunk this needs would expose the imbalance of the snake deal (a wave's tiles
// are not equally far along at a given k), so the rows are dealt differently: the 16-row tiles
// are grouped in eights FROM THE LAST TILE DOWN (virtual tile index = real + shift, shift =
// (8 - NT mod 8) mod 8, so that every group is complete but possibly the first, cheapest one),
// group G's diagonal block falls into chunk G, and wave q owns in every group the tile at
// position s = min(q, 7 - q) for two of the four walker tiles and the tile at position 7 - s for
// the other two (q < 4: walker tiles {0, 1} | {2, 3}; q >= 4: {2, 3} | {0, 1}): in every chunk
// every wave has the same work -- 2 (s + 1) + 2 (8 - s) = 18 pair-iterations of two walker tiles in
// the diagonal group, 16 of four walker tiles in each group above --, so the barrier costs
// nothing but its latency.  The partial sums are the chains p[pos][c] over the groups ascending
// (oracle: binned_class(R) = (R + shift) mod 8).  The tiles of L^-1 stream from L2 as before,
// one 16-byte load per half-tile and pair, re-issued right after their use (one pair ahead; two
// from the second chunk on, in the registers of the finished group).
//
// Round 5 (tools/pl_clocks.py: the shader clock of every wave at the phase boundaries of a set;
// profiles/r05_pl_fused_experiments.txt): the kernel lost 22 % of its cycles beside the MFMAs --
// 0.497 ms per launch -- and most of that in the producers:
//   * the gathers of dtheta sat in the arm of a conditional: eight branches, eight round trips to
//     memory in a row per producer call.  Now requested together, ONCE per set (with its first
//     chunk), and kept in the producing wave's corner of LDS (20 KB beside the 128 KB of residuals);
//   * (X_b) was requested behind the MFMAs (the compiler had sunk the loads into the conditional
//     store), one more round trip per pair of tiles; (Bc0, X) are read as 4 x 64 B per tile;
//   * the older wave of a SIMD produced BEFORE its consumption and the younger behind: the older
//     is through a chunk first anyway, so the younger produced beside nothing.  Now the other way
//     round (see the kernel);
//   * the first chunk of a set was produced with every SIMD idle: it is now produced beside the
//     last chunk of the set before (the two LDS buffers alternate across sets);
//   * the operand streams were addressed as (stream base + lane offset) kept in 20 VGPRs; the lane
//     offset is now laundered per pair-iteration, the position goes into the scalar base.
// What the register allocator makes of all this decides as much as the design: tools/check_pl_spills.py
// (and tests/test_host_logic.py) hold the build to MFMA loops without scratch traffic.
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// a 16-byte load at (wave-uniform base) + (per-lane 32-bit byte offset): global_load_dwordx4 with an
// SGPR base -- the address costs one VGPR for all the streams of the kernel
__device__ __forceinline__ double2 ld16(const void* ubase, unsigned voff)
{
    return *(const double2*)((const char*)ubase + voff);
}

// Experiment hooks (tools/exp_pl_variants.sh); the shipped values are the defaults.
#ifndef PL_DEEP_FROM
#define PL_DEEP_FROM 1         // operands of L^-1 two pairs ahead of their MFMAs from this chunk on (one before)
#endif
#ifndef PL_PRODUCER_PRIO
#define PL_PRODUCER_PRIO 1     // producers at a raised priority
#endif
#ifndef PL_OLDER_FIRST
#define PL_OLDER_FIRST 1       // the waves q < 4 at a higher priority than the waves q >= 4 throughout
#endif
#ifndef PL_ES_COMPACT
#define PL_ES_COMPACT 1        // (Bc0, X): the 16 lanes of a class read the same 16 bytes
#endif
#ifndef PL_CROSS_SET
#define PL_CROSS_SET 1         // the first chunk of the NEXT set is produced beside the last chunk of this one
#endif
// (timing experiment, tools/pl_clocks.py: the shader clock of every wave at the phase boundaries of
// its SECOND set of walkers; not compiled into libmcmc_hip.so)
#ifdef PL_DEBUG_CLOCKS
__device__ unsigned long long pl_clock_log[256 * 8 * 64];
#define PL_STAMP(i)                                                                         \
    do {                                                                                    \
        if (bt == 1 && blockIdx.x < 256) {                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
            if (lane == 0) pl_clock_log[((size_t)blockIdx.x * 8 + q) * 64 + (i)] = t_;      \
        }                                                                                   \
    } while (0)
#else
#define PL_STAMP(i) ((void)0)
#endif
#ifdef PL_DEBUG_NO_BARRIER
#define PL_BARRIER() do {} while (0)   // (timing experiment: races)
#else
#define PL_BARRIER() __syncthreads()
#endif
template <int NG, int NP>
__global__ void __launch_bounds__(512, 2) pl_fused_kernel(const PlFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double2 lds[];   // [2][16][4][64] + [4][NP + 1][64]
    const int tid = threadIdx.x, lane = tid & 63, c = lane >> 4, n = lane & 15;
    const unsigned l16 = (unsigned)lane * 16u;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s_pos = q < 4 ? q : 7 - q;
    const int ws = q < 4 ? 0 : 2, wl = 2 - ws;          // walker tiles of the short / long half
    const int sh = a.shift, W = a.W, calib = a.calib;
    constexpr int CP = kPlChunkPairs;
    int nS[NG], nL[NG];                                  // real pairs of the half-tiles (0: absent)
    const char* pS[NG];                                  // their streams (wave-uniform)
    const char* pL[NG];
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        nS[G] = max(a.a_pairs[q][G][0] - 2 * sh, 0); nL[G] = max(a.a_pairs[q][G][1] - 2 * sh, 0);
        pS[G] = (const char*)(a.Astream + a.a_off[q][G][0]);
        pL[G] = (const char*)(a.Astream + a.a_off[q][G][1]);
    }
    // The two waves of a SIMD are q and q + 4.  Each produces half of the next chunk's residuals
    // for the walker tile q & 3: the wave q >= 4 BEFORE it consumes the current chunk, the wave
    // q < 4 behind its consumption -- a producer (a chain of latencies with 32 MFMAs in it) then
    // runs beside a consumer (MFMA-bound) on its SIMD, and the wave that starts a chunk late is the
    // one that ends it late.  The waves q < 4 are served first throughout, by priority and not
    // only by age.  (Round 4 had it the other way round -- q < 4 before, q >= 4 behind: the older
    // wave, through its consumption first anyway, left the younger one to produce beside nothing,
    // 8 k clocks per chunk; tools/pl_clocks.py.)
    if (PL_OLDER_FIRST && q < 4) __builtin_amdgcn_s_setprio(1);
    int par = 0;                                         // LDS buffer of this set's first chunk
    for (int bt = 0; bt < a.batches; ++bt) {
        const int wg = blockIdx.x * a.batches + bt;
        if (wg >= a.n_sets) break;                       // (uniform over the workgroup)
        // ---- producer: the residuals of four bin tiles of chunk M (virtual tiles 8 M + 4 (q / 4) .. + 3)
        // of set `wg` for walker tile q & 3 -> LDS buffer `slot` in B-operand order (pl_residual_mfma_kernel)
        auto produce = [&](int wg, auto M, int slot) {
#ifdef PL_DEBUG_SKIP_PRODUCE
            return;       // (timing experiment: the residuals are garbage)
#endif
            constexpr int m = decltype(M)::value, NT = 2, NTL = 4;   // NT of the wave's NTL tiles in flight together
            const int t0 = q < 4 ? 0 : 4;
            // a producer is a chain of latencies with a few MFMAs in it; at a raised priority those
            // do not queue behind the 8-40 MFMAs per iteration of the consumer it runs beside
            if (PL_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(3);
            // (the lane index goes through an empty asm: everything derived from it below is then
            // recomputed here instead of being hoisted out of the chunk loop, where it would sit in
            // registers through the MFMA loops)
            unsigned lw = l16;
            asm volatile("; producer lane" : "+v"(lw));
            const int wq = q & 3;    // the walker tile
            double2* dst = (double2*)((char*)lds + ((size_t)slot * CP * 256 + wq * 64) * 16 + lw);
            // dtheta_p = theta_p - theta0_p of the lane's walker (p = 4 j + c) and 1 / A^2: gathered
            // with the set's first chunk -- ALL the requests in flight together (round 4 had them
            // inside the `p < n_lin` arm of a conditional: eight branches, eight round trips in a
            // row, most of a producer's time) -- and left in the wave's own corner of LDS for the
            // other chunks of the set (a wave reads what it wrote itself: no barrier)
            double2* const keep = (double2*)((char*)lds + kPlFusedChunkBytes + (size_t)wq * (NP + 1) * 1024 + lw);
            double dth[2 * NP];
            double A = 1.0, iA2 = 0.0;
            double traw[2 * NP], th0[2 * NP];
            if (m == 0) {
                const int cc = (int)(lw >> 8), nn = (int)((lw >> 4) & 15u);
                const unsigned w8 = ((unsigned)wg * 64u + (unsigned)wq * 16u + (unsigned)nn) * 8u;
#pragma unroll
                for (int j = 0; j < 2 * NP; ++j) {
                    const int p = min(4 * j + cc, a.n_lin - 1), i = p + (p >= calib ? 1 : 0);
#ifdef PL_DEBUG_NO_DTH
                    traw[j] = (double)(i + (int)w8);
#else
                    traw[j] = *(const double*)((const char*)a.trial + ((unsigned)i * (unsigned)W * 8u + w8));
#endif
                    th0[j] = a.theta0[p];
                }
                A = *(const double*)((const char*)a.trial + ((unsigned)calib * (unsigned)W * 8u + w8));
            } else {
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
                    const double2 v = keep[jp * 64];
                    dth[2 * jp] = v.x;
                    dth[2 * jp + 1] = v.y;
                }
                iA2 = keep[NP * 64].x;
            }
            // (Bc0, X) of a bin tile do not depend on the walker: the 16 lanes of a class read the same
            // 16 bytes (the n = 0 entries) -- 4 x 64 B per tile through the L1 instead of 4 KB
            const unsigned le = PL_ES_COMPACT ? (lw & ~255u) : lw;
            // NT tiles at a time.  Registers are what a producer is short of (it runs between the
            // MFMA loops, with the accumulators of the unfinished groups alive): Bc0 is requested
            // straight into the accumulators, the operands of BJ two k-step pairs ahead of their
            // MFMAs into the registers those just released, X behind the last but one pair --
            // 16 + 24 NT registers instead of 16 + (16 + 4 NP) NT
#pragma unroll
            for (int j = 0; j < NTL; j += NT) {
                const char* bjT[NT];
                const char* esT[NT];
                double2 av[NT][2], X[NT][2];
                d4 y[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int T = 8 * m + t0 + j + u - sh;        // real bin tile
                    const int Tc = T >= 0 && T < a.n_tiles ? T : 0;   // (uniform; an absent tile's rows are never read)
                    bjT[u] = (const char*)(a.bjs + (size_t)Tc * NP * 128);
                    esT[u] = (const char*)(a.es + (size_t)Tc * 4 * 128);
                    const double2 e0 = ld16(esT[u], le), e1 = ld16(esT[u] + 1024, le);
                    y[u] = d4{e0.x, e0.y, e1.x, e1.y};
                    av[u][0] = ld16(bjT[u], lw);
                    if (NP > 1) av[u][1] = ld16(bjT[u] + 1024, lw);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (m == 0 && j == 0) {     // (behind the first tiles' requests: one round trip for both)
#pragma unroll
                    for (int jj = 0; jj < 2 * NP; ++jj) {
                        const int cc = (int)(lw >> 8);
                        dth[jj] = 4 * jj + cc < a.n_lin ? traw[jj] - th0[jj] : 0.0;
                    }
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp)
                        if (q < 4) keep[jp * 64] = make_double2(dth[2 * jp], dth[2 * jp + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        y[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][jp & 1].x, dth[2 * jp], y[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        y[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][jp & 1].y, dth[2 * jp + 1], y[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (jp + 2 < NP) {
#pragma unroll
                        for (int u = 0; u < NT; ++u) av[u][jp & 1] = ld16(bjT[u] + (jp + 2) * 1024, lw);
                    }
                    if (jp == (NP >= 2 ? NP - 2 : 0)) {
#pragma unroll
                        for (int u = 0; u < NT; ++u) {
                            X[u][0] = ld16(esT[u] + 2048, le);
                            X[u][1] = ld16(esT[u] + 3072, le);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m == 0 && j == 0) {     // (1 / A^2 behind the first MFMAs: its division waits for A there, not in front of the requests)
                    double A2 = A * A;
                    asm volatile("; calibration" : "+v"(A2));
                    iA2 = 1.0 / A2;
                    if (q < 4) keep[NP * 64] = make_double2(iA2, 0.0);
                }
                // (the rows of an absent tile -- below the first or beyond the last bin -- are written
                // too: nobody reads them, and a conditional store would have the compiler sink the
                // requests of X behind the MFMAs, a round trip of their own)
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int tl = t0 + j + u;                // tile of the chunk: pairs 2 tl, 2 tl + 1
                    dst[(size_t)(2 * tl) * 256] = make_double2(fma(-y[u][0], iA2, X[u][0].x), fma(-y[u][1], iA2, X[u][0].y));
                    dst[(size_t)(2 * tl + 1) * 256] = make_double2(fma(-y[u][2], iA2, X[u][1].x), fma(-y[u][3], iA2, X[u][1].y));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PL_PRODUCER_PRIO) {
                if (PL_OLDER_FIRST && q < 4) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        };
        // ---- consumer state
        d4 acc[NG][4];
        double2 aS[2][NG], aL[2][NG];   // operands of the next pair(s) of every half-tile ([1]: from chunk PL_DEEP_FROM on)
#pragma unroll
        for (int G = 0; G < NG; ++G) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[G][j] = d4{0.0, 0.0, 0.0, 0.0};
            aS[0][G] = ld16(pS[G], l16);  // real pair 0 (absent half-tiles point at zeros)
            aL[0][G] = ld16(pL[G], l16);
        }
        double pch[4] = {0.0, 0.0, 0.0, 0.0};   // the chains p[pos][c] of the wave's four columns
        PL_STAMP(0);
        if (!PL_CROSS_SET || bt == 0) {
            produce(wg, std::integral_constant<int, 0>{}, par);
            PL_STAMP(1);
            PL_BARRIER();
        }
        PL_STAMP(2);
        static_for<NG>([&](auto M_) {
            constexpr int m = decltype(M_)::value;
            // what is produced beside chunk m: the next chunk of this set, or (PL_CROSS_SET) the
            // first chunk of the next set -- into the buffer chunk m does not read
            const bool nxt_set = m + 1 == NG;
            const bool prod = !nxt_set || (PL_CROSS_SET && bt + 1 < a.batches && wg + 1 < a.n_sets);   // (uniform)
            const int p_wg = nxt_set ? wg + 1 : wg, p_slot = (par + m + 1) & 1;
            using PM = std::integral_constant<int, (m + 1 == NG ? 0 : m + 1)>;
            if (prod && q >= 4) produce(p_wg, PM{}, p_slot);
            PL_STAMP(3 + 6 * m);
            // (as in the producer: the lane offset is laundered per chunk, so that the LDS and
            // stream addresses of the five unrolled chunks are not all computed up front and
            // kept in registers for the whole set)
            unsigned lm = l16;
            asm volatile("; chunk lane" : "+v"(lm));
            const char* const buf = (const char*)lds + (size_t)((par + m) & 1) * CP * 4096 + lm;
            // pairs of this chunk: virtual [16 m, 16 m + 16); real = virtual - 2 shift
            const int base = CP * m - 2 * sh;            // real pair of i = 0
            const int i0 = m == 0 ? 2 * sh : 0;
            // (wave-uniform, and said so: as a lane value a loop bound was spilled and re-read per
            // iteration behind a full s_waitcnt -- every request drained before the next iteration)
            const int cS = __builtin_amdgcn_readfirstlane(min(max(nS[m] - base, 0), CP));
            const int cL = __builtin_amdgcn_readfirstlane(min(max(nL[m] - base, 0), CP));
            // one pair-iteration: B operands from LDS, for every active half-tile two MFMAs per
            // walker tile, then the operands of its next pair
            // B operands of pair i0; from then on each pair's operands are re-read for the NEXT pair
            // right behind their last use (the short half-tiles' behind the short MFMAs, ...), so
            // that the LDS latency hides behind the other half's MFMAs
            double2 b0 = *(const double2*)(buf + (size_t)i0 * 4096 + ws * 1024);
            double2 b1 = *(const double2*)(buf + (size_t)i0 * 4096 + ws * 1024 + 1024);
            double2 b2 = *(const double2*)(buf + (size_t)i0 * 4096 + wl * 1024);
            double2 b3 = *(const double2*)(buf + (size_t)i0 * 4096 + wl * 1024 + 1024);
            // From chunk PL_DEEP_FROM on a finished group's registers hold a SECOND pair of operands:
            // pair i's are re-requested for pair i + 2 (slot i & 1; all the loop bounds are even).  A
            // wave that has its SIMD to itself -- the other producing, or through with the chunk --
            // has only its own MFMAs to cover a request: 4 to 16 of them one pair ahead.
            constexpr int DEPTH = m >= PL_DEEP_FROM ? 2 : 1;
            if (m == PL_DEEP_FROM) {
#pragma unroll
                for (int G = m; G < NG; ++G) {
                    aS[1][G] = ld16(pS[G] + (size_t)min(base + i0 + 1, max(nS[G] - 1, 0)) * 1024, lm);
                    aL[1][G] = ld16(pL[G] + (size_t)min(base + i0 + 1, max(nL[G] - 1, 0)) * 1024, lm);
                }
            }
            auto step = [&](int i, auto DS, auto DL, auto K) {
                constexpr int k = decltype(K)::value;
                // (the lane offset is taken anew in every pair-iteration: were it loop-invariant, the
                // compiler would keep (stream base + lane offset) of all ten streams in registers --
                // 20 VGPRs -- instead of adding the stream's position to the base in scalar registers)
                unsigned li = lm;
                asm volatile("; pair lane" : "+v"(li));
                const int Pn = base + i + DEPTH;         // the real pair requested behind this one's MFMAs
                const char* const bn = buf + (size_t)min(i + 1, CP - 1) * 4096;
                // first k-step of every active short half-tile, then the second (a dependent MFMA is
                // then 2 x groups apart from its predecessor, as in pl_chi2_kernel), each followed by
                // the reload of its operands
#pragma unroll
                for (int G = m; G < NG; ++G)
                    if (G > m || decltype(DS)::value) {
                        acc[G][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[k][G].x, b0.x, acc[G][0], 0, 0, 0);
                        acc[G][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[k][G].x, b1.x, acc[G][1], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int G = m; G < NG; ++G)
                    if (G > m || decltype(DS)::value) {
                        acc[G][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[k][G].y, b0.y, acc[G][0], 0, 0, 0);
                        acc[G][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[k][G].y, b1.y, acc[G][1], 0, 0, 0);
                        aS[k][G] = ld16(pS[G] + (size_t)min(Pn, max(nS[G] - 1, 0)) * 1024, li);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                b0 = *(const double2*)(bn + ws * 1024);
                b1 = *(const double2*)(bn + ws * 1024 + 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int G = m; G < NG; ++G)
                    if (G > m || decltype(DL)::value) {
                        acc[G][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[k][G].x, b2.x, acc[G][2], 0, 0, 0);
                        acc[G][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[k][G].x, b3.x, acc[G][3], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int G = m; G < NG; ++G)
                    if (G > m || decltype(DL)::value) {
                        acc[G][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[k][G].y, b2.y, acc[G][2], 0, 0, 0);
                        acc[G][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[k][G].y, b3.y, acc[G][3], 0, 0, 0);
                        aL[k][G] = ld16(pL[G] + (size_t)min(Pn, max(nL[G] - 1, 0)) * 1024, li);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                b2 = *(const double2*)(bn + wl * 1024);
                b3 = *(const double2*)(bn + wl * 1024 + 1024);
                __builtin_amdgcn_sched_barrier(0);
            };
            using T1 = std::integral_constant<bool, true>;
            using T0 = std::integral_constant<bool, false>;
            // (cS <= cL always: the long half-tile sits at the higher position of its group, and
            // the absent virtual tiles are the lowest positions of group 0)
            int i = i0;
#ifdef PL_DEBUG_SKIP_CONSUME
            i = CP;       // (timing experiment: no triangular product)
#endif
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            auto run = [&](int upto, auto DS, auto DL) {
                if constexpr (DEPTH == 1) {
#pragma unroll 1
                    for (; i < upto; ++i) step(i, DS, DL, K0{});
                } else {
#pragma unroll 1
                    for (; i < upto; i += 2) {
                        step(i, DS, DL, K0{});
                        step(i + 1, DS, DL, K1{});
                    }
                }
            };
            run(cS, T1{}, T1{});
            PL_STAMP(4 + 6 * m);
            run(cL, T0{}, T1{});
            PL_STAMP(5 + 6 * m);
            if (m + 1 < NG) run(CP, T0{}, T0{});
            PL_STAMP(6 + 6 * m);
            // group m is complete (its diagonal block lies in this chunk): its rows join the
            // chains -- groups ascending, r = 0..3 -- and its 32 accumulator registers are free
            // for the producer of the next chunk
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) pch[j] = fma(acc[m][j][r], acc[m][j][r], pch[j]);
            if (prod && q < 4) produce(p_wg, PM{}, p_slot);
            PL_STAMP(7 + 6 * m);
            PL_BARRIER();
            PL_STAMP(8 + 6 * m);
        });
        par = (par + NG) & 1;
        unsigned le = l16;
        asm volatile("; epilogue lane" : "+v"(le));
        const unsigned ce = le >> 8, ne = (le >> 4) & 15u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = j < 2 ? s_pos : 7 - s_pos;
            const int wt = (j < 2 ? ws : wl) + (j & 1);
            a.psum[(size_t)(pos * 4 + ce) * a.W + (size_t)wg * 64 + wt * 16 + ne] = pch[j];
        }
    }
}

// chi2 of a walker from the 32 partial sums pl_chi2_kernel left (oracle: orc_binned_chi2_of_delta)
__device__ __forceinline__ double pl_combine(const double* __restrict__ psum, size_t W, size_t w)
{
    double sq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        sq[q] = (psum[(size_t)(4 * q + 0) * W + w] + psum[(size_t)(4 * q + 1) * W + w]) +
                (psum[(size_t)(4 * q + 2) * W + w] + psum[(size_t)(4 * q + 3) * W + w]);
    return ((sq[0] + sq[1]) + (sq[2] + sq[3])) + ((sq[4] + sq[5]) + (sq[6] + sq[7]));
}

__global__ void __launch_bounds__(64) pl_combine_kernel(const double* __restrict__ psum,
                                                       double* __restrict__ chi2, int n)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w < n) chi2[w] = pl_combine(psum, (size_t)n, (size_t)w);
}

}  // namespace
}  // namespace mcmc

using namespace mcmc;

extern "C" hipError_t mcmc_hip_launch_pl_walker(const PlWalkerArgs* a, int accept, int propose,
                                                hipStream_t st)
{
    const dim3 g(a->s.W / 64), b(64);
    if (accept && propose) hipLaunchKernelGGL((pl_walker_kernel<true, true>), g, b, 0, st, *a);
    else if (accept) hipLaunchKernelGGL((pl_walker_kernel<true, false>), g, b, 0, st, *a);
    else hipLaunchKernelGGL((pl_walker_kernel<false, true>), g, b, 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_pl_prior(const double* t, int n, int d, const double* C,
                                               uint32_t norm_mask, double uniform_logp, double* lp,
                                               hipStream_t st)
{
    hipLaunchKernelGGL(pl_prior_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t, n, d, C, norm_mask,
                       uniform_logp, lp);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_pl_residual(const PlResidualArgs* a, hipStream_t st)
{
    const dim3 g(a->W / 64), b(256);
    switch (a->nlp) {
    case 4: hipLaunchKernelGGL(pl_residual_kernel<4>, g, b, 0, st, *a); break;
    case 8: hipLaunchKernelGGL(pl_residual_kernel<8>, g, b, 0, st, *a); break;
    case 12: hipLaunchKernelGGL(pl_residual_kernel<12>, g, b, 0, st, *a); break;
    case 16: hipLaunchKernelGGL(pl_residual_kernel<16>, g, b, 0, st, *a); break;
    case 20: hipLaunchKernelGGL(pl_residual_kernel<20>, g, b, 0, st, *a); break;
    case 24: hipLaunchKernelGGL(pl_residual_kernel<24>, g, b, 0, st, *a); break;
    case 28: hipLaunchKernelGGL(pl_residual_kernel<28>, g, b, 0, st, *a); break;
    case 32: hipLaunchKernelGGL(pl_residual_kernel<32>, g, b, 0, st, *a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_pl_residual_mfma(const PlResidualMfmaArgs* a, hipStream_t st)
{
    const dim3 g(a->W / 64), b(256);
    switch (a->np) {
    case 1: hipLaunchKernelGGL(pl_residual_mfma_kernel<1>, g, b, 0, st, *a); break;
    case 2: hipLaunchKernelGGL(pl_residual_mfma_kernel<2>, g, b, 0, st, *a); break;
    case 3: hipLaunchKernelGGL(pl_residual_mfma_kernel<3>, g, b, 0, st, *a); break;
    case 4: hipLaunchKernelGGL(pl_residual_mfma_kernel<4>, g, b, 0, st, *a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_pl_bin(const PlBinArgs* a, hipStream_t st)
{
    const int padded = (a->n_pts + 63) & ~63;
    hipLaunchKernelGGL(pl_bin_kernel, dim3((4 * a->KT + 63) / 64, padded), dim3(64), 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_pl_chi2(const PlChi2Args* a, hipStream_t st)
{
    // one workgroup per CU for the whole launch where the ensemble is large enough
    PlChi2Args b = *a;
    b.batches = (a->n_sets + 255) / 256;
    const dim3 g((a->n_sets + b.batches - 1) / b.batches), blk(512);
    switch (a->ntw) {
    case 1: hipLaunchKernelGGL(pl_chi2_kernel<1>, g, blk, 0, st, b); break;
    case 2: hipLaunchKernelGGL(pl_chi2_kernel<2>, g, blk, 0, st, b); break;
    case 3: hipLaunchKernelGGL(pl_chi2_kernel<3>, g, blk, 0, st, b); break;
    case 4: hipLaunchKernelGGL(pl_chi2_kernel<4>, g, blk, 0, st, b); break;
    case 5: hipLaunchKernelGGL(pl_chi2_kernel<5>, g, blk, 0, st, b); break;
    default: return hipErrorInvalidValue;
    }
    static const char* const names[5] = {"mcmc::pl_chi2_kernel<1>", "mcmc::pl_chi2_kernel<2>",
                                         "mcmc::pl_chi2_kernel<3>", "mcmc::pl_chi2_kernel<4>",
                                         "mcmc::pl_chi2_kernel<5>"};
    mcmc_hip_note_step_kernel(names[a->ntw - 1]);
    return hipGetLastError();
}

template <int NG>
static hipError_t launch_pl_fused_ng(const PlFusedArgs& b, dim3 g, hipStream_t st)
{
    constexpr size_t lds = kPlFusedLdsBytes;   // 128 KB of residuals + 20 KB
    auto go = [&](auto kern) -> hipError_t {
        {   // per device and per call: the attribute is the current device's
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, g, dim3(512), lds, st, b);
        return hipGetLastError();
    };
    switch (b.np) {
    case 1: return go(pl_fused_kernel<NG, 1>);
    case 2: return go(pl_fused_kernel<NG, 2>);
    case 3: return go(pl_fused_kernel<NG, 3>);
    case 4: return go(pl_fused_kernel<NG, 4>);
    default: return hipErrorInvalidValue;
    }
}

extern "C" hipError_t mcmc_hip_launch_pl_fused(const PlFusedArgs* a, hipStream_t st)
{
    PlFusedArgs b = *a;
    b.batches = (a->n_sets + 255) / 256;     // one workgroup per CU for the whole launch
    const dim3 g((a->n_sets + b.batches - 1) / b.batches);
    static const char* const names[5] = {"mcmc::pl_fused_kernel<1>", "mcmc::pl_fused_kernel<2>",
                                         "mcmc::pl_fused_kernel<3>", "mcmc::pl_fused_kernel<4>",
                                         "mcmc::pl_fused_kernel<5>"};
    hipError_t e;
    switch (a->ng) {
    case 1: e = launch_pl_fused_ng<1>(b, g, st); break;
    case 2: e = launch_pl_fused_ng<2>(b, g, st); break;
    case 3: e = launch_pl_fused_ng<3>(b, g, st); break;
    case 4: e = launch_pl_fused_ng<4>(b, g, st); break;
    case 5: e = launch_pl_fused_ng<5>(b, g, st); break;
    default: return hipErrorInvalidValue;
    }
    if (e == hipSuccess) mcmc_hip_note_step_kernel(names[a->ng - 1]);
    return e;
}

#ifdef PL_DEBUG_CLOCKS
extern "C" __attribute__((visibility("default"))) int mcmc_hip_debug_pl_clocks(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mcmc::pl_clock_log), sizeof(unsigned long long) * 256 * 8 * 64);
}
#endif

extern "C" hipError_t mcmc_hip_launch_pl_combine(const double* psum, double* chi2, int n, hipStream_t st)
{
    hipLaunchKernelGGL(pl_combine_kernel, dim3((n + 63) / 64), dim3(64), 0, st, psum, chi2, n);
    return hipGetLastError();
}
