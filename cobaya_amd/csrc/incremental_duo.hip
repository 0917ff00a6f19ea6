// mcmc_hip -- INCREMENTAL EVALUATION of mixtures, TWO lanes per walker (round 6; gfx950 only).
//
// The same step as step_inc_mix_kernel (incremental_kernels.hip) -- the same arithmetic bit for bit,
// specification: oracle/mcmc_oracle.c step_core_inc with `carries_modes` -- in another layout.
// step_inc_mix_kernel gives a walker FOUR lanes (lane class c holds the dimensions i = 4 kk + c): 16
// walkers per wave, and everything that is not a sum over the dimensions -- the log-sum-exp
// (gaussian_mixture.py:158-163) with its exponential and logarithm, the accept test, the selects of
// the carried values, the bookkeeping of weight / rejections / burn-in, the fetch of the variates --
// is executed once per LANE, four times per walker: ~120 of the ~175 vector instructions of a
// wave-step at d = 30, K = 2.  Its state (dq (K + 1) doubles per lane beside a ~100-register body)
// holds it to three waves per SIMD -- 65 536 walkers are four: a second round of a quarter of the
// workgroups at one wave per SIMD --, and the SIMDs wait 38 % of their cycles
// (profiles/r05_variant_counters.txt).
// Here a walker has TWO lanes: lane half h = lane & 1 holds the dimensions i = 4 kk + 2 h + j,
// j = 0, 1 -- two of the specification's four interleaved chains --, a wave serves 32 walkers, and the
// per-lane work is done twice per walker.  The sums over the dimensions stay the specification's
// (p0 + p1) + (p2 + p3): a lane adds its own two chains, one DPP swap adds the neighbour's.  The
// planes (v, u_1 .. u_K) of a column give a lane the operands of both its chains in ONE ds_read_b128.
// 65 536 walkers are 2 048 waves of this layout: two per SIMD in one round, 256 registers each --
// which hold x and y_1, y_2 of a two-mode walker at d <= 32; with three modes above d = 24 and four above d = 20 x
// moves to LDS (XLDS below) and the registers hold y_1 .. y_K: two modes up to d = 48, three up to d = 32,
// four up to d = 24 (kernels.h: duo_serves, duo_x_in_lds).
//
// Measured (same box, 65 536 walkers, tools/mix_bench.py; step kernel ms per 40 d steps, four lanes ->
// two; profiles/r06_duo.txt): d = 30: K = 2 2.907 -> 1.855 (2.71 -> 4.24e10 evals/s), K = 3 3.568 -> 2.815;
// d = 24: K = 2 1.978 -> 1.250, K = 3 2.605 -> 1.527, K = 4 2.882 -> 2.222; d = 16: K = 3 1.359 -> 1.054,
// K = 4 1.563 -> 1.019; two modes above d = 32 (x in LDS): d = 36 3.66 -> 3.04, d = 40 4.38 -> 3.47, d = 48 5.66 -> 4.99.
// Four modes at d = 30 do not fit (y_1 .. y_4 alone are 128 registers: with x in
// registers as well the step loop spilled, 4.07 -> 19.9 ms) and stay on step_inc_mix_kernel.  One mode
// gains nothing (0.893 -> 0.889 ms at d = 30: its per-lane work is small, the kernel is bound by its
// FP64 fmas in either layout) and was not kept.
//
// Served: Metropolis steps, no periodic parameter, no emitted rows, no block of one parameter, whole
// workgroups of 128 walkers inside one basis group -- for ensembles that fill the chip with it
// (capi.hip: kDuoMinWalkers); smaller ensembles keep the four-lane kernel, whose twice as many
// waves cover their latencies better.
#include <string>

#include "incremental_common.h"

#ifndef MCMC_DUO_DEPK
#define MCMC_DUO_DEPK(tuned) (tuned)   // (experiment hooks: _exp/inc_experiment.h)
#endif
#ifndef MCMC_DUO_KEEPV
#define MCMC_DUO_KEEPV(tuned) (tuned)
#endif

namespace mcmc {
namespace {

// the neighbour lane's value (lane ^ 1)
__device__ __forceinline__ double pair_swap(double v) { return quad_perm<0xB1>(v); }
// (p0 + p1) + (p2 + p3): s = this lane's two chains added, the neighbour holds the other two
__device__ __forceinline__ double duo_sum(double p_even, double p_odd)
{
    const double s = p_even + p_odd;
    return s + pair_swap(s);
}
__device__ __forceinline__ unsigned pair_max_u32(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    return v > o ? v : o;
}
// lane mask -> the mask of the lanes whose PAIR is held completely
__device__ __forceinline__ unsigned long long pair_all_mask(unsigned long long m)
{
    m &= m >> 1;
    m &= 0x5555555555555555ull;
    return m * 3ull;
}

// The (r, Ea) variates of an octet of steps (StagedVariates of incremental_common.h for 32 walkers
// per wave): lane half h draws the Philox blocks of the step pairs 4 * octet + 2 h and + 2 h + 1
// and leaves their four pairs in sRE[wave][step of the octet][walker of the wave].  Row stride 33
// pairs = 528 bytes: the eight lanes a ds_write_b128 serves together (four walkers x two halves)
// then cover the 128-byte bank window once (the halves' rows are 4 x 528 = 64 bytes apart mod 128).
constexpr int kDuoRow = 33;
constexpr int kDuoStagedPairs = 4 * 8 * kDuoRow;   // 16.5 KB per workgroup of four waves
struct DuoVariates {
    unsigned base, off;
    __device__ __forceinline__ void init(const pair_t* sRE, int wave, int lane)
    {
        base = lds_offset(sRE + (wave * 8 * kDuoRow + (lane >> 1)));
        off = base;
    }
    // the two pairs of the step pair 4 * octet + 2 h + q (q = 0, 1), as soon as they are drawn
    __device__ __forceinline__ void put(pair_t* sRE, int wave, int lane, int h, int q, const PairRng& p)
    {
        pair_t* const mine = sRE + ((wave * 8 + 4 * h + 2 * q) * kDuoRow + (lane >> 1));
        mine[0] = pair_t{p.r[0], p.Ea[0]};
        mine[kDuoRow] = pair_t{p.r[1], p.Ea[1]};
    }
    __device__ __forceinline__ void seek(unsigned long long S) { off = base + (unsigned)(S & 7ull) * (16u * kDuoRow); }
    __device__ __forceinline__ void fetch(double& r, double& Ea) const
    {
        const pair_t re = *(lds_pairs)(unsigned long long)off;
        r = re.x;
        Ea = re.y;
    }
    __device__ __forceinline__ void next() { off += 16u * kDuoRow; }
};

// ---------------------------------------------------------------- mixtures
// The log-sum-exp (gaussian_mixture.py:158-163; mixture_lse of the oracle): lane half h takes the
// exponential of mode h and of mode h + 2, pair broadcasts hand them to the neighbour, the weighted sum runs in
// the order of the specification.
// columns of one LDS chunk (a multiple of 4): two workgroups per CU have 80 KB each -- what the staged
// variates (16.5 KB), the tables (2.5 KB), the bounds (2 KB) and, where x lives in LDS (duo_x_in_lds,
// kernels.h), its 128 dq bytes per walker leave, for two buffers of planes
__host__ __device__ constexpr int duo_chunk_mix(int dq, int km)
{
    const int avail = 80 * 1024 - 16896 - 2560 - 2048 - (duo_x_in_lds(km, dq) ? dq * 4096 : 0);
    int c = ((avail / 16) / ((1 + km) * 4 * dq)) & ~3;
    return c < 4 ? 4 : (c > 64 ? 64 : c);
}
// the v plane of a step kept in registers from the trial to the commit where the registers hold it
// beside the state: up to 64 doubles of state + plane per lane (measured at d = 30, K = 2: 1.93 -> 1.83 ms
// per 1200 steps)
__host__ __device__ constexpr bool duo_keep_v(int dq, int km)
{
    return 2 * dq * (km + (duo_x_in_lds(km, dq) ? 0 : 1)) + 2 * dq <= 64;
}

template <int DQ, int KM, bool UNIT_T, bool BOX0>
__global__ void __launch_bounds__(256, 2) step_duo_mix_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int dpad = 4 * DQ;
    constexpr int NE = 2 * DQ;                    // dimensions per lane: e = 2 kk + j <-> i = 4 kk + 2 h + j
    constexpr int COL = (1 + KM) * dpad;          // doubles per column
    constexpr int C = duo_chunk_mix(DQ, KM);
    constexpr int CHUNK = C * COL;
    // the element of a plane whose result the next plane's reads wait for (measured at d = 30, K = 2:
    // first, middle and last element within 1 %)
    constexpr int DEPK = MCMC_DUO_DEPK(DQ / 2);
    // XLDS (two modes from d = 33 on, three from d = 25 on, four from d = 21 on): x lives in LDS, [kk][lane] pairs
    // (x_even, x_odd) -- 16 reads and 8 writes of 16 bytes per lane and step --, so that the registers
    // hold y_1 .. y_KM as they hold x, y_1, y_2 of a two-mode walker (with x in registers too three
    // modes spilled ~45 doubles inside the step loop: 7.6 ms per 1200 steps at d = 30 against
    // step_inc_mix_kernel's 3.5; with x in LDS 2.8)
    constexpr bool XLDS = duo_x_in_lds(KM, DQ);
    constexpr bool KEEPV = MCMC_DUO_KEEPV(duo_keep_v(DQ, KM));
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, h = tid & 1, wave = tid >> 6, lane = tid & 63;
    const int W = s.W, d = a.d;
    const int w = blockIdx.x * 128 + (tid >> 1);
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
    // BOX0: every prior uniform on the same [0, hi]; else per-parameter bounds and normal priors
    __shared__ double2 sLH[BOX0 ? 1 : 4 * DQ];     // the bounds as (lo, hi) pairs
    __shared__ double2 sNA[BOX0 ? 1 : 4 * DQ];     // normal priors: (loc, 1/scale) and -log(scale sqrt(2 pi))
    __shared__ double sNM[BOX0 ? 1 : 4 * DQ];
    if (!BOX0)
        for (int i = tid; i < dpad; i += 256) {
            sLH[i] = make_double2(a.prior[i], a.prior[dpad + i]);
            sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
            sNM[i] = a.prior[4 * dpad + i];
        }
    double x[XLDS ? 1 : NE], y[KM][NE];
    typedef pair_t __attribute__((address_space(3))) * lds_pairs_rw;
    __shared__ pair_t sX[XLDS ? DQ * 256 : 1];
    const unsigned xoff0 = lds_offset(sX + tid);
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        double xe[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = 2 * kk + j, i = 4 * kk + 2 * h + j;
            const bool in = i < d;
            // (one box for all dimensions: a padded dimension rests at its middle, inside for every step)
            xe[j] = in ? s.x[(size_t)i * W + w] : (BOX0 ? 0.5 * a.box_hi : 0.0);
            if (!XLDS) x[XLDS ? 0 : e] = xe[j];
#pragma unroll
            for (int k = 0; k < KM; ++k) y[k][e] = in ? a.y[((size_t)k * d + i) * W + w] : 0.0;
        }
        if (XLDS) sX[kk * 256 + tid] = pair_t{xe[0], xe[1]};
    }
    double cn[KM], wk[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) { cn[k] = s.cblock[a.cnorm_off + k]; wk[k] = s.cblock[a.weight_off + k]; }
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;
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
    const unsigned long long half1 = 0xAAAAAAAAAAAAAAAAull;   // the lanes with h = 1
    auto lse = [&](const double (&ak)[KM]) {
        double amax = ak[0];
#pragma unroll
        for (int k = 1; k < KM; ++k) amax = fmax(ak[k], amax);
        const double mine = sel(half1, ak[1], ak[0]);
        const double e_mine = dexp_tab(mine - amax, etab);
        double Ssum = fma(wk[0], quad_perm<0xA0>(e_mine), 0.0);    // [0,0,2,2]: the pair's h = 0 lane
        Ssum = fma(wk[1], quad_perm<0xF5>(e_mine), Ssum);          // [1,1,3,3]: its h = 1 lane
        if (KM > 2) {   // modes 2, 3: a second exponential per lane
            const double mine2 = KM > 3 ? sel(half1, ak[3 < KM ? 3 : 0], ak[2 < KM ? 2 : 0]) : ak[2 < KM ? 2 : 0];
            const double e2 = dexp_tab(mine2 - amax, etab);
            Ssum = fma(wk[2 < KM ? 2 : 0], quad_perm<0xA0>(e2), Ssum);
            if (KM > 3) Ssum = fma(wk[3 < KM ? 3 : 0], quad_perm<0xF5>(e2), Ssum);
        }
        return dlog_tab(Ssum, slog) + amax;
    };
    // the carried log-density of every mode (the same value in both lanes of a walker)
    double am[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) am[k] = a.amode[(size_t)k * W + w];
    if (a.anchor) {   // (wave-uniform) y has just been refreshed from x: orc_anchor_modes
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            double pa0 = 0.0, pa1 = 0.0;
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                pa0 = fma(y[k][2 * kk], y[k][2 * kk], pa0);
                pa1 = fma(y[k][2 * kk + 1], y[k][2 * kk + 1], pa1);
            }
            am[k] = -0.5 * (cn[k] + duo_sum(pa0, pa1));
        }
        llik = lse(am);
        lpost = lpri + llik;
    }
    const cdoubles gUU = (cdoubles)(unsigned long long)(a.UU + (size_t)g * ncols * KM);
    // (the box [0, hi]: the support test on the high words of the trial coordinates, as in
    // step_inc_mix_kernel)
    const unsigned bhi_word = (unsigned)__double2hiint(a.box_hi);
    const int hw_slot = hw_wave_slot();
    bool burning = lanes(burn > 0) != 0ull;   // wave-uniform
    unsigned long long cur_oct = ~0ull;
    __shared__ pair_t sRE[kDuoStagedPairs];
    DuoVariates sv;
    sv.init(sRE, wave, lane);

    for (int base = 0, kc = 0; base < ncols; base += C, ++kc) {
        const double* __restrict__ cur = smem + (kc & 1) * CHUNK;
        stage(kc + 1);
        const int cols = __builtin_amdgcn_readfirstlane(ncols - base < C ? ncols - base : C);
        unsigned coff = lds_offset(cur + 2 * h);
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl, coff += COL * 8) {
            const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
            if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step
                cur_oct = S >> 3;
                rotate_priority<2>(hw_slot);
#pragma unroll 1
                for (int q = 0; q < 2; ++q) {   // (rolled: one Philox block's registers at a time)
                    PairRng pr;
                    pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)(2 * h + q), slog);
                    sv.put(sRE, wave, lane, h, q, pr);
                }
                sv.seek(S);
            }
            double r, Ea;
            sv.fetch(r, Ea);
            sv.next();
            const lds_pairs col = (lds_pairs)(unsigned long long)coff;   // plane v: [2 kk] = (v_even, v_odd)
            unsigned long long inb = ~0ull;
            double sc0 = 0.0, sc1 = 0.0;
            double dep;   // what the next plane's reads are ordered behind
            pair_t vk[KEEPV ? DQ : 1];
            // (XLDS: the offset of the walker's x passes through an empty asm at every use -- an
            // address the compiler takes as new: nothing is promoted back to registers, and the
            // reads stay behind the writes of the step before)
            unsigned xo = xoff0;
            if (XLDS) asm volatile("" : "+v"(xo));
            const lds_pairs xs = (lds_pairs)(unsigned long long)xo;
            auto xpair = [&](int kk) {
                if constexpr (XLDS) return xs[kk * 256];
                else return pair_t{x[2 * kk], x[2 * kk + 1]};
            };
            if constexpr (BOX0) {
                unsigned hmx = 0u;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const pair_t v = col[2 * kk];
                    const pair_t xp = xpair(kk);
                    if (KEEPV) vk[kk] = v;
                    if (kk == DEPK) dep = fma(r, v.x, xp.x);
                    const unsigned h0 = (unsigned)__double2hiint(fma(r, v.x, xp.x));
                    const unsigned h1 = (unsigned)__double2hiint(fma(r, v.y, xp.y));
                    hmx = hmx > h0 ? hmx : h0;
                    hmx = hmx > h1 ? hmx : h1;
                }
                inb = lanes(pair_max_u32(hmx) < bhi_word);
                if (inb != lanes(true)) {   // (wave-uniform, rare) the exact comparisons
                    unsigned xoff = coff;
                    asm volatile("" : "+v"(xoff));
                    const lds_pairs colx = (lds_pairs)(unsigned long long)xoff;
                    inb = ~0ull;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const pair_t v = colx[2 * kk];
                        const pair_t xp = xpair(kk);
                        const double t0 = fma(r, v.x, xp.x), t1 = fma(r, v.y, xp.y);
                        inb &= lanes(t0 <= a.box_hi) & lanes(t0 >= 0.0);
                        inb &= lanes(t1 <= a.box_hi) & lanes(t1 >= 0.0);
                    }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const pair_t v = col[2 * kk];
                    const pair_t xp = xpair(kk);
                    if (KEEPV) vk[kk] = v;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int i = 4 * kk + 2 * h + j;
                        const double t = fma(r, j ? v.y : v.x, j ? xp.y : xp.x);
                        if (kk == DEPK && j == 0) dep = t;
                        const double2 lh = sLH[i];
                        inb &= lanes(t <= lh.y) & lanes(t >= lh.x);
                        if (a.has_norm) {   // wave-uniform; branch-free inside (1/scale = 0: no term)
                            const double2 li = sNA[i];
                            const double qq = (t - li.x) * li.y;
                            if (j) sc1 = sc1 + fma(-0.5 * qq, qq, sNM[i]);
                            else sc0 = sc0 + fma(-0.5 * qq, qq, sNM[i]);
                        }
                    }
                }
            }
            // (the reads of a plane are ordered behind the MIDDLE result of the plane before -- half
            // of its arithmetic covers these reads' latency --: issued all up
            // front, the (1 + KM) NE operands of a step do not fit beside the (1 + KM) NE of state)
            double ak[KM];
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                unsigned uoff = coff + (unsigned)((1 + k) * dpad * 8);
                asm volatile("" : "+v"(uoff) : "v"(dep));
                const lds_pairs uk = (lds_pairs)(unsigned long long)uoff;
                double pc0 = 0.0, pc1 = 0.0;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {   // y_k . u_k
                    const pair_t u = uk[2 * kk];
                    pc0 = fma(y[k][2 * kk], u.x, pc0);
                    pc1 = fma(y[k][2 * kk + 1], u.y, pc1);
                    if (kk == DEPK) dep = pc0;
                }
                const double yu = duo_sum(pc0, pc1);
                const double uu = gUU[(size_t)(base + sl) * KM + k];   // (a scalar load)
                ak[k] = fma(-0.5 * r, fma(r, uu, yu + yu), am[k]);
            }
            const unsigned long long inside_m = pair_all_mask(inb);
            const double lp = s.uniform_logp + ((!BOX0 && a.has_norm) ? duo_sum(sc0, sc1) : 0.0);
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
            // the commit reads the planes AGAIN (x, then y_1 .. y_KM): the pointer passes through
            // an empty asm behind the accept decision, nothing is kept from the trial
            if constexpr (XLDS) {
                unsigned xw = xoff0, poff = coff;
                asm volatile("" : "+v"(xw), "+v"(poff) : "v"(ra));
                const lds_pairs_rw px = (lds_pairs_rw)(unsigned long long)xw;
                const lds_pairs pv = (lds_pairs)(unsigned long long)poff;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    pair_t xp = px[kk * 256];
                    pair_t v;
                    if constexpr (KEEPV) v = vk[kk];
                    else v = pv[2 * kk];
                    xp.x = fma(ra, v.x, xp.x);
                    xp.y = fma(ra, v.y, xp.y);
                    px[kk * 256] = xp;
                }
            } else if constexpr (KEEPV) {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    x[2 * kk] = fma(ra, vk[kk].x, x[2 * kk]);
                    x[2 * kk + 1] = fma(ra, vk[kk].y, x[2 * kk + 1]);
                }
            } else {
                unsigned poff = coff;
                asm volatile("" : "+v"(poff) : "v"(ra));
                const lds_pairs pv = (lds_pairs)(unsigned long long)poff;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const pair_t v = pv[2 * kk];
                    x[2 * kk] = fma(ra, v.x, x[2 * kk]);
                    x[2 * kk + 1] = fma(ra, v.y, x[2 * kk + 1]);
                }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                unsigned poff = coff + (unsigned)((1 + k) * dpad * 8);
                // (behind the middle result of the plane before, as in the trial)
                asm volatile("" : "+v"(poff) : "v"(k == 0 ? ((KEEPV || XLDS) ? ra : x[XLDS ? 0 : 2 * DEPK]) : y[k > 0 ? k - 1 : 0][2 * DEPK]));
                const lds_pairs pu = (lds_pairs)(unsigned long long)poff;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const pair_t u = pu[2 * kk];
                    y[k][2 * kk] = fma(ra, u.x, y[k][2 * kk]);
                    y[k][2 * kk + 1] = fma(ra, u.y, y[k][2 * kk + 1]);
                }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k) am[k] = sel(acc_m, ak[k], am[k]);
            lpri = sel(acc_m, lp, lpri);
            llik = sel(acc_m, ll, llik);
            lpost = sel(acc_m, lt, lpost);
            prej = sel(acc_m, 0, prej + sel(inside_m, 0, 1));
            wt = sel(acc_m, 1, wt + 1);
            nacc += sel(acc_m, 1, 0);
            if (wt - prej > lim && h == 0) atomicCAS(s.stuck, 0, 1 + (int)gid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // (the walker index passes through an empty asm: the addresses of the stores below are then
    // formed HERE -- else the compiler keeps the addresses of the prologue's (1 + KM) NE loads alive
    // through the whole step loop to reuse them, two registers each)
    int we = w, he = h;
    unsigned xe_off = xoff0;
    asm volatile("" : "+v"(we), "+v"(he), "+v"(xe_off));
    const lds_pairs xend = (lds_pairs)(unsigned long long)xe_off;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int i = 4 * (e >> 1) + 2 * he + (e & 1);
        if (i < d) {
            double xv;
            if constexpr (XLDS) { const pair_t xp = xend[(e >> 1) * 256]; xv = (e & 1) ? xp.y : xp.x; }
            else xv = x[XLDS ? 0 : e];
            s.x[(size_t)i * W + we] = xv;
#pragma unroll
            for (int k = 0; k < KM; ++k) a.y[((size_t)k * d + i) * W + we] = y[k][e];
        }
    }
    if (h == 0) {
#pragma unroll
        for (int k = 0; k < KM; ++k) a.amode[(size_t)k * W + we] = am[k];
        s.logpost[we] = lpost; s.logprior[we] = lpri; s.loglike[we] = llik;
        s.weight[we] = wt; s.prior_rej[we] = prej; s.burn_left[we] = burn;
        s.n_accept[we] = nacc0 + nacc;
    }
    wave_add_accepts(s.accept_total, (h == 0) ? nacc : 0);
}

template <int DQ, int KM>
hipError_t launch_duo_mix(const IncStepArgs& a, hipStream_t st)
{
    constexpr int C = duo_chunk_mix(DQ, KM);
    const size_t lds = sizeof(double) * 2 * C * (1 + KM) * 4 * DQ;
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    const bool box0 = a.box && a.box_lo == 0.0 && a.box_hi > 0.0 && a.box_hi < INFINITY;
    static const kern_t kerns[4] = {
        step_duo_mix_kernel<DQ, KM, false, false>, step_duo_mix_kernel<DQ, KM, true, false>,
        step_duo_mix_kernel<DQ, KM, false, true>, step_duo_mix_kernel<DQ, KM, true, true>};
    const std::string stem = "mcmc::step_duo_mix_kernel<" + std::to_string(DQ) + ", " + std::to_string(KM);
    static const std::string names[4] = {stem + ", false>", stem + ", true>", stem + ", false, box>",
                                         stem + ", true, box>"};
    const int v = (unit_t ? 1 : 0) + (box0 ? 2 : 0);
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 128), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int DQ>
hipError_t dispatch_duo_mix(const IncStepArgs& a, hipStream_t st)
{
    if constexpr (DQ > MCMC_DUO_DQ_HI) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) {
            if (!duo_serves(a.n_modes, DQ)) return hipErrorInvalidValue;
            if constexpr (duo_serves(4, DQ)) {
                if (a.n_modes == 4) return launch_duo_mix<DQ, 4>(a, st);
            }
            if constexpr (duo_serves(3, DQ)) {
                if (a.n_modes == 3) return launch_duo_mix<DQ, 3>(a, st);
            }
            return a.n_modes == 2 ? launch_duo_mix<DQ, 2>(a, st) : hipErrorInvalidValue;
        }
        return dispatch_duo_mix<DQ + 1>(a, st);
    }
}

}  // namespace
}  // namespace mcmc

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)
// one translation unit per range of DQ (build.py: -DMCMC_DUO_DQ_LO=.. -DMCMC_DUO_DQ_HI=..)
extern "C" hipError_t MCMC_CAT(mcmc_hip_launch_inc_duo_, MCMC_DUO_DQ_LO)(const mcmc::IncStepArgs* a,
                                                                       hipStream_t st)
{
    if (a->dq < MCMC_DUO_DQ_LO || a->dq > MCMC_DUO_DQ_HI || a->n_drag > 0 || a->colflag || a->s.rows ||
        a->s.W % 128 != 0 || a->s.group_size % 128 != 0)
        return hipErrorInvalidValue;
    for (int q = 0; q < 4; ++q)
        if (a->periodic_mask4[q]) return hipErrorInvalidValue;
    if (a->n_modes < 2 || !a->amode || a->vu_cols > 0) return hipErrorInvalidValue;
    return mcmc::dispatch_duo_mix<MCMC_DUO_DQ_LO>(*a, st);
}
