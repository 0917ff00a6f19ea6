// mcmc_hip -- incremental step kernel for ONE Gaussian mode with PERIODIC parameters (gfx950).
//
// prior.py:658-676 (Prior.reduce_periodic) in incremental mode; specification: oracle/mcmc_oracle.c,
// step_core_inc with `carry_periodic`.  Round 5: the kernel is step_inc_kernel's body -- x and y
// in registers, four lanes per walker, the (v, u) pairs of a step read from LDS for the trial
// and, re-read, for the commit, the log-likelihood CARRIED along the whitened direction -- with
// the bounds of a periodic dimension set to [lo, pred(hi)]: a step on which every lane of the
// wave is inside all its bounds (most steps) is step_inc_kernel's step.  Only when some lane is
// outside (wave-uniform branch) the rows are looked at again, branch-free:
//   * a periodic coordinate that LEFT [lo, hi) is wrapped, t' = ((t - lo) / w - floor(.)) w + lo
//     (inside the interval the reference's expression returns t up to its own rounding; here it
//     returns t), and when the winding number changes (floor != 0) the move sh = t' - t is
//     carried into the whitened residual, y'_j += sh L^-1[j][i] for j >= i (ascending i), and
//     the walker's chi2 is summed from that residual instead of moved;
//   * the division by the period without a division: with R = RN(1 / w) from the prologue,
//     q0 = a R, q1 = fma(fma(-q0, w, a), R, q0), q2 = fma(fma(-q1, w, a), R, q1) is the correctly
//     rounded a / w (q1 is faithful -- its exact argument is within 2^-52 ulp of a / w -- and a
//     faithful quotient corrected once with the correctly rounded reciprocal is the IEEE quotient:
//     Markstein 1990) -- bit for bit the oracle's `/`, 5 instructions instead of 30.
// Rounds 2-4 wrapped every periodic coordinate at every step behind a scalar test per row (25
// branches per step at d = 30) and summed chi2 from the trial residual: 2.5 ms per 1200 steps with
// one periodic parameter against 1.37 for the same bounds without the flag.
// Up to kMaxPeriodic periodic parameters; more run on the general kernel (incremental_any.hip).
#include <string>

#include "incremental_common.h"

namespace mcmc {
namespace {

constexpr int kMaxPeriodic = 8;   // periodic parameters this kernel serves (capi.hip checks)
__host__ __device__ constexpr int inc_periodic_min_waves(int dq)
{
    // (measured, 65 536 walkers: d = 30 at four waves 2.6 ms per 1200 steps, at two 4.4; d = 100 at
    // two waves -- with some spilled registers -- 34 ms per 4000 steps, at one 59)
    return MCMC_EXP_WAVES(PERIODIC, dq <= 8 ? 4 : 2);
}

// the largest double below a finite x
__device__ __forceinline__ double pred_double(double x)
{
    const long long b = __double_as_longlong(x);
    if (x > 0.0) return __longlong_as_double(b - 1);
    if (x < 0.0) return __longlong_as_double(b + 1);
    return -4.9406564584124654e-324;
}

// a / w given R = RN(1 / w): the correctly rounded quotient (see the header)
__device__ __forceinline__ double div_by(double a, double w, double R)
{
    double q = a * R;
    q = fma(fma(-q, w, a), R, q);
    return fma(fma(-q, w, a), R, q);
}

template <int DQ, bool NORMP, bool ONED>
__global__ void __launch_bounds__(256, inc_periodic_min_waves(DQ))
step_inc_periodic_kernel(const IncStepArgs a, const int C)
{
    extern __shared__ __attribute__((aligned(16))) double2 smem2[];
    constexpr int COLB = 4 * DQ;
    const int CHUNK = C * COLB;   // (C: columns per chunk, inc_chunk(DQ) or what the LDS leaves)
    constexpr int dpad = 4 * DQ;
    __shared__ double2 sLH[dpad];       // (lo, hi); beyond d: (-inf, +inf); a periodic dimension: [lo, pred(hi)]
    __shared__ double4 sPer[dpad];      // periodic dimensions: (lo, hi, w, RN(1 / w)), w = hi - lo
    __shared__ double2 sNA[NORMP ? dpad : 1];   // normal priors: (loc, 1/scale); 1/scale = 0: none here
    __shared__ double sNM[NORMP ? dpad : 1];    //                -log(scale sqrt(2 pi))
    __shared__ int sPdim[kMaxPeriodic];   // the periodic dimensions, ascending
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, c = tid & 3, wave = tid >> 6, lane = tid & 63;
    const int W = s.W, d = a.d;
    const int w = blockIdx.x * 64 + (tid >> 2);
    const int g = __builtin_amdgcn_readfirstlane(w / s.group_size);
    const int ncols = s.n_steps;
    const double2* __restrict__ gVU = (const double2*)a.VU + (size_t)g * ncols * COLB;
    double2* const sVU = smem2;
    int np = 0;
    for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
    // behind the two column chunks, sized by the number of periodic parameters (every KiB counts:
    // four workgroups share the 160 KiB of a CU) --
    // the wrap moves of a step: [walker of the workgroup][periodic parameter], written by the
    // lane that owns the dimension, read by its quad
    double* const sShift = (double*)(smem2 + 2 * CHUNK);                 // [64][np]
    // L^-1[j][i_q] for j >= i_q, the q-th periodic dimension: what a wrap moves y by
    double* const sLc = sShift + 64 * np;                                // [np][dpad]
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
    };
    stage(0);
    auto is_periodic = [&](int i) { return (a.periodic_mask4[i >> 5] >> (i & 31)) & 1u; };
    for (int i = tid; i < dpad; i += 256) {
        const double lo = a.prior[i], hi = a.prior[dpad + i];
        const bool per = i < d && is_periodic(i);
        // (a periodic dimension: [lo, pred(hi)] -- "t <= pred(hi)" is "t < hi": inside means
        // that nothing has to be wrapped)
        sLH[i] = per ? make_double2(lo, pred_double(hi)) : make_double2(lo, hi);
        sPer[i] = per ? make_double4(lo, hi, hi - lo, 1.0 / (hi - lo)) : make_double4(0.0, 1.0, 1.0, 1.0);
        if (NORMP) {
            sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
            sNM[i] = a.prior[4 * dpad + i];
        }
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
    double x[DQ], y[DQ];
    unsigned mine = 0;     // bit kk: dimension 4 kk + c of this lane is periodic
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        x[kk] = in ? s.x[(size_t)i * W + w] : 0.0;     // (beyond d: bounds -inf / +inf)
        y[kk] = in ? a.y[(size_t)i * W + w] : 0.0;
        if (in && is_periodic(i)) mine |= 1u << kk;
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
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    if (a.anchor) {   // (wave-uniform) y has just been refreshed from x: the carried log-likelihood
        double pa = 0.0;   // is re-anchored on it (orc_anchor_loglike)
#pragma unroll
        for (int kk = 0; kk < DQ; ++kk) pa = fma(y[kk], y[kk], pa);
        llik = -0.5 * (s.cnorm0 + quad_sum(pa));
        lpost = lpri + llik;
    }
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;
    const uint32_t gid = s.walker0 + (uint32_t)w;
    const double mt10 = s.max_tries * 10.0;
    const int lim1 = s.max_tries < 2.0e9 ? (int)floor(s.max_tries) : 0x7fffffff;
    const int lim10 = mt10 < 2.0e9 ? (int)floor(mt10) : 0x7fffffff;
    const bool unit_t = s.temperature == 1.0;
    // |u|^2 of the launch's columns by scalar loads (see step_inc_kernel)
    const cdoubles gUU = (cdoubles)(unsigned long long)(a.UU + (size_t)g * ncols);
    __shared__ dpair_t short_log_lds[SHORT_LOG_TABLE_SIZE];
    const short_log_tab slog = short_log_load(short_log_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool burning = lanes(burn > 0) != 0ull;   // wave-uniform
    unsigned long long cur_oct = ~0ull;
    __shared__ pair_t sRE[kStagedPairs];   // the (r, Ea) pairs of the current octet (StagedVariates)
    StagedVariates sv;
    sv.init(sRE, wave, lane);
    const int hw_slot = hw_wave_slot();
    double* const myShift = sShift + (tid >> 2) * np;

    for (int base = 0, k = 0; base < ncols; base += C, ++k) {
        const double2* __restrict__ cur = sVU + (k & 1) * CHUNK;
        stage(k + 1);
        const int cols = ncols - base < C ? ncols - base : C;
        unsigned long long oned_cols = 0;
        if (ONED)   // (the instantiations chosen when the blocking has a one-parameter block)
            oned_cols = lanes(lane < cols && a.colflag[(size_t)g * ncols + base + lane] != 0);
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl) {
            const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
            if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step (see step_inc_kernel)
                cur_oct = S >> 3;
                rotate_priority<inc_periodic_min_waves(DQ)>(hw_slot);
                PairRng pr;
                pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)c, slog);
                sv.fill(sRE, wave, lane, c, pr, S);
            }
            double r, Ea;
            if (ONED && ((oned_cols >> sl) & 1ull)) {   // wave-uniform
                step_variates(s.key0, s.key1, gid, S, 0, true, r, Ea);
            } else
            {
                sv.fetch(r, Ea);
            }
            sv.next();
            const double uu = gUU[base + sl];   // (wave-uniform address: a scalar load)
            const double2* __restrict__ col = cur + sl * COLB + c;
            unsigned long long inb = ~0ull;
            double pc = 0.0, sc = 0.0;
            // ---- the trial as in step_inc_kernel: the bounds of a periodic dimension are
            // [lo, pred(hi)] here, so "every lane inside" also says that no coordinate has to be
            // wrapped (round 5: a periodic coordinate is wrapped only where it LEAVES [lo, hi)),
            // and the log-likelihood moves by r (2 y.u + r |u|^2) / 2 (carried)
            // (four rows at a time: the pointers of the next rows depend, through an empty
            // asm, on the chain of these -- else the pairs and the bounds of all DQ rows are
            // fetched up front into 8 DQ registers)
            lds_pairs colt = relaunder(col), lht = relaunder(&sLH[c]);
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                if (kk % 4 == 0 && kk) {
                    colt = relaunder_after(colt, pc);
                    lht = relaunder_after(lht, pc);
                }
                const pair_t p = colt[4 * kk];
                const pair_t lh = lht[4 * kk];
                const double tk = fma(r, p.x, x[kk]);
                inb &= lanes(tk <= lh.y) & lanes(tk >= lh.x);
                pc = fma(y[kk], p.y, pc);   // y . u
                if (NORMP) {   // branch-free inside (1/scale = 0: no term -- a periodic dimension)
                    const int i = 4 * kk + c;
                    const double2 li = sNA[i];
                    const double qq = (tk - li.x) * li.y;
                    sc = sc + fma(-0.5 * qq, qq, sNM[i]);
                }
            }
            const double yu = quad_sum(pc);
            double ll = fma(-0.5 * r, fma(r, uu, yu + yu), llik);
            // ---- some lane of the wave is outside some bound (wave-uniform; rare away from the
            // walls except for walkers at the seam of a periodic parameter): row by row,
            // branch-free -- a periodic coordinate that left [lo, hi) is wrapped (prior.py:675,
            // the division by the period as div_by), its move goes to the walker's slot in LDS;
            // the support test is taken on the wrapped coordinates
            unsigned long long wound = 0ull;
            const bool slow = inb != ~0ull;
            if (slow) {
                inb = ~0ull;
                lds_pairs colw = relaunder(col), lhw = relaunder(&sLH[c]);
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const bool per = (mine >> kk) & 1u;
                    const double tk = fma(r, colw[4 * kk].x, x[kk]);
                    const pair_t lh = lhw[4 * kk];
                    const bool out = !((tk <= lh.y) & (tk >= lh.x));
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
            // the trial residual with the wrap moves (a wrap in the wave only): every row's
            // fma(r, u, y), then the moves in ascending dimension -- the columns of L^-1 of the
            // periodic dimensions sit in LDS (sLc) -- for the lanes `on`
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
            // (the walkers whose residual took a move: any lane of the quad wrote a non-zero one)
            const unsigned long long wq = ~quad_all_mask(~wound);
            if (wound != 0ull) {   // wave-uniform, rarer still: chi2 of those walkers is summed anew
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes
                double ytw[DQ];
                lds_pairs colw = relaunder(col);
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) ytw[kk] = fma(r, colw[4 * kk].y, y[kk]);
                shift_rows(ytw, ~0ull);
                double ps = 0.0;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) ps = fma(ytw[kk], ytw[kk], ps);
                const double ll_w = -0.5 * (s.cnorm0 + quad_sum(ps));
                ll = sel(wq, ll_w, ll);
            }
            const bool inside = quad_all(inb);
            const double lp = s.uniform_logp + (NORMP ? quad_sum(sc) : 0.0);
            // (outside the support lt is not used; lt = -inf fails both comparisons like the
            // specification's explicit lt != -inf)
            const double lt = lp + ll;
            double delta = lpost - lt;
            if (!unit_t) {   // wave-uniform: a division only where the run is tempered
                asm volatile("" ::: "memory");
                delta = delta / s.temperature;
            }
            const bool accept = inside & ((lt > lpost) | (Ea > delta));
            int lim = lim1;
            if (burning) {   // wave-uniform (see step_inc_kernel)
                lim = burn > 0 ? lim10 : lim1;
                burn -= (accept & (burn > 0)) ? 1 : 0;
                burning = lanes(burn > 0) != 0ull;
            }
            // ---- commit: the pairs are read again (see step_inc_kernel)
            const double ra = accept ? r : 0.0;
            lds_pairs col2 = relaunder(col);
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                if (kk % 4 == 0 && kk) col2 = relaunder_after(col2, y[kk - 1]);
                const pair_t p = col2[4 * kk];
                x[kk] = fma(ra, p.x, x[kk]);
                y[kk] = fma(ra, p.y, y[kk]);
            }
            if (slow) {   // wave-uniform: an accepted coordinate that left [lo, hi) is the wrapped one
                lds_pairs lhw = relaunder(&sLH[c]);
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const bool per = (mine >> kk) & 1u;
                    const pair_t lh = lhw[4 * kk];
                    const bool out = !((x[kk] <= lh.y) & (x[kk] >= lh.x));
                    const double4 pw = sPer[4 * kk + c];
                    const double yv = div_by(x[kk] - pw.x, pw.z, pw.w);
                    const double fl = floor(yv);
                    const double tw = (yv - fl) * pw.z + pw.x;
                    x[kk] = (per & out & accept) ? tw : x[kk];   // (a walker that stays keeps its x)
                }
                // ... and the residual of a walker that accepted a wrapping trial takes the moves
                // (on top of fma(r, u, y): the same operations in the same order as for the trial)
                if (wound != 0ull)
                    shift_rows(y, lanes(accept));
            }
            llik = accept ? ll : llik;
            if (NORMP) lpri = accept ? lp : lpri;
            lpost = accept ? lt : lpost;
            prej = accept ? 0 : (prej + (inside ? 0 : 1));
            wt = accept ? 1 : wt + 1;
            nacc += accept ? 1 : 0;
            if (wt - prej > lim && c == 0) atomicCAS(s.stuck, 0, 1 + (int)gid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc : 0);
}

template <int DQ>
hipError_t launch_periodic_dq(const IncStepArgs& a, hipStream_t st)
{
    int np = 0;
    for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
    // columns per chunk: inc_chunk(DQ), or fewer where the wrap moves and the columns of L^-1
    // would push the workgroup over its share of the LDS (a fourth / half / all of a CU's 160 KiB
    // by the occupancy the registers are held to: one workgroup less per CU is a second round
    // of workgroups at 65 536 walkers)
    const size_t fixed = (size_t)4 * DQ * (16 + 32 + (a.has_norm ? 24 : 0)) + 16 * SHORT_LOG_TABLE_SIZE + 64 +
                         sizeof(pair_t) * kStagedPairs +
                         sizeof(double) * (size_t)np * (64 + 4 * DQ);
    const size_t share = ((size_t)160 << 10) / inc_periodic_min_waves(DQ);
    int C = inc_chunk(DQ);
    while (C > 4 && fixed + sizeof(double2) * 2 * C * 4 * DQ > share) C -= 4;
    const size_t lds = sizeof(double2) * 2 * C * 4 * DQ + sizeof(double) * (size_t)np * (64 + 4 * DQ);
    const std::string stem = "mcmc::step_inc_periodic_kernel<" + std::to_string(DQ);
    static const std::string names[4] = {stem + ", false>", stem + ", true>",
                                         stem + ", false, 1-D blocks>", stem + ", true, 1-D blocks>"};
    typedef void (*kern_t)(const IncStepArgs, const int);
    static const kern_t kerns[4] = {
        step_inc_periodic_kernel<DQ, false, false>, step_inc_periodic_kernel<DQ, true, false>,
        step_inc_periodic_kernel<DQ, false, true>, step_inc_periodic_kernel<DQ, true, true>};
    const int v = (a.has_norm ? 1 : 0) + (a.colflag ? 2 : 0);
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kerns[v],
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(names[v].c_str());
    hipLaunchKernelGGL(kerns[v], dim3(a.s.W / 64), dim3(256), lds, st, a, C);
    return hipGetLastError();
}

template <int DQ>
hipError_t dispatch_periodic(const IncStepArgs& a, hipStream_t st)
{
    if constexpr (DQ > 32) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) return launch_periodic_dq<DQ>(a, st);
        return dispatch_periodic<DQ + 1>(a, st);
    }
}

}  // namespace
}  // namespace mcmc

extern "C" hipError_t mcmc_hip_launch_inc_periodic(const mcmc::IncStepArgs* a, hipStream_t st)
{
    int n_periodic = 0;
    for (int q = 0; q < 4; ++q) n_periodic += __builtin_popcount(a->periodic_mask4[q]);
    if (n_periodic < 1 || n_periodic > mcmc::kMaxPeriodic || a->n_modes != 1 || a->n_drag > 0)
        return hipErrorInvalidValue;
    return mcmc::dispatch_periodic<1>(*a, st);
}
