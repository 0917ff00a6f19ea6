// mcmc_hip -- the GENERAL incremental step kernel (gfx950 only), 2 <= d <= 128.
//
// What the tuned incremental kernels (incremental_kernels.hip: step_inc_kernel<.., PER>, step_inc_mix_kernel) leave
// out: mixtures of more than four modes (gaussian_mixture.py:138-163, up to kMaxModes), mixtures
// above d = 64, periodic parameters (prior.py:658-676) together with a mixture, more than eight
// periodic parameters, and emitted rows of anything but one non-periodic mode.  Same specification (oracle/mcmc_oracle.c, step_core_inc), same O(d) step: the trial
// t = x + r v has the whitened residuals yt_k = y_k + r u_k, u_k = L_k^-1 v shared by the walkers
// of a basis group; a periodic coordinate that changes its winding number by the wrap carries the
// move sh = t' - t into every residual, yt_k[j] += sh L_k^-1[j][i] for j >= i.
//
// Layout: FOUR lanes per walker as in the tuned kernels (lane class c = lane & 3 owns the
// dimensions i = 4 kk + c; chi2 and the normal-prior terms are four interleaved chains combined
// (p0 + p1) + (p2 + p3)), x in registers -- but the K residuals of a walker live in LDS, lane-major
// ([mode][kk][lane]: conflict-free 8-byte accesses), because K dq doubles per lane do not fit the
// register file at an occupancy worth having, and K and the periodic set are run-time values.
// A workgroup is 1, 2 or 4 waves (16 walkers each) of ONE basis group: the launcher picks the
// width that puts the most waves on a CU given the LDS the state needs.  The columns
// (v, u_1 .. u_K) of the launch are staged through LDS in chunks, double-buffered, by
// global->LDS DMA, as PLANES ([1 + K][4 dq], whiten_directions_planes_kernel below).
// The trial point and the trial residuals are NOT kept: the commit recomputes them from the
// column (x += ra v, y_k += ra u_k with ra = r where the walker accepts and 0 elsewhere; with a
// wrap in the wave the residuals are selected instead).
//
// What fits the register file -- up to 2 / 4 / 8 / 16 modes while dq (modes + 1) <= 208 doubles per
// lane, i.e. 4 modes at d = 128, 8 at d <= 92, 16 at d <= 48, and periodic sets whose columns of
// L^-1 take at most 24 KiB of LDS -- runs on step_inc_regs_kernel<DQ, KM, PER> instead:
// step_inc_mix_kernel's design (everything in registers, KM = 2, 4, 8 or 16 register planes of
// which the first n_modes are live; PER adds the round-3 scheme for periodic
// parameters), which does not pay the LDS round trips and is not held to one wave per SIMD by
// the LDS the state takes.  One translation unit per group of KM (build.py: -DANY_PART=0 / 1 / 2).
// Both kernels emit rows at run time (`emit: chains`, s.rows).
#include <string>

#include "incremental_common.h"

#ifndef ANY_REGS_TWO_WAVES
#define ANY_REGS_TWO_WAVES 80   // doubles of state per lane up to which two waves share a SIMD
#endif
#ifndef ANY_PART
#define ANY_PART 0   // 0: the LDS kernel, the planes, KM = 2 and 4; 1: KM = 8; 2: KM = 16
#endif
namespace mcmc {
namespace {

#if ANY_PART == 0
// LDS the kernel needs (bytes) for `nw` waves: the column chunks, the residuals, the per-mode
// log-densities of a step and the wrap moves; `fixed` = the statically allocated part
struct AnyGeom {
    int nw, C, npad;
    int lcols;   // the columns of L_k^-1 of the periodic dimensions sit in LDS (else: read from HBM)
    size_t dynamic;
};
constexpr size_t kAnyStatic = 16 * 128 * 2 + 8 * 128 + 4 * 128 + 16 * SHORT_LOG_TABLE_SIZE + 16 * kMaxModes + 256 + 512 +
                              16 * kStagedPairs;   // (+ the staged variates, StagedVariates)
constexpr size_t kLdsPerCu = 160u << 10;
constexpr size_t kLdsPerWg = 160u << 10;   // (a single workgroup may hold all of it)

inline int any_chunk(int K, int dq)
{
    const int col_bytes = (1 + K) * 4 * dq * 8;
    int c = (4096 / col_bytes) & ~3;
    return c < 4 ? 4 : (c > 32 ? 32 : c);
}
inline size_t any_dynamic(int K, int dq, int npad, int nw, int C, int lcols)
{
    const size_t col = (size_t)(1 + K) * 4 * dq;
    return sizeof(double) * (2 * (size_t)C * col + (size_t)nw * K * dq * 64 + (size_t)nw * K * 64 +
                             (size_t)nw * 16 * npad + (lcols ? (size_t)K * npad * 4 * dq : 0));
}
inline bool any_geometry(int K, int dq, int n_periodic, int W, int group_size, AnyGeom& g)
{
    g.C = any_chunk(K, dq);
    g.npad = n_periodic > 0 ? n_periodic : 1;
    // a wrap moves every residual by a column of L_k^-1: the columns of the periodic dimensions
    // are kept in LDS where that takes at most 32 KiB (else each element is a load from HBM
    // in the middle of a step)
    for (int lcols = (n_periodic > 0 && (size_t)K * n_periodic * 4 * dq * 8 <= (32u << 10)) ? 1 : 0;
         lcols >= 0; --lcols) {
        int best = 0;
        for (int nw = 4; nw >= 1; nw >>= 1) {
            if (W % (16 * nw) != 0 || group_size % (16 * nw) != 0) continue;
            const size_t need = any_dynamic(K, dq, g.npad, nw, g.C, lcols) + kAnyStatic;
            if (need > kLdsPerWg) continue;
            const int waves = (int)(kLdsPerCu / need) * nw;
            if (waves > best) { best = waves; g.nw = nw; g.dynamic = need - kAnyStatic; }
        }
        g.lcols = lcols;
        if (best > 0) return true;
    }
    return false;
}

__device__ __forceinline__ double wrap_into(double t, double lo, double w, double& fl)
{
    const double yv = (t - lo) / w;
    fl = floor(yv);
    return (yv - fl) * w + lo;
}

template <int DQ>
__global__ void __launch_bounds__(256) step_inc_any_kernel(const IncStepArgs a, int C, int npad, int lcols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int dpad = 4 * DQ;
    __shared__ double2 sLH[dpad];      // (lo, hi); beyond d: (-inf, +inf)
    __shared__ double2 sNA[dpad];      // normal priors: (loc, 1/scale); 1/scale = 0: none here
    __shared__ double sNM[dpad];       //                -log(scale sqrt(2 pi))
    __shared__ int sPdim[dpad];        // the periodic dimensions, ascending
    __shared__ double2 sMode[kMaxModes];   // (log-normalisation, weight) of the modes
    __shared__ dpair_t short_log_lds[SHORT_LOG_TABLE_SIZE];
    const StepArgs& s = a.s;
    const int K = a.n_modes, d = a.d, W = s.W;
    const int COL = (1 + K) * dpad;
    const int CHUNK = C * COL;
    const int nw = (int)(blockDim.x >> 6);
    const int tid = threadIdx.x, c = tid & 3, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wl = lane >> 2;                                   // walker of the wave
    const int w = blockIdx.x * 16 * nw + (tid >> 2);
    const int g = __builtin_amdgcn_readfirstlane(w / s.group_size);
    const int ncols = s.n_steps;
    const double* __restrict__ gVU = a.VU + (size_t)g * ncols * COL;
    double* const sCol = smem;                                                   // [2][CHUNK]
    double* const sY = smem + 2 * CHUNK + (size_t)wave * K * DQ * 64 + lane;     // [K][DQ][64]
    double* const sA = smem + 2 * CHUNK + (size_t)nw * K * DQ * 64 + (size_t)wave * K * 64;   // [K][64]
    double* const sSh = smem + 2 * CHUNK + (size_t)nw * K * (DQ + 1) * 64 +
                        ((size_t)wave * 16 + wl) * npad;                         // [16][npad]
    // [K][np][dpad]: L_k^-1[j][i_q] for j >= i_q, the q-th periodic dimension (lcols)
    double* const sLc = smem + 2 * CHUNK + (size_t)nw * K * (DQ + 1) * 64 + (size_t)nw * 16 * npad;
    // (read through an LDS-typed pointer: beside the HBM read of the other branch a generic one
    // would be merged with it into a flat load)
    typedef const double __attribute__((address_space(3))) * lds_cdoubles;
    const lds_cdoubles pLc = (lds_cdoubles)(unsigned long long)(unsigned)(unsigned long long)sLc;
    auto stage = [&](int k) {
        const int first = k * C;
        if (first >= ncols) return;
        const int cols = ncols - first < C ? ncols - first : C;
        const int bytes = cols * COL * 8;
        const char* src = (const char*)(gVU + (size_t)first * COL);
        char* dst = (char*)(sCol + (k & 1) * CHUNK);
        for (int kb = wave; kb * 1024 < bytes; kb += nw) {
            if (kb * 1024 + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane * 16),
                    (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
        }
    };
    stage(0);
    for (int i = tid; i < dpad; i += (int)blockDim.x) {
        sLH[i] = make_double2(a.prior[i], a.prior[dpad + i]);
        sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
        sNM[i] = a.prior[4 * dpad + i];
    }
    auto is_periodic = [&](int i) { return (a.periodic_mask4[i >> 5] >> (i & 31)) & 1u; };
    int np = 0;
    for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
    if (tid < K) sMode[tid] = make_double2(s.cblock[a.cnorm_off + tid], s.cblock[a.weight_off + tid]);
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < d; ++i)
            if (is_periodic(i)) sPdim[n++] = i;
    }
    if (lcols) {   // (wave-uniform; the barrier makes sPdim visible to every thread)
        __syncthreads();
        for (int e = tid; e < K * np * dpad; e += (int)blockDim.x) {
            const int j = e % dpad, q = (e / dpad) % np, k = e / (dpad * np);
            const int i = sPdim[q];
            sLc[e] = (j >= i && j < d) ? a.Lrow[((size_t)k * d + j) * d + i] : 0.0;
        }
    }
    double x[DQ];
    unsigned mine = 0;     // bit kk: dimension 4 kk + c of this lane is periodic
    unsigned anyp = 0;     // bit kk: one of the dimensions 4 kk .. 4 kk + 3 is (wave-uniform)
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        x[kk] = in ? s.x[(size_t)i * W + w] : 0.0;
        for (int k = 0; k < K; ++k) sY[(k * DQ + kk) * 64] = in ? a.y[((size_t)k * d + i) * W + w] : 0.0;
        if (in && is_periodic(i)) mine |= 1u << kk;
        if ((a.periodic_mask4[(4 * kk) >> 5] >> ((4 * kk) & 31)) & 0xFu) anyp |= 1u << kk;
    }
    anyp = (unsigned)__builtin_amdgcn_readfirstlane((int)anyp);
    // the slot of a periodic dimension in sPdim = the number of periodic dimensions below it
    auto slot_of = [&](int i) {
        int n = 0;
        for (int q = 0; q < (i >> 5); ++q) n += __builtin_popcount(a.periodic_mask4[q]);
        return n + __builtin_popcount(a.periodic_mask4[i >> 5] & ((1u << (i & 31)) - 1u));
    };
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;
    // `emit: chains` (run-time here: s.rows): every accepted step past the burn-in stores the
    // point it LEAVES with its weight (mcmc.py:691-707), as step_inc_kernel<.., EMIT> does
    int nrow = s.rows ? s.n_rows[w] : 0;
    // thinned emission (round 6 here; collection.py:1373-1383 -- OneSamplePoint.add_to_collection with
    // output_thin --, as step_inc_kernel<.., EMIT> does it): a walker's weights add up in thin_acc, a
    // row goes out when the sum reaches `thin`, with weight sum / thin (the quotient by a
    // reciprocal, set right by the remainder), the remainder carried
    const bool thinning = s.rows && s.thin > 1;   // (wave-uniform)
    int tacc = thinning ? s.thin_acc[w] : 0;
    const double inv_thin = thinning ? 1.0 / (double)s.thin : 1.0;
    const uint32_t gid = s.walker0 + (uint32_t)w;
    const double mt10 = s.max_tries * 10.0;
    const int lim1 = s.max_tries < 2.0e9 ? (int)floor(s.max_tries) : 0x7fffffff;
    const int lim10 = mt10 < 2.0e9 ? (int)floor(mt10) : 0x7fffffff;
    const short_log_tab slog = short_log_load(short_log_lds);
    __shared__ double exp64_lds[64];   // 2^(j / 64): the log-sum-exp's table-driven exponential
    const exp_tab etab = exp_tab_load(exp64_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long cur_oct = ~0ull;
    __shared__ pair_t sRE[kStagedPairs];   // the (r, Ea) pairs of the current octet (StagedVariates)
    StagedVariates sv;
    sv.init(sRE, wave, lane);

    for (int base = 0, kc = 0; base < ncols; base += C, ++kc) {
        const double* __restrict__ cur = sCol + (kc & 1) * CHUNK;
        stage(kc + 1);
        const int cols = ncols - base < C ? ncols - base : C;
        unsigned long long oned_cols = 0;   // bit sl: the column belongs to a one-parameter block
        if (a.colflag)
            oned_cols = lanes(lane < cols && a.colflag[(size_t)g * ncols + base + lane] != 0);
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl) {
            const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
            if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step (see step_inc_kernel)
                cur_oct = S >> 3;
                PairRng pr;
                pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)c, slog);
                sv.fill(sRE, wave, lane, c, pr, S);
            }
            double r, Ea;
            if ((oned_cols >> sl) & 1ull) {   // wave-uniform: the un-paired 1-D variates
                step_variates(s.key0, s.key1, gid, S, 0, true, r, Ea);
            } else
            {
                sv.fetch(r, Ea);
            }
            sv.next();
            const double* __restrict__ col = cur + sl * COL + c;
            // ---- the trial point: support, normal priors, wraps
            unsigned long long inb = ~0ull, wound = 0ull;
            double sc = 0.0;
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                const double2 lh = sLH[4 * kk + c];
                double tk = fma(r, col[4 * kk], x[kk]);
                if ((anyp >> kk) & 1u) {   // wave-uniform: some lane's dimension here is periodic
                    double fl;
                    const double tw = wrap_into(tk, lh.x, lh.y - lh.x, fl);
                    const bool per = (mine >> kk) & 1u;
                    const double shk = (per & (fl != 0.0)) ? tw - tk : 0.0;
                    tk = per ? tw : tk;
                    wound |= lanes(shk != 0.0);
                    if (per) sSh[slot_of(4 * kk + c)] = shk;
                }
                inb &= lanes(tk <= lh.y) & lanes(tk >= lh.x);
                if (a.has_norm) {   // wave-uniform; branch-free inside
                    const double2 li = sNA[4 * kk + c];
                    const double qq = (tk - li.x) * li.y;
                    sc = sc + fma(-0.5 * qq, qq, sNM[4 * kk + c]);
                }
            }
            // the wrap moves of this step that are not zero somewhere in the wave, as slot masks
            unsigned long long act0 = 0ull, act1 = 0ull;
            if (wound != 0ull) {   // wave-uniform
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes
                for (int q = 0; q < np; ++q)
                    if (lanes(sSh[q] != 0.0) != 0ull) {
                        if (q < 64) act0 |= 1ull << q;
                        else act1 |= 1ull << (q - 64);
                    }
            }
            // the residual of the trial in mode k with a wrap in the wave: fma(r, u, y) for every row,
            // then the wrap moves slot by slot -- ascending dimension, the order of the
            // specification for every element -- with the rows unrolled (their reads of the
            // column of L^-1 are independent; lcols: from LDS, else from HBM)
            auto shifted = [&](int k, const double* __restrict__ uk, const double* __restrict__ yk,
                               double (&yt)[DQ]) {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) yt[kk] = fma(r, uk[4 * kk], yk[kk * 64]);
                for (int half = 0; half < 2; ++half) {
                    unsigned long long m = half ? act1 : act0;
                    while (m != 0ull) {
                        const int q = __builtin_ctzll(m) + 64 * half;
                        m &= m - 1ull;
                        const double sv = sSh[q];
                        const int i = sPdim[q];
                        if (lcols) {
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) {
                                const int j = 4 * kk + c;
                                const bool on = (sv != 0.0) & (j >= i) & (j < d);
                                const double lji = pLc[(k * np + q) * dpad + j];
                                yt[kk] = on ? fma(sv, lji, yt[kk]) : yt[kk];
                            }
                        } else {
#pragma unroll
                            for (int kk = 0; kk < DQ; ++kk) {
                                const int j = 4 * kk + c;
                                const bool on = (sv != 0.0) & (j >= i) & (j < d);
                                const double lji = on ? a.Lrow[((size_t)k * d + j) * d + i] : 0.0;
                                yt[kk] = on ? fma(sv, lji, yt[kk]) : yt[kk];
                            }
                        }
                    }
                }
            };
            // ---- chi2 per mode, log-sum-exp (eval_point / step_core_inc)
            double amax = -INFINITY, a_last = 0.0;
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double* __restrict__ uk = col + (1 + k) * dpad;
                const double* __restrict__ yk = sY + (size_t)k * DQ * 64;
                double pc = 0.0;
                if (wound != 0ull) {   // wave-uniform
                    double yt[DQ];
                    shifted(k, uk, yk, yt);
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) pc = fma(yt[kk], yt[kk], pc);
                } else {
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double yt = fma(r, uk[4 * kk], yk[kk * 64]);
                        pc = fma(yt, yt, pc);
                    }
                }
                a_last = -0.5 * (sMode[k].x + quad_sum(pc));
                amax = fmax(a_last, amax);
                if (K > 1) sA[k * 64 + lane] = a_last;
            }
            double ll = a_last;
            if (K > 1) {
                // one exponential per (walker, mode): lane class c takes the modes k = c mod 4
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int k = c; k < K; k += 4) sA[k * 64 + lane] = dexp_tab(sA[k * 64 + lane] - amax, etab);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double Ssum = 0.0;
                for (int k = 0; k < K; ++k) Ssum = fma(sMode[k].y, sA[k * 64 + (lane & ~3) + (k & 3)], Ssum);
                ll = dlog_tab(Ssum, slog) + amax;
            }
            const unsigned long long inside_m = quad_all_mask(inb);
            const double lp = s.uniform_logp + (a.has_norm ? quad_sum(sc) : 0.0);
            const double lt = lp + ll;
            const double delta = (lpost - lt) / s.temperature;   // (T = 1: x / 1.0 == x)
            const unsigned long long acc_m =
                inside_m & lanes(lt != -INFINITY) & (lanes(lt > lpost) | lanes(Ea > delta));
            if (s.rows) {   // wave-uniform
                const unsigned long long em_m = acc_m & lanes(burn <= 0);
                if (em_m != 0ull) {   // some walker of the wave emits
                    bool em = __builtin_amdgcn_inverse_ballot_w64(em_m);
                    int ew = wt;   // the weight the row is written with
                    if (thinning) {
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
                        double* __restrict__ row = s.rows + ((size_t)w * s.row_cap + nrow) * (size_t)(d + 4);
                        row[c] = c == 0 ? (double)ew : c == 1 ? lpost : c == 2 ? lpri : llik;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk)
                            if (4 * kk + c < d) row[4 + 4 * kk + c] = x[kk];
                    }
                    nrow += em ? 1 : 0;   // rows beyond the capacity are counted as dropped
                }
            }
            const int lim = sel(lanes(burn > 0), lim10, lim1);
            burn -= sel(acc_m & lanes(burn > 0), 1, 0);
            // ---- commit: recomputed from the column
            const double ra = sel(acc_m, r, 0.0);
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) {
                if ((anyp >> kk) & 1u) {   // wave-uniform
                    const double2 lh = sLH[4 * kk + c];
                    const double tk = fma(r, col[4 * kk], x[kk]);
                    double fl;
                    const double tw = wrap_into(tk, lh.x, lh.y - lh.x, fl);
                    x[kk] = sel(acc_m, ((mine >> kk) & 1u) ? tw : tk, x[kk]);
                } else {
                    x[kk] = fma(ra, col[4 * kk], x[kk]);
                }
            }
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double* __restrict__ uk = col + (1 + k) * dpad;
                double* __restrict__ yk = sY + (size_t)k * DQ * 64;
                if (wound != 0ull) {
                    double yt[DQ];
                    shifted(k, uk, yk, yt);
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) yk[kk * 64] = sel(acc_m, yt[kk], yk[kk * 64]);
                } else {
                    // (every read before the first write: the compiler cannot tell that the
                    // residuals and the column chunk are different parts of the LDS)
                    double u_[DQ], y_[DQ];
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) { u_[kk] = uk[4 * kk]; y_[kk] = yk[kk * 64]; }
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) yk[kk * 64] = fma(ra, u_[kk], y_[kk]);
                }
            }
            lpri = sel(acc_m, lp, lpri);
            llik = sel(acc_m, ll, llik);
            lpost = sel(acc_m, lt, lpost);
            prej = sel(acc_m, 0, prej + sel(inside_m, 0, 1));
            wt = sel(acc_m, 1, wt + 1);
            nacc += sel(acc_m, 1, 0);
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
            for (int k = 0; k < K; ++k) a.y[((size_t)k * d + i) * W + w] = sY[(k * DQ + kk) * 64];
        }
    }
    if (c == 0) {
        s.logpost[w] = lpost; s.logprior[w] = lpri; s.loglike[w] = llik;
        s.weight[w] = wt; s.prior_rej[w] = prej; s.burn_left[w] = burn;
        s.n_accept[w] = nacc0 + nacc;
        if (s.rows) s.n_rows[w] = nrow;
        if (thinning) s.thin_acc[w] = tacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc : 0);
}

#endif  // ANY_PART == 0

// ---------------------------------------------------------------- mixtures in registers
__host__ __device__ constexpr int regs_chunk(int dq, int km)
{
    int c = (1792 / ((1 + km) * 4 * dq)) & ~3;   // (as inc_chunk_mix: 8 KB go to the staged variates)
    return c < 4 ? 4 : (c > 64 ? 64 : c);
}
__host__ __device__ constexpr int regs_min_waves(int dq, int km)
{
    // (the state is dq (km + 1) doubles per lane: step_inc_mix_kernel's table, continued)
    // (more than four planes: the per-mode log-densities and exponentials of a step alone take
    // some forty registers -- never more than two waves.  Two waves up to ANY_REGS_TWO_WAVES
    // doubles of state per lane: measured, K = 5 at d = 30 14.9 -> 8.2 ms per 1200 steps)
    return km > 4 ? (dq * (km + 1) <= ANY_REGS_TWO_WAVES ? 2 : 1)
                  : dq * (km + 1) <= 18 ? 4 : dq * (km + 1) <= 24 ? 3
                  : dq * (km + 1) <= ANY_REGS_TWO_WAVES ? 2 : 1;
}
// what fits the 512 registers of a lane at one wave per SIMD
__host__ __device__ constexpr bool regs_fits(int dq, int km) { return dq * (km + 1) <= 208; }
__host__ __device__ constexpr int regs_bucket(int K) { return K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : 16; }

// a / w given R = RN(1 / w): the correctly rounded quotient in five operations (see
// DESIGN.md 2 "Periodic parameters": q1 is faithful, and a faithful quotient corrected once with the
// correctly rounded reciprocal is the IEEE quotient)
__device__ __forceinline__ double regs_div_by(double a, double w, double R)
{
    double q = a * R;
    q = fma(fma(-q, w, a), R, q);
    return fma(fma(-q, w, a), R, q);
}

// KM register planes, the first a.n_modes of them live (a plane beyond that is never touched:
// the tests on k < K are wave-uniform branches around fully unrolled code).  The columns are
// planes of 4 DQ doubles, 1 + n_modes of them.  One-parameter blocks and the temperature are
// run-time properties here.  PER: periodic parameters (the round-3 scheme; the tuned one-mode kernel folds them
// into step_inc_kernel<.., PER> instead) -- the trial
// and commit loops stay branch-free (a periodic dimension has the bounds (-inf, +inf) there);
// behind each, the rows that hold a periodic dimension wrap their coordinate; a wrap in the
// wave sends the trial residual of a mode to registers, which takes the wrap moves in ascending
// dimension (columns of L_k^-1 in LDS) and is formed again for the commit.
template <int DQ, int KM, bool PER>
__global__ void __launch_bounds__(256, regs_min_waves(DQ, KM)) step_inc_regs_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int dpad = 4 * DQ;
    constexpr int C = regs_chunk(DQ, KM);
    const StepArgs& s = a.s;
    const int K = a.n_modes;
    const int COL = (1 + K) * dpad;               // doubles per column
    const int CHUNK = C * (1 + KM) * dpad;        // (the buffers are sized for KM planes)
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
    __shared__ double2 sLH[4 * DQ];
    __shared__ double2 sNA[4 * DQ];     // normal priors: (loc, 1/scale) and -log(scale sqrt(2 pi))
    __shared__ double sNM[4 * DQ];
    __shared__ double2 sMode[kMaxModes];   // (log-normalisation, weight) of the modes
    __shared__ double4 sPer[PER ? 4 * DQ : 1];   // periodic dimensions: (lo, hi, w, RN(1 / w))
    __shared__ int sPdim[PER ? 4 * DQ : 1];      // the periodic dimensions, ascending
    auto is_periodic = [&](int i) { return (a.periodic_mask4[i >> 5] >> (i & 31)) & 1u; };
    int np = 0;
    if (PER)
        for (int q = 0; q < 4; ++q) np += __builtin_popcount(a.periodic_mask4[q]);
    // PER, behind the two column chunks: the wrap moves of a step [walker][periodic parameter]
    // and the columns L_k^-1[j][i_q], j >= i_q, of the periodic dimensions [mode][q][4 DQ]
    double* const sShift = smem + 2 * CHUNK;
    double* const sLc = sShift + 64 * np;
    const bool box = a.box && !PER;
    for (int i = tid; i < dpad; i += 256) {
        const double lo = a.prior[i], hi = a.prior[dpad + i];
        const bool per = PER && i < d && is_periodic(i);
        sLH[i] = per ? make_double2(-INFINITY, INFINITY) : make_double2(lo, hi);
        if (PER) sPer[i] = per ? make_double4(lo, hi, hi - lo, 1.0 / (hi - lo)) : make_double4(0.0, 1.0, 1.0, 1.0);
        sNA[i] = make_double2(a.prior[2 * dpad + i], a.prior[3 * dpad + i]);
        sNM[i] = a.prior[4 * dpad + i];
    }
    if (tid < K) sMode[tid] = make_double2(s.cblock[a.cnorm_off + tid], s.cblock[a.weight_off + tid]);
    if (PER) {
        if (tid == 0) {
            int n = 0;
            for (int i = 0; i < d; ++i)
                if (is_periodic(i)) sPdim[n++] = i;
        }
        __syncthreads();
        for (int e = tid; e < K * np * dpad; e += 256) {
            const int j = e % dpad, q = (e / dpad) % np, k = e / (dpad * np);
            const int i = sPdim[q];
            sLc[e] = (j >= i && j < d) ? a.Lrow[((size_t)k * d + j) * d + i] : 0.0;
        }
    }
    double x[DQ], y[KM][DQ];
    unsigned mine = 0;     // PER, bit kk: dimension 4 kk + c of this lane is periodic
    unsigned anyp = 0;     //      bit kk: one of the dimensions 4 kk .. 4 kk + 3 is (wave-uniform)
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        // (one box for all dimensions: a padded dimension rests at its middle, inside for every step)
        x[kk] = in ? s.x[(size_t)i * W + w] : (box ? 0.5 * (a.box_lo + a.box_hi) : 0.0);
#pragma unroll
        for (int k = 0; k < KM; ++k)
            y[k][kk] = (in && k < K) ? a.y[((size_t)k * d + i) * W + w] : 0.0;
        if (PER) {
            if (in && is_periodic(i)) mine |= 1u << kk;
            if ((a.periodic_mask4[(4 * kk) >> 5] >> ((4 * kk) & 31)) & 0xFu) anyp |= 1u << kk;
        }
    }
    if (PER) anyp = (unsigned)__builtin_amdgcn_readfirstlane((int)anyp);
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
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    const long long nacc0 = s.n_accept[w];
    int nacc = 0;     // accepted steps of this launch
    int nrow = s.rows ? s.n_rows[w] : 0;   // `emit: chains`, a run-time property here (s.rows)
    // thinned emission (round 6 here; collection.py:1373-1383 -- OneSamplePoint.add_to_collection with
    // output_thin --, as step_inc_kernel<.., EMIT> does it): a walker's weights add up in thin_acc, a
    // row goes out when the sum reaches `thin`, with weight sum / thin (the quotient by a
    // reciprocal, set right by the remainder), the remainder carried
    const bool thinning = s.rows && s.thin > 1;   // (wave-uniform)
    int tacc = thinning ? s.thin_acc[w] : 0;
    const double inv_thin = thinning ? 1.0 / (double)s.thin : 1.0;
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
        unsigned long long oned_cols = 0;   // bit sl: the column belongs to a one-parameter block
        if (a.colflag)
            oned_cols = lanes(lane < cols && a.colflag[(size_t)g * ncols + base + lane] != 0);
#pragma unroll 1
        for (int sl = 0; sl < cols; ++sl) {
            const unsigned long long S = s.step0 + (unsigned long long)(base + sl);
            if ((S >> 3) != cur_oct) {   // wave-uniform: every eighth step (see step_inc_kernel)
                cur_oct = S >> 3;
                rotate_priority<regs_min_waves(DQ, KM)>(hw_slot);
                PairRng pr;
                pr.run(s.key0, s.key1, gid, (cur_oct << 2) + (unsigned long long)c, slog);
                sv.fill(sRE, wave, lane, c, pr, S);
            }
            double r, Ea;
            if ((oned_cols >> sl) & 1ull) {   // wave-uniform: the un-paired 1-D variates
                step_variates(s.key0, s.key1, gid, S, 0, true, r, Ea);
            } else
            {
                sv.fetch(r, Ea);
            }
            sv.next();
            const double* __restrict__ col = cur + sl * COL + c;
            unsigned long long inb = ~0ull;   // the support test as a lane mask
            unsigned long long wound = 0ull;  // PER: lanes whose periodic coordinate changed its winding
            double sc = 0.0;
            if (box) {   // wave-uniform: one box for every dimension, no normal priors
                double tmx = -INFINITY, tmn = INFINITY;
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const double t = fma(r, col[4 * kk], x[kk]);
                    tmx = __builtin_fmax(tmx, t);
                    tmn = __builtin_fmin(tmn, t);
                }
                inb = lanes(tmx <= a.box_hi) & lanes(tmn >= a.box_lo);
            } else {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) {
                    const double t = fma(r, col[4 * kk], x[kk]);
                    const double2 lh = sLH[4 * kk + c];
                    inb &= lanes(t <= lh.y) & lanes(t >= lh.x);
                    if (a.has_norm) {   // wave-uniform; branch-free inside (1/scale = 0: no term)
                        const int i = 4 * kk + c;
                        const double2 li = sNA[i];
                        const double qq = (t - li.x) * li.y;
                        sc = sc + fma(-0.5 * qq, qq, sNM[i]);
                    }
                }
            }
            // PER: the wrapped coordinate of a periodic row (prior.py:675, the division by the
            // period as regs_div_by)
            auto wrapped = [&](int kk, double tk, double& fl) {
                const double4 pw = sPer[PER ? 4 * kk + c : 0];
                const double yv = regs_div_by(tk - pw.x, pw.z, pw.w);
                fl = floor(yv);
                return (yv - fl) * pw.z + pw.x;
            };
            if (PER) {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk)
                    if ((anyp >> kk) & 1u) {   // wave-uniform: a periodic dimension in this row
                        const bool per = (mine >> kk) & 1u;
                        const double tk = fma(r, col[4 * kk], x[kk]);
                        double fl;
                        const double tw = wrapped(kk, tk, fl);
                        const double4 pw = sPer[PER ? 4 * kk + c : 0];
                        inb &= lanes(!per | ((tw <= pw.y) & (tw >= pw.x)));
                        const double shk = (per & (fl != 0.0)) ? tw - tk : 0.0;
                        wound |= lanes(shk != 0.0);
                        if (per) myShift[slot_of(kk)] = shk;
                    }
                if (wound != 0ull) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes
            }
            // PER, a wrap in the wave: the trial residual of mode k -- fma(r, u, y) for every row,
            // then the wrap moves in ascending dimension
            auto shifted = [&](int k, const double* __restrict__ uk, const double (&yk)[DQ], double (&yt)[DQ]) {
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk) yt[kk] = fma(r, uk[4 * kk], yk[kk]);
#pragma unroll 1
                for (int q = 0; q < np; ++q) {
                    const double sv = myShift[q];                // the same in the walker's quad
                    if (lanes(sv != 0.0) == 0ull) continue;      // wave-uniform
                    const int i = sPdim[PER ? q : 0];
                    const double* __restrict__ lc = sLc + ((size_t)k * np + q) * dpad + c;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const int j = 4 * kk + c;
                        const bool on = (sv != 0.0) & (j >= i) & (j < d);
                        yt[kk] = on ? fma(sv, lc[4 * kk], yt[kk]) : yt[kk];
                    }
                }
            };
            double ak[KM], amax = -INFINITY;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                ak[k] = -INFINITY;
                if (k < K) {   // wave-uniform
                    const double* __restrict__ uk = col + (1 + k) * dpad;
                    double pc = 0.0;
                    if (PER && wound != 0ull) {   // wave-uniform
                        double yt[DQ];
                        shifted(k, uk, y[k], yt);
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) pc = fma(yt[kk], yt[kk], pc);
                    } else {
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double yt = fma(r, uk[4 * kk], y[k][kk]);
                        pc = fma(yt, yt, pc);
                    }
                    }
                    ak[k] = -0.5 * (sMode[k].x + quad_sum(pc));
                    amax = fmax(ak[k], amax);
                }
            }
            const unsigned long long inside_m = quad_all_mask(inb);
            const double lp = s.uniform_logp + (a.has_norm ? quad_sum(sc) : 0.0);
            // one exponential per lane and four modes: lane class q takes the modes 4 j + q, and
            // the weighted sum gathers them by quad broadcasts in the order of the specification
            double ll = ak[0];
            if (K > 1) {   // wave-uniform
                constexpr int NJ = (KM + 3) / 4;
                double e[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (4 * j < K) {
                        double mine = ak[4 * j];
                        if (4 * j + 1 < KM) mine = sel(class1, ak[4 * j + 1 < KM ? 4 * j + 1 : 0], mine);
                        if (4 * j + 2 < KM) mine = sel(class2, ak[4 * j + 2 < KM ? 4 * j + 2 : 0], mine);
                        if (4 * j + 3 < KM) mine = sel(class3, ak[4 * j + 3 < KM ? 4 * j + 3 : 0], mine);
                        e[j] = dexp_tab(mine - amax, etab);
                    }
                double Ssum = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (4 * j + 0 < K) Ssum = fma(sMode[4 * j + 0].y, quad_perm<0x00>(e[j]), Ssum);
                    if (4 * j + 1 < K) Ssum = fma(sMode[4 * j + 1].y, quad_perm<0x55>(e[j]), Ssum);
                    if (4 * j + 2 < K) Ssum = fma(sMode[4 * j + 2].y, quad_perm<0xAA>(e[j]), Ssum);
                    if (4 * j + 3 < K) Ssum = fma(sMode[4 * j + 3].y, quad_perm<0xFF>(e[j]), Ssum);
                }
                ll = dlog_tab(Ssum, slog) + amax;
            }
            const double lt = lp + ll;
            const double delta = (lpost - lt) / s.temperature;   // (T = 1: x / 1.0 == x)
            const unsigned long long acc_m =
                inside_m & lanes(lt != -INFINITY) & (lanes(lt > lpost) | lanes(Ea > delta));
            const bool accept = __builtin_amdgcn_inverse_ballot_w64(acc_m);
            if (s.rows) {   // wave-uniform: the point the walker leaves, with its weight
                bool em = accept & (burn <= 0);
                if (lanes(em) != 0ull) {   // some walker of the wave emits
                    int ew = wt;   // the weight the row is written with
                    if (thinning) {
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
                        double* __restrict__ row = s.rows + ((size_t)w * s.row_cap + nrow) * (size_t)(d + 4);
                        row[c] = c == 0 ? (double)ew : c == 1 ? lpost : c == 2 ? lpri : llik;
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk)
                            if (4 * kk + c < d) row[4 + 4 * kk + c] = x[kk];
                    }
                    nrow += em ? 1 : 0;   // rows beyond the capacity are counted as dropped
                }
            }
            int lim = lim1;
            if (burning) {   // wave-uniform (see step_inc_kernel)
                lim = burn > 0 ? lim10 : lim1;
                burn -= (accept & (burn > 0)) ? 1 : 0;
                burning = lanes(burn > 0) != 0ull;
            }
            const double ra = sel(acc_m, r, 0.0);
            const lds_doubles col2 = relaunder(col);
#pragma unroll
            for (int kk = 0; kk < DQ; ++kk) x[kk] = fma(ra, col2[4 * kk], x[kk]);
            if (PER) {   // an accepted periodic coordinate holds the moved value: wrapped now
#pragma unroll
                for (int kk = 0; kk < DQ; ++kk)
                    if ((anyp >> kk) & 1u) {   // wave-uniform
                        double fl;
                        const double tw = wrapped(kk, x[kk], fl);
                        x[kk] = (accept & (bool)((mine >> kk) & 1u)) ? tw : x[kk];
                    }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {   // wave-uniform
                    if (PER && wound != 0ull) {   // wave-uniform: the residual that took the wrap moves
                        double yt[DQ];
                        shifted(k, col + (1 + k) * dpad, y[k], yt);
#pragma unroll
                        for (int kk = 0; kk < DQ; ++kk) y[k][kk] = accept ? yt[kk] : y[k][kk];
                    } else {
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk)
                        y[k][kk] = fma(ra, col2[(1 + k) * dpad + 4 * kk], y[k][kk]);
                    }
                }
            lpri = sel(acc_m, lp, lpri);
            llik = sel(acc_m, ll, llik);
            lpost = sel(acc_m, lt, lpost);
            prej = sel(acc_m, 0, prej + sel(inside_m, 0, 1));
            wt = sel(acc_m, 1, wt + 1);
            nacc += sel(acc_m, 1, 0);
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
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) a.y[((size_t)k * d + i) * W + w] = y[k][kk];
        }
    }
    if (c == 0) {
        s.logpost[w] = lpost; s.logprior[w] = lpri; s.loglike[w] = llik;
        s.weight[w] = wt; s.prior_rej[w] = prej; s.burn_left[w] = burn;
        s.n_accept[w] = nacc0 + nacc;
        if (s.rows) s.n_rows[w] = nrow;
        if (thinning) s.thin_acc[w] = tacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc : 0);
}

// LDS of the periodic instantiations behind the column chunks: the wrap moves and the columns of
// L_k^-1 of the periodic dimensions
inline size_t regs_periodic_lds(int K, int dq, int np) { return sizeof(double) * (size_t)np * (64 + (size_t)K * 4 * dq); }

template <int DQ, int KM>
hipError_t launch_regs(const IncStepArgs& a, int np, hipStream_t st)
{
    constexpr int C = regs_chunk(DQ, KM);
    const size_t lds = sizeof(double) * 2 * C * (1 + KM) * 4 * DQ + regs_periodic_lds(a.n_modes, DQ, np);
    const std::string stem = "mcmc::step_inc_regs_kernel<" + std::to_string(DQ) + ", " + std::to_string(KM);
    static const std::string names[4] = {stem + ">", stem + ", emit>", stem + ", periodic>",
                                         stem + ", periodic, emit>"};
    const std::string& name = names[(a.s.rows ? 1 : 0) + (np > 0 ? 2 : 0)];
    auto kern = np > 0 ? step_inc_regs_kernel<DQ, KM, true> : step_inc_regs_kernel<DQ, KM, false>;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(name.c_str());
    hipLaunchKernelGGL(kern, dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}

// the (DQ, KM) this translation unit instantiates: every DQ that fits, for its KM
template <int DQ, int KM>
hipError_t dispatch_regs(const IncStepArgs& a, int np, hipStream_t st)
{
    if constexpr (DQ > 32 || !regs_fits(DQ, KM)) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) return launch_regs<DQ, KM>(a, np, st);
        return dispatch_regs<DQ + 1, KM>(a, np, st);
    }
}

inline int regs_count_periodic(const IncStepArgs& a)
{
    int n = 0;
    for (int q = 0; q < 4; ++q) n += __builtin_popcount(a.periodic_mask4[q]);
    return n;
}

#if ANY_PART == 0
// (v, u_k = L_k^-1 v) of every step of the launch as PLANES: VU[g][step][0] = v, [1 + k] = u_k,
// each 4 dq doubles (zero beyond d); one ascending fma chain from +0.0 per element
// (orc_whiten_directions).  One thread per (column, mode): 64 columns per workgroup, the modes
// dealt to blockIdx.z.
__global__ void __launch_bounds__(64) whiten_directions_planes_kernel(const IncDirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sv[];   // [d][64]
    const int l = threadIdx.x;
    const int sr = blockIdx.x * 64 + l;
    const int g = blockIdx.y, k = blockIdx.z;
    const int d = a.d, K = a.n_modes, dpad = 4 * a.dq;
    if (sr >= a.n_steps) return;
    const unsigned long long step = a.step0 + (unsigned long long)sr;
    const int cyc = (int)(step / (unsigned long long)a.cps - a.cycle0);
    const int col = (int)(step % (unsigned long long)a.cps);
    const double* __restrict__ v = a.V + ((size_t)g * a.ncyc + cyc) * a.slab + (size_t)col * a.ld;
    for (int i = 0; i < d; ++i) sv[i * 64 + l] = v[i];
    double* __restrict__ out = a.VU + ((size_t)g * a.n_steps + sr) * (size_t)((1 + K) * dpad);
    if (k == 0) {
        if (a.colflag)
            a.colflag[(size_t)g * a.n_steps + sr] =
                a.vflag ? a.vflag[((size_t)g * a.ncyc + cyc) * a.cps + col] : 0;
        for (int j = 0; j < dpad; ++j) out[j] = j < d ? sv[j * 64 + l] : 0.0;
    }
    double* __restrict__ uk = out + (size_t)(1 + k) * dpad;
    for (int j = 0; j < d; ++j) {
        const double* __restrict__ row = a.Lrow + ((size_t)k * d + j) * d;
        double acc = 0.0;
        for (int i = 0; i <= j; ++i) acc = fma(row[i], sv[i * 64 + l], acc);
        uk[j] = acc;
    }
    for (int j = d; j < dpad; ++j) uk[j] = 0.0;
}

template <int DQ>
hipError_t launch_any(const IncStepArgs& a, const AnyGeom& g, hipStream_t st)
{
    static const std::string names[2] = {"mcmc::step_inc_any_kernel<" + std::to_string(DQ) + ">",
                                         "mcmc::step_inc_any_kernel<" + std::to_string(DQ) + ", emit>"};
    const std::string& name = names[a.s.rows ? 1 : 0];
    auto kern = step_inc_any_kernel<DQ>;
    if (g.dynamic > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)g.dynamic);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(name.c_str());
    hipLaunchKernelGGL(kern, dim3(a.s.W / (16 * g.nw)), dim3(64 * g.nw), g.dynamic, st, a, g.C, g.npad,
                       g.lcols);
    return hipGetLastError();
}

template <int DQ>
hipError_t dispatch_any(const IncStepArgs& a, const AnyGeom& g, hipStream_t st)
{
    if constexpr (DQ > 32) {
        return hipErrorInvalidValue;
    } else {
        if (a.dq == DQ) return launch_any<DQ>(a, g, st);
        return dispatch_any<DQ + 1>(a, g, st);
    }
}

int count_periodic(const IncStepArgs& a)
{
    int n = 0;
    for (int q = 0; q < 4; ++q) n += __builtin_popcount(a.periodic_mask4[q]);
    return n;
}

#endif  // ANY_PART == 0

}  // namespace
}  // namespace mcmc

#if ANY_PART == 1
extern "C" hipError_t mcmc_hip_launch_inc_regs_8(const mcmc::IncStepArgs* a, hipStream_t st)
{
    return mcmc::dispatch_regs<1, 8>(*a, mcmc::regs_count_periodic(*a), st);
}
#elif ANY_PART == 2
extern "C" hipError_t mcmc_hip_launch_inc_regs_16(const mcmc::IncStepArgs* a, hipStream_t st)
{
    return mcmc::dispatch_regs<1, 16>(*a, mcmc::regs_count_periodic(*a), st);
}
#else
extern "C" hipError_t mcmc_hip_launch_inc_regs_8(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_regs_16(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));

namespace mcmc {
namespace {
// the register kernel serves what capi.hip sends here -- mixtures that the tuned
// step_inc_mix_kernel (K <= 4, dq <= 16) does not serve, and emitted rows (`emit: chains`) of
// anything but the one Gaussian mode of step_inc_kernel<.., EMIT> -- without periodic parameters,
// as far as the registers hold the residuals
bool regs_serves(int K, int dq, int n_periodic)
{
    if (K < 1 || K > 16) return false;   // (more than 16 modes: the LDS kernel, if its geometry fits)
    const int km = regs_bucket(K);
    if (!regs_fits(dq, km)) return false;
    // periodic parameters: their wrap moves and columns of L_k^-1 beside the column chunks in LDS
    if (n_periodic > 0 && regs_periodic_lds(K, dq, n_periodic) > (24u << 10)) return false;
    return km <= 4 || (km == 8 ? mcmc_hip_launch_inc_regs_8 != nullptr : mcmc_hip_launch_inc_regs_16 != nullptr);
}
}  // namespace
}  // namespace mcmc

// 1 if the general incremental kernels serve (d, n_modes, n_periodic) at this ensemble shape
extern "C" int mcmc_hip_inc_any_fits(int d, int n_modes, int n_periodic, int n_walkers, int group_size)
{
    mcmc::AnyGeom g{};
    if (d < 2 || d > 128 || n_modes < 1 || n_modes > mcmc::kMaxModes) return 0;
    if (n_walkers % 64 == 0 && group_size % 64 == 0 && mcmc::regs_serves(n_modes, (d + 3) / 4, n_periodic))
        return 1;
    return mcmc::any_geometry(n_modes, (d + 3) / 4, n_periodic, n_walkers, group_size, g) ? 1 : 0;
}

extern "C" hipError_t mcmc_hip_launch_inc_any(const mcmc::IncStepArgs* a, hipStream_t st)
{
    const int np = mcmc::count_periodic(*a);
    if (a->s.W % 64 == 0 && a->s.group_size % 64 == 0 && mcmc::regs_serves(a->n_modes, a->dq, np)) {
        const int km = mcmc::regs_bucket(a->n_modes);
        return km == 2 ? mcmc::dispatch_regs<1, 2>(*a, np, st)
             : km == 4 ? mcmc::dispatch_regs<1, 4>(*a, np, st)
             : km == 8 ? mcmc_hip_launch_inc_regs_8(a, st) : mcmc_hip_launch_inc_regs_16(a, st);
    }
    mcmc::AnyGeom g{};
    if (!mcmc::any_geometry(a->n_modes, a->dq, np, a->s.W, a->s.group_size, g))
        return hipErrorInvalidValue;
    return mcmc::dispatch_any<1>(*a, g, st);
}

extern "C" hipError_t mcmc_hip_launch_whiten_directions_planes(const mcmc::IncDirArgs* a, int n_groups,
                                                               hipStream_t st)
{
    const size_t lds = sizeof(double) * 64 * (size_t)a->d;
    if (lds > 40 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_directions_planes_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mcmc::whiten_directions_planes_kernel,
                       dim3((a->n_steps + 63) / 64, n_groups, a->n_modes), dim3(64), lds, st, *a);
    return hipGetLastError();
}
#endif  // ANY_PART
