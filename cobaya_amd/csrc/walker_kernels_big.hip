// Walker kernels for 32 < d <= MCMC_DP (gfx950); one translation unit per padded size DP, the
// actual d is a run-time argument <= DP (rows / columns beyond d are zero operands: exact no-ops).
//
//   basis_big_kernel      Haar directions V = T R per (group, cycle): four lanes per row of H,
//                         barrier-free reflections, row-blocked product
//   step_mfma_kernel<N>   the step for ensembles of whole 256-walker workgroups: y = L^-1 dev on
//                         the matrix cores (v_mfma_f64_16x16x4_f64), 16 walkers per wave; N = with
//                         normal priors.  One Gaussian mode, non-periodic priors, no emitted rows.
//   step_big_reg_kernel   (DP <= 112) lane-per-walker fallback for other ensemble sizes, uniform
//                         priors: a column sweep over 4x4 tiles of L^-1 broadcast from LDS, the
//                         accumulators y_j in VGPRs and x in the accumulation half of the
//                         register file; each y_j is the same ascending fma chain
//   evaluate_big_kernel   batch log-prior / log-likelihood (general: mixtures, normal priors)
//   group_moments_big_kernel, pool_moments_big_kernel   streaming sufficient statistics; the
//                         group passes through LDS in slices, so any group size serves any d
// Every sum follows the d > 32 order of the specification (four interleaved chains), so the
// results equal the oracle's bit for bit.  What these kernels leave out (mixtures, periodic
// parameters, emitted rows, ...) runs on general_kernels.hip; for 32 < d <= 56 the step of whole
// 256-walker workgroups is served by walker_kernels.hip's two-wave kernel instead.
#include <type_traits>
#include "det_math.h"
#include "kernels.h"

#ifndef MCMC_DP
#error "compile with -DMCMC_DP=<max dimension>"
#endif

namespace mcmc {
namespace {

constexpr int DP = MCMC_DP;
constexpr int NB = DP / 4;
static_assert(DP % 4 == 0 && DP <= 128, "DP must be a multiple of 4, at most 128");

// ---------------------------------------------------------------- Haar basis (run-time d <= DP)
// Same construction as basis_kernel / orc_haar_from_normals, in the d > 32 order of the
// specification: the projection of a row on a reflector is formed as FOUR chains over the columns
// j = c (mod 4), combined (t0 + t1) + (t2 + t3) -- so that FOUR lanes serve one row of H, lane
// class c holding its columns j = c (mod 4) in DP/4 registers.
//   A  the (d+2)(d-1)/2 normals (Philox + Box-Muller), all threads;
//   B  thread n: norm, sign, pivot and denominator of reflection n (sequential chain, all
//      reflections in parallel); then every reflector normalised and stored zero-padded to the
//      DP layout (reflection n at n DP - n(n-1)/2, indexed by the absolute column);
//   C  the d-1 reflections: no barrier -- a row needs only its own four lanes (DPP) and the
//      reflectors, which are read-only by now;
//   D  R = D H to LDS (over the space of the normals), then V = T R: a wave takes 4 rows i of T
//      (wave-uniform, scalar loads) x 64 columns c (lanes) at a time, four independent chains
//      over k in the specified ascending order.
constexpr int kBasisRows = (DP + 15) / 16 * 16;   // whole waves: 16 rows x 4 classes
constexpr int kBasisThreads = 4 * kBasisRows;
constexpr int kBasisQ = DP / 4;
__host__ __device__ constexpr int refl_offset(int n, int dim) { return n * dim - n * (n - 1) / 2; }

__device__ __forceinline__ double quad_bcast(double v, int cc)
{
    const int lane = __lane_id();
    return __shfl(v, (lane & ~3) | cc);
}

__global__ void __launch_bounds__(kBasisThreads) basis_big_kernel(const BasisArgs a, int d)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nz = (d + 2) * (d - 1) / 2;
    constexpr int nzp = refl_offset(DP - 1, DP) + 1;   // padded reflectors: sum_{n<DP-1} (DP-n) (+1)
    double* sz = smem;                        // [nz + 2] normals
    double* sxp = sz + ((nz + 3) & ~1);       // [nzp] normalised reflectors, DP layout
    double* sDn = sxp + ((nzp + 1) & ~1);     // [DP] sign D_n
    double* sPv = sDn + DP;                   // [DP] pivot x0 + D_n sqrt(norm2)
    double* sDen = sPv + DP;                  // [DP] denominator
    double* sH = smem;                        // [d][d] overlays everything, final phase only
    const int tid = threadIdx.x, nt = kBasisThreads;
    const int t = tid >> 2, c = tid & 3;
    const uint32_t group = a.group0 + blockIdx.x;
    const uint32_t cycle = a.cycle0 + blockIdx.y;
    const int ldv = v_ld(d);
    double* __restrict__ Vout = a.V + ((size_t)blockIdx.x * a.ncyc + blockIdx.y) * v_slab_big(d);
    typedef const double __attribute__((address_space(4))) * cptr;
    const cptr T = (cptr)(unsigned long long)a.T;

    // ---- A
    for (int j = tid; 2 * j < nz; j += nt) {
        const u32x4 w4 = philox4x32_10(a.key0, a.key1, group, kStreamBasis, cycle, (uint32_t)j);
        const uint64_t ka = ((uint64_t)w4.w0 << 20) | (w4.w1 >> 12);
        const uint64_t kb = ((uint64_t)w4.w2 << 20) | (w4.w3 >> 12);
        const double rad = sqrt(-2.0 * dlog(u52(ka)));
        double sn, cs;
        sincos2pi(kb, sn, cs);
        sz[2 * j] = rad * cs;
        sz[2 * j + 1] = rad * sn;
    }
    __syncthreads();
    // ---- B
    if (tid < d - 1) {
        const int n = tid, m = d - n;
        const int ix = refl_offset(n, d);
        double norm2 = 0.0;
#pragma unroll 8
        for (int k = 0; k < m; ++k) norm2 = fma(sz[ix + k], sz[ix + k], norm2);
        const double x0 = sz[ix];
        const double Dn = (x0 < 0.0) ? -1.0 : 1.0;
        const double x0n = x0 + Dn * sqrt(norm2);
        double tt = norm2 - x0 * x0;
        tt = tt + x0n * x0n;
        sDn[n] = Dn;
        sPv[n] = x0n;
        sDen[n] = sqrt(0.5 * tt);
    }
    __syncthreads();
    {   // wave w normalises the reflectors n = w, w + #waves, ...; lanes over the columns
        const int lane = tid & 63, wv = tid >> 6;
        for (int n = wv; n < DP - 1; n += nt / 64) {
            const bool live = n < d - 1;
            const int m = d - n;
            const double pv = live ? sPv[n] : 0.0, den = live ? sDen[n] : 1.0;
            const int ix = live ? refl_offset(n, d) : 0;
            for (int k = lane; k < DP - n; k += 64)
                sxp[refl_offset(n, DP) + k] =
                    (live && k < m) ? ((k == 0) ? pv : sz[ix + k]) / den : 0.0;
        }
    }
    double Dmine = 1.0;   // sign of row t: D_t, the last one closes det = +1
    {
        double dprod = 1.0;   // product of the signs, in order (exact: +-1)
        for (int n = 0; n < d - 1; ++n) dprod *= sDn[n];
        if (t < d - 1) Dmine = sDn[t];
        else if (t == d - 1) Dmine = (((d - 1) & 1) ? -1.0 : 1.0) * dprod;
    }
    __syncthreads();
    // ---- C
    double h[kBasisQ];    // h[q] = H[t][4 q + c]
#pragma unroll
    for (int q = 0; q < kBasisQ; ++q) h[q] = (4 * q + c == t) ? 1.0 : 0.0;
#pragma unroll
    for (int n = 0; n < DP - 1; ++n) {
        if (n < d - 1) {   // uniform (the padded reflectors beyond are zero anyway)
            const double* __restrict__ xr = sxp + (refl_offset(n, DP) - n);   // xr[j], j >= n
            const int q0 = n / 4;
            double xv[kBasisQ];
            double tmp = 0.0;
#pragma unroll
            for (int q = q0; q < kBasisQ; ++q) {
                const int j = 4 * q + c;
                // the first group of four straddles column n: the columns below it are not part
                // of the reflector (a zero term is an exact no-op)
                xv[q] = (q == q0 && (n & 3) && j < n) ? 0.0 : xr[j];
                tmp = fma(h[q], xv[q], tmp);
            }
            const double t0 = quad_bcast(tmp, 0), t1 = quad_bcast(tmp, 1), t2 = quad_bcast(tmp, 2),
                         t3 = quad_bcast(tmp, 3);
            const double tot = (t0 + t1) + (t2 + t3);
#pragma unroll
            for (int q = q0; q < kBasisQ; ++q) h[q] = fma(-tot, xv[q], h[q]);
        }
    }
    __syncthreads();      // normals and reflectors are dead: R takes their place in LDS
    // ---- D
    if (t < d) {
#pragma unroll
        for (int q = 0; q < kBasisQ; ++q)
            if (4 * q + c < d) sH[t * d + 4 * q + c] = Dmine * h[q];
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nrb = (d + 3) / 4, nct = (d + 63) / 64;
        for (int it = wv; it < nrb * nct; it += nt / 64) {
            const int rb = it / nct, ct = it - rb * nct;
            const int i0 = 4 * rb;
            const int col = 64 * ct + lane;
            const int cc = col < d ? col : d - 1;
            // rows i0 .. i0+3 (clamped: a duplicate row is computed and not stored)
            const int r1 = i0 + 1 < d ? i0 + 1 : d - 1, r2 = i0 + 2 < d ? i0 + 2 : d - 1,
                      r3 = i0 + 3 < d ? i0 + 3 : d - 1;
            const cptr T0 = T + i0 * d, T1 = T + r1 * d, T2 = T + r2 * d, T3 = T + r3 * d;
            const int K = (i0 + 4 < d) ? i0 + 4 : d;   // T is lower triangular: zeros above
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 8
            for (int k = 0; k < K; ++k) {
                const double hk = sH[k * d + cc];
                s0 = fma(T0[k], hk, s0);
                s1 = fma(T1[k], hk, s1);
                s2 = fma(T2[k], hk, s2);
                s3 = fma(T3[k], hk, s3);
            }
            if (col < d) {
                double* o = Vout + (size_t)col * ldv + i0;
                o[0] = s0;
                if (i0 + 1 < d) o[1] = s1;
                if (i0 + 2 < d) o[2] = s2;
                if (i0 + 3 < d) o[3] = s3;
            }
        }
    }
}

// ---------------------------------------------------------------- the column-sweep kernel
// Lane-per-walker fallback for ensembles that are not a multiple of 256 walkers; needs 2 * DP
// doubles of registers per lane, hence not compiled for DP > 112.
#if MCMC_DP <= 112
struct BigStepArgs {
    StepArgs s;
    const double* Lcol;  // [DP/4 column blocks][DP/4 row blocks][4 cols][4 rows] tiles of L^-1
    int d;
};

// x and the accumulators both stay in registers for the whole launch (the compiler parks x in
// the accumulation half of the unified 512-entry register file and moves an element through
// v_accvgpr_read/write only a few times per step), every loop is static, and n_steps are
// fused: nothing but the direction column crosses the memory system inside a launch.  Needs
// 2 * DP doubles of registers per lane: DP <= 112.
// The 4x4 tiles a pass touches, in the order it touches them (column block ascending, then
// row block ascending), as a compile-time table.
template <int RB0, int RB1>
struct TileSeq {
    static constexpr int count()
    {
        int n = 0;
        for (int cb = 0; cb < RB1; ++cb)
            for (int B = (cb > RB0 ? cb : RB0); B < RB1; ++B) ++n;
        return n;
    }
    static constexpr int N = count();
    unsigned char cb[N];
    unsigned char rb[N];
};
template <int RB0, int RB1>
constexpr TileSeq<RB0, RB1> make_tile_seq()
{
    TileSeq<RB0, RB1> t{};
    int n = 0;
    for (int cb = 0; cb < RB1; ++cb)
        for (int B = (cb > RB0 ? cb : RB0); B < RB1; ++B) {
            t.cb[n] = (unsigned char)cb;
            t.rb[n] = (unsigned char)B;
            ++n;
        }
    return t;
}
template <int RB0, int RB1>
__device__ constexpr TileSeq<RB0, RB1> kTileSeq = make_tile_seq<RB0, RB1>();

typedef const double __attribute__((address_space(3))) * lptr;
// LDS tile pointer passed through an empty asm together with a value computed just before:
// the tile's loads can be issued neither earlier (all 2500 of them hoisted to the top and
// spilled) nor later (latency exposed) than this point of the instruction stream.
__device__ __forceinline__ lptr after(lptr p, double& anchor)
{
    unsigned v = (unsigned)(unsigned long long)p;
    asm volatile("; next tile" : "+v"(v), "+v"(anchor));
    return (lptr)__builtin_assume_aligned((lptr)(unsigned long long)v, 16);
}

// One pass over all columns for the row blocks [RB0, RB1): with only half of the accumulators
// live, the other half of the VGPRs holds L^-1 tiles in flight: tile k+2 is requested while
// tile k is consumed (LDS returns in order, so the waits are counted, not drained).  dev is
// recomputed per pass (cheap) rather than kept.  Columns beyond d carry dev = 0 and zero tiles.
template <int RB0, int RB1>
__device__ __forceinline__ void sweep_pass(double (&y)[4 * (RB1 - RB0)], const double (&x)[DP],
                                           double r, lptr v, lptr sE, lptr sL, int d)
{
    constexpr int N = TileSeq<RB0, RB1>::N;
    const auto& seq = kTileSeq<RB0, RB1>;
    auto tile_ptr = [&](int k) { return sL + (seq.cb[k] * NB + seq.rb[k]) * 16; };
    double cur[16], n1[16], n2[16];
    double anchor0 = r;
    {
        const lptr p0 = after(tile_ptr(0), anchor0);
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = p0[e];
        if (N > 1) {
            const lptr p1 = after(tile_ptr(1), anchor0);
#pragma unroll
            for (int e = 0; e < 16; ++e) n1[e] = p1[e];
        }
    }
    double dev[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int cb = seq.cb[k], B = seq.rb[k];
        if (k == 0 || seq.cb[k - 1] != cb) {  // first tile of a column block: its four dev
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * cb + u;
                const double ti = fma(r, v[i], x[i]);
                const double dv = ((ti <= sE[4 * i + 1]) & (ti >= sE[4 * i + 0])) ? ti - sE[4 * i + 2]
                                                                              : INFINITY;
                dev[u] = (i < d) ? dv : 0.0;
            }
        }
        double* yy = y + 4 * (B - RB0);
        yy[0] = fma(cur[0], dev[0], yy[0]);
        if (k + 2 < N) {
            const lptr p2 = after(tile_ptr(k + 2), yy[0]);
#pragma unroll
            for (int e = 0; e < 16; ++e) n2[e] = p2[e];
        }
        yy[0] = fma(cur[4], dev[1], yy[0]);
        yy[0] = fma(cur[8], dev[2], yy[0]);
        yy[0] = fma(cur[12], dev[3], yy[0]);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            double acc = yy[q];
            acc = fma(cur[0 * 4 + q], dev[0], acc);
            acc = fma(cur[1 * 4 + q], dev[1], acc);
            acc = fma(cur[2 * 4 + q], dev[2], acc);
            acc = fma(cur[3 * 4 + q], dev[3], acc);
            yy[q] = acc;
        }
        asm volatile("; tile end" : "+v"(yy[0]), "+v"(yy[1]), "+v"(yy[2]), "+v"(yy[3]));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            cur[e] = n1[e];
            n1[e] = n2[e];
        }
    }
}

__global__ void __launch_bounds__(256) step_big_reg_kernel(const BigStepArgs b)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const lptr lsm = (lptr)__builtin_assume_aligned(
        (const __attribute__((address_space(3))) double*)smem, 16);
    const StepArgs& a = b.s;
    const int d = b.d;
    const int tid = threadIdx.x, bs = blockDim.x;
    const int w = blockIdx.x * bs + tid;
    const int W = a.W;
    const int group = __builtin_amdgcn_readfirstlane(w / a.group_size);
    const int gpb = bs / a.group_size;
    const int gib = __builtin_amdgcn_readfirstlane(tid / a.group_size);
    const int wpg = a.group_size >> 6;
    const int part = __builtin_amdgcn_readfirstlane((tid >> 6) % wpg);
    const ConstLayout cl{d, 1};
    double* sL = smem;           // L^-1 tiles [DP*DP]
    double* sE = sL + DP * DP;   // {lo, hi, mu, 0}[DP]
    double* sVr = sE + 4 * DP;   // direction ring [2][gpb][128]
    for (int i = tid; i < DP * DP; i += bs) sL[i] = b.Lcol[i];
    for (int i = tid; i < DP; i += bs) {
        const bool in = i < d;
        sE[4 * i + 0] = in ? a.cblock[cl.lo() + i] : -INFINITY;
        sE[4 * i + 1] = in ? a.cblock[cl.hi() + i] : INFINITY;
        sE[4 * i + 2] = in ? a.cblock[cl.mean(0) + i] : 0.0;
        sE[4 * i + 3] = 0.0;
    }
    const int ldv = v_ld(d);
    const double* const Vgrp = a.V + (size_t)group * a.ncyc * v_slab_big(d);
    auto stage_col = [&](int cycle, int column, int slot) {
        if (part == 0) {
            const char* g = (const char*)(Vgrp + (size_t)cycle * v_slab_big(d) + (size_t)column * ldv) +
                            (tid & 63) * 16;
            char* l = (char*)(sVr + (slot * gpb + gib) * 128);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    unsigned long long step = a.step0;
    int col = (int)(step % (unsigned long long)d);
    int cyc = 0;
    stage_col(0, col, 0);
    double x[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) x[i] = (i < d) ? a.x[(size_t)i * W + w] : 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int slot = 0;

    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    const uint32_t gid = a.walker0 + (uint32_t)w;

    for (int s = 0; s < a.n_steps; ++s) {
        {
            int ncol = col + 1, ncyc = cyc;
            if (ncol == d) { ncol = 0; ++ncyc; }
            if (s + 1 < a.n_steps) stage_col(ncyc, ncol, slot ^ 1);
        }
        StepRng rng;
        rng.begin(a.key0, a.key1, gid, step);
        rng.run_all();
        const double r = rng.r, Ea = rng.Ea;
        const double* __restrict__ v = sVr + (slot * gpb + gib) * 128;
        const lptr lv = lsm + (DP * DP + 4 * DP) + (slot * gpb + gib) * 128;
        const lptr lE = lsm + DP * DP;
        // chi2: four interleaved chains over the rows j = c (mod 4) (the specification for
        // d > 32, see step_mfma_kernel), rows of both passes in ascending order
        double pc[4] = {0.0, 0.0, 0.0, 0.0};
        {
            constexpr int H = (NB + 1) / 2;
            double y[4 * H];
#pragma unroll
            for (int j = 0; j < 4 * H; ++j) y[j] = 0.0;
            sweep_pass<0, H>(y, x, r, lv, lE, lsm, d);
#pragma unroll
            for (int j = 0; j < 4 * H; ++j) pc[j & 3] = fma(y[j], y[j], pc[j & 3]);
        }
        {
            constexpr int H = (NB + 1) / 2;
            double y[4 * (NB - H)];
#pragma unroll
            for (int j = 0; j < 4 * (NB - H); ++j) y[j] = 0.0;
            sweep_pass<H, NB>(y, x, r, lv, lE, lsm, d);
#pragma unroll
            for (int j = 0; j < 4 * (NB - H); ++j) pc[j & 3] = fma(y[j], y[j], pc[j & 3]);
        }
        const double chi2 = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        const bool inb = chi2 < INFINITY;
        const double lp = a.uniform_logp + 0.0;
        const double ll = -0.5 * (a.cnorm0 + chi2);
        const double lt = inb ? lp + ll : -INFINITY;
        const bool accept = inb & (lt != -INFINITY) &
                            ((lt > lpost) | (Ea > (lpost - lt) / a.temperature));
        burn -= (accept & (burn > 0)) ? 1 : 0;
        const double ra = accept ? r : 0.0;  // fma(0, v, x) == x exactly
#pragma unroll
        for (int i = 0; i < DP; ++i) x[i] = fma(ra, v[i], x[i]);
        lpri = accept ? lp : lpri;
        llik = accept ? ll : llik;
        lpost = accept ? lt : lpost;
        prej = accept ? 0 : (prej + (inb ? 0 : 1));
        wt = accept ? 1 : wt + 1;
        nacc += accept ? 1 : 0;
        if (!accept) {
            const double max_now = a.max_tries * (burn > 0 ? 10.0 : 1.0);
            if ((double)(wt - prej) > max_now) atomicCAS(a.stuck, 0, 1 + (int)gid);
        }
        ++step;
        if (++col == d) { col = 0; ++cyc; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next direction has landed
        __syncthreads();
        slot ^= 1;
    }
#pragma unroll
    for (int i = 0; i < DP; ++i)
        if (i < d) a.x[(size_t)i * W + w] = x[i];
    a.logpost[w] = lpost; a.logprior[w] = lpri; a.loglike[w] = llik;
    a.weight[w] = wt; a.prior_rej[w] = prej; a.burn_left[w] = burn;
    a.n_accept[w] = nacc;
    wave_add_accepts(a.accept_total, nacc - nacc0);
}

#endif  // MCMC_DP <= 112

// ---------------------------------------------------------------- the matrix-core Metropolis kernel
// y = L^-1 dev for 16 walkers at a time as FP64 MFMAs (v_mfma_f64_16x16x4_f64): the whitening
// of d > 32 is a triangular GEMM (rows x walkers), and the matrix core delivers its uniform
// operand -- the L^-1 tile -- to all 16 walkers from 64 lane registers, where the VALU form
// needs one LDS broadcast per two FMAs.  The instruction accumulates k in ascending order with
// one rounding per product-sum (verified bit for bit against an fma chain,
// tools/probes/mfma_f64_order.hip), so y_j is exactly the oracle's chain; tile entries above
// the diagonal are zero (fma(0, dev, y) == y).
//
// Layout.  A wave owns 16 walkers.  Lane l = 16 c + n serves walker n of the wave and the
// dimensions i = 4 kk + c (kk = 0 .. KT-1): x[kk] is lane-resident, and dev for k-step kk is
// exactly the MFMA's B operand (B[k = l >> 4][col = l & 15]).  A = the 16 x 4 tile
// (rows 16 R - kRowShift .., columns 4 kk ..) of L^-1 from LDS in lane order.  D: lane l holds
// rows 16 R - kRowShift + 4 r + c (r = 0..3) of walker n, i.e. rows j = c (mod 4): its chain p_c.
// chi2 = (p0 + p1) + (p2 + p3) through four cross-lane reads; all four lanes of a walker take
// the same decision and commit their own quarter of x.  16 waves (256 walkers) per workgroup,
// one workgroup per CU, so every SIMD holds four waves; the row tiles go in two passes so that
// x, the accumulators of one pass and the temporaries (nearly) fit the 128 registers that
// occupancy allows.  Left to the compiler's own schedule on purpose: pinning the LDS loads
// behind asm anchors (as the VALU kernels do), three passes, or parking the walker scalars in
// LDS all measured SLOWER (5.1-5.8 ms vs 4.07 ms per 200 steps at d = 100) -- the unpinned
// schedule pairs adjacent tiles into ds_read2st64_b64.
typedef double d4 __attribute__((ext_vector_type(4)));
// Row tile R holds the rows 16 R - kRowShift .. + 15: when DP is not a multiple of 16 the PARTIAL
// row tile is the FIRST one (rows < 0 are zero padding), where it costs 4 - kRowShift/4 tiles,
// instead of the last one, where it would cost KT (DP = 100: 91 tiles instead of 109).  The
// shift is a multiple of 4, so lane class c still holds the rows j = c (mod 4).
constexpr int kRowShift = (16 - (DP + 3) / 4 * 4 % 16) % 16;
constexpr int RT = (DP + kRowShift + 15) / 16;     // row tiles
constexpr int KT = (DP + 3) / 4;       // k-steps
constexpr int RH = (RT + 1) / 2;       // row tiles of the first pass (RT / 2 -- fewer repeated trial
                                       // deviations -- measured 14.95 against 14.0 ms at d = 100)
// tile (R, kk) exists for kk <= kk_max(R) (its last row is 16 R - kRowShift + 15); R-major
constexpr int kk_max(int R) { return 4 * R + 3 - kRowShift / 4; }
constexpr int tiles_of_row(int R) { return (kk_max(R) + 1 < KT) ? kk_max(R) + 1 : KT; }
constexpr int tile_offset(int R)
{
    int n = 0;
    for (int q = 0; q < R; ++q) n += tiles_of_row(q);
    return n;
}
constexpr int kTiles = tile_offset(RT);

struct MfmaStepArgs {
    StepArgs s;
    const double* Ltiles;  // [kTiles][64] tiles of L^-1 in A-operand lane order
    int d;
    uint32_t norm_mask4[4];  // one bit per dimension with a normal prior
};

// SEL: mask the trial deviation of the padded dimensions with a select.  The padded part of V is
// zero (zero_tail below), so the select is redundant -- but whether dropping it helps is decided
// by the register allocation it leads to, measured per instantiation (ms per 8 d steps, with /
// without): d = 80: 7.35 / 7.20, 100: 14.65 / 14.04, 112: 19.7 / 21.4, 120: 25.2 / 26.2; normal
// priors at d = 100: 2.5 % slower without.
template <int R0, int R1, bool SEL>
__device__ __forceinline__ void mfma_pass(d4 (&acc)[R1 - R0], const double (&x)[KT], double r,
                                          const double* __restrict__ sv,
                                          const double* __restrict__ sE,
                                          const double* __restrict__ sL, int c, int lane, int d)
{
#pragma unroll
    for (int q = 0; q < R1 - R0; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        if (kk > kk_max(R1 - 1)) continue;
        const int i = 4 * kk + c;
        const double ti = fma(r, sv[i], x[kk]);
        const double lo = sE[3 * i], hi = sE[3 * i + 1], mu = sE[3 * i + 2];
        // (a padded dimension i >= d has v = x = mu = 0 and infinite bounds: dev = 0)
        double dev = ((ti <= hi) & (ti >= lo)) ? ti - mu : INFINITY;
        if (SEL) dev = (i < d) ? dev : 0.0;
#pragma unroll
        for (int R = R0; R < R1; ++R) {
            if (kk > kk_max(R)) continue;
            const double a = sL[(tile_offset(R) + kk) * 64 + lane];
            acc[R - R0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, dev, acc[R - R0], 0, 0, 0);
        }
    }
}

// kMfmaWaves waves of 16 walkers per workgroup.  16 (256 walkers, one workgroup per CU, 128
// registers per lane, two passes with 73 spilled registers) measured 4.04 ms per 200 steps at
// d = 100; 8 (128 walkers, two workgroups per CU, 256 registers: one pass, no spills, no second
// evaluation of the trial) 4.33 ms -- four waves per SIMD hide more than the spills cost.
#ifndef MCMC_MFMA_WAVES
#define MCMC_MFMA_WAVES 16
#endif
#ifndef MCMC_MFMA_RNG4
#define MCMC_MFMA_RNG4 (MCMC_DP <= 80)
#endif
constexpr int kMfmaWaves = MCMC_MFMA_WAVES;
constexpr int kMfmaWalkers = 16 * kMfmaWaves;
constexpr int kMfmaThreads = 64 * kMfmaWaves;
constexpr int kMfmaFirstPass = kMfmaWaves <= 8 ? RT : RH;

// NORMP: some priors are normal (its own instantiation: the uniform-only kernel keeps its
// registers)
template <bool NORMP>
__global__ void __launch_bounds__(kMfmaThreads) step_mfma_kernel(const MfmaStepArgs b)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const StepArgs& a = b.s;
    const int d = b.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int c = lane >> 4;
    const int wl = wv * 16 + (lane & 15);
    const int w = blockIdx.x * kMfmaWalkers + wl;
    const int W = a.W;
    const int group = __builtin_amdgcn_readfirstlane(w / a.group_size);
    const int gpb = (kMfmaWalkers + a.group_size - 1) / a.group_size;  // groups touching the block
    const int gib = __builtin_amdgcn_readfirstlane(wl / a.group_size);
    const bool stager = __builtin_amdgcn_readfirstlane(wl % a.group_size) == 0;
    const ConstLayout cl{d, 1};
    double* sL = smem;
    double* sE = sL + kTiles * 64;
    double* sVr = sE + 3 * 4 * KT;
    constexpr bool has_norm = NORMP;
    constexpr bool kSel = NORMP || DP > 100;   // see mfma_pass
    double* sN = sVr + 2 * gpb * 128;   // {loc, scale, mls}[4 KT]; scale = +inf marks "uniform"
    if (has_norm)
        for (int i = tid; i < 4 * KT; i += kMfmaThreads) {
            const bool nrm = i < d && ((b.norm_mask4[i >> 5] >> (i & 31)) & 1u);
            sN[3 * i + 0] = nrm ? a.cblock[cl.loc() + i] : 0.0;
            sN[3 * i + 1] = nrm ? a.cblock[cl.scale() + i] : INFINITY;
            sN[3 * i + 2] = nrm ? a.cblock[cl.mls() + i] : 0.0;
        }
    for (int i = tid; i < kTiles * 64; i += kMfmaThreads) sL[i] = b.Ltiles[i];
    for (int i = tid; i < 4 * KT; i += kMfmaThreads) {
        const bool in = i < d;
        sE[3 * i + 0] = in ? a.cblock[cl.lo() + i] : -INFINITY;
        sE[3 * i + 1] = in ? a.cblock[cl.hi() + i] : INFINITY;
        sE[3 * i + 2] = in ? a.cblock[cl.mean(0) + i] : 0.0;
    }
    const int ldv = v_ld(d);
    const double* const Vgrp = a.V + (size_t)group * a.ncyc * v_slab_big(d);
    auto stage_col = [&](int cycle, int column, int slot) {
        if (stager) {
            const char* g = (const char*)(Vgrp + (size_t)cycle * v_slab_big(d) + (size_t)column * ldv) +
                            lane * 16;
            char* l = (char*)(sVr + (slot * gpb + gib) * 128);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    // The DMA brings 128 doubles: beyond the d of the column lies the next column.  The padded
    // dimensions d .. 4 KT - 1 are zeroed by the staging wave once its DMA has landed, so that
    // the `i < d` select on the trial deviation can be dropped (mfma_pass<.., SEL>).
    auto zero_tail = [&](int slot) {
        if (stager)
            for (int k = d + lane; k < 4 * KT; k += 64) sVr[(slot * gpb + gib) * 128 + k] = 0.0;
    };
    unsigned long long step = a.step0;
    int col = (int)(step % (unsigned long long)d);
    int cyc = 0;
    stage_col(0, col, 0);
    double x[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int i = 4 * kk + c;
        x[kk] = (i < d) ? a.x[(size_t)i * W + w] : 0.0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_tail(0);
    __syncthreads();
    int slot = 0;
    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    const uint32_t gid = a.walker0 + (uint32_t)w;
    double r4 = 0.0, Ea4 = 0.0;   // MCMC_MFMA_RNG4: the variates this lane class drew
    (void)r4; (void)Ea4;

    for (int s = 0; s < a.n_steps; ++s) {
        {
            int ncol = col + 1, ncyc = cyc;
            if (ncol == d) { ncol = 0; ++ncyc; }
            if (s + 1 < a.n_steps) stage_col(ncyc, ncol, slot ^ 1);
        }
#if MCMC_MFMA_RNG4
        // The four lanes of a walker draw the variates of FOUR consecutive steps at once: lane
        // class c evaluates the Philox block of step (this step + c) every fourth step of the
        // launch, and each step fetches its pair from the class that drew it.  Pays where the
        // variates are a large share of a step (small d); at d = 100 it measured 2 % slower.
        if ((s & 3) == 0) {
            StepRng rng;
            rng.begin(a.key0, a.key1, gid, step + (unsigned long long)c);
            rng.run_all();
            r4 = rng.r;
            Ea4 = rng.Ea;
        }
        const int src = (lane & 15) + 16 * (s & 3);
        const double r = __shfl(r4, src), Ea = __shfl(Ea4, src);
#else
        StepRng rng;   // the four lanes of a walker draw the same variates
        rng.begin(a.key0, a.key1, gid, step);
        rng.run_all();
        const double r = rng.r, Ea = rng.Ea;
#endif
        const double* __restrict__ v = sVr + (slot * gpb + gib) * 128;
        double p = 0.0;
        {
            constexpr int R1 = kMfmaFirstPass;
            d4 acc[R1];
            mfma_pass<0, R1, kSel>(acc, x, r, v, sE, sL, c, lane, d);
#pragma unroll
            for (int q = 0; q < R1; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) p = fma(acc[q][e], acc[q][e], p);
        }
        if (RT > kMfmaFirstPass) {   // (R0 is clamped so that the dead instantiation is valid)
            constexpr int R0 = RT > kMfmaFirstPass ? kMfmaFirstPass : RT - 1;
            d4 acc[RT - R0];
            mfma_pass<R0, RT, kSel>(acc, x, r, v, sE, sL, c, lane, d);
#pragma unroll
            for (int q = 0; q < RT - R0; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) p = fma(acc[q][e], acc[q][e], p);
        }
        const int n = lane & 15;
        const double p0 = __shfl(p, n), p1 = __shfl(p, n + 16), p2 = __shfl(p, n + 32),
                     p3 = __shfl(p, n + 48);
        const double chi2 = (p0 + p1) + (p2 + p3);
        const bool inb = chi2 < INFINITY;
        // Normal priors (prior.py:746-761): lane class c chains the terms of its dimensions
        // i = c (mod 4) in ascending order; the four chains combine like chi2 (specification
        // for d > 32).  Wave-uniform branch: skipped when every prior is uniform.
        double psum = 0.0;
        if (has_norm) {
            double sc = 0.0;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const int i = 4 * kk + c;
                const double scale = sN[3 * i + 1];
                if (scale < INFINITY) {   // (trial of this dimension recomputed: same fma)
                    const double q = (fma(r, v[i], x[kk]) - sN[3 * i]) / scale;
                    sc = sc + fma(-0.5 * q, q, sN[3 * i + 2]);
                }
            }
            const double s0 = __shfl(sc, n), s1 = __shfl(sc, n + 16), s2 = __shfl(sc, n + 32),
                         s3 = __shfl(sc, n + 48);
            psum = (s0 + s1) + (s2 + s3);
        }
        const double lp = a.uniform_logp + psum;
        const double ll = -0.5 * (a.cnorm0 + chi2);
        const double lt = inb ? lp + ll : -INFINITY;
        const bool accept = inb & (lt != -INFINITY) &
                            ((lt > lpost) | (Ea > (lpost - lt) / a.temperature));
        burn -= (accept & (burn > 0)) ? 1 : 0;
        const double ra = accept ? r : 0.0;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int i = 4 * kk + c;
            // (guarded on purpose: the unconditional form lets the compiler hoist the 25 LDS
            // loads and spill 30 more registers -- 16.6 against 14.1 ms per 800 steps at d = 100)
            x[kk] = (i < d) ? fma(ra, v[i], x[kk]) : 0.0;
        }
        lpri = accept ? lp : lpri;
        llik = accept ? ll : llik;
        lpost = accept ? lt : lpost;
        prej = accept ? 0 : (prej + (inb ? 0 : 1));
        wt = accept ? 1 : wt + 1;
        nacc += accept ? 1 : 0;
        if (!accept && c == 0) {
            const double max_now = a.max_tries * (burn > 0 ? 10.0 : 1.0);
            if ((double)(wt - prej) > max_now) atomicCAS(a.stuck, 0, 1 + (int)gid);
        }
        ++step;
        if (++col == d) { col = 0; ++cyc; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        zero_tail(slot ^ 1);
        __syncthreads();
        slot ^= 1;
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int i = 4 * kk + c;
        if (i < d) a.x[(size_t)i * W + w] = x[kk];
    }
    if (c == 0) {
        a.logpost[w] = lpost; a.logprior[w] = lpri; a.loglike[w] = llik;
        a.weight[w] = wt; a.prior_rej[w] = prej; a.burn_left[w] = burn;
        a.n_accept[w] = nacc;
    }
    wave_add_accepts(a.accept_total, (c == 0) ? nacc - nacc0 : 0);
}

// ---------------------------------------------------------------- batch evaluator (run-time d)
// General: normal priors and mixtures included.  One thread per point; operands from the
// "big" constant block (row-major L^-1 per mode) through global memory.
struct BigEvalArgs {
    EvalArgs e;
    const double* Lrow;  // [K][d][d] row-major L^-1
    int d;
    double* scratch;     // [n][K] mode log-pdfs for the log-sum-exp
};

__global__ void __launch_bounds__(64) evaluate_big_kernel(const BigEvalArgs b)
{
    const EvalArgs& a = b.e;
    const int d = b.d, K = a.n_modes;
    const ConstLayout cl{d, K};
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= a.n) return;
    const double* __restrict__ t = a.x + (size_t)p * d;
    const double* __restrict__ C = a.cblock;
    bool in = true;
    for (int i = 0; i < d; ++i) in = in & (t[i] <= C[cl.hi() + i]) & (t[i] >= C[cl.lo() + i]);
    double sc[4] = {0.0, 0.0, 0.0, 0.0};   // d > 32: four interleaved chains (specification)
    if (a.norm_mask4[0] | a.norm_mask4[1] | a.norm_mask4[2] | a.norm_mask4[3]) {
        for (int i = 0; i < d; ++i)
            if ((a.norm_mask4[i >> 5] >> (i & 31)) & 1u) {
                const double q = (t[i] - C[cl.loc() + i]) / C[cl.scale() + i];
                sc[i & 3] = sc[i & 3] + fma(-0.5 * q, q, C[cl.mls() + i]);
            }
    }
    const double lp = a.uniform_logp + ((sc[0] + sc[1]) + (sc[2] + sc[3]));
    double ll = 0.0;
    if (K >= 1) {
        double amax = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ Lk = b.Lrow + (size_t)k * d * d;
            const double* __restrict__ mu = C + cl.mean(k);
            double pc[4] = {0.0, 0.0, 0.0, 0.0};  // d > 32: four interleaved chi2 chains
            for (int j = 0; j < d; ++j) {
                double y = 0.0;
                for (int i = 0; i <= j; ++i) y = fma(Lk[j * d + i], t[i] - mu[i], y);
                if (a.derived) a.derived[((size_t)p * K + k) * d + j] = y;
                pc[j & 3] = fma(y, y, pc[j & 3]);
            }
            const double chi2 = (pc[0] + pc[1]) + (pc[2] + pc[3]);
            const double ak = -0.5 * (C[cl.cnorm() + k] + chi2);
            b.scratch[(size_t)p * K + k] = ak;
            amax = (ak > amax) ? ak : amax;
        }
        if (K == 1) {
            ll = amax;
        } else {
            double S = 0.0;
            for (int k = 0; k < K; ++k)
                S = fma(C[cl.weight() + k], dexp(b.scratch[(size_t)p * K + k] - amax), S);
            ll = dlog(S) + amax;
        }
    }
    a.logprior[p] = in ? lp : -INFINITY;
    a.loglike[p] = in ? ll : -INFINITY;
}

// ---------------------------------------------------------------- moments (run-time d)
// The group's walkers pass through LDS in slices of `slice` walkers (the whole group when its
// tile fits 160 KiB): every sum is ONE chain over the walkers in ascending order, carried from
// slice to slice (the pair sums through Sg, which this block owns), so the result does not
// depend on the slicing.
__global__ void __launch_bounds__(256) group_moments_big_kernel(const MomentArgs a, int d, int slice)
{
    extern __shared__ __attribute__((aligned(16))) double sX[];  // [slice][d|1]
    const int ldx = d | 1;
    const int npair = d * (d + 1) / 2;
    const int tid = threadIdx.x, gs = blockDim.x, g = blockIdx.x;
    double gpart[2] = {0.0, 0.0};   // this thread's group sums, carried over the slices
    for (int l0 = 0; l0 < gs; l0 += slice) {
        if (l0) __syncthreads();
        // slice x d values, walker-fastest across the threads (coalesced)
        for (int e = tid; e < slice * d; e += gs) {
            const int l = e % slice, i = e / slice;
            sX[l * ldx + i] = a.x[(size_t)i * a.W + (size_t)g * gs + l0 + l] - a.shift[i];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // d <= 128, gs >= 64: at most two sums per thread
            const int i = tid + q * gs;
            if (i < d) {
                double s = gpart[q];
                for (int l = 0; l < slice; ++l) s = s + sX[l * ldx + i];
                gpart[q] = s;
                if (l0 + slice >= gs) a.group_sum[(size_t)g * d + i] += s;
            }
        }
        // pair sums in 4 x 4 register tiles over the lower triangle (tile (I, J), I >= J: the
        // pairs i = 4I + a, j = 4J + b): a walker costs a tile 8 LDS reads for 16 fmas instead
        // of 32 -- the kernel is bound by LDS bandwidth.  Every pair is still ONE chain over
        // the walkers in ascending order.
        const int nb = (d + 3) / 4, ntile = nb * (nb + 1) / 2;
        double* __restrict__ Sg = a.Sg + (size_t)g * npair;
        for (int t = tid; t < ntile; t += gs) {
            int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            while (I * (I + 1) / 2 > t) --I;
            const int J = t - I * (I + 1) / 2;
            const int i0 = 4 * I, j0 = 4 * J;
            double acc[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = i0 + u, j = j0 + v;
                    acc[u][v] = (l0 && i < d && j <= i) ? Sg[(size_t)i * (i + 1) / 2 + j] : 0.0;
                }
            // (columns beyond d: read inside the row -- ldx >= d + 1 only for even d -- and not
            // stored; clamp the index instead of branching)
            int iu[4], jv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                iu[u] = i0 + u < d ? i0 + u : d - 1;
                jv[u] = j0 + u < d ? j0 + u : d - 1;
            }
            for (int l = 0; l < slice; ++l) {
                const double* __restrict__ row = sX + l * ldx;
                const double xi0 = row[iu[0]], xi1 = row[iu[1]], xi2 = row[iu[2]], xi3 = row[iu[3]];
                const double xj0 = row[jv[0]], xj1 = row[jv[1]], xj2 = row[jv[2]], xj3 = row[jv[3]];
                acc[0][0] = fma(xi0, xj0, acc[0][0]); acc[0][1] = fma(xi0, xj1, acc[0][1]);
                acc[0][2] = fma(xi0, xj2, acc[0][2]); acc[0][3] = fma(xi0, xj3, acc[0][3]);
                acc[1][0] = fma(xi1, xj0, acc[1][0]); acc[1][1] = fma(xi1, xj1, acc[1][1]);
                acc[1][2] = fma(xi1, xj2, acc[1][2]); acc[1][3] = fma(xi1, xj3, acc[1][3]);
                acc[2][0] = fma(xi2, xj0, acc[2][0]); acc[2][1] = fma(xi2, xj1, acc[2][1]);
                acc[2][2] = fma(xi2, xj2, acc[2][2]); acc[2][3] = fma(xi2, xj3, acc[2][3]);
                acc[3][0] = fma(xi3, xj0, acc[3][0]); acc[3][1] = fma(xi3, xj1, acc[3][1]);
                acc[3][2] = fma(xi3, xj2, acc[3][2]); acc[3][3] = fma(xi3, xj3, acc[3][3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = i0 + u, j = j0 + v;
                    if (i < d && j <= i) Sg[(size_t)i * (i + 1) / 2 + j] = acc[u][v];
                }
        }
    }
}

__global__ void __launch_bounds__(64) pool_moments_big_kernel(const MomentArgs a, int d)
{
    const int npair = d * (d + 1) / 2;
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npair) return;
    double acc = a.pooled[p];
    int g = 0;
    for (; g + 16 <= a.G; g += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = a.Sg[(size_t)(g + u) * npair + p];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; g < a.G; ++g) acc += a.Sg[(size_t)g * npair + p];
    a.pooled[p] = acc;
}

// ---------------------------------------------------------------- launchers
hipError_t launch_step(const StepArgs& a, const double* Lcol, int d, const uint32_t* norm_mask4,
                       hipStream_t st)
{
    const bool has_norm = (norm_mask4[0] | norm_mask4[1] | norm_mask4[2] | norm_mask4[3]) != 0u;
    if (a.W % kMfmaWalkers == 0 && (kMfmaWalkers % a.group_size == 0 || a.group_size % kMfmaWalkers == 0)) {
        // matrix-core kernel: its tiles follow the column-sweep copy of L^-1 in `Lcol`
        MfmaStepArgs m{a, Lcol + (size_t)DP * DP, d, {norm_mask4[0], norm_mask4[1], norm_mask4[2], norm_mask4[3]}};
        const int gpb = (kMfmaWalkers + a.group_size - 1) / a.group_size;
        const size_t lds = sizeof(double) * (size_t)(kTiles * 64 + 24 * KT + 2 * gpb * 128);
        // placement: 256 walkers per CU at W = 65 536 -- ask for just under 1/n of the LDS so
        // that exactly n = 256 / walkers-per-workgroup workgroups share a CU
        const size_t share = ((size_t)(160 << 10) / (size_t)(256 / kMfmaWalkers)) - 2048;
        const size_t want = lds > share ? lds : share;
        hipError_t e = hipFuncSetAttribute(
            has_norm ? (const void*)step_mfma_kernel<true> : (const void*)step_mfma_kernel<false>,
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e != hipSuccess) return e;
        mcmc_hip_note_step_kernel(has_norm ? "mcmc::step_mfma_kernel<true>"
                                           : "mcmc::step_mfma_kernel<false>");
        if (has_norm)
            hipLaunchKernelGGL(step_mfma_kernel<true>, dim3(a.W / kMfmaWalkers), dim3(kMfmaThreads),
                               want, st, m);
        else
            hipLaunchKernelGGL(step_mfma_kernel<false>, dim3(a.W / kMfmaWalkers), dim3(kMfmaThreads),
                               want, st, m);
        return hipGetLastError();
    }
#if MCMC_DP > 112
    return hipErrorInvalidValue;                 // no column-sweep fallback at this size
#else
    if (has_norm) return hipErrorInvalidValue;   // the column-sweep fallback has uniform priors only
    BigStepArgs b{a, Lcol, d};
    const int bs = (a.W % 256 == 0) ? 256 : (a.W % 128 == 0) ? 128 : 64;
    const size_t lds = sizeof(double) * (size_t)(DP * DP + 4 * DP + 2 * (bs / a.group_size) * 128);
    {
        hipError_t e = hipFuncSetAttribute((const void*)step_big_reg_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel("mcmc::step_big_reg_kernel");
    hipLaunchKernelGGL(step_big_reg_kernel, dim3(a.W / bs), dim3(bs), lds, st, b);
    return hipGetLastError();
#endif
}

hipError_t launch_basis(const BasisArgs& a, int n_groups, int d, hipStream_t st)
{
    const int nz = (d + 2) * (d - 1) / 2;
    constexpr int nzp = refl_offset(DP - 1, DP) + 1;
    const size_t phase1 = (size_t)(((nz + 3) & ~1) + ((nzp + 1) & ~1) + 3 * DP);
    const size_t phase2 = (size_t)d * d;
    const size_t lds = sizeof(double) * (phase1 > phase2 ? phase1 : phase2);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)basis_big_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(basis_big_kernel, dim3(n_groups, a.ncyc), dim3(kBasisThreads), lds, st, a, d);
    return hipGetLastError();
}

hipError_t launch_evaluate(const EvalArgs& a, const double* Lrow, int d, double* scratch,
                           hipStream_t st)
{
    BigEvalArgs b{a, Lrow, d, scratch};
    hipLaunchKernelGGL(evaluate_big_kernel, dim3((a.n + 63) / 64), dim3(64), 0, st, b);
    return hipGetLastError();
}

hipError_t launch_moments(const MomentArgs& a, int group_size, int d, hipStream_t st)
{
    int slice = group_size;   // walkers per pass through LDS
    while (sizeof(double) * (size_t)slice * (d | 1) > 160 * 1024) slice /= 2;
    const size_t lds = sizeof(double) * (size_t)slice * (d | 1);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)group_moments_big_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(group_moments_big_kernel, dim3(a.G), dim3(group_size), lds, st, a, d, slice);
    const int npair = d * (d + 1) / 2;
    hipLaunchKernelGGL(pool_moments_big_kernel, dim3((npair + 63) / 64), dim3(64), 0, st, a, d);
    return hipGetLastError();
}

const BigKernels kKernels = {DP, launch_step, launch_basis, launch_evaluate, launch_moments,
                             kTiles, kRowShift};

}  // namespace
}  // namespace mcmc

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)
extern "C" const mcmc::BigKernels* MCMC_CAT(mcmc_hip_big_, MCMC_DP)() { return &mcmc::kKernels; }
