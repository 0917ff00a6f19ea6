// Lane-per-walker Metropolis kernels for one compile-time dimension MCMC_D (gfx950).
//
// One wavefront lane owns one walker: its parameter vector x[D] and trial t[D] live in
// VGPRs (all loops over D are fully unrolled); the problem constants the 64 lanes share -- the
// inverse-Cholesky whitening stream, means and prior bounds -- come in through the scalar
// data cache as SGPR operands, the cycle's proposal directions V through LDS (DMA'd one cycle
// ahead); and n_steps Metropolis steps are fused into one launch so that the state crosses
// HBM once per launch (coalesced, dimension-major).  DESIGN.md section 4 has the reasoning
// and the measurements behind each of these choices.
//
// Kernels: step_kernel<MULTI, GENERAL> (one wave per 64 walkers; every feature),
// step_pair_kernel<UNIT_T, NORMP> (the hot variant: two waves per 64 walkers), drag_kernel
// (dragging steps), basis_kernel (Haar directions), evaluate_kernel, moment kernels.
//
// Restates (paths relative to the reference checkout):
//   cobaya/samplers/mcmc/mcmc.py:545-562 (step) 670-683 (accept) 685-748 (bookkeeping)
//   cobaya/samplers/mcmc/proposal.py:59-82,222-224 ; cobaya/functions.py:35-61 (Haar basis)
//   cobaya/prior.py:658-676,733-763 ; cobaya/tools.py:720-729
//   cobaya/likelihoods/gaussian_mixture/gaussian_mixture.py:138-163 ; gaussian/gaussian.py:96-112
//   cobaya/collection.py:926-934,970-981 (moments, as streaming sufficient statistics)
// The arithmetic order is the one fixed in DESIGN.md "Ensemble specification".
#include "det_math.h"
#include "kernels.h"

#ifndef MCMC_CH
#define MCMC_CH 16
#endif
#ifndef MCMC_D
#error "compile with -DMCMC_D=<dimension>"
#endif

namespace mcmc {
namespace {

constexpr int D = MCMC_D;
constexpr int NPAIR = D * (D + 1) / 2;
// 32 < D <= kMaxDimPair: this translation unit only provides the two-wave step kernel (the other
// kernels of such a dimension are the walker_kernels_big.hip ones, whose layout of V it reads).
// The sums over dimensions follow the d > 32 specification: four interleaved chains.
constexpr bool kBigD = D > kMaxDimLane;
constexpr int LDV = kBigD ? v_ld(D) : D;   // doubles between two columns of V

// Wave-uniform read-only operands (problem constants, proposal directions) are read through
// the SCALAR data path: pointers in the constant address space make every load an s_load into
// SGPRs, which v_fma_f64 / v_cmp_f64 take directly as a source -- no LDS traffic, no VGPRs,
// no vector-memory issue slots for data that all 64 lanes share.
typedef const double __attribute__((address_space(4))) * cptr;
__device__ __forceinline__ cptr as_const(const double* p) { return (cptr)(unsigned long long)p; }
// The operands are invariant over the step loop, so LICM would hoist ALL their loads out of it
// (hundreds of SGPRs -> spilled to VGPR lanes).  Passing the base pointer through an empty asm
// once per step makes the loads belong to that step.
__device__ __forceinline__ cptr launder(cptr p)
{
    unsigned long long v = (unsigned long long)p;
    asm volatile("; operand base of this step" : "+s"(v));
    return (cptr)v;
}

// ---------------------------------------------------------------- operand streaming
// With one lane per walker, W = 65 536 walkers give exactly ONE wave per SIMD (the paired kernel
// below splits a walker set over two waves for that reason), so little but the wave's own
// instruction stream can hide the latency of the scalar loads.  The operand streams are
// therefore consumed in chunks with an explicit software pipeline:
//     [use FIRST operand of chunk c]  -> the only s_waitcnt, for chunk c alone
//     [issue the loads of chunk c+1]  -> in flight behind ...
//     [use the rest of chunk c]       -> ... a chunk's worth of FP64 work
// The order is pinned by DATA dependences the compiler cannot break: the base pointer of
// chunk c+1 is passed through an empty asm together with the result of the first operation of
// chunk c (`after`), so those loads can neither be hoisted to the top (SGPR spills) nor out
// of the step loop (LICM), and nothing of chunk c+1 can start before chunk c was waited for.
__device__ __forceinline__ cptr after(cptr p, double& anchor)
{
    unsigned long long v = (unsigned long long)p;
    asm volatile("; next operand chunk" : "+s"(v), "+v"(anchor));
    return (cptr)v;
}

__device__ __forceinline__ void after2(cptr& p, cptr& q, double& anchor)
{
    unsigned long long v = (unsigned long long)p, u = (unsigned long long)q;
    asm volatile("; next operand chunk" : "+s"(v), "+s"(u), "+v"(anchor));
    p = (cptr)v;
    q = (cptr)u;
}
// The proposal directions of the current cycle live in LDS (staged by DMA one cycle ahead):
// wave-uniform LDS addresses, read as broadcasts.
typedef const double __attribute__((address_space(3))) * lptr;
typedef double __attribute__((address_space(3))) * lds_t;
__device__ __forceinline__ lptr after(lptr p, double& anchor)
{
    unsigned v = (unsigned)(unsigned long long)p;
    asm volatile("; next operand chunk" : "+s"(v), "+v"(anchor));
    return (lptr)(unsigned long long)v;
}
__device__ __forceinline__ void after2(lptr& p, cptr& q, double& anchor)
{
    unsigned v = (unsigned)(unsigned long long)p;
    unsigned long long u = (unsigned long long)q;
    asm volatile("; next operand chunk" : "+s"(v), "+s"(u), "+v"(anchor));
    p = (lptr)(unsigned long long)v;
    q = (cptr)u;
}

constexpr int NT = D * (D + 1) / 2;  // operands of one whitening factor
constexpr int CH = MCMC_CH;           // doubles per chunk (16 = two s_load_dwordx16)
constexpr int NCH = (NT + CH - 1) / CH;

struct TriMap {
    unsigned char j[NT];
    unsigned char i[NT];
};
constexpr TriMap make_tri_map()
{
    TriMap m{};
    int idx = 0;
    for (int jb = 0; jb < D; jb += kRowBlock)
        for (int i = 0; i < jb + kRowBlock && i < D; ++i)
            for (int r = 0; r < kRowBlock; ++r)
                if (jb + r < D && i <= jb + r) {
                    m.j[idx] = (unsigned char)(jb + r);
                    m.i[idx] = (unsigned char)i;
                    ++idx;
                }
    return m;
}
__device__ constexpr TriMap kTriMap = make_tri_map();

// ---------------------------------------------------------------- log-posterior of a point
// prior support test, 8 dimensions (lo, hi) per chunk
__device__ __forceinline__ void bounds_stream(const double (&t)[D], cptr lo, cptr hi, bool& inb)
{
    bool in = true;
    constexpr int NC = (D + 7) / 8;
    double cl_[8], ch_[8], nl_[8], nh_[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cl_[k] = (k < D) ? lo[k] : 0.0;
        ch_[k] = (k < D) ? hi[k] : 0.0;
    }
    double anchor = t[0];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = c * 8;
        in = in & (anchor <= ch_[0]) & (anchor >= cl_[0]);
        if (c + 1 < NC) {
            anchor = t[b + 8];
            after2(lo, hi, anchor);  // the (unchanged) bases; offsets stay immediates
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                nl_[k] = (b + 8 + k < D) ? lo[b + 8 + k] : 0.0;
                nh_[k] = (b + 8 + k < D) ? hi[b + 8 + k] : 0.0;
            }
        }
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (b + k < D) in = in & (t[b + k] <= ch_[k]) & (t[b + k] >= cl_[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cl_[k] = nl_[k];
            ch_[k] = nh_[k];
        }
    }
    inb = in;
}

// dev = t - mu, 16 dimensions per chunk
__device__ __forceinline__ void dev_stream(double (&dev)[D], const double (&t)[D], cptr mu)
{
    constexpr int NC = (D + 15) / 16;
    double cur[16], nxt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] = (k < D) ? mu[k] : 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = c * 16;
        dev[b] = t[b] - cur[0];
        if (c + 1 < NC) {
            mu = after(mu, dev[b]);
#pragma unroll
            for (int k = 0; k < 16; ++k) nxt[k] = (b + 16 + k < D) ? mu[b + 16 + k] : 0.0;
        }
#pragma unroll
        for (int k = 1; k < 16; ++k)
            if (b + k < D) dev[b + k] = t[b + k] - cur[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
}

// One mode: triangular whitening y = L^-1 (t - mu), chi2 = |y|^2, as fma chains in ascending
// index order; kRowBlock rows advance together (independent chains in flight); Lk is the
// operand stream packed in exactly this order (kernels.h tri_stream_for_each).
// chi2 = |L^-1 dev|^2 from the operand stream Lk.  `anchor` is any value computed just before
// (its producer precedes the first chunk's loads).  TAIL: while the LAST chunk is being
// consumed, the first 16 doubles at `tail_ptr` are fetched into `tail` (the next phase's
// first chunk).
// [S0, S1) is the part of the operand stream this call consumes (whole row blocks; the paired
// kernel splits the rows between two waves); SUMSQ = false only delivers the y_j (`derived`).
template <bool DERIVED, bool TAIL, bool PRELOADED, bool RNG, typename TP, int S0 = 0, int S1 = NT,
          bool SUMSQ = true>
__device__ __forceinline__ double tri_stream(const double (&dev)[D], cptr Lk, double& anchor,
                                             double* derived, TP tail_ptr, double (&tail)[16],
                                             const double (&first)[CH], StepRng& rng,
                                             double* pcs = nullptr)
{
    constexpr int RB = kRowBlock;
    constexpr int NCH = (S1 - S0 + CH - 1) / CH;
    static_assert(RB == 4, "the d > 32 chains are the rows j mod 4");
    // chi2 chains: one (d <= 32) or four over the rows j = c (mod 4) (d > 32; the caller
    // combines pcs[0..3])
    double pc[4] = {0.0, 0.0, 0.0, 0.0};
    double y[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) y[r] = 0.0;
    double cur[CH], nxt[CH];
    if (PRELOADED) {
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = first[k];
    } else {
        Lk = after(Lk, anchor);
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = (S0 + k < S1) ? Lk[S0 + k] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int base = S0 + c * CH;
        auto op = [&](int k) {
            const int j = kTriMap.j[base + k], i = kTriMap.i[base + k], r = j % RB;
            y[r] = fma(cur[k], dev[i], (i == 0) ? 0.0 : y[r]);
            return r;
        };
        auto fin = [&](int k) {
            const int j = kTriMap.j[base + k], i = kTriMap.i[base + k], r = j % RB;
            if (i == j) {
                if (DERIVED) derived[j] = y[r];
                if (SUMSQ) pc[kBigD ? r : 0] = fma(y[r], y[r], pc[kBigD ? r : 0]);
            }
        };
        // first operand of the chunk: the only wait; then the next chunk's loads go out
        const int r0 = op(0);
        if (c + 1 < NCH) {
            Lk = after(Lk, y[r0]);
#pragma unroll
            for (int q = 0; q < CH; ++q) nxt[q] = (base + CH + q < S1) ? Lk[base + CH + q] : 0.0;
        }
        if (TAIL && c + 1 == NCH) {
            const TP T2 = after(tail_ptr, y[r0]);
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = (q < D) ? T2[q] : 0.0;
        }
        if (RNG) {  // the next step's RNG arithmetic rides behind this chunk's loads
#pragma unroll
            for (int st = 0; st < StepRng::kStages; ++st)
                if ((st * NCH) / StepRng::kStages == c) rng.stage(st);
        }
        fin(0);
#pragma unroll
        for (int k = 1; k < CH; ++k)
            if (base + k < S1) {
                op(k);
                fin(k);
            }
        // chunk fence: every accumulator passes through an empty asm, so no FMA of this chunk
        // can be delayed past it (its SGPR operands die here) and none of the next can start
        asm volatile("; chunk end" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(pc[0]));
        if (kBigD && SUMSQ) asm volatile("; chunk end" : "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]));
#pragma unroll
        for (int k = 0; k < CH; ++k) cur[k] = nxt[k];
    }
    if (kBigD && SUMSQ) {
#pragma unroll
        for (int c = 0; c < 4; ++c) pcs[c] = pc[c];
    }
    return pc[0];
}

// One mode: triangular whitening y = L^-1 (t - mu), chi2 = |y|^2, as fma chains in ascending
// index order; kRowBlock rows advance together (independent chains in flight); Lk is the
// operand stream packed in exactly this order (kernels.h tri_stream_for_each).
template <bool DERIVED>
__device__ __forceinline__ double mode_logpdf(const double (&t)[D], cptr mu, cptr Lk,
                                              double cnorm, double* derived)
{
    double dev[D];
    dev_stream(dev, t, mu);
    double tail[16], first[CH];
    StepRng none;
    const double chi2 = tri_stream<DERIVED, false, false, false, cptr>(dev, Lk, dev[D - 1], derived,
                                                                      Lk, tail, first, none);
    return -0.5 * (cnorm + chi2);
}

// Hot path (one mode, uniform priors): ONE pass over the dimensions computes the trial
// t_i = fma(r, v_i, x_i), tests the prior support and forms dev_i = t_i - mu_i, 4 dimensions
// per chunk from the interleaved stream elem[i] = {lo_i, hi_i, mu_i}.  A dimension outside
// its bounds gets dev_i = +inf, which makes chi2 non-finite: "outside the prior support" is
// recovered as !(chi2 < inf) with no mask reduction over the dimensions.
__device__ __forceinline__ void propose_fused(double (&dev)[D], double r, lptr v, cptr E,
                                              const double (&x)[D], cptr Lk, double (&lfirst)[CH])
{
    constexpr int NC = (D + 3) / 4;
    double cv[4], ce[12], nv[4], ne[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) cv[k] = (k < D) ? v[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) ce[k] = (k < 3 * D) ? E[k] : 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = 4 * c;
        double t0 = fma(r, cv[0], x[b]);
        if (c + 1 < NC) {
            after2(v, E, t0);
#pragma unroll
            for (int k = 0; k < 4; ++k) nv[k] = (b + 4 + k < D) ? v[b + 4 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                ne[k] = (3 * (b + 4) + k < 3 * D) ? E[3 * (b + 4) + k] : 0.0;
        } else {  // last chunk: fetch the first chunk of the whitening stream behind it
            Lk = after(Lk, t0);
#pragma unroll
            for (int k = 0; k < CH; ++k) lfirst[k] = (k < NT) ? Lk[k] : 0.0;
        }
        dev[b] = ((t0 <= ce[1]) & (t0 >= ce[0])) ? t0 - ce[2] : INFINITY;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (b + k < D) {
                const double tk = fma(r, cv[k], x[b + k]);
                dev[b + k] = ((tk <= ce[3 * k + 1]) & (tk >= ce[3 * k])) ? tk - ce[3 * k + 2]
                                                                        : INFINITY;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) cv[k] = nv[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) ce[k] = ne[k];
    }
}

template <bool MULTI, bool DERIVED, bool GENERAL>
__device__ __forceinline__ void eval_point(const double (&t)[D], cptr C, const ConstLayout& cl,
                                           uint32_t norm_mask, double uniform_logp,
                                           double* __restrict__ sA, int astride, bool& inb,
                                           double& lp, double& ll, double* derived)
{
    bounds_stream(t, C + cl.lo(), C + cl.hi(), inb);
    bool in = inb;
    inb = in;
    double s = 0.0;
    if (GENERAL && norm_mask) {
#pragma unroll
        for (int i = 0; i < D; ++i)
            if ((norm_mask >> i) & 1u) {
                const double q = (t[i] - C[cl.loc() + i]) / C[cl.scale() + i];
                s = s + fma(-0.5 * q, q, C[cl.mls() + i]);
            }
    }
    lp = uniform_logp + s;
    if (!MULTI) {
        if (cl.K == 0) {
            ll = 0.0;
        } else {
            ll = mode_logpdf<DERIVED>(t, C + cl.mean(0), C + cl.linv(0), C[cl.cnorm()], derived);
        }
    } else {
        double amax = -INFINITY;
        for (int k = 0; k < cl.K; ++k) {
            const double a = mode_logpdf<DERIVED>(t, C + cl.mean(k), C + cl.linv(k),
                                                  C[cl.cnorm() + k],
                                                  DERIVED ? derived + k * D : nullptr);
            sA[k * astride] = a;
            amax = (a > amax) ? a : amax;
        }
        double S = 0.0;
        for (int k = 0; k < cl.K; ++k) S = fma(C[cl.weight() + k], dexp(sA[k * astride] - amax), S);
        ll = dlog(S) + amax;
    }
}

__device__ __forceinline__ double wrap_periodic(double t, double lo, double hi)
{
    const double w = hi - lo;
    const double y = (t - lo) / w;
    const double m = y - floor(y);
    return m * w + lo;
}

// out[i] = fma(r, v[i], x[i]) with v streamed 16 dimensions per chunk (out may alias x);
// PRELOADED: the first chunk is already in `first` (fetched by the previous phase).
template <bool PRELOADED, int N = D>
__device__ __forceinline__ void axpy_stream(double (&out)[D], double r, lptr v,
                                            const double (&x)[D], const double (&first)[16])
{
    constexpr int NC = (N + 15) / 16;
    double cur[16], nxt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] = PRELOADED ? first[k] : ((k < D) ? v[k] : 0.0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = c * 16;
        out[b] = fma(r, cur[0], x[b]);
        if (c + 1 < NC) {
            v = after(v, out[b]);
#pragma unroll
            for (int k = 0; k < 16; ++k) nxt[k] = (b + 16 + k < N) ? v[b + 16 + k] : 0.0;
        }
#pragma unroll
        for (int k = 1; k < 16; ++k)
            if (b + k < N) out[b + k] = fma(r, cur[k], x[b + k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
}

#if MCMC_D <= 32
// ---------------------------------------------------------------- the Metropolis kernel
// FAST (= !MULTI && !GENERAL) is the hot variant: exactly one mode, uniform priors only,
// nothing periodic, no row emission; no control flow inside a step, the trial is not kept in
// registers but recomputed (same fma) when the step is accepted.  The GENERAL variants cover
// everything else (normal priors, periodic parameters, `one`, mixtures, emitted rows).
// OWN (round 6; GENERAL only): `shared_basis: False`, the to-the-letter control -- every walker
// proposes along the columns of its OWN Haar basis (proposal.py:59-69): no slab of directions in
// LDS; lane w reads its column [w][cycle][col][0 .. D) straight from HBM -- D contiguous doubles,
// fetched one step ahead into registers (one wave per SIMD: 512 of them) --, everything else is
// the GENERAL step on registers.  (Rounds 1-5 ran this on step_general_kernel: state and trial in
// LDS, L^-1 through the scalar cache row by row, 0.57 ms per cycle of 30 steps at 65 536 walkers.)
template <bool MULTI, bool GENERAL, bool OWN = false>
__global__ void __launch_bounds__(256) step_kernel(const StepArgs a)
{
    static_assert(!OWN || GENERAL, "own bases: the GENERAL step");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int SLAB = a.slab, cps = a.cps;
    const ConstLayout cl{D, a.n_modes};
    const cptr C0 = as_const(a.cblock);
    const int tid = threadIdx.x, gs = blockDim.x;
    const int w = blockIdx.x * gs + tid;
    const int W = a.W;
    // workgroup != group: the block is as wide as W allows (4 waves = one per SIMD of a CU, so
    // the dispatcher cannot pile single-wave workgroups onto one SIMD); the basis group of a
    // wave is wave-uniform, hence the readfirstlane.
    const int group = __builtin_amdgcn_readfirstlane(w / a.group_size);
    const int gpb = gs / a.group_size;                       // groups per block
    const int gib = __builtin_amdgcn_readfirstlane(tid / a.group_size);   // group in block
    const int wpg = a.group_size >> 6;                       // waves per group
    const int part = __builtin_amdgcn_readfirstlane((tid >> 6) % wpg);
    // LDS: two slabs of proposal directions per group of the block (current cycle and the
    // next one, which a global->LDS DMA fills while the current one is used), then sA.
    double* sV[2] = {smem, smem + gpb * SLAB};
    double* const sA = OWN ? smem : smem + 2 * gpb * SLAB;   // MULTI only: [K][blockDim]
    const double* const Vgrp = a.V + (size_t)group * a.ncyc * SLAB;
    auto stage_dma = [&](int cycle, double* dst) {
        // this wave moves every wpg-th KiB of its group's slab; each lane carries 16 bytes
        for (int kb = part; kb < SLAB / 128; kb += wpg) {
            const char* g = (const char*)(Vgrp + (size_t)cycle * SLAB) + kb * 1024 + (tid & 63) * 16;
            char* l = (char*)(dst + gib * SLAB) + kb * 1024;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)g,
                (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    if (!OWN) {
        stage_dma(0, sV[0]);
        if (a.ncyc > 1) stage_dma(1, sV[1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int cur_buf = 0;

    double x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = a.x[(size_t)i * W + w];
    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    int nrow = a.rows ? a.n_rows[w] : 0;
    const uint32_t gid = a.walker0 + (uint32_t)w;
    unsigned long long step = a.step0;
    int col = (int)(step % (unsigned long long)cps);
    int cyc = 0;
    // OWN: this walker's columns, and the one of the first step in registers
    const double* const Vown = OWN ? a.V + (size_t)w * a.ncyc * (size_t)SLAB : nullptr;
    double vown[OWN ? D : 1];
    auto fetch_own = [&](int cy, int co, double (&dst)[OWN ? D : 1]) {
        const double* __restrict__ p = Vown + (size_t)cy * SLAB + (size_t)co * D;
        if (D % 2 == 0) {   // (slabs and even columns start on 16 bytes: two dimensions per load)
#pragma unroll
            for (int i = 0; i < D; i += 2) {
                const double2 q = *(const double2*)(p + i);
                dst[i] = q.x;
                dst[i + 1 < D ? i + 1 : i] = q.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) dst[i] = p[i];
        }
    };
    if (OWN) fetch_own(0, col, vown);
    StepRng rng;
    if (!MULTI && !GENERAL && D >= 2) {  // hot variant: the first step's variates up front
        rng.begin(a.key0, a.key1, gid, step);
        rng.run_all();
    }

    for (int s = 0; s < a.n_steps; ++s) {
        // ---- random variates of (walker, step): one Philox block
        constexpr bool FAST = !MULTI && !GENERAL && D >= 2;
        double r, Ea;
        if (FAST) {  // computed during the previous step's whitening stream (or the prologue)
            r = rng.r;
            Ea = rng.Ea;
            rng.begin(a.key0, a.key1, gid, step + 1);
        } else {
            rng.begin(a.key0, a.key1, gid, step);
            rng.run_all();
            r = rng.r;
            Ea = rng.Ea;
            // a column of a one-parameter block (wave-uniform flag written with the directions)
            const bool oned = D == 1 || (GENERAL && a.vflag != nullptr &&
                                         a.vflag[((size_t)group * a.ncyc + cyc) * cps + col] != 0);
            if (oned) {
                double sn, cs;
                sincos2pi(rng.ka, sn, cs);
                const double rr = rng.expo ? rng.Er : sqrt(2.0 * rng.Er) * fabs(cs);
                r = (rng.c0 & 0x80u) ? rr : -rr;
                const u32x4 q4 = philox4x32_10(a.key0, a.key1, gid, kStreamStep | 0x100u,
                                               (uint32_t)step, (uint32_t)(step >> 32));
                Ea = -dlog(u52(((uint64_t)q4.w0 << 20) | (q4.w1 >> 12)));
            }
        }
        // ---- proposal: t = x + r * v, v = T R[:, col] shared by the group
        const lptr v = (lptr)(sV[cur_buf] + gib * SLAB + col * D);
        const cptr C = launder(C0);
        bool inb;
        double lp, ll;
        double t[D];
        double vhead[16];  // FAST: first chunk of v, re-fetched for the commit
        if (FAST) {
            double dev[D], lfirst[CH];
            propose_fused(dev, r, v, C + cl.elem(), x, C + cl.linv(0), lfirst);
            lp = a.uniform_logp + 0.0;
            const double chi2 = tri_stream<false, true, true, true, lptr>(
                dev, C + cl.linv(0), dev[D - 1], nullptr, v, vhead, lfirst, rng);
            inb = chi2 < INFINITY;  // false for +inf and NaN: some dimension was out of bounds
            ll = -0.5 * (a.cnorm0 + chi2);
        } else if (OWN) {
            // the next step's column travels while this one is evaluated (a launch ends with its
            // last column: nothing beyond it is read)
            double vnext[OWN ? D : 1];
            const bool more = s + 1 < a.n_steps;
            const int ncol = col + 1 == cps ? 0 : col + 1, ncy = col + 1 == cps ? cyc + 1 : cyc;
            if (more) fetch_own(ncy, ncol, vnext);
#pragma unroll
            for (int i = 0; i < D; ++i) t[i] = fma(r, vown[OWN ? i : 0], x[i]);
            if (a.periodic_mask) {
#pragma unroll
                for (int i = 0; i < D; ++i)
                    if ((a.periodic_mask >> i) & 1u)
                        t[i] = wrap_periodic(t[i], C[cl.lo() + i], C[cl.hi() + i]);
            }
            eval_point<MULTI, false, GENERAL>(t, C, cl, a.norm_mask, a.uniform_logp, sA + tid, gs,
                                              inb, lp, ll, nullptr);
            if (more) {
#pragma unroll
                for (int i = 0; i < (OWN ? D : 1); ++i) vown[i] = vnext[i];
            }
        } else {
            axpy_stream<false>(t, r, v, x, vhead);
            if (GENERAL && a.periodic_mask) {
#pragma unroll
                for (int i = 0; i < D; ++i)
                    if ((a.periodic_mask >> i) & 1u)
                        t[i] = wrap_periodic(t[i], C[cl.lo() + i], C[cl.hi() + i]);
            }
            // ---- log-posterior of the trial
            eval_point<MULTI, false, GENERAL>(t, C, cl, a.norm_mask, a.uniform_logp, sA + tid, gs,
                                              inb, lp, ll, nullptr);
        }
        const double lt = inb ? lp + ll : -INFINITY;
        // ---- Metropolis test (mcmc.py:678-683)
        const bool accept = inb & (lt != -INFINITY) &
                            ((lt > lpost) | (Ea > (lpost - lt) / a.temperature));
        // ---- bookkeeping (mcmc.py:685-748)
        if (FAST) {
            burn -= (accept & (burn > 0)) ? 1 : 0;
        } else if (accept) {
            if (burn <= 0) {
                if (a.rows) {
                    if (nrow < a.row_cap) {
                        double* row = a.rows + ((size_t)w * a.row_cap + nrow) * (D + 4);
                        row[0] = (double)wt; row[1] = lpost; row[2] = lpri; row[3] = llik;
#pragma unroll
                        for (int i = 0; i < D; ++i) row[4 + i] = x[i];
                    }
                    ++nrow;  // rows beyond the capacity are counted as dropped
                }
            } else {
                --burn;
            }
        }
        if (FAST) {
            const double ra = accept ? r : 0.0;  // fma(0, v, x) == x exactly (v finite)
            axpy_stream<true>(x, ra, v, x, vhead);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = accept ? t[i] : x[i];
        }
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
        if (++col == cps) {  // next cycle: its slab was DMA'd during this one
            col = 0;
            ++cyc;
            if (!OWN && s + 1 < a.n_steps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (cyc + 1 < a.ncyc) stage_dma(cyc + 1, sV[cur_buf]);
                cur_buf ^= 1;
            }
        }
    }

    // The output pointers are re-read from the kernarg segment here (behind an asm the loads
    // cannot be hoisted over) so that they do not occupy ~30 SGPRs for the whole step loop.
    typedef const StepArgs __attribute__((address_space(4))) * kaptr;
    unsigned long long kav = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue" : "+s"(kav));
    const kaptr ka = (kaptr)kav;
    double* const ox = ka->x;
#pragma unroll
    for (int i = 0; i < D; ++i) ox[(size_t)i * W + w] = x[i];
    ka->logpost[w] = lpost; ka->logprior[w] = lpri; ka->loglike[w] = llik;
    ka->weight[w] = wt; ka->prior_rej[w] = prej; ka->burn_left[w] = burn;
    ka->n_accept[w] = nacc;
    wave_add_accepts(ka->accept_total, nacc - nacc0);
    if (ka->rows) ka->n_rows[w] = nrow;
}

#endif  // MCMC_D <= 32
// ---------------------------------------------------------------- the paired Metropolis kernel
// The hot variant again, but with TWO waves per 64 walkers so that a SIMD always holds two
// waves (at W = 65 536 the one-wave-per-walker-set kernel leaves every SIMD with a single wave
// and nothing hides its scalar-load and LDS latencies).  The rows of the whitening factor are
// split at kSplit (a whole number of row blocks):
//   role 0: trial + support test of dimensions [0, kSplit), rows [0, kSplit) of y = L^-1 dev,
//           the partial chi2 of those rows, the walker's random variates and bookkeeping;
//   role 1: trial of all dimensions (support test only for [kSplit, D)), rows [kSplit, D).
// Once per step the two waves meet at a workgroup barrier and exchange, through LDS, role 0's
// partial chi2 and the next step's variates against role 1's y_j; both then finish
// chi2 = fma(y_j, y_j, chi2) for j = kSplit .. D-1 -- the SAME ascending chain as the one-wave
// kernel and the oracle, so the result is bit-identical -- take the same accept decision and
// commit their own copy of the state.
// The split pays from d ~ 14: per 40d steps of 65 536 walkers, two waves vs one wave per walker
// set run 0.387 / 0.385 ms at d = 8, 0.698 / 0.655 at 12, 1.09 / 1.25 at 16, 1.61 / 1.93 at 20,
// 2.22 / 2.84 at 24.
#ifndef MCMC_PAIR_MIN
#define MCMC_PAIR_MIN 14
#endif
constexpr bool kPair = D >= MCMC_PAIR_MIN;
// Measured at d = 30 (W = 65 536): splits 16 / 20 / 24 run 3.47 / 3.32 / 3.46 ms per 1200 steps,
// 12 runs 4.1 ms -- role 0 (which also generates the variates) takes about two thirds of the rows.
// With normal priors role 1 also forms the prior terms (a division each), and one more row
// block moves to role 0: at the config-5 shape (d = 27, 21 normal priors) splits 16 / 20 / 24
// run 4.42 / 4.25 / 4.03 ms per 1080 steps.
constexpr int pair_split(bool normp)
{
    if (!kPair) return D;
    int h = kRowBlock * ((2 * D + 6) / 12);  // multiple of the row block nearest 2D/3
    // 32 < d <= 48, measured (10^10 evals/s at W = 65 536; the matrix-core kernel runs 1.15 at
    // every one of these d): d = 33: split 24 / 28 give 2.01 / 1.85; d = 36: 24 / 28 / 32 give
    // 1.85 / 1.76 / 1.78; d = 40: 24 / 28 / 32 / 36 give 1.60 / 1.63 / 1.68 / 1.51; d = 44:
    // 28 / 32 / 36 give 1.46 / 1.52 / 1.48; d = 48: 28 / 32 / 36 / 40 give 1.31 / 1.31 / 1.38 / 1.26
    // (d = 38 with split 24 dips to 1.56)
    // d = 50: 32 / 36 / 40 give 1.16 / 1.18 / 1.20; d = 52: 32 / 36 / 40 give 0.89 / 1.00 / 1.15;
    // d = 54: 40 / 44 give 1.02 / 1.07; d = 56: 40 / 44 / 48 give 0.85 / 0.97 / 0.84 (matrix
    // cores, padded to 56: 0.93) -- past d = 48 role 1 runs out of registers first
    if (kBigD) h = D < 37 ? 24 : (D < 47 ? 32 : (D < 49 ? 36 : (D < 53 ? 40 : 44)));
    if (normp) h += kRowBlock;
    const int hmax = D - 1 - (D - 1) % kRowBlock;   // largest whole number of row blocks below D
    return h < kRowBlock ? kRowBlock : (h > hmax ? hmax : h);
}
// the split and what follows from it, per instantiation
template <bool NORMP>
struct PairGeom {
    static constexpr int split = pair_split(NORMP);
    static constexpr int nta = split * (split + 1) / 2;   // operands of rows [0, split)
    static constexpr int db = D - split;
    // exchanged doubles per walker and step: chi2 of role 0, r, Ea, the y_j of role 1; with
    // normal priors also role 1's prior sum
    // (d > 32: the three other chi2 chains of role 0 at the end)
    static constexpr int xf = db + (NORMP ? 4 : 3) + (kBigD ? 3 : 0);
};

// `ok` collects the support test as a wave mask on the scalar ALU (one bit per walker):
// v_cmp writes an SGPR pair, s_and folds it in -- no per-dimension VALU select.
__device__ __forceinline__ void support_and(unsigned long long& ok, double t, double lo, double hi)
{
    ok &= __builtin_amdgcn_ballot_w64(t <= hi) & __builtin_amdgcn_ballot_w64(t >= lo);
    asm volatile("; support" : "+s"(ok));  // keeps the AND a chain (a tree would hold 2D masks)
}
// a if this lane's bit of `mask` is set, else b
__device__ __forceinline__ double select_by_mask(unsigned long long mask, double a, double b)
{
    const unsigned long long ab = (unsigned long long)__double_as_longlong(a);
    const unsigned long long bb = (unsigned long long)__double_as_longlong(b);
    unsigned lo, hi;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)bb), "v"((unsigned)ab), "s"(mask));
    asm("v_cndmask_b32 %0, %1, %2, %3"
        : "=v"(hi) : "v"((unsigned)(bb >> 32)), "v"((unsigned)(ab >> 32)), "s"(mask));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// NORMP: some priors are normal (prior.py:746-761).  Role 1 forms every trial t_i anyway (and
// carries less of the whitening), so it chains the terms fma(-q/2, q, mls_i),
// q = (t_i - loc_i) / scale_i, of the normal dimensions in ascending order into s0.
template <int ROLE, bool NORMP>
__device__ __forceinline__ void propose_pair(double (&dev)[D], double r, lptr v, cptr E, cptr MU,
                                             const double (&x)[D], cptr Lk, double (&lfirst)[CH],
                                             unsigned long long& ok, unsigned long long nmask,
                                             cptr Cn, double (&sc)[4])
{
    constexpr int kSplit = PairGeom<NORMP>::split, kNTA = PairGeom<NORMP>::nta;
    const ConstLayout cl{D, 1};
    auto prior_term = [&](int dim, double t) {
        if (NORMP && ROLE == 1 && ((nmask >> dim) & 1ull)) {
            const double q = (t - Cn[cl.loc() + dim]) / Cn[cl.scale() + dim];
            const int c = kBigD ? (dim & 3) : 0;   // d > 32: chains over i mod 4
            sc[c] = sc[c] + fma(-0.5 * q, q, Cn[cl.mls() + dim]);
        }
    };
    constexpr int N = ROLE == 0 ? kSplit : D;
    constexpr int C0 = ROLE == 0 ? 0 : kSplit / 4;  // first chunk with the support test
    constexpr int NC = (N + 3) / 4;
    constexpr int S0 = ROLE == 0 ? 0 : kNTA, S1 = ROLE == 0 ? kNTA : NT;
    double cv[4], ce[12], nv[4], ne[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) cv[k] = (k < N) ? v[k] : 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k)
        ce[k] = (C0 == 0) ? ((k < 3 * D) ? E[k] : 0.0) : ((k < 4) ? MU[k] : 0.0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int b = 4 * c;
        const bool test = c >= C0;
        double t0 = fma(r, cv[0], x[b]);
        if (c + 1 < NC) {
            const bool test2 = c + 1 >= C0;
            if (test2) after2(v, E, t0);
            else after2(v, MU, t0);
#pragma unroll
            for (int k = 0; k < 4; ++k) nv[k] = (b + 4 + k < N) ? v[b + 4 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                ne[k] = test2 ? ((3 * (b + 4) + k < 3 * D) ? E[3 * (b + 4) + k] : 0.0)
                              : ((k < 4 && b + 4 + k < D) ? MU[b + 4 + k] : 0.0);
        } else {  // last chunk: fetch the first chunk of this role's whitening stream behind it
            Lk = after(Lk, t0);
#pragma unroll
            for (int k = 0; k < CH; ++k) lfirst[k] = (S0 + k < S1) ? Lk[S0 + k] : 0.0;
        }
        unsigned long long m = ~0ull;  // this chunk's tests: folded into `ok` once per chunk
        if (test) {
            m &= __builtin_amdgcn_ballot_w64(t0 <= ce[1]) & __builtin_amdgcn_ballot_w64(t0 >= ce[0]);
            dev[b] = t0 - ce[2];
        } else {
            dev[b] = t0 - ce[0];
        }
        prior_term(b, t0);
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (b + k < N) {
                const double tk = fma(r, cv[k], x[b + k]);
                prior_term(b + k, tk);
                if (test) {
                    m &= __builtin_amdgcn_ballot_w64(tk <= ce[3 * k + 1]) &
                         __builtin_amdgcn_ballot_w64(tk >= ce[3 * k]);
                    dev[b + k] = tk - ce[3 * k + 2];
                } else {
                    dev[b + k] = tk - ce[k];
                }
            }
        if (test) {
            ok &= m;
            asm volatile("; support" : "+s"(ok));  // a chain over the chunks, not a tree
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cv[k] = nv[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) ce[k] = ne[k];
    }
}

// all waves of the workgroup have issued their exchange stores; LDS only (the global->LDS DMA
// of the next cycle's slab stays in flight across this barrier)
__device__ __forceinline__ void exchange_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// UNIT_T: temperature == 1 (x / 1.0 == x exactly, so the division is dropped)
template <int ROLE, bool UNIT_T, bool NORMP>
__device__ __forceinline__ void pair_steps(const StepArgs& a, lds_t smem)
{
    constexpr int kSplit = PairGeom<NORMP>::split, kNTA = PairGeom<NORMP>::nta,
                  kDB = PairGeom<NORMP>::db, kXF = PairGeom<NORMP>::xf;
    const int SLAB = a.slab;
    constexpr int NX = ROLE == 0 ? kSplit : D;      // dimensions of the state this role holds
    const ConstLayout cl{D, 1};
    const cptr C0 = as_const(a.cblock);
    const int tid = threadIdx.x;
    const int wl = tid & 255;                        // walker within the block
    const int w = blockIdx.x * 256 + wl;
    const int W = a.W;
    typedef const StepArgs __attribute__((address_space(4))) * kaptr;
    // LDS (32-bit addresses throughout): two slabs of proposal directions per group of the
    // block (current cycle and the next), then the exchange area [parity][kXF][256].
    // group_size is 64, 128 or 256: divisions are shifts.
    const int lg = __builtin_ctz((unsigned)a.group_size);
    const int slab2 = (256 >> lg) * SLAB;                                  // one buffer
    const int vbase = __builtin_amdgcn_readfirstlane(wl >> lg) * SLAB;     // this group's slab
    const lds_t sX = smem + 2 * slab2;
    // The slab DMA runs once per cycle: everything it needs is recomputed from the kernarg
    // segment there, so that nothing of it stays live (in SGPRs) across the step loop.
    auto stage_dma = [&](kaptr k, int cycle, int buf) {
        const int lgs = __builtin_ctz((unsigned)k->group_size), SLAB = k->slab;
        const int wpg = 1 << (lgs - 6);
        const int part = __builtin_amdgcn_readfirstlane((wl >> 6) & (wpg - 1)) + wpg * ROLE;
        const int group = __builtin_amdgcn_readfirstlane(w >> lgs);
        const double* const src = k->V + ((size_t)group * k->ncyc + cycle) * SLAB;
        const lds_t dst = smem + buf * ((256 >> lgs) * SLAB) +
                          __builtin_amdgcn_readfirstlane(wl >> lgs) * SLAB;
        for (int kb = part; kb < SLAB / 128; kb += 2 * wpg) {
            const char* g = (const char*)src + kb * 1024 + (tid & 63) * 16;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)g,
                (__attribute__((address_space(3))) void*)(dst + kb * 128), 16, 0, 0);
        }
    };
    {
        const kaptr k0 = (kaptr)(unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        stage_dma(k0, 0, 0);
        if (a.ncyc > 1) stage_dma(k0, 1, 1);
    }

    double x[D];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = a.x[(size_t)i * W + w];
    double lpost = a.logpost[w];
    double llik = 0.0, lpri = 0.0;
    int wt = 0, prej = 0, burn = 0;
    long long nacc = 0;
    if (ROLE == 0) {
        llik = a.loglike[w];
        lpri = a.logprior[w];
        wt = a.weight[w]; prej = a.prior_rej[w]; burn = a.burn_left[w];
        nacc = a.n_accept[w];
    }
    const long long nacc0 = nacc;
    const uint32_t gid = a.walker0 + (uint32_t)w;
    int col = (int)(a.step0 % (unsigned long long)a.cps);
    int cyc = 0;
    StepRng rng;
    double r = 0.0, Ea = 0.0;
    if (ROLE == 0) {  // the first step's variates, handed to role 1 through the parity-1 slot
        rng.begin(a.key0, a.key1, gid, a.step0);
        rng.run_all();
        sX[(kXF + 1) * 256 + wl] = rng.r;
        sX[(kXF + 2) * 256 + wl] = rng.Ea;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ROLE == 1) {
        r = sX[(kXF + 1) * 256 + wl];
        Ea = sX[(kXF + 2) * 256 + wl];
    }

    const int n_steps = a.n_steps;
    for (int s = 0; s < n_steps; ++s) {
        // The scalars of the step (keys, temperature, norm, ...) are re-read from the kernarg
        // segment every step, behind an asm the loads cannot be hoisted over: as loop
        // invariants they would be spilled to VGPR lanes and cost a v_readlane per use.
        unsigned long long kas = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("; step scalars" : "+s"(kas));
        const kaptr ks = (kaptr)kas;
        const lds_t X = sX + (s & 1) * (kXF * 256) + wl;
        if (ROLE == 0) {  // computed during the previous step's whitening stream
            r = rng.r;
            Ea = rng.Ea;
            rng.begin(ks->key0, ks->key1, gid, ks->step0 + (unsigned)(s + 1));
        }
        const lptr v = (lptr)(smem + (cyc & 1) * slab2 + vbase + col * LDV);
        const cptr C = launder(C0);
        double dev[D], lfirst[CH], vhead[16], yb[D];
        unsigned long long ok = ~0ull;  // walkers whose trial is inside the prior support
        double sc[4] = {0.0, 0.0, 0.0, 0.0};   // normal priors' terms (role 1 forms them)
        unsigned long long nmask = 0ull;
        if (NORMP) nmask = ks->norm_mask | (kBigD ? (unsigned long long)ks->norm_mask_hi << 32 : 0ull);
        propose_pair<ROLE, NORMP>(dev, r, v, C + cl.elem(), C + cl.mean(0), x, C + cl.linv(0),
                                  lfirst, ok, nmask, C, sc);
        double s0 = kBigD ? (sc[0] + sc[1]) + (sc[2] + sc[3]) : sc[0];
        constexpr int kXP = 3 + kDB + (NORMP ? 1 : 0);   // slots of chi2 chains 1..3 (d > 32)
        double pc[4] = {0.0, 0.0, 0.0, 0.0};
        double chi2, r_next = 0.0, Ea_next = 0.0;
        if (ROLE == 0) {
            chi2 = tri_stream<false, true, true, true, lptr, 0, kNTA, true>(
                dev, C + cl.linv(0), dev[kSplit - 1], nullptr, v, vhead, lfirst, rng, pc);
            chi2 = select_by_mask(ok, chi2, INFINITY);  // outside: chi2 is not finite
            X[0] = chi2;
            X[256] = rng.r;
            X[512] = rng.Ea;
            if (kBigD) {
#pragma unroll
                for (int c = 1; c < 4; ++c) X[(kXP + c - 1) * 256] = pc[c];
            }
            exchange_barrier();
#pragma unroll
            for (int q = 0; q < kDB; ++q) yb[kSplit + q] = X[(3 + q) * 256];
            if (NORMP) s0 = X[(3 + kDB) * 256];
        } else {
            StepRng none;
            tri_stream<true, true, true, false, lptr, kNTA, NT, false>(
                dev, C + cl.linv(0), dev[D - 1], yb, v, vhead, lfirst, none);
            yb[D - 1] = select_by_mask(ok, yb[D - 1], INFINITY);
#pragma unroll
            for (int q = 0; q < kDB; ++q) X[(3 + q) * 256] = yb[kSplit + q];
            if (NORMP) X[(3 + kDB) * 256] = s0;
            exchange_barrier();
            chi2 = X[0];
            r_next = X[256];
            Ea_next = X[512];
            if (kBigD) {
#pragma unroll
                for (int c = 1; c < 4; ++c) pc[c] = X[(kXP + c - 1) * 256];
            }
        }
        if (kBigD) {  // kSplit is a whole number of row blocks: row kSplit + q is chain q mod 4
            pc[0] = chi2;
#pragma unroll
            for (int q = 0; q < kDB; ++q)
                pc[q & 3] = fma(yb[kSplit + q], yb[kSplit + q], pc[q & 3]);
            chi2 = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        } else {
#pragma unroll
            for (int q = 0; q < kDB; ++q) chi2 = fma(yb[kSplit + q], yb[kSplit + q], chi2);
        }
        const bool inb = chi2 < INFINITY;  // false for +inf and NaN: outside the prior support
        const double lp = ks->uniform_logp + s0;
        const double ll = -0.5 * (ks->cnorm0 + chi2);
        const double lt = inb ? lp + ll : -INFINITY;
        // ---- Metropolis test (mcmc.py:678-683), identical in both roles
        const bool accept = inb & (lt != -INFINITY) &
                            ((lt > lpost) |
                             (Ea > (UNIT_T ? lpost - lt : (lpost - lt) / ks->temperature)));
        const double ra = accept ? r : 0.0;  // fma(0, v, x) == x exactly (v finite)
        axpy_stream<true, NX>(x, ra, v, x, vhead);
        lpost = accept ? lt : lpost;
        if (ROLE == 0) {  // bookkeeping (mcmc.py:685-748)
            burn -= (accept & (burn > 0)) ? 1 : 0;
            llik = accept ? ll : llik;
            lpri = accept ? lp : lpri;
            prej = accept ? 0 : (prej + (inb ? 0 : 1));
            wt = accept ? 1 : wt + 1;
            nacc += accept ? 1 : 0;
            if (!accept) {
                const double max_now = ks->max_tries * (burn > 0 ? 10.0 : 1.0);
                if ((double)(wt - prej) > max_now) atomicCAS(ks->stuck, 0, 1 + (int)gid);
            }
        } else {
            r = r_next;
            Ea = Ea_next;
        }
        if (++col == ks->cps) {  // next cycle: its slab was DMA'd during this one
            col = 0;
            ++cyc;
            if (s + 1 < n_steps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // the buffer of the cycle just finished receives the cycle after the next
                if (cyc + 1 < ks->ncyc) stage_dma(ks, cyc + 1, (cyc + 1) & 1);
            }
        }
    }

    typedef const StepArgs __attribute__((address_space(4))) * kaptr;
    unsigned long long kav = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue" : "+s"(kav));
    const kaptr ka = (kaptr)kav;
    double* const ox = ka->x;
    if (ROLE == 0) {
#pragma unroll
        for (int i = 0; i < kSplit; ++i) ox[(size_t)i * W + w] = x[i];
        ka->logpost[w] = lpost;
        ka->logprior[w] = lpri;
        ka->loglike[w] = llik;
        ka->weight[w] = wt; ka->prior_rej[w] = prej; ka->burn_left[w] = burn;
        ka->n_accept[w] = nacc;
        wave_add_accepts(ka->accept_total, nacc - nacc0);
    } else {
#pragma unroll
        for (int i = kSplit; i < D; ++i) ox[(size_t)i * W + w] = x[i];
    }
}

template <bool UNIT_T, bool NORMP>
__global__ void __launch_bounds__(512) step_pair_kernel(const StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if constexpr (kPair) {
        const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
        if (role == 0) pair_steps<0, UNIT_T, NORMP>(a, (lds_t)smem);
        else pair_steps<1, UNIT_T, NORMP>(a, (lds_t)smem);
    }
}

#if MCMC_D <= 32
// ---------------------------------------------------------------- the dragging kernel
// One dragging step per walker and iteration (mcmc.py:564-668), the arithmetic and order of
// oracle/mcmc_oracle.c drag_core: a slow proposal to the end point, n_drag interpolation
// steps that move start and end point together along fast directions under the interpolated
// posterior, and the final test on the averaged log-posteriors.  Not a hot kernel: general
// evaluator, directions read from global memory, the start point kept in LDS.
__device__ __forceinline__ void drag_variates(uint32_t key0, uint32_t key1, uint32_t gid,
                                              unsigned long long step, uint32_t sub, bool oned,
                                              double& r, double& Ea)
{
    StepRng rng;
    rng.begin(key0, key1, gid, step, sub);
    rng.run_all();
    r = rng.r;
    Ea = rng.Ea;
    if (oned) {  // RandProposer1D, proposal.py:85-93
        double sn, cs;
        sincos2pi(rng.ka, sn, cs);
        const double rr = rng.expo ? rng.Er : sqrt(2.0 * rng.Er) * fabs(cs);
        r = (rng.c0 & 0x80u) ? rr : -rr;
        const u32x4 q4 = philox4x32_10(key0, key1, gid, kStreamStep | (sub << 16) | 0x100u,
                                       (uint32_t)step, (uint32_t)(step >> 32));
        Ea = -dlog(u52(((uint64_t)q4.w0 << 20) | (q4.w1 >> 12)));
    }
}

__device__ __forceinline__ bool metropolis(double trial, double current, double T, double Ea)
{
    return (trial != -INFINITY) & ((trial > current) | (Ea > (current - trial) / T));
}

template <bool MULTI>
__global__ void __launch_bounds__(256) drag_kernel(const DragArgs da)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const StepArgs& a = da.s;
    const ConstLayout cl{D, a.n_modes};
    const cptr C = as_const(a.cblock);
    // blocks are 256 wide when W allows (one wave per SIMD of a CU: single-wave workgroups
    // get piled onto one SIMD by the dispatcher), else 64
    const int tid = threadIdx.x, bs = blockDim.x;
    const int w = blockIdx.x * bs + tid;
    const int W = a.W;
    const int group = __builtin_amdgcn_readfirstlane(w / a.group_size);
    double* const sC = smem + tid;             // start point: sC[i * bs]
    double* const sX = smem + D * bs + tid;    // current point, restored on rejection
    double* const sA = smem + 2 * D * bs + tid;  // MULTI: mode log-densities [K][bs]
    double ce[D], t[D];
#pragma unroll
    for (int i = 0; i < D; ++i) ce[i] = a.x[(size_t)i * W + w];
    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    int nrow = a.rows ? a.n_rows[w] : 0;
    const uint32_t gid = a.walker0 + (uint32_t)w;
    const int n = da.n_drag;
    auto evaluate = [&](double& lp, double& ll) -> double {
        bool inb;
        eval_point<MULTI, false, true>(t, C, cl, a.norm_mask, a.uniform_logp, sA, bs, inb, lp, ll,
                                       nullptr);
        return inb ? lp + ll : -INFINITY;
    };
    for (int s = 0; s < a.n_steps; ++s) {
        const unsigned long long step = a.step0 + (unsigned long long)s;
        const unsigned long long cyc = step / (unsigned long long)a.cps;
        const int col = (int)(step % (unsigned long long)a.cps);
        const size_t slot = ((size_t)group * a.ncyc + (size_t)(cyc - da.cyc0)) ;
        const double* __restrict__ vs = a.V + slot * a.slab + (size_t)col * D;
        const bool oned0 = a.vflag != nullptr && a.vflag[slot * a.cps + col] != 0;
        double r0, Ea0;
        drag_variates(a.key0, a.key1, gid, step, 0, oned0, r0, Ea0);
        // start point = current point (LDS), end point = slow proposal
#pragma unroll
        for (int i = 0; i < D; ++i) {
            sC[i * bs] = ce[i];
            sX[i * bs] = ce[i];
            t[i] = fma(r0, vs[i], ce[i]);
        }
        if (a.periodic_mask) {
#pragma unroll
            for (int i = 0; i < D; ++i)
                if ((a.periodic_mask >> i) & 1u)
                    t[i] = wrap_periodic(t[i], C[cl.lo() + i], C[cl.hi() + i]);
        }
        double ce_lp, ce_ll;
        double ce_lt = evaluate(ce_lp, ce_ll);
        const bool dead = ce_lt == -INFINITY;   // mcmc.py:590-592
        // ce <- end point (the current point stays in LDS until the final test)
#pragma unroll
        for (int i = 0; i < D; ++i) ce[i] = t[i];
        double cs_lt = lpost;
        double start_acc = cs_lt, end_acc = ce_lt;
        for (int i = 1; i <= n; ++i) {
            const unsigned long long f = step * (unsigned long long)n + (unsigned long long)(i - 1);
            const unsigned long long fc = f / (unsigned long long)da.cps_f;
            const int fcol = (int)(f % (unsigned long long)da.cps_f);
            const size_t fslot = (size_t)group * da.ncyc_f + (size_t)(fc - da.cyc0_f);
            const double* __restrict__ vf = da.Vf + fslot * da.slab_f + (size_t)fcol * D;
            const bool oned = da.vflag_f != nullptr && da.vflag_f[fslot * da.cps_f + fcol] != 0;
            double ri, Eai;
            drag_variates(a.key0, a.key1, gid, step, (uint32_t)i, oned, ri, Eai);
            auto delta = [&](int k) -> double {
                double dk = ri * vf[k];
                if ((a.periodic_mask >> k) & 1u)   // the reference wraps the DELTA (mcmc.py:606)
                    dk = wrap_periodic(dk, C[cl.lo() + k], C[cl.hi() + k]);
                return dk;
            };
#pragma unroll
            for (int k = 0; k < D; ++k) t[k] = sC[k * bs] + delta(k);
            double ps_lp, ps_ll;
            const double ps_lt = evaluate(ps_lp, ps_ll);
#pragma unroll
            for (int k = 0; k < D; ++k) t[k] = ce[k] + delta(k);
            double pe_lp, pe_ll;
            const double pe_lt = evaluate(pe_lp, pe_ll);
            const double frac = (double)i / (double)(1 + n);
            const double pi = (1.0 - frac) * ps_lt + frac * pe_lt;
            const double ci = (1.0 - frac) * cs_lt + frac * ce_lt;
            const bool ok = !dead & (ps_lt != -INFINITY) & (pe_lt != -INFINITY) &
                            metropolis(pi, ci, a.temperature, Eai);
            if (ok) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    sC[k * bs] = sC[k * bs] + delta(k);
                    ce[k] = t[k];
                }
                cs_lt = ps_lt;
                ce_lp = pe_lp; ce_ll = pe_ll; ce_lt = pe_lt;
            }
            start_acc += cs_lt;
            end_acc += ce_lt;
        }
        const double navg = (double)(1 + n);
        const bool accept = !dead & metropolis(end_acc / navg, start_acc / navg, a.temperature, Ea0);
        // bookkeeping (mcmc.py:685-748); a dead slow proposal only adds weight
        if (accept) {
            // the point that is left enters the collection with its weight
            // (mcmc.py:660-668 -> process_accept_or_reject, 691-707); sX still holds it
            if (burn > 0) {
                --burn;
            } else if (a.rows) {
                if (nrow < a.row_cap) {
                    double* row = a.rows + ((size_t)w * a.row_cap + nrow) * (D + 4);
                    row[0] = (double)wt; row[1] = lpost; row[2] = lpri; row[3] = llik;
#pragma unroll
                    for (int i = 0; i < D; ++i) row[4 + i] = sX[i * bs];
                }
                ++nrow;  // rows beyond the capacity are counted as dropped
            }
            lpri = ce_lp; llik = ce_ll; lpost = ce_lt;
            wt = 1; prej = 0; ++nacc;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) ce[i] = sX[i * bs];
            wt += 1;
            if (!dead) {   // the end point is always inside the prior support here
                const double max_now = a.max_tries * (burn > 0 ? 10.0 : 1.0);
                if ((double)(wt - prej) > max_now) atomicCAS(a.stuck, 0, 1 + (int)gid);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) a.x[(size_t)i * W + w] = ce[i];
    a.logpost[w] = lpost; a.logprior[w] = lpri; a.loglike[w] = llik;
    a.weight[w] = wt; a.prior_rej[w] = prej; a.burn_left[w] = burn;
    a.n_accept[w] = nacc;
    if (a.rows) a.n_rows[w] = nrow;
    wave_add_accepts(a.accept_total, nacc - nacc0);
}

// ---------------------------------------------------------------- Haar basis kernel
// Two (group, cycle) problems per 64-lane workgroup, one per half-wave (D <= 32 rows each):
// Box-Muller normals on the basis Philox stream, Householder construction of
// functions.py:45-61 with lane l owning row l of H in VGPRs, then
// V[c][i] = sum_{k<=i} T[i][k] R[k][c] (proposal.py:222-224, 256-260).
__global__ void __launch_bounds__(64) basis_kernel(const BasisArgs a, int n_problems)
{
    constexpr int NZ = (D + 2) * (D - 1) / 2;
    constexpr int LDH = D | 1;  // odd leading dimension: conflict-free column reads
    // (round 6: the normals and R share one buffer -- the normals are dead when R is written --, so a
    // workgroup holds 17 KB of LDS instead of 24 at D = 30 and nine of them fit a CU instead of six:
    // the kernel is one wave per workgroup of dependent fma chains, only more waves cover them)
    constexpr int NZP = NZ + 2;
    constexpr int kBuf = 2 * NZP > 2 * D * LDH ? 2 * NZP : 2 * D * LDH;
    __shared__ double sbuf[kBuf];
    double (*const sz)[NZP] = (double (*)[NZP])sbuf;
    double (*const sR)[D * LDH] = (double (*)[D * LDH])sbuf;
    __shared__ double sx[2][D + 1];
    const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
    const int prob = 2 * blockIdx.x + half;          // (group, cycle) pair of this half-wave
    const bool valid = prob < n_problems;
    const int pg = valid ? prob / a.ncyc : 0, pc = valid ? prob % a.ncyc : 0;
    const uint32_t group = a.group0 + (uint32_t)pg;
    const uint32_t cycle = a.cycle0 + (uint32_t)pc;
    double* __restrict__ Vout = a.V + ((size_t)pg * a.ncyc + pc) * v_slab(D);

    // (round 6: the transform through the constant address space -- wave-uniform addresses, scalar
    // loads, scalar operands of the fmas below -- instead of a copy in LDS read once per term:
    // 465 ds_reads per lane and 7 KB of LDS per workgroup less at D = 30)
    const cptr cT = as_const(a.T);
    if (D == 1) {
        if (valid && l == 0) Vout[0] = a.T[0];
        return;
    }
    for (int j = l; 2 * j < NZ; j += 32) {
        const u32x4 w4 = philox4x32_10(a.key0, a.key1, group, kStreamBasis, cycle, (uint32_t)j);
        const uint64_t ka = ((uint64_t)w4.w0 << 20) | (w4.w1 >> 12);
        const uint64_t kb = ((uint64_t)w4.w2 << 20) | (w4.w3 >> 12);
        const double rad = sqrt(-2.0 * dlog(u52(ka)));
        double sn, cs;
        sincos2pi(kb, sn, cs);
        sz[half][2 * j] = rad * cs;
        sz[half][2 * j + 1] = rad * sn;  // sz has room for the unused odd tail
    }
    __syncthreads();

    const double* __restrict__ z = sz[half];
    double* __restrict__ xs = sx[half];
    // The scalars of the D - 1 reflections -- norm, sign, pivot, denominator -- depend on the
    // normals alone, not on each other: lane n forms those of reflection n (its own ascending fma
    // chain over the m = D - n normals of that reflection, two square roots), all reflections at
    // once, instead of every lane repeating them one reflection after the other inside the
    // sequential loop below (round 4: two dependent square roots per reflection were half of
    // the kernel's 25 us -- it is one wave per SIMD of pure latency).  Same operations in the same
    // order as before (orc_haar_from_normals): the results are bit-identical.
    // (round 6) ... and RN(1 / den): the m normals of a reflection are divided by `den` through it
    // (div_by, det_math.h: the IEEE quotient in 5 instructions instead of the ~35 of the division
    // sequence -- 29 of those per lane were a fifth of this kernel's instructions; one true
    // division per reflection is left, here)
    __shared__ double sPivot[2][D], sDen[2][D], sSign[2][D], sRcp[2][D];
    if (l < D - 1) {
        const int n = l, m = D - n, ix = n * D - n * (n - 1) / 2;
        double norm2 = 0.0;
        for (int k = 0; k < m; ++k) norm2 = fma(z[ix + k], z[ix + k], norm2);
        const double x0 = z[ix];
        const double Dn = (x0 < 0.0) ? -1.0 : 1.0;
        const double x0n = x0 + Dn * sqrt(norm2);
        double tt = norm2 - x0 * x0;
        tt = tt + x0n * x0n;
        sPivot[half][n] = x0n;
        const double den_n = sqrt(0.5 * tt);
        sDen[half][n] = den_n;
        sRcp[half][n] = 1.0 / den_n;
        sSign[half][n] = Dn;
    }
    __syncthreads();
    double H[D];
#pragma unroll
    for (int k = 0; k < D; ++k) H[k] = (k == l) ? 1.0 : 0.0;
    double dprod = 1.0, Dmine = 1.0;
    int ix = 0;
#pragma unroll
    for (int n = 0; n < D - 1; ++n) {
        const int m = D - n;
        const double Dn = sSign[half][n];
        dprod *= Dn;
        if (l == n) Dmine = Dn;
        const double x0n = sPivot[half][n], den = sDen[half][n], rcp = sRcp[half][n];
        if (n) __syncthreads();   // (the previous reflection's xs has been used)
        if (l < m) xs[l] = div_by((l == 0) ? x0n : z[ix + l], den, rcp);
        __syncthreads();
        double tmp = 0.0;
#pragma unroll
        for (int k = 0; k < m; ++k) tmp = fma(H[n + k], xs[k], tmp);
#pragma unroll
        for (int k = 0; k < m; ++k) H[n + k] = fma(-tmp, xs[k], H[n + k]);
        ix += m;
    }
    if (l == D - 1) Dmine = (((D - 1) & 1) ? -1.0 : 1.0) * dprod;
    __syncthreads();   // (the last reflection has read its normals: R may overwrite them)
    if (l < D) {
#pragma unroll
        for (int k = 0; k < D; ++k) sR[half][l * LDH + k] = Dmine * H[k];
    }
    __syncthreads();
    // l = column c of R; V[c][i] for i ascending.  (round 6) The column goes back to LDS -- R is in
    // registers by then -- and the slab leaves the half-wave as D * D CONTIGUOUS doubles: written
    // straight from the lanes, Vout[l * D + i], every store touched 30 cache lines 240 bytes apart
    // (with a basis per walker: 1.9 GB of such stores per 4 d steps, TCP_PENDING_STALL_CYCLES 82 % of the
    // kernel's duration, profiles/r06_basis_kernel.txt)
    double Rc[D];
    if (l < D) {
#pragma unroll
        for (int k = 0; k < D; ++k) Rc[k] = sR[half][k * LDH + l];
    }
    __syncthreads();   // (every column has been read: the buffer takes V)
    if (l < D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) s = fma(cT[i * D + k], Rc[k], s);
            sR[half][l * LDH + i] = s;
        }
    }
    __syncthreads();
    if (valid) {
        const double* __restrict__ sV = sR[half];
        for (int j = l; j < D * D; j += 32) {
            const int c = j / D, i = j - c * D;
            Vout[j] = sV[c * LDH + i];
        }
    }
}

// ---------------------------------------------------------------- batch evaluator
// logprior / loglike / derived of n arbitrary points through the same device functions as
// the step kernel: the parity hook for model.logposterior (model.py:579-678).
template <bool MULTI, bool DERIVED>
__global__ void __launch_bounds__(64) evaluate_kernel(const EvalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sA[];  // MULTI only: [K][64]
    const ConstLayout cl{D, a.n_modes};
    const cptr C = as_const(a.cblock);
    const int tid = threadIdx.x;
    const int p = blockIdx.x * 64 + tid;
    if (p >= a.n) return;
    double t[D];
#pragma unroll
    for (int i = 0; i < D; ++i) t[i] = a.x[(size_t)p * D + i];
    bool inb;
    double lp, ll;
    double* der = DERIVED ? a.derived + (size_t)p * (cl.K > 0 ? cl.K : 1) * D : nullptr;
    eval_point<MULTI, DERIVED, true>(t, C, cl, a.norm_mask, a.uniform_logp, sA + tid, 64, inb, lp,
                                     ll, der);
    a.logprior[p] = inb ? lp : -INFINITY;
    a.loglike[p] = inb ? ll : -INFINITY;
}

// ---------------------------------------------------------------- moments
// Per group: sum_x[i] and S_g[i][j] = sum_l x_i x_j over the group's walkers in ascending
// order (sequential fma chains: deterministic), X tile staged in LDS.
__global__ void __launch_bounds__(256) group_moments_kernel(const MomentArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sX[];  // [gs][D|1]
    constexpr int LDX = D | 1;
    const int tid = threadIdx.x, gs = blockDim.x, g = blockIdx.x;
    const int w = g * gs + tid;
#pragma unroll
    for (int i = 0; i < D; ++i) sX[tid * LDX + i] = a.x[(size_t)i * a.W + w] - a.shift[i];
    __syncthreads();
    // Every sum is ONE chain over the group's walkers in ascending order (the specification); a
    // chain's operands are read from LDS eight walkers at a time, so that the reads of a batch
    // travel together instead of one LDS round trip per term (round 4: the kernel has one wave
    // per SIMD and nothing else to cover them: 14.6 -> ~6 us at group_size 256, d = 30).  A thread
    // runs two chains side by side: the pairs p and p + gs, or -- the last D threads, whose
    // second pair does not exist when NPAIR <= 2 gs - D -- a pair and a group sum.
    auto pair_of = [](int p, int& i, int& j) {   // p = i(i+1)/2 + j, i >= j
        i = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= p) ++i;
        while (i * (i + 1) / 2 > p) --i;
        j = p - i * (i + 1) / 2;
    };
    const bool sums_ride = NPAIR + D <= 2 * gs;   // the group sums fit beside the second pairs
    for (int p0 = tid; p0 < NPAIR || (sums_ride && p0 < gs); p0 += 2 * gs) {
        const int p1 = p0 + gs;
        const bool has0 = p0 < NPAIR, has1 = p1 < NPAIR;
        const int sidx = tid - (gs - D);                       // the group sum this thread carries
        const bool sum1 = sums_ride && !has1 && p0 < gs && sidx >= 0 && sidx < D;
        int i0 = 0, j0 = 0, i1 = 0, j1 = 0;
        if (has0) pair_of(p0, i0, j0);
        if (has1) pair_of(p1, i1, j1);
        if (sum1) i1 = sidx;
        double s0 = 0.0, s1 = 0.0;
        int l = 0;
        for (; l + 8 <= gs; l += 8) {
            double a0[8], b0[8], a1[8], b1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a0[q] = sX[(l + q) * LDX + i0]; b0[q] = sX[(l + q) * LDX + j0];
                a1[q] = sX[(l + q) * LDX + i1]; b1[q] = sX[(l + q) * LDX + j1];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                s0 = fma(a0[q], b0[q], s0);
                s1 = sum1 ? s1 + a1[q] : fma(a1[q], b1[q], s1);
            }
        }
        for (; l < gs; ++l) {
            s0 = fma(sX[l * LDX + i0], sX[l * LDX + j0], s0);
            s1 = sum1 ? s1 + sX[l * LDX + i1] : fma(sX[l * LDX + i1], sX[l * LDX + j1], s1);
        }
        if (has0) a.Sg[(size_t)g * NPAIR + p0] = s0;
        if (has1) a.Sg[(size_t)g * NPAIR + p1] = s1;
        if (sum1) a.group_sum[(size_t)g * D + sidx] += s1;
    }
    if (!sums_ride && tid < D) {   // (small groups: the sums on their own)
        double s = 0.0;
        for (int l = 0; l < gs; ++l) s = s + sX[l * LDX + tid];
        a.group_sum[(size_t)g * D + tid] += s;
    }
}

__global__ void __launch_bounds__(64) pool_moments_kernel(const MomentArgs a)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= NPAIR) return;
    double acc = a.pooled[p];
    // loads batched 64 deep (a batch is one L2 round trip: 16 deep the 256 groups of config 2 took
    // 11 us), additions strictly in ascending group order (spec)
    int g = 0;
    for (; g + 64 <= a.G; g += 64) {
        double v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = a.Sg[(size_t)(g + u) * NPAIR + p];
#pragma unroll
        for (int u = 0; u < 64; ++u) acc += v[u];
    }
    for (; g + 16 <= a.G; g += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = a.Sg[(size_t)(g + u) * NPAIR + p];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; g < a.G; ++g) acc += a.Sg[(size_t)g * NPAIR + p];
    a.pooled[p] = acc;
}

#endif  // MCMC_D <= 32
// ---------------------------------------------------------------- launchers
constexpr size_t kLdsMax = 160 * 1024;
size_t pair_lds(const StepArgs& a)
{
    return sizeof(double) * (size_t)(2 * (256 / a.group_size) * a.slab +
        2 * ((a.norm_mask | a.norm_mask_hi) != 0u ? PairGeom<true>::xf : PairGeom<false>::xf) * 256);
}
// the two-wave kernel serves: one mode, non-periodic priors (normal ones have their own
// instantiation), no emitted rows, no one-parameter blocks, whole 256-walker workgroups
bool pair_fits(const StepArgs& a)
{
    // (with normal priors the matrix-core kernel is ahead again from d = 55: 7.3 against 7.0)
    if (D >= 55 && (a.norm_mask | a.norm_mask_hi) != 0u) return false;
    return kPair && a.periodic_mask == 0u && a.n_modes == 1 && a.rows == nullptr &&
           a.vflag == nullptr && a.W % 256 == 0 && 256 % a.group_size == 0 &&
           pair_lds(a) <= kLdsMax;
}
hipError_t launch_pair(const StepArgs& a, hipStream_t st)
{
    // two waves per 64 walkers: 512-thread workgroups of 256 walkers
    const bool normp = (a.norm_mask | a.norm_mask_hi) != 0u;
    size_t plds = pair_lds(a);
    const int nwg = a.W / 256;
    const int per_cu = (nwg + 255) / 256;      // even placement, see launch_step
    size_t want = ((size_t)(160 * 1024) / (size_t)per_cu / 1024) * 1024;
    if (per_cu == 1) want = 96 * 1024;         // > half of the LDS: one workgroup per CU
    if (want > plds) plds = want;
    const bool unit_t = a.temperature == 1.0;
    typedef void (*kern_t)(const StepArgs);
    const kern_t kern = normp ? (unit_t ? step_pair_kernel<true, true> : step_pair_kernel<false, true>)
                              : (unit_t ? step_pair_kernel<true, false> : step_pair_kernel<false, false>);
    hipError_t e = hipFuncSetAttribute((const void*)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
    if (e != hipSuccess) return e;
    mcmc_hip_note_step_kernel(normp ? (unit_t ? "mcmc::step_pair_kernel<true, true>"
                                              : "mcmc::step_pair_kernel<false, true>")
                                    : (unit_t ? "mcmc::step_pair_kernel<true, false>"
                                              : "mcmc::step_pair_kernel<false, false>"));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), plds, st, a);
    return hipGetLastError();
}

#if MCMC_D <= 32
hipError_t launch_step(const StepArgs& a, int group_size, hipStream_t st)
{
    const bool multi = a.n_modes > 1;
    size_t lds = 0;
    const int bs = (a.W % 256 == 0) ? 256 : (a.W % 128 == 0) ? 128 : 64;
    const dim3 grid(a.W / bs), block(bs);
    lds = sizeof(double) * (size_t)(2 * (bs / a.group_size) * a.slab + (multi ? a.n_modes * bs : 0));
    // Placement: the waves of this kernel run for the whole launch, and at W = 65 536 there
    // are exactly as many waves as SIMDs.  Requesting (otherwise unused) LDS so that only
    // ceil(#workgroups / 256 CUs) workgroups fit on a CU makes the dispatcher spread them
    // evenly instead of doubling up waves on some SIMDs while others idle.
    {
        const int per_cu = (int)((grid.x + 255) / 256);
        size_t want = (size_t)(160 * 1024) / (size_t)per_cu;
        want = (want / 1024) * 1024;
        if (want > 64 * 1024) want = 64 * 1024;
        if (want > lds) lds = want;
    }
    const bool general = (a.norm_mask | a.periodic_mask) != 0u || a.n_modes == 0 ||
                         a.rows != nullptr || D == 1 || a.vflag != nullptr;
    if (a.own_basis) {
        // `shared_basis: False`: no directions in LDS (the mode log-densities of a mixture and the
        // placement request only); the GENERAL step with every walker's own columns from HBM
        size_t own_lds = sizeof(double) * (size_t)(multi ? a.n_modes * bs : 0);
        const int per_cu = (int)((grid.x + 255) / 256);
        size_t want = (((size_t)(160 * 1024) / (size_t)per_cu) / 1024) * 1024;
        if (want > 64 * 1024) want = 64 * 1024;
        if (want > own_lds) own_lds = want;
        if (own_lds > kLdsMax) return hipErrorInvalidValue;
        const void* ofn = multi ? (const void*)step_kernel<true, true, true> : (const void*)step_kernel<false, true, true>;
        if (own_lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(ofn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)own_lds);
            if (e != hipSuccess) return e;
        }
        mcmc_hip_note_step_kernel(multi ? "mcmc::step_kernel<true, true, own basis>"
                                        : "mcmc::step_kernel<false, true, own basis>");
        if (multi) hipLaunchKernelGGL((step_kernel<true, true, true>), grid, block, own_lds, st, a);
        else hipLaunchKernelGGL((step_kernel<false, true, true>), grid, block, own_lds, st, a);
        return hipGetLastError();
    }
    if (lds > kLdsMax) return hipErrorInvalidValue;   // the cycle's directions do not fit LDS
    if (pair_fits(a)) return launch_pair(a, st);
    const void* fn = multi ? (general ? (const void*)step_kernel<true, true>
                                      : (const void*)step_kernel<true, false>)
                           : (general ? (const void*)step_kernel<false, true>
                                      : (const void*)step_kernel<false, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(multi ? (general ? "mcmc::step_kernel<true, true>"
                                               : "mcmc::step_kernel<true, false>")
                                    : (general ? "mcmc::step_kernel<false, true>"
                                               : "mcmc::step_kernel<false, false>"));
    if (multi) {
        if (general) hipLaunchKernelGGL((step_kernel<true, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((step_kernel<true, false>), grid, block, lds, st, a);
    } else {
        if (general) hipLaunchKernelGGL((step_kernel<false, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((step_kernel<false, false>), grid, block, lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_basis(const BasisArgs& a, int n_groups, hipStream_t st)
{
    const int n_problems = n_groups * a.ncyc;
    hipLaunchKernelGGL(basis_kernel, dim3((n_problems + 1) / 2), dim3(64), 0, st, a, n_problems);
    return hipGetLastError();
}

hipError_t launch_evaluate(const EvalArgs& a, hipStream_t st)
{
    const bool multi = a.n_modes > 1;
    const size_t lds = sizeof(double) * (size_t)(multi ? a.n_modes * 64 : 0);
    const dim3 grid((a.n + 63) / 64), block(64);
    if (multi) {
        if (a.derived) hipLaunchKernelGGL((evaluate_kernel<true, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((evaluate_kernel<true, false>), grid, block, lds, st, a);
    } else {
        if (a.derived) hipLaunchKernelGGL((evaluate_kernel<false, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((evaluate_kernel<false, false>), grid, block, lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_moments(const MomentArgs& a, int group_size, hipStream_t st)
{
    const size_t lds = sizeof(double) * (size_t)group_size * (D | 1);
    hipLaunchKernelGGL(group_moments_kernel, dim3(a.G), dim3(group_size), lds, st, a);
    hipLaunchKernelGGL(pool_moments_kernel, dim3((NPAIR + 63) / 64), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_drag(const DragArgs& a, hipStream_t st)
{
    const bool multi = a.s.n_modes > 1;
    const int bs = (a.s.W % 256 == 0) ? 256 : 64;
    const size_t lds = sizeof(double) * bs * (size_t)(2 * D + (multi ? a.s.n_modes : 0));
    const dim3 grid(a.s.W / bs), block(bs);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(multi ? (const void*)drag_kernel<true> : (const void*)drag_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel(multi ? "mcmc::drag_kernel<true>" : "mcmc::drag_kernel<false>");
    if (multi) hipLaunchKernelGGL(drag_kernel<true>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(drag_kernel<false>, grid, block, lds, st, a);
    return hipGetLastError();
}

const DimKernels kKernels = {launch_step, launch_basis, launch_evaluate, launch_moments,
                             launch_drag};
#else
const PairKernels kPairKernels = {pair_fits, launch_pair};
#endif

}  // namespace
}  // namespace mcmc

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)
#if MCMC_D <= 32
extern "C" const mcmc::DimKernels* MCMC_CAT(mcmc_hip_dim_, MCMC_D)() { return &mcmc::kKernels; }
#else
static_assert(MCMC_D <= mcmc::kMaxDimPair, "two-wave kernel: d <= kMaxDimPair");
extern "C" const mcmc::PairKernels* MCMC_CAT(mcmc_hip_pair_, MCMC_D)() { return &mcmc::kPairKernels; }
#endif
