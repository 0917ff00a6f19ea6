// The GENERAL Metropolis step (gfx950), run-time d, compiled once.  It serves
//   * 32 < d <= 128: everything the specialised d > 32 kernels leave out -- Gaussian mixtures
//     (and `one`), periodic parameters, emitted rows with burn-in, normal priors and ensembles
//     that are not whole 256-walker workgroups;
//   * 2 <= d <= 128 with MCMC_HIP_FLAG_OWN_BASIS (`shared_basis: False`): every walker reads the
//     columns of its OWN Haar basis (proposal.py:59-69 to the letter) -- the reference-faithful
//     control of the shared-basis design, not a fast path.
// Not a hot kernel: one lane per walker with state and trial in LDS (dimension-major,
// conflict-free), the problem constants read from global memory at wave-uniform addresses.
//
// Restates the same reference lines as walker_kernels.hip (mcmc.py:545-562, 670-748;
// prior.py:658-676, 733-763; gaussian_mixture.py:138-163) in the d > 32 order of the
// specification (DESIGN.md "Ensemble specification", oracle/mcmc_oracle.c eval_point): sums over
// dimensions / rows as four interleaved chains combined (s0 + s1) + (s2 + s3) -- and for d <= 32
// (own-basis mode only) in the d <= 32 order: one ascending chain.
#include "det_math.h"
#include "kernels.h"

namespace mcmc {
namespace {

__device__ __forceinline__ double wrap_periodic(double t, double lo, double hi)
{
    const double w = hi - lo;
    const double y = (t - lo) / w;
    const double m = y - floor(y);
    return m * w + lo;
}

// chi2 chains of one mode for the point in `st` (LDS, stride 64): pc[j & cm] takes y_j^2 in
// ascending j, y_j = sum_{i <= j} L^-1[j][i] (t_i - mu_i) (eval_point's order)
typedef const double __attribute__((address_space(4))) * cdbl;
__device__ __forceinline__ void general_tri_chi2(const double* __restrict__ Lk, const cdbl mu,
                                                 const double* st, const int d, const int cm,
                                                 double (&pc)[4])
{
    // y_j = sum_{i <= j} L^-1[j][i] (t_i - mu_i): one ascending chain per row (the
    // specification).  Round 5: FOUR rows at a time -- four independent chains share
    // every deviation read from LDS -- and eight terms per batch: the elements of a
    // row and the means are contiguous and wave-uniform and arrive with one wide scalar
    // load per batch (through the constant address space: as plain global pointers the
    // compiler issued one VECTOR load per element and lane, and the kernel's time was
    // those loads: 5.76 -> 3.12 ms per 120 steps of 65 536 walkers on this change alone)
    int j = 0;
    for (; j + 4 <= d; j += 4) {
        const cdbl r0 = (cdbl)(unsigned long long)(Lk + (size_t)j * d);
        const cdbl r1 = r0 + d, r2 = r1 + d, r3 = r2 + d;
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
        int i = 0;
        for (; i + 8 <= j + 1; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double dv = st[64 * (i + u)] - mu[i + u];
                y0 = fma(r0[i + u], dv, y0);
                y1 = fma(r1[i + u], dv, y1);
                y2 = fma(r2[i + u], dv, y2);
                y3 = fma(r3[i + u], dv, y3);
            }
        }
        for (; i <= j; ++i) {
            const double dv = st[64 * i] - mu[i];
            y0 = fma(r0[i], dv, y0);
            y1 = fma(r1[i], dv, y1);
            y2 = fma(r2[i], dv, y2);
            y3 = fma(r3[i], dv, y3);
        }
        const double d1 = st[64 * (j + 1)] - mu[j + 1], d2 = st[64 * (j + 2)] - mu[j + 2],
                     d3 = st[64 * (j + 3)] - mu[j + 3];
        y1 = fma(r1[j + 1], d1, y1);
        y2 = fma(r2[j + 1], d1, y2);
        y3 = fma(r3[j + 1], d1, y3);
        y2 = fma(r2[j + 2], d2, y2);
        y3 = fma(r3[j + 2], d2, y3);
        y3 = fma(r3[j + 3], d3, y3);
        pc[j & cm] = fma(y0, y0, pc[j & cm]);
        pc[(j + 1) & cm] = fma(y1, y1, pc[(j + 1) & cm]);
        pc[(j + 2) & cm] = fma(y2, y2, pc[(j + 2) & cm]);
        pc[(j + 3) & cm] = fma(y3, y3, pc[(j + 3) & cm]);
    }
    for (; j < d; ++j) {
        const cdbl row = (cdbl)(unsigned long long)(Lk + (size_t)j * d);
        double y = 0.0;
        for (int i = 0; i <= j; ++i) y = fma(row[i], st[64 * i] - mu[i], y);
        pc[j & cm] = fma(y, y, pc[j & cm]);
    }
}

__global__ void __launch_bounds__(64) step_general_kernel(const GeneralStepArgs b)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const StepArgs& a = b.s;
    const int d = b.d, K = a.n_modes;
    const ConstLayout cl{d, K};
    const int tid = threadIdx.x;
    const int w = blockIdx.x * 64 + tid;
    const int W = a.W;
    double* const sx = smem + tid;            // x[i] at sx[64 i]
    double* const st = smem + 64 * d + tid;   // trial
    double* const sa = smem + 128 * d + tid;  // mode log-pdfs a_k at sa[64 k]
    // (the problem constants through the constant address space: wave-uniform addresses, scalar
    // loads -- as plain global pointers the compiler issued a vector load per lane for each)
    const cdbl C = (cdbl)(unsigned long long)a.cblock;
    // shared basis: the group of the wave (group_size >= 64); own basis: one "group" per walker
    const int group = b.own_basis ? w : __builtin_amdgcn_readfirstlane(w / a.group_size);
    const int ldv = b.ld;
    const int cm = d > 32 ? 3 : 0;   // chain of dimension / row i: i & cm
    const double* const Vgrp = a.V + (size_t)group * a.ncyc * (size_t)a.slab;

    for (int i = 0; i < d; ++i) sx[64 * i] = a.x[(size_t)i * W + w];
    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    int nrow = a.rows ? a.n_rows[w] : 0;
    const uint32_t gid = a.walker0 + (uint32_t)w;
    unsigned long long step = a.step0;
    // columns per cycle: d for one block, sum_b oversample_b n_b with parameter blocks
    // (proposal.py:96-224; the columns of a one-parameter block draw RandProposer1D variates)
    const int cps = a.cps;
    int col = (int)(step % (unsigned long long)cps);
    int cyc = 0;

    for (int s = 0; s < a.n_steps; ++s) {
        const bool oned = a.vflag != nullptr &&
                          a.vflag[((size_t)group * a.ncyc + cyc) * (size_t)cps + col] != 0;
        double r, Ea;
        step_variates(a.key0, a.key1, gid, step, 0, oned, r, Ea);
        const double* __restrict__ v = Vgrp + (size_t)cyc * a.slab + (size_t)col * ldv;
        // ---- trial, periodic wrap, prior support and normal priors (prior.py:658-676, 733-763)
        bool inb = true;
        double sc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < d; ++i) {
            double t = fma(r, v[i], sx[64 * i]);
            const double lo = C[cl.lo() + i], hi = C[cl.hi() + i];
            if ((b.periodic_mask4[i >> 5] >> (i & 31)) & 1u) t = wrap_periodic(t, lo, hi);
            st[64 * i] = t;
            inb = inb & (t <= hi) & (t >= lo);
            if ((b.norm_mask4[i >> 5] >> (i & 31)) & 1u) {
                const double q = (t - C[cl.loc() + i]) / C[cl.scale() + i];
                sc[i & cm] = sc[i & cm] + fma(-0.5 * q, q, C[cl.mls() + i]);
            }
        }
        const double lp = a.uniform_logp + (d > 32 ? (sc[0] + sc[1]) + (sc[2] + sc[3]) : sc[0]);
        // ---- likelihood (gaussian_mixture.py:138-163); lanes outside the support skip it
        double ll = 0.0;
        if (inb && K >= 1) {
            double amax = -INFINITY;
            for (int k = 0; k < K; ++k) {
                const double* __restrict__ Lk = b.Lrow + (size_t)k * d * d;
                const cdbl mu = C + cl.mean(k);
                double pc[4] = {0.0, 0.0, 0.0, 0.0};
                general_tri_chi2(Lk, mu, st, d, cm, pc);
                const double chi2 = d > 32 ? (pc[0] + pc[1]) + (pc[2] + pc[3]) : pc[0];
                const double ak = -0.5 * (C[cl.cnorm() + k] + chi2);
                sa[64 * k] = ak;
                amax = (ak > amax) ? ak : amax;
            }
            if (K == 1) {
                ll = sa[0];
            } else {
                double S = 0.0;
                for (int k = 0; k < K; ++k) S = fma(C[cl.weight() + k], dexp(sa[64 * k] - amax), S);
                ll = dlog(S) + amax;
            }
        }
        const double lt = inb ? lp + ll : -INFINITY;
        // ---- Metropolis test (mcmc.py:678-683) and bookkeeping (mcmc.py:685-748)
        const bool accept = inb & (lt != -INFINITY) &
                            ((lt > lpost) | (Ea > (lpost - lt) / a.temperature));
        if (accept) {
            if (burn <= 0) {
                if (a.rows) {
                    if (nrow < a.row_cap) {
                        double* row = a.rows + ((size_t)w * a.row_cap + nrow) * (d + 4);
                        row[0] = (double)wt; row[1] = lpost; row[2] = lpri; row[3] = llik;
                        for (int i = 0; i < d; ++i) row[4 + i] = sx[64 * i];
                    }
                    ++nrow;  // rows beyond the capacity are counted as dropped
                }
            } else {
                --burn;
            }
            for (int i = 0; i < d; ++i) sx[64 * i] = st[64 * i];
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
        if (++col == cps) { col = 0; ++cyc; }
    }

    for (int i = 0; i < d; ++i) a.x[(size_t)i * W + w] = sx[64 * i];
    a.logpost[w] = lpost; a.logprior[w] = lpri; a.loglike[w] = llik;
    a.weight[w] = wt; a.prior_rej[w] = prej; a.burn_left[w] = burn;
    a.n_accept[w] = nacc;
    wave_add_accepts(a.accept_total, nacc - nacc0);
    if (a.rows) a.n_rows[w] = nrow;
}

// ---------------------------------------------------------------- the general dragging step
// mcmc.py:564-668 for what the tuned dragging kernels leave out (oracle: drag_core): 32 < d <= 128
// from scratch -- any mixture, periodic parameters (the reference wraps the DELTA of an
// interpolation step, mcmc.py:606), one-parameter blocks, emitted rows.  One lane per walker;
// the END point and the trial live in LDS, the START point in a scratch array in HBM (read and
// written once per interpolation step), the walker's own point stays in `x` until the final test.
// Not a hot kernel.
__device__ __forceinline__ bool metropolis_rule(double trial, double current, double T, double Ea)
{
    return (trial != -INFINITY) & ((trial > current) | (Ea > (current - trial) / T));
}

// log-posterior of the point in `st` (LDS, stride 64): prior support and normal priors
// (prior.py:733-763), then the mixture (gaussian_mixture.py:138-163) -- the arithmetic and
// order of step_general_kernel
__device__ __forceinline__ double general_logpost(const GeneralStepArgs& b, const double* st, double* sa,
                                                  double& lp, double& ll)
{
    const StepArgs& a = b.s;
    const int d = b.d, K = a.n_modes;
    const ConstLayout cl{d, K};
    const double* __restrict__ C = a.cblock;
    const int cm = d > 32 ? 3 : 0;
    bool inb = true;
    double sc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < d; ++i) {
        const double t = st[64 * i];
        inb = inb & (t <= C[cl.hi() + i]) & (t >= C[cl.lo() + i]);
        if ((b.norm_mask4[i >> 5] >> (i & 31)) & 1u) {
            const double q = (t - C[cl.loc() + i]) / C[cl.scale() + i];
            sc[i & cm] = sc[i & cm] + fma(-0.5 * q, q, C[cl.mls() + i]);
        }
    }
    lp = a.uniform_logp + (d > 32 ? (sc[0] + sc[1]) + (sc[2] + sc[3]) : sc[0]);
    ll = 0.0;
    if (inb && K >= 1) {
        double amax = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const double* __restrict__ Lk = b.Lrow + (size_t)k * d * d;
            const cdbl mu = (cdbl)(unsigned long long)(C + cl.mean(k));
            double pc[4] = {0.0, 0.0, 0.0, 0.0};
            general_tri_chi2(Lk, mu, st, d, cm, pc);
            const double chi2 = d > 32 ? (pc[0] + pc[1]) + (pc[2] + pc[3]) : pc[0];
            const double ak = -0.5 * (C[cl.cnorm() + k] + chi2);
            sa[64 * k] = ak;
            amax = (ak > amax) ? ak : amax;
        }
        if (K == 1) {
            ll = sa[0];
        } else {
            double S = 0.0;
            for (int k = 0; k < K; ++k) S = fma(C[cl.weight() + k], dexp(sa[64 * k] - amax), S);
            ll = dlog(S) + amax;
        }
    }
    if (!inb) { lp = -INFINITY; ll = -INFINITY; }
    return inb ? lp + ll : -INFINITY;
}

__global__ void __launch_bounds__(64) drag_general_kernel(const GeneralDragArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const GeneralStepArgs& b = g.g;
    const StepArgs& a = b.s;
    const int d = b.d, K = a.n_modes, n = g.n_drag;
    const ConstLayout cl{d, K};
    const int tid = threadIdx.x;
    const int w = blockIdx.x * 64 + tid;
    const int W = a.W;
    double* const se = smem + tid;            // end point ce[i] at se[64 i]
    double* const st = smem + 64 * d + tid;   // trial
    double* const sa = smem + 128 * d + tid;  // mode log-pdfs
    double* const cs = g.cs + w;              // start point cs[i] at cs[i W] (HBM scratch)
    const double* __restrict__ C = a.cblock;
    const int group = __builtin_amdgcn_readfirstlane(w / a.group_size);
    const int ldv = b.ld;
    auto periodic = [&](int i) { return ((b.periodic_mask4[i >> 5] >> (i & 31)) & 1u) != 0u; };

    double lpost = a.logpost[w], lpri = a.logprior[w], llik = a.loglike[w];
    int wt = a.weight[w], prej = a.prior_rej[w], burn = a.burn_left[w];
    long long nacc = a.n_accept[w];
    const long long nacc0 = nacc;
    int nrow = a.rows ? a.n_rows[w] : 0;
    const uint32_t gid = a.walker0 + (uint32_t)w;

    for (int s = 0; s < a.n_steps; ++s) {
        const unsigned long long step = a.step0 + (unsigned long long)s;
        const unsigned long long cyc = step / (unsigned long long)a.cps;
        const int col = (int)(step % (unsigned long long)a.cps);
        const size_t slot = (size_t)group * a.ncyc + (size_t)(cyc - g.cyc0);
        const double* __restrict__ vs = a.V + slot * a.slab + (size_t)col * ldv;
        const bool oned0 = a.vflag != nullptr && a.vflag[slot * a.cps + col] != 0;
        double r0, Ea0;
        step_variates(a.key0, a.key1, gid, step, 0, oned0, r0, Ea0);
        // start point = the walker's point; end point = the slow proposal (wrapped, prior.py:658-676)
        for (int i = 0; i < d; ++i) {
            const double xi = a.x[(size_t)i * W + w];
            cs[(size_t)i * W] = xi;
            double t = fma(r0, vs[i], xi);
            if (periodic(i)) t = wrap_periodic(t, C[cl.lo() + i], C[cl.hi() + i]);
            se[64 * i] = t;
        }
        double ce_lp, ce_ll;
        double ce_lt = general_logpost(b, se, sa, ce_lp, ce_ll);
        const bool dead = ce_lt == -INFINITY;   // mcmc.py:590-592
        double cs_lt = lpost;
        double start_acc = cs_lt, end_acc = ce_lt;
        for (int i = 1; i <= n; ++i) {
            const unsigned long long f = step * (unsigned long long)n + (unsigned long long)(i - 1);
            const unsigned long long fc = f / (unsigned long long)g.cps_f;
            const int fcol = (int)(f % (unsigned long long)g.cps_f);
            const size_t fslot = (size_t)group * g.ncyc_f + (size_t)(fc - g.cyc0_f);
            const double* __restrict__ vf = g.Vf + fslot * g.slab_f + (size_t)fcol * ldv;
            const bool oned = g.vflag_f != nullptr && g.vflag_f[fslot * g.cps_f + fcol] != 0;
            double ri, Eai;
            step_variates(a.key0, a.key1, gid, step, (uint32_t)i, oned, ri, Eai);
            auto delta = [&](int k) -> double {
                double dk = ri * vf[k];
                if (periodic(k)) dk = wrap_periodic(dk, C[cl.lo() + k], C[cl.hi() + k]);
                return dk;
            };
            for (int k = 0; k < d; ++k) st[64 * k] = cs[(size_t)k * W] + delta(k);
            double ps_lp, ps_ll;
            const double ps_lt = general_logpost(b, st, sa, ps_lp, ps_ll);
            for (int k = 0; k < d; ++k) st[64 * k] = se[64 * k] + delta(k);
            double pe_lp, pe_ll;
            const double pe_lt = general_logpost(b, st, sa, pe_lp, pe_ll);
            const double frac = (double)i / (double)(1 + n);
            const double pi = (1.0 - frac) * ps_lt + frac * pe_lt;
            const double ci = (1.0 - frac) * cs_lt + frac * ce_lt;
            const bool ok = !dead & (ps_lt != -INFINITY) & (pe_lt != -INFINITY) &
                            metropolis_rule(pi, ci, a.temperature, Eai);
            if (ok) {
                for (int k = 0; k < d; ++k) {
                    cs[(size_t)k * W] = cs[(size_t)k * W] + delta(k);
                    se[64 * k] = st[64 * k];
                }
                cs_lt = ps_lt;
                ce_lp = pe_lp; ce_ll = pe_ll; ce_lt = pe_lt;
            }
            start_acc += cs_lt;
            end_acc += ce_lt;
        }
        const double navg = (double)(1 + n);
        const bool accept = !dead & metropolis_rule(end_acc / navg, start_acc / navg, a.temperature, Ea0);
        // bookkeeping (mcmc.py:685-748); a dead slow proposal only adds weight
        if (accept) {
            if (burn > 0) {
                --burn;
            } else if (a.rows) {
                if (nrow < a.row_cap) {
                    double* row = a.rows + ((size_t)w * a.row_cap + nrow) * (d + 4);
                    row[0] = (double)wt; row[1] = lpost; row[2] = lpri; row[3] = llik;
                    for (int i = 0; i < d; ++i) row[4 + i] = a.x[(size_t)i * W + w];
                }
                ++nrow;
            }
            for (int i = 0; i < d; ++i) a.x[(size_t)i * W + w] = se[64 * i];
            lpri = ce_lp; llik = ce_ll; lpost = ce_lt;
            wt = 1; prej = 0; ++nacc;
        } else {
            wt += 1;
            if (!dead) {
                const double max_now = a.max_tries * (burn > 0 ? 10.0 : 1.0);
                if ((double)(wt - prej) > max_now) atomicCAS(a.stuck, 0, 1 + (int)gid);
            }
        }
    }
    a.logpost[w] = lpost; a.logprior[w] = lpri; a.loglike[w] = llik;
    a.weight[w] = wt; a.prior_rej[w] = prej; a.burn_left[w] = burn;
    a.n_accept[w] = nacc;
    if (a.rows) a.n_rows[w] = nrow;
    wave_add_accepts(a.accept_total, nacc - nacc0);
}

}  // namespace
}  // namespace mcmc

// ---------------------------------------------------------------- packing of emitted rows
// rows[W][cap][d+4] (what the step kernels fill) -> out[offset[w] + r][d+5] =
// (global walker id, weight, logpost, logprior, loglike, x...): only the rows that exist cross
// PCIe, already in the layout mcmc_hip_drain_samples hands out.  One wave per walker.
namespace mcmc {
namespace {
__global__ void __launch_bounds__(64) pack_rows_kernel(const double* __restrict__ rows,
                                                      const int* __restrict__ n_rows,
                                                      const long long* __restrict__ offset,
                                                      double* __restrict__ out, int cap, int d,
                                                      uint32_t walker0)
{
    const int w = blockIdx.x, l = threadIdx.x;
    const int n = n_rows[w] < cap ? n_rows[w] : cap;
    const int rl = d + 4;
    const double* __restrict__ src = rows + (size_t)w * cap * rl;
    double* __restrict__ dst = out + (size_t)offset[w] * (d + 5);
    for (int e = l; e < n * (d + 5); e += 64) {
        const int r = e / (d + 5), c = e - r * (d + 5);
        dst[e] = c == 0 ? (double)(walker0 + (uint32_t)w) : src[(size_t)r * rl + c - 1];
    }
}
}  // namespace
}  // namespace mcmc

extern "C" hipError_t mcmc_hip_launch_pack_rows(const double* rows, const int* n_rows,
                                                const long long* offset, double* out, int W,
                                                int cap, int d, uint32_t walker0, hipStream_t st)
{
    hipLaunchKernelGGL(mcmc::pack_rows_kernel, dim3(W), dim3(64), 0, st, rows, n_rows, offset, out,
                       cap, d, walker0);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_general_step(const mcmc::GeneralStepArgs* b, hipStream_t st)
{
    using namespace mcmc;
    const size_t lds = sizeof(double) * 64 * (size_t)(2 * b->d + (b->s.n_modes > 0 ? b->s.n_modes : 1));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)step_general_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel("mcmc::step_general_kernel");
    hipLaunchKernelGGL(step_general_kernel, dim3(b->s.W / 64), dim3(64), lds, st, *b);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_general_drag(const mcmc::GeneralDragArgs* g, hipStream_t st)
{
    using namespace mcmc;
    const GeneralStepArgs* b = &g->g;
    const size_t lds = sizeof(double) * 64 * (size_t)(2 * b->d + (b->s.n_modes > 0 ? b->s.n_modes : 1));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)drag_general_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    mcmc_hip_note_step_kernel("mcmc::drag_general_kernel");
    hipLaunchKernelGGL(drag_general_kernel, dim3(b->s.W / 64), dim3(64), lds, st, *g);
    return hipGetLastError();
}
