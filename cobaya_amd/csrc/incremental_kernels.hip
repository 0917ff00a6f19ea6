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
// permutes -- no LDS, no barrier inside a step.  The random variates of four consecutive steps
// are drawn at once, one step per lane class, and fetched by quad broadcasts.  Per step a lane
// reads its (v_i, u_i) pairs from LDS, where the columns of the launch are staged in chunks
// (double-buffered, one workgroup barrier per chunk).
#include <string>

#include "det_math.h"
#include "kernels.h"

namespace mcmc {
namespace {

// ---------------------------------------------------------------- DPP quad helpers
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// (p0 + p1) + (p2 + p3) in every lane of the quad; lane c holds p_c
__device__ __forceinline__ double quad_sum(double p)
{
    const double q = p + quad_perm<0xB1>(p);   // [1,0,3,2]: p0+p1 | p0+p1 | p2+p3 | p2+p3
    return q + quad_perm<0x4E>(q);             // [2,3,0,1]
}

// columns of one LDS chunk: a multiple of 4 (the variates come in fours), <= 16 KiB of pairs
__host__ __device__ constexpr int inc_chunk(int dq)
{
    int c = (1024 / (4 * dq)) & ~3;
    return c < 4 ? 4 : (c > 64 ? 64 : c);
}

// ---------------------------------------------------------------- the step kernel
// 256 threads = 64 walkers of ONE group (group_size is a multiple of 64).
template <int DQ, bool NORMP, bool UNIT_T>
__global__ void __launch_bounds__(256) step_inc_kernel(const IncStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double2 sVU[];
    constexpr int COLB = 4 * DQ;                 // (v, u) pairs per column
    constexpr int C = inc_chunk(DQ);
    constexpr int PF = (C * COLB + 255) / 256;   // pairs a thread prefetches per chunk
    constexpr bool kKeepVU = DQ <= 12;           // keep the step's pairs for the commit
    const StepArgs& s = a.s;
    const int tid = threadIdx.x, c = tid & 3;
    const int W = s.W, d = a.d;
    const int w = blockIdx.x * 64 + (tid >> 2);
    const int g = w / s.group_size;
    const int ncols = s.n_steps;
    const double2* __restrict__ gVU = (const double2*)a.VU + (size_t)g * ncols * COLB;
    const int dpad = 4 * DQ;

    double x[DQ], y[DQ], lo[DQ], hi[DQ];
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
        const int i = 4 * kk + c;
        const bool in = i < d;
        x[kk] = in ? s.x[(size_t)i * W + w] : 0.0;
        y[kk] = in ? a.y[(size_t)i * W + w] : 0.0;
        lo[kk] = a.prior[i];               // padded: -inf / +inf beyond d
        hi[kk] = a.prior[dpad + i];
    }
    double lpost = s.logpost[w], lpri = s.logprior[w], llik = s.loglike[w];
    int wt = s.weight[w], prej = s.prior_rej[w], burn = s.burn_left[w];
    long long nacc = s.n_accept[w];
    const long long nacc0 = nacc;
    const uint32_t gid = s.walker0 + (uint32_t)w;

    {   // first chunk straight into buffer 0
        const int cnt = (ncols < C ? ncols : C) * COLB;
        for (int e = tid; e < cnt; e += 256) sVU[e] = gVU[e];
    }
    __syncthreads();

    for (int base = 0; base < ncols; base += C) {
        const int buf = (base / C) & 1;
        const double2* __restrict__ cur = sVU + buf * (C * COLB);
        // the next chunk travels to registers while this one is consumed
        double2 pf[PF];
        int nextcnt = ncols - base - C;
        nextcnt = (nextcnt < 0 ? 0 : (nextcnt > C ? C : nextcnt)) * COLB;
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int e = tid + 256 * p;
            if (e < nextcnt) pf[p] = gVU[(size_t)(base + C) * COLB + e];
        }
        const int cols = ncols - base < C ? ncols - base : C;
        for (int s4 = 0; s4 < cols; s4 += 4) {
            // lane class c draws the variates of step base + s4 + c
            StepRng rng;
            rng.begin(s.key0, s.key1, gid, s.step0 + (unsigned long long)(base + s4 + c));
            rng.run_all();
            const double r4 = rng.r, E4 = rng.Ea;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (s4 + k < cols) {
                    const double r = k == 0 ? quad_perm<0x00>(r4) : k == 1 ? quad_perm<0x55>(r4)
                                   : k == 2 ? quad_perm<0xAA>(r4) : quad_perm<0xFF>(r4);
                    const double Ea = k == 0 ? quad_perm<0x00>(E4) : k == 1 ? quad_perm<0x55>(E4)
                                    : k == 2 ? quad_perm<0xAA>(E4) : quad_perm<0xFF>(E4);
                    const double2* __restrict__ col = cur + (s4 + k) * COLB + c;
                    double2 vu[kKeepVU ? DQ : 1];
                    double pc = 0.0, sc = 0.0;
                    bool inb = true;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double2 p = col[4 * kk];
                        if (kKeepVU) vu[kk] = p;
                        const double t = fma(r, p.x, x[kk]);
                        inb = inb & (t <= hi[kk]) & (t >= lo[kk]);
                        const double yt = fma(r, p.y, y[kk]);
                        pc = fma(yt, yt, pc);
                        if (NORMP) {
                            const int i = 4 * kk + c;
                            const double scale = a.prior[3 * dpad + i];
                            if (scale < INFINITY) {
                                const double q = (t - a.prior[2 * dpad + i]) / scale;
                                sc = sc + fma(-0.5 * q, q, a.prior[4 * dpad + i]);
                            }
                        }
                    }
                    // a walker outside the prior support anywhere gets chi2 = +inf
                    const double chi2 = quad_sum(inb ? pc : INFINITY);
                    const bool inside = chi2 < INFINITY;
                    const double lp = s.uniform_logp + (NORMP ? quad_sum(sc) : 0.0);
                    const double ll = -0.5 * (s.cnorm0 + chi2);
                    const double lt = inside ? lp + ll : -INFINITY;
                    const double delta = UNIT_T ? (lpost - lt) : (lpost - lt) / s.temperature;
                    const bool accept = inside & (lt != -INFINITY) & ((lt > lpost) | (Ea > delta));
                    burn -= (accept & (burn > 0)) ? 1 : 0;
                    const double ra = accept ? r : 0.0;
#pragma unroll
                    for (int kk = 0; kk < DQ; ++kk) {
                        const double2 p = kKeepVU ? vu[kk] : col[4 * kk];
                        x[kk] = fma(ra, p.x, x[kk]);
                        y[kk] = fma(ra, p.y, y[kk]);
                    }
                    lpri = accept ? lp : lpri;
                    llik = accept ? ll : llik;
                    lpost = accept ? lt : lpost;
                    prej = accept ? 0 : (prej + (inside ? 0 : 1));
                    wt = accept ? 1 : wt + 1;
                    nacc += accept ? 1 : 0;
                    if (!accept && c == 0) {
                        const double max_now = s.max_tries * (burn > 0 ? 10.0 : 1.0);
                        if ((double)(wt - prej) > max_now) atomicCAS(s.stuck, 0, 1 + (int)gid);
                    }
                }
            }
        }
        double2* __restrict__ nxt = sVU + (buf ^ 1) * (C * COLB);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int e = tid + 256 * p;
            if (e < nextcnt) nxt[e] = pf[p];
        }
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
        s.n_accept[w] = nacc;
    }
    wave_add_accepts(s.accept_total, (c == 0) ? nacc - nacc0 : 0);
}

// ---------------------------------------------------------------- y = L^-1 (x - mu)
// One thread per walker, 64 walkers per workgroup; the deviations of the workgroup sit in LDS
// ([i][lane]) and the rows of L^-1 are read at wave-uniform addresses (scalar loads).  One
// ascending fma chain per row from +0.0 (orc_whiten).  Runs once per `refresh_every` steps.
__global__ void __launch_bounds__(64) whiten_state_kernel(const double* __restrict__ x,
                                                          double* __restrict__ y,
                                                          const double* __restrict__ mean,
                                                          const double* __restrict__ Lrow, int d,
                                                          int W)
{
    extern __shared__ __attribute__((aligned(16))) double sdev[];
    const int l = threadIdx.x, w = blockIdx.x * 64 + l;
    if (w < W)
        for (int i = 0; i < d; ++i) sdev[i * 64 + l] = x[(size_t)i * W + w] - mean[i];
    if (w >= W) return;
    for (int j = 0; j < d; ++j) {
        const double* __restrict__ row = Lrow + (size_t)j * d;
        double acc = 0.0;
        for (int i = 0; i <= j; ++i) acc = fma(row[i], sdev[i * 64 + l], acc);
        y[(size_t)j * W + w] = acc;
    }
}

// ---------------------------------------------------------------- (v, u = L^-1 v) per step
// One thread per (group, step of the launch): reads the step's direction column from the basis
// kernels' buffer V, forms u_j = sum_{i<=j} L^-1[j][i] v_i (ascending chain from +0.0,
// orc_whiten_directions) and writes the column in the step kernel's layout
// VU[g][step][kk][c] = (v_{4kk+c}, u_{4kk+c}), zero beyond d.
__global__ void __launch_bounds__(64) whiten_directions_kernel(const IncDirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sv[];   // [d][64]
    const int l = threadIdx.x;
    const int sr = blockIdx.x * 64 + l;          // step of the launch
    const int g = blockIdx.y;
    const int d = a.d;
    const bool live = sr < a.n_steps;
    if (live) {
        const unsigned long long step = a.step0 + (unsigned long long)sr;
        const int cyc = (int)(step / (unsigned long long)d - a.cycle0);
        const int col = (int)(step % (unsigned long long)d);
        const double* __restrict__ v = a.V + ((size_t)g * a.ncyc + cyc) * a.slab + (size_t)col * a.ld;
        for (int i = 0; i < d; ++i) sv[i * 64 + l] = v[i];
    }
    if (!live) return;
    double2* __restrict__ out = (double2*)a.VU + ((size_t)g * a.n_steps + sr) * (4 * a.dq);
    for (int j = 0; j < d; ++j) {
        const double* __restrict__ row = a.Lrow + (size_t)j * d;
        double acc = 0.0;
        for (int i = 0; i <= j; ++i) acc = fma(row[i], sv[i * 64 + l], acc);
        out[j] = make_double2(sv[j * 64 + l], acc);
    }
    for (int j = d; j < 4 * a.dq; ++j) out[j] = make_double2(0.0, 0.0);
}

template <int DQ>
hipError_t launch_inc_dq(const IncStepArgs& a, hipStream_t st)
{
    constexpr int C = inc_chunk(DQ);
    const size_t lds = sizeof(double2) * 2 * C * 4 * DQ;
    const bool unit_t = a.s.temperature == 1.0;
    typedef void (*kern_t)(const IncStepArgs);
    const kern_t kern = a.has_norm ? (unit_t ? step_inc_kernel<DQ, true, true> : step_inc_kernel<DQ, true, false>)
                                   : (unit_t ? step_inc_kernel<DQ, false, true> : step_inc_kernel<DQ, false, false>);
    static const std::string names[4] = {
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", false, false>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", false, true>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", true, false>",
        "mcmc::step_inc_kernel<" + std::to_string(DQ) + ", true, true>"};
    mcmc_hip_note_step_kernel(names[(a.has_norm ? 2 : 0) + (unit_t ? 1 : 0)].c_str());
    hipLaunchKernelGGL(kern, dim3(a.s.W / 64), dim3(256), lds, st, a);
    return hipGetLastError();
}

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

}  // namespace
}  // namespace mcmc

#define MCMC_CAT2(a, b) a##b
#define MCMC_CAT(a, b) MCMC_CAT2(a, b)
// one translation unit per range of DQ = ceil(d / 4) (build.py: -DMCMC_DQ_LO=.. -DMCMC_DQ_HI=..)
extern "C" hipError_t MCMC_CAT(mcmc_hip_launch_inc_step_, MCMC_DQ_LO)(const mcmc::IncStepArgs* a,
                                                                    hipStream_t st)
{
    if (a->dq < MCMC_DQ_LO || a->dq > MCMC_DQ_HI) return hipErrorInvalidValue;
    return mcmc::dispatch_inc<MCMC_DQ_LO>(*a, st);
}

#if MCMC_DQ_LO == 1
extern "C" hipError_t mcmc_hip_launch_whiten_state(const double* x, double* y, const double* mean,
                                                   const double* Lrow, int d, int W, hipStream_t st)
{
    const size_t lds = sizeof(double) * 64 * (size_t)d;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_state_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mcmc::whiten_state_kernel, dim3((W + 63) / 64), dim3(64), lds, st, x, y,
                       mean, Lrow, d, W);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_whiten_directions(const mcmc::IncDirArgs* a, int n_groups,
                                                        hipStream_t st)
{
    const size_t lds = sizeof(double) * 64 * (size_t)a->d;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mcmc::whiten_directions_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mcmc::whiten_directions_kernel, dim3((a->n_steps + 63) / 64, n_groups),
                       dim3(64), lds, st, *a);
    return hipGetLastError();
}
#endif
