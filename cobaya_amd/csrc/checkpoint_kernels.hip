// The learn / convergence checkpoint of MCMC.check_convergence_and_learn_proposal
// (cobaya/samplers/mcmc/mcmc.py:773-1032) on the device (gfx950): the window sums over the
// checkpoint intervals, the sufficient statistics the all-reduce carries, the R-1 of
// the means (mcmc.py:856-889, functions.py:81-89) and the refresh of the proposal transform
// (proposal.py:226-260, tools.py:761-788) -- written in place, so that a refreshed proposal needs
// no host round trip.  Three launches:
//   ckpt_window_kernel   interval accumulators -> ring slot, zeroed; sums over the window's slots
//   ckpt_payload_kernel  group means, sum_m, sum_mm, sum_N cov: the buffer the all-reduce carries
//   ckpt_solve_kernel    W, B, R-1 = max eig(L^-1 Bhat L^-T); T = scale diag(std) chol(corr(W))
// The dense linear algebra (d <= 128) is one workgroup: Cholesky, triangular inverse and the two
// products in the operation order of the host routines in capi.hip (cholesky_lower,
// tri_inverse_lower, mcmc_hip_gelman_rubin, mcmc_hip_set_proposal_cov) -- so the transform the
// device writes is bit for bit the one the host would compute from the same statistics --, the
// largest eigenvalue by Householder tridiagonalisation + Sturm bisection (the host uses QL: R-1
// agrees to rounding, it only feeds thresholds and logs).
#include <type_traits>

#include "checkpoint_args.h"

namespace mcmc {
namespace {

// ---------------------------------------------------------------- window sums
__global__ void __launch_bounds__(256) ckpt_window_kernel(const CkptWindowArgs a)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_elem) return;
    // this interval -> its ring slot; the accumulators start the next interval at zero
    a.ring[(size_t)a.slot * a.n_elem + e] = a.acc[e];
    a.acc[e] = 0.0;
    // window = the slots first .. first + n - 1 (mod cap), summed in that (ascending) order from
    // +0: the order of the host's `sum(iv[1] for iv in intervals)`
    double s = 0.0;
    for (int k = 0; k < a.n_slots; ++k) {
        const int sl = (a.first + k) % a.cap;
        s = s + a.ring[(size_t)sl * a.n_elem + e];
    }
    a.wsum[e] = s;
    // the chain (= group) means of the window: one division per element, once
    if (e < a.n_mean) a.means[e] = s / a.n_per_chain;
}

// ---------------------------------------------------------------- payload of the all-reduce
// per rank: [G, N_c G, accepted since the last checkpoint, steps since x W, accepted |
//            sum_g N_c cov_g = S - N_c sum_mm (d x d) | sum_g m_g (d) | sum_g m_g m_g^T (d x d)]
// (mcmc.py:791-793 gathers N, mean, cov per chain; here a chain is a group of walkers)
// The chain means are staged through LDS a tile of groups at a time (read straight from L2 the
// dependent chain over the groups waits for memory at every batch: 41 us at G = 256, d = 30); the
// terms are still added group by group in ascending order.
constexpr int kPayloadTileDoubles = 6144;     // 48 KB of LDS

__global__ void __launch_bounds__(256) ckpt_payload_kernel(const CkptPayloadArgs a)
{
    __shared__ double tile[kPayloadTileDoubles];
    const int d = a.d, G = a.G;
    const double Nc = a.n_per_chain;
    const double* __restrict__ ms = a.means;             // [G][d] chain means of the window
    const double* __restrict__ S = a.wsum + (size_t)G * d;   // lower triangle i(i+1)/2 + j
    double* __restrict__ P = a.payload;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if (tid == 0) {
        const unsigned long long acc = *a.accept_total;
        P[0] = (double)G;
        P[1] = Nc * (double)G;
        P[2] = (double)(acc - *a.accept_prev);
        P[3] = a.steps_since * (double)a.W;
        P[4] = (double)acc;
        *a.accept_prev = acc;
    }
    const int npair = d * (d + 1) / 2;
    int i = 0, j = 0;
    if (tid < npair) {           // (i, j), i >= j
        i = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > tid) --i;
        while ((i + 1) * (i + 2) / 2 <= tid) ++i;
        j = tid - i * (i + 1) / 2;
    } else if (tid < npair + d) {
        i = tid - npair;
    }
    double mm = 0.0, sm = 0.0;
    const int gt = kPayloadTileDoubles / d;               // groups per tile
    for (int g0 = 0; g0 < G; g0 += gt) {
        const int ng = G - g0 < gt ? G - g0 : gt;
        __syncthreads();
        for (int e = threadIdx.x; e < ng * d; e += 256) tile[e] = ms[(size_t)g0 * d + e];
        __syncthreads();
        if (tid < npair) {
            for (int g = 0; g < ng; ++g) mm = fma(tile[g * d + i], tile[g * d + j], mm);
        } else if (tid < npair + d) {
            for (int g = 0; g < ng; ++g) sm = sm + tile[g * d + i];
        }
    }
    if (tid < npair) {
        const double ncov = S[tid] - Nc * mm;
        P[5 + i * d + j] = P[5 + j * d + i] = ncov;
        P[5 + d * d + d + i * d + j] = P[5 + d * d + d + j * d + i] = mm;
    } else if (tid < npair + d) {
        P[5 + d * d + i] = sm;
    }
}

// ---------------------------------------------------------------- dense linear algebra, one workgroup
// lower Cholesky, row-major (capi.hip cholesky_lower: the same operations in the same order);
// returns false if not positive definite.  One thread per row, two barriers per column.
// (PT: `double*` in global memory, or an LDS pointer -- ds_read / ds_write, which the compiler
// pipelines freely; through a generic pointer every access is a flat instruction it serialises)
template <typename PT>
__device__ bool wg_cholesky(int n, PT A, PT L, int* flag)
{
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += blockDim.x) L[e] = 0.0;
    if (t == 0) *flag = 1;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        double v = 0.0;
        if (t >= j && t < n) {
            v = A[t * n + j];
            int k = 0;
            for (; k + 8 <= j; k += 8) {      // (loads of eight terms in flight; the chain in order)
                double a8[8], b8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { a8[q] = L[t * n + k + q]; b8[q] = L[j * n + k + q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) v -= a8[q] * b8[q];
            }
            for (; k < j; ++k) v -= L[t * n + k] * L[j * n + k];
        }
        if (t == j) {
            if (!(v > 0.0) || !isfinite(v)) *flag = 0;
            L[j * n + j] = sqrt(v);
        }
        __syncthreads();
        if (*flag == 0) return false;
        if (t > j && t < n) L[t * n + j] = v / L[j * n + j];
        __syncthreads();
    }
    return true;
}

// inverse of a lower-triangular matrix (capi.hip tri_inverse_lower): one thread per column
template <typename PT>
__device__ void wg_tri_inverse(int n, PT L, PT Li)
{
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += blockDim.x) Li[e] = 0.0;
    __syncthreads();
    if (t < n) {
        const int j = t;
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            int k = j;
            for (; k + 8 <= i; k += 8) {
                double a8[8], b8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { a8[q] = L[i * n + k + q]; b8[q] = Li[(k + q) * n + j]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) s += a8[q] * b8[q];
            }
            for (; k < i; ++k) s += L[i * n + k] * Li[k * n + j];
            Li[i * n + j] = -s / L[i * n + i];
        }
    }
    __syncthreads();
}

// sum over the workgroup (wave shuffles, then the four wave sums through LDS)
__device__ double wg_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int t = threadIdx.x;
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// largest eigenvalue of the symmetric n x n matrix A (destroyed): Householder reduction to
// tridiagonal form (full storage, rank-2 updates), then bisection on the Sturm count
template <typename PT>
__device__ double wg_lambda_max(int n, PT A, PT dg, PT eg, PT v, PT p, double* red)
{
    const int t = threadIdx.x, nt = blockDim.x;
    for (int k = 0; k + 2 < n; ++k) {
        const int m = n - k - 1;                 // x = A[k+1.., k]
        double part = 0.0;
        for (int i = t; i < m; i += nt) { const double x = A[(k + 1 + i) * n + k]; part += x * x; }
        const double nx2 = wg_sum(part, red);
        const double x0 = A[(k + 1) * n + k];
        const double alpha = x0 > 0.0 ? -sqrt(nx2) : sqrt(nx2);
        const double vnorm2 = nx2 - 2.0 * alpha * x0 + alpha * alpha;   // |x - alpha e1|^2
        if (t == 0) { dg[k] = A[k * n + k]; eg[k] = alpha; }
        if (!(vnorm2 > 0.0) || nx2 - x0 * x0 == 0.0) {   // column already in tridiagonal form
            if (t == 0) eg[k] = x0;
            __syncthreads();
            continue;
        }
        const double beta = 2.0 / vnorm2;
        for (int i = t; i < m; i += nt) v[i] = A[(k + 1 + i) * n + k] - (i == 0 ? alpha : 0.0);
        __syncthreads();
        double kp = 0.0;
        for (int i = t; i < m; i += nt) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += A[(k + 1 + i) * n + (k + 1 + j)] * v[j];
            p[i] = beta * s;
            kp += v[i] * p[i];
        }
        const double K = 0.5 * beta * wg_sum(kp, red);
        for (int i = t; i < m; i += nt) p[i] -= K * v[i];       // w
        __syncthreads();
        for (int e = t; e < m * m; e += nt) {
            const int i = e / m, j = e - i * m;
            A[(k + 1 + i) * n + (k + 1 + j)] -= v[i] * p[j] + p[i] * v[j];
        }
        __syncthreads();
    }
    if (t == 0) {
        if (n >= 2) { dg[n - 2] = A[(n - 2) * n + (n - 2)]; eg[n - 2] = A[(n - 1) * n + (n - 2)]; }
        dg[n - 1] = A[(n - 1) * n + (n - 1)];
        // Gershgorin bounds of the tridiagonal matrix
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) {
            const double r = (i > 0 ? fabs(eg[i - 1]) : 0.0) + (i + 1 < n ? fabs(eg[i]) : 0.0);
            lo = fmin(lo, dg[i] - r);
            hi = fmax(hi, dg[i] + r);
        }
        red[8] = lo; red[9] = hi;
    }
    __syncthreads();
    // multisection on the Sturm count (count(x) = eigenvalues below x): every thread takes one of
    // 256 points of the bracket, the bracket shrinks 257-fold per round -- eight rounds reach
    // the last bit from any Gershgorin width
    double lo = red[8], hi = red[9];
    __syncthreads();
    for (int round = 0; round < 8 && hi > lo; ++round) {
        const double x = lo + (hi - lo) * ((double)(t + 1) / (double)(nt + 1));
        int cnt = 0;
        double q = dg[0] - x;
        if (q < 0.0) ++cnt;
        for (int i = 1; i < n; ++i) {
            const double den = fabs(q) < 1e-300 ? (q < 0.0 ? -1e-300 : 1e-300) : q;
            // (hardware reciprocal + one Newton step instead of an IEEE division: the COUNT only
            // depends on the sign of q, which a relative error of 1e-15 flips only within that
            // distance of an eigenvalue -- below the resolution of the statistic)
            double r = __builtin_amdgcn_rcp(den);
            r = fma(fma(-den, r, 1.0), r, r);
            q = dg[i] - x - eg[i - 1] * eg[i - 1] * r;
            if (q < 0.0) ++cnt;
        }
        // the first point with all n eigenvalues below it bounds the largest from above
        const unsigned long long m = __builtin_amdgcn_ballot_w64(cnt >= n);
        int* first = (int*)(red + 16);
        if ((t & 63) == 0) first[t >> 6] = m ? (t & ~63) + __builtin_ctzll(m) : nt;
        __syncthreads();
        int f = nt;
        for (int w = 0; w < (nt + 63) / 64; ++w) f = first[w] < f ? first[w] : f;
        const double nlo = f == 0 ? lo : lo + (hi - lo) * ((double)f / (double)(nt + 1));
        const double nhi = f == nt ? hi : lo + (hi - lo) * ((double)(f + 1) / (double)(nt + 1));
        __syncthreads();
        lo = nlo; hi = nhi;
    }
    if (t == 0) red[0] = 0.5 * (lo + hi);
    __syncthreads();
    const double lam = red[0];
    __syncthreads();
    return lam;
}

// out: [0] R-1 of the group means (raw), [1] status (0 ok, 1 B not positive, 2 W not PD,
// 3 eigenvalues not finite), [2] 1 if the proposal was refreshed, [3] n_chains, [4] sum_N,
// [5] accepted since the last checkpoint, [6] steps x walkers since, [7] accepted so far,
// [8 ..] mean_of_covs (d x d), then the transform T (d x d) that is in force after this kernel
typedef double __attribute__((address_space(3))) * lds_dp;

template <bool IN_LDS>
__global__ void __launch_bounds__(256) ckpt_solve_kernel(const CkptSolveArgs a)
{
    typedef typename std::conditional<IN_LDS, lds_dp, double*>::type PT;
    __shared__ double red[32];
    __shared__ int flag;
    const int n = a.d, t = threadIdx.x, nt = blockDim.x, nn = n * n;
    const double* __restrict__ P = a.payload;
    double* __restrict__ out = a.out;
    // the workspace: LDS where 7 n^2 + 5 n doubles fit (n <= 50), else global memory (L2)
    extern __shared__ __attribute__((aligned(16))) double ck_lds[];
    PT Wm;                           // mean of covs
    if constexpr (IN_LDS) Wm = (lds_dp)ck_lds; else Wm = a.ws;
    PT cB = Wm + nn, nW = cB + nn, L = nW + nn, Li = L + nn, tmp = Li + nn, M = tmp + nn;
    PT sd = M + nn, dg = sd + n, eg = dg + n, hv = eg + n, hp = hv + n;
    const double n_chains = P[0], sum_N = P[1];
    const double* __restrict__ sum_Ncov = P + 5;
    const double* __restrict__ sum_mean = P + 5 + nn;
    const double* __restrict__ sum_mm = P + 5 + nn + n;
    if (t < 8) out[t] = t == 3 ? n_chains : t == 4 ? sum_N : t >= 5 ? P[t - 3] : 0.0;
    __syncthreads();
    // W = sum N cov / sum N (mcmc.py:856); B = cov of the chain means, ddof 1 (860)
    for (int e = t; e < nn; e += nt) {
        const int i = e / n, j = e - i * n;
        const double w = sum_Ncov[e] / sum_N;
        Wm[e] = w;
        out[8 + e] = w;
        tmp[e] = (sum_mm[e] - sum_mean[i] * sum_mean[j] / n_chains) / (n_chains - 1.0);   // B
    }
    __syncthreads();
    int status = 0;
    if (t == 0) {
        flag = 1;
        for (int i = 0; i < n; ++i)
            if (!(tmp[i * n + i] > 0.0)) flag = 0;
    }
    __syncthreads();
    if (!flag) status = 1;
    double rm1 = NAN;
    if (status == 0) {
        if (t < n) sd[t] = sqrt(tmp[t * n + t]);
        __syncthreads();
        for (int e = t; e < nn; e += nt) {
            const int i = e / n, j = e - i * n;
            cB[e] = tmp[e] / sd[i] / sd[j];       // mcmc.py:865
            nW[e] = Wm[e] / sd[i] / sd[j];        // mcmc.py:866
        }
        __syncthreads();
        if (!wg_cholesky(n, nW, L, &flag)) status = 2;     // mcmc.py:871
    }
    if (status == 0) {
        wg_tri_inverse(n, L, Li);
        for (int e = t; e < nn; e += nt) {
            const int i = e / n, j = e - i * n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Li[i * n + k] * cB[k * n + j];
            tmp[e] = s;
        }
        __syncthreads();
        for (int e = t; e < nn; e += nt) {
            const int i = e / n, j = e - i * n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += tmp[i * n + k] * Li[j * n + k];
            M[e] = s;
        }
        __syncthreads();
        for (int e = t; e < nn; e += nt) {       // symmetrise (into cB: no longer needed)
            const int i = e / n, j = e - i * n;
            cB[e] = i == j ? M[e] : 0.5 * (M[i * n + j] + M[j * n + i]);
        }
        __syncthreads();
        // the statistic is max |eig| (mcmc.py:889); M is positive semi-definite (a congruence of
        // B), so that is its largest eigenvalue
        rm1 = n == 1 ? cB[0] : wg_lambda_max(n, cB, dg, eg, hv, hp, red);
        if (!isfinite(rm1)) status = 3;
    }
    // ---- proposal refresh (mcmc.py:1009-1023) when R-1 per walker is inside the learning window
    const double rq = rm1 * a.group_size;
    bool learn = status == 0 && rq >= a.learn_lo && rq <= a.learn_hi;
    if (learn) {
        // BlockedProposer.set_covariance (proposal.py:226-260) of `W` (already tempered): reorder by
        // i_of_j, std, corr with unit diagonal (tools.py:779-788), Cholesky, T = scale diag(std) L
        PT cov = nW;    // sorted covariance
        PT corr = M;
        for (int e = t; e < nn; e += nt) {
            const int i = e / n, j = e - i * n;
            cov[e] = a.i_of_j ? Wm[a.i_of_j[i] * n + a.i_of_j[j]] : Wm[e];
        }
        __syncthreads();
        if (t == 0) {
            flag = 1;
            for (int i = 0; i < n; ++i)
                if (!(cov[i * n + i] > 0.0) || !isfinite(cov[i * n + i])) flag = 0;
        }
        __syncthreads();
        learn = flag != 0;
        if (learn) {
            if (t < n) sd[t] = sqrt(cov[t * n + t]);
            __syncthreads();
            for (int e = t; e < nn; e += nt) {
                const int i = e / n, j = e - i * n;
                corr[e] = i == j ? 1.0 : (1.0 / sd[i]) * cov[e] * (1.0 / sd[j]);
            }
            __syncthreads();
            learn = wg_cholesky(n, corr, L, &flag);
        }
        if (learn) {
            for (int e = t; e < nn; e += nt) {
                const int i = e / n, j = e - i * n;
                a.T[e] = j <= i ? a.proposal_scale * (sd[i] * L[e]) : 0.0;
            }
        }
    }
    __syncthreads();
    for (int e = t; e < nn; e += nt) out[8 + nn + e] = a.T[e];
    if (t == 0) {
        out[0] = rm1;
        out[1] = (double)status;
        out[2] = learn ? 1.0 : 0.0;
    }
}

}  // namespace
}  // namespace mcmc

using namespace mcmc;

namespace {
// ---------------------------------------------------------------- R-1 of the bounds
// mcmc.py:918-1002: per chain the lower / upper confidence bound of every parameter
// (GetDist `MCSamples.confidence(i, limfrac, upper)` = the sample at which the cumulative weight
// first reaches limfrac * norm, resp. (1 - limfrac) * norm; oracle/ref_numpy.py `confidence`),
// then std over chains / sigma.  Here a chain is a walker group and its samples are the
// ensemble snapshots of the window (weight 1 each), so a bound is an ORDER STATISTIC of
// n = n_slots * gs values: selected exactly, without sorting, by fixing the bits of its
// order-preserving integer key from the top -- 64 counting passes over keys held in LDS.
// One workgroup per (chain, parameter); both bounds in the same passes.
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_value(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ void __launch_bounds__(256) ckpt_bounds_kernel(const CkptBoundsArgs a)
{
    extern __shared__ unsigned long long keys[];      // [n]
    __shared__ int cnt[2][4];
    const int g = blockIdx.x, i = blockIdx.y, tid = threadIdx.x;
    const int n = a.n_slots * a.gs;
    for (int e = tid; e < n; e += 256) {
        const int s = e / a.gs, w = e - s * a.gs;
        keys[e] = order_key(a.ring[((size_t)a.slots[s] * a.d + i) * a.W + (size_t)g * a.gs + w]);
    }
    __syncthreads();
    unsigned long long p_lo = 0, p_hi = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long c_lo = p_lo | (1ull << bit), c_hi = p_hi | (1ull << bit);
        int n_lo = 0, n_hi = 0;
        for (int e = tid; e < n; e += 256) {
            const unsigned long long k = keys[e];
            n_lo += k < c_lo;
            n_hi += k < c_hi;
        }
        for (int o = 32; o > 0; o >>= 1) {
            n_lo += __shfl_down(n_lo, o);
            n_hi += __shfl_down(n_hi, o);
        }
        if ((tid & 63) == 0) {
            cnt[0][tid >> 6] = n_lo;
            cnt[1][tid >> 6] = n_hi;
        }
        __syncthreads();
        n_lo = cnt[0][0] + cnt[0][1] + cnt[0][2] + cnt[0][3];
        n_hi = cnt[1][0] + cnt[1][1] + cnt[1][2] + cnt[1][3];
        __syncthreads();
        // the k-th smallest key (0-based) is >= c  iff  fewer than k + 1 keys are below c
        if (n_lo <= a.k_lo) p_lo = c_lo;
        if (n_hi <= a.k_hi) p_hi = c_hi;
    }
    if (tid == 0) {
        double* b = a.bounds + ((size_t)g * a.d + i) * 2;
        b[0] = key_value(p_lo);
        b[1] = key_value(p_hi);
    }
}

// sums over this rank's chains, ascending, from +0 (what the all-reduce carries; the statistic
// is formed from the reduced sums: std over ALL chains, mcmc.py:977)
__global__ void __launch_bounds__(128) ckpt_bounds_reduce_kernel(const CkptBoundsReduceArgs a)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i == 0) a.payload[0] = (double)a.G;
    if (i >= a.d) return;
    const double sh = a.shift[i];
    double s_lo = 0.0, s_hi = 0.0, q_lo = 0.0, q_hi = 0.0;
    for (int g = 0; g < a.G; ++g) {
        const double lo = a.bounds[((size_t)g * a.d + i) * 2] - sh;
        const double hi = a.bounds[((size_t)g * a.d + i) * 2 + 1] - sh;
        s_lo = s_lo + lo;
        s_hi = s_hi + hi;
        q_lo = q_lo + lo * lo;
        q_hi = q_hi + hi * hi;
    }
    a.payload[1 + i] = s_lo;
    a.payload[1 + a.d + i] = s_hi;
    a.payload[1 + 2 * a.d + i] = q_lo;
    a.payload[1 + 3 * a.d + i] = q_hi;
}
}  // namespace

extern "C" hipError_t mcmc_hip_launch_ckpt_window(const CkptWindowArgs* a, hipStream_t st)
{
    hipLaunchKernelGGL(ckpt_window_kernel, dim3((unsigned)((a->n_elem + 255) / 256)), dim3(256), 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_ckpt_payload(const CkptPayloadArgs* a, hipStream_t st)
{
    const int n = a->d * (a->d + 1) / 2 + a->d;
    hipLaunchKernelGGL(ckpt_payload_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_ckpt_solve(const CkptSolveArgs* a, hipStream_t st)
{
    const size_t need = sizeof(double) * (7 * (size_t)a->d * a->d + 5 * (size_t)a->d);
    if (need <= 150 * 1024) {
        if (need > 40 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)ckpt_solve_kernel<true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(ckpt_solve_kernel<true>, dim3(1), dim3(256), need, st, *a);
    } else {
        hipLaunchKernelGGL(ckpt_solve_kernel<false>, dim3(1), dim3(256), 0, st, *a);
    }
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_ckpt_bounds(const mcmc::CkptBoundsArgs* a, int G, hipStream_t st)
{
    using namespace mcmc;
    const size_t lds = sizeof(unsigned long long) * (size_t)a->n_slots * a->gs;
    // the attribute belongs to the CURRENT device's copy of the kernel: set it before every
    // launch that needs it (cheap, and right for engines on several devices and threads)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ckpt_bounds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kBoundsLdsBytes);
        if (e != hipSuccess) return e;
    }
    ckpt_bounds_kernel<<<dim3(G, a->d), 256, lds, st>>>(*a);
    return hipGetLastError();
}

extern "C" hipError_t mcmc_hip_launch_ckpt_bounds_reduce(const mcmc::CkptBoundsReduceArgs* a, hipStream_t st)
{
    ckpt_bounds_reduce_kernel<<<(a->d + 127) / 128, 128, 0, st>>>(*a);
    return hipGetLastError();
}
