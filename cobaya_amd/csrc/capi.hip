// libmcmc_hip.so: engine context, host-side small dense linear algebra and the C ABI
// declared in include/mcmc_hip.h.  gfx950 only; no CPU fallback.
#include "../../include/mcmc_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "kernels.h"
#include "pliklite_args.h"
#include "checkpoint_args.h"
#include "comm.h"

MCMC_DECLARE_DIM(1) MCMC_DECLARE_DIM(2) MCMC_DECLARE_DIM(3) MCMC_DECLARE_DIM(4)
MCMC_DECLARE_DIM(5) MCMC_DECLARE_DIM(6) MCMC_DECLARE_DIM(7) MCMC_DECLARE_DIM(8)
MCMC_DECLARE_DIM(9) MCMC_DECLARE_DIM(10) MCMC_DECLARE_DIM(11) MCMC_DECLARE_DIM(12)
MCMC_DECLARE_DIM(13) MCMC_DECLARE_DIM(14) MCMC_DECLARE_DIM(15) MCMC_DECLARE_DIM(16)
MCMC_DECLARE_DIM(17) MCMC_DECLARE_DIM(18) MCMC_DECLARE_DIM(19) MCMC_DECLARE_DIM(20)
MCMC_DECLARE_DIM(21) MCMC_DECLARE_DIM(22) MCMC_DECLARE_DIM(23) MCMC_DECLARE_DIM(24)
MCMC_DECLARE_DIM(25) MCMC_DECLARE_DIM(26) MCMC_DECLARE_DIM(27) MCMC_DECLARE_DIM(28)
MCMC_DECLARE_DIM(29) MCMC_DECLARE_DIM(30) MCMC_DECLARE_DIM(31) MCMC_DECLARE_DIM(32)

MCMC_DECLARE_BIG(48) MCMC_DECLARE_BIG(56) MCMC_DECLARE_BIG(64) MCMC_DECLARE_BIG(72)
MCMC_DECLARE_BIG(80) MCMC_DECLARE_BIG(88) MCMC_DECLARE_BIG(96) MCMC_DECLARE_BIG(100)
MCMC_DECLARE_BIG(112) MCMC_DECLARE_BIG(120) MCMC_DECLARE_BIG(128)

MCMC_DECLARE_PAIR(33) MCMC_DECLARE_PAIR(34) MCMC_DECLARE_PAIR(35) MCMC_DECLARE_PAIR(36)
MCMC_DECLARE_PAIR(37) MCMC_DECLARE_PAIR(38) MCMC_DECLARE_PAIR(39) MCMC_DECLARE_PAIR(40)
MCMC_DECLARE_PAIR(41) MCMC_DECLARE_PAIR(42) MCMC_DECLARE_PAIR(43) MCMC_DECLARE_PAIR(44)
MCMC_DECLARE_PAIR(45) MCMC_DECLARE_PAIR(46) MCMC_DECLARE_PAIR(47) MCMC_DECLARE_PAIR(48)
MCMC_DECLARE_PAIR(49) MCMC_DECLARE_PAIR(50) MCMC_DECLARE_PAIR(51) MCMC_DECLARE_PAIR(52)
MCMC_DECLARE_PAIR(53) MCMC_DECLARE_PAIR(54) MCMC_DECLARE_PAIR(55) MCMC_DECLARE_PAIR(56)

namespace {

using mcmc::BigKernels;
using mcmc::ConstLayout;
using mcmc::DimKernels;

constexpr int kMaxDimBig = 128;  // basis_big_kernel keeps H (d*d doubles) in 160 KiB of LDS

// smallest compiled padded size that serves dimension d (32 < d <= 128)
const BigKernels* big_for_dim(int d)
{
    typedef const BigKernels* (*getter)();
    static const getter table[] = {mcmc_hip_big_48,  mcmc_hip_big_56,  mcmc_hip_big_64,
                                   mcmc_hip_big_72,  mcmc_hip_big_80,  mcmc_hip_big_88,
                                   mcmc_hip_big_96,  mcmc_hip_big_100, mcmc_hip_big_112,
                                   mcmc_hip_big_120, mcmc_hip_big_128};
    if (d <= mcmc::kMaxDimLane || d > kMaxDimBig) return nullptr;
    for (getter g : table)
        if (g != nullptr && g()->dp >= d) return g();
    return nullptr;
}

// the two-wave step kernel of a dimension 32 < d <= kMaxDimPair, if compiled
const mcmc::PairKernels* pair_for_dim(int d)
{
    typedef const mcmc::PairKernels* (*getter)();
    static const getter table[] = {mcmc_hip_pair_33, mcmc_hip_pair_34, mcmc_hip_pair_35,
                                   mcmc_hip_pair_36, mcmc_hip_pair_37, mcmc_hip_pair_38,
                                   mcmc_hip_pair_39, mcmc_hip_pair_40, mcmc_hip_pair_41,
                                   mcmc_hip_pair_42, mcmc_hip_pair_43, mcmc_hip_pair_44,
                                   mcmc_hip_pair_45, mcmc_hip_pair_46, mcmc_hip_pair_47,
                                   mcmc_hip_pair_48, mcmc_hip_pair_49, mcmc_hip_pair_50,
                                   mcmc_hip_pair_51, mcmc_hip_pair_52, mcmc_hip_pair_53,
                                   mcmc_hip_pair_54, mcmc_hip_pair_55, mcmc_hip_pair_56};
    static_assert(sizeof(table) / sizeof(table[0]) == mcmc::kMaxDimPair - mcmc::kMaxDimLane, "");
    if (d <= mcmc::kMaxDimLane || d > mcmc::kMaxDimPair) return nullptr;
    const getter g = table[d - mcmc::kMaxDimLane - 1];
    return g != nullptr ? g() : nullptr;
}

const DimKernels* kernels_for_dim(int d)
{
    typedef const DimKernels* (*getter)();
    static const getter table[33] = {
        nullptr,          mcmc_hip_dim_1,  mcmc_hip_dim_2,  mcmc_hip_dim_3,  mcmc_hip_dim_4,
        mcmc_hip_dim_5,   mcmc_hip_dim_6,  mcmc_hip_dim_7,  mcmc_hip_dim_8,  mcmc_hip_dim_9,
        mcmc_hip_dim_10,  mcmc_hip_dim_11, mcmc_hip_dim_12, mcmc_hip_dim_13, mcmc_hip_dim_14,
        mcmc_hip_dim_15,  mcmc_hip_dim_16, mcmc_hip_dim_17, mcmc_hip_dim_18, mcmc_hip_dim_19,
        mcmc_hip_dim_20,  mcmc_hip_dim_21, mcmc_hip_dim_22, mcmc_hip_dim_23, mcmc_hip_dim_24,
        mcmc_hip_dim_25,  mcmc_hip_dim_26, mcmc_hip_dim_27, mcmc_hip_dim_28, mcmc_hip_dim_29,
        mcmc_hip_dim_30,  mcmc_hip_dim_31, mcmc_hip_dim_32};
    if (d < 1 || d > 32 || table[d] == nullptr) return nullptr;
    return table[d]();
}

std::string g_create_error;

// slots per cycle of the blocked proposer's three sequences (oracle: orc_block_slots)
int block_slots(const mcmc_hip_ctx* h, int which);

// ------------------------------------------------------------------ small dense LA (host)
// lower Cholesky, row-major; false if not positive definite (np.linalg.cholesky semantics)
bool cholesky_lower(int n, const double* A, double* L)
{
    std::fill(L, L + (size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double ljj = std::sqrt(s);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / ljj;
        }
    }
    return true;
}

// inverse of a lower-triangular matrix (LAPACK dtrtri semantics, functions.py:81-89)
void tri_inverse_lower(int n, const double* L, double* Li)
{
    std::fill(Li, Li + (size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += L[i * n + k] * Li[k * n + j];
            Li[i * n + j] = -s / L[i * n + i];
        }
    }
}

// eigenvalues of a symmetric matrix (np.linalg.eigvalsh, mcmc.py:881): Householder reduction
// to tridiagonal form followed by the implicit-shift QL iteration (the classic EISPACK
// tred1 / tql1 pair, eigenvalues only).  A is destroyed; returns false if QL fails to converge.
bool symmetric_eigenvalues(int n, double* A, double* ev)
{
    std::vector<double> e(n, 0.0);
    double* d = ev;
    for (int i = n - 1; i > 0; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += std::fabs(A[i * n + k]);
            if (scale == 0.0) {
                e[i] = A[i * n + l];
            } else {
                for (int k = 0; k <= l; ++k) {
                    A[i * n + k] /= scale;
                    h += A[i * n + k] * A[i * n + k];
                }
                double f = A[i * n + l];
                const double g = (f >= 0.0) ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A[i * n + l] = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    double gg = 0.0;
                    for (int k = 0; k <= j; ++k) gg += A[j * n + k] * A[i * n + k];
                    for (int k = j + 1; k <= l; ++k) gg += A[k * n + j] * A[i * n + k];
                    e[j] = gg / h;
                    f += e[j] * A[i * n + j];
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = A[i * n + j];
                    const double gg = e[j] - hh * f;
                    e[j] = gg;
                    for (int k = 0; k <= j; ++k) A[j * n + k] -= f * e[k] + gg * A[i * n + k];
                }
            }
        } else {
            e[i] = A[i * n + l];
        }
        d[i] = h;
    }
    for (int i = 0; i < n; ++i) d[i] = A[i * n + i];
    // QL with implicit shifts on (d, e)
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= std::numeric_limits<double>::epsilon() * dd) break;
            }
            if (m != l) {
                if (++iter > 60) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

// np.allclose(A.T, A) (rtol 1e-5, atol 1e-8), proposal.py:243
bool is_symmetric(int n, const double* A)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double a = A[j * n + i], b = A[i * n + j];
            if (!(std::fabs(a - b) <= 1e-8 + 1e-5 * std::fabs(b))) return false;
        }
    return true;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t resize(size_t count)
    {
        if (count <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void**)&p, sizeof(T) * count);
        if (e == hipSuccess) n = count;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

extern "C" hipError_t mcmc_hip_launch_general_step(const mcmc::GeneralStepArgs* b, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_general_drag(const mcmc::GeneralDragArgs* g, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pack_rows(const double* rows, const int* n_rows,
                                                const long long* offset, double* out, int W,
                                                int cap, int d, uint32_t walker0, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_blocked_basis(const mcmc::BlockedBasisArgs* a, int n_groups,
                                                    hipStream_t st);

// incremental_kernels.hip: one translation unit per range of dq = ceil(d / 4)
extern "C" hipError_t mcmc_hip_launch_inc_step_1(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_step_9(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_step_17(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_step_25(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
// ... and their EMIT instantiations (accepted rows stored, `emit: chains`): -DMCMC_INC_EMIT_TU
extern "C" hipError_t mcmc_hip_launch_inc_emit_1(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_emit_9(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_emit_17(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_emit_25(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
// incremental_any.hip: the general incremental kernel (any number of modes / periodic parameters)
extern "C" hipError_t mcmc_hip_launch_inc_any(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
// incremental_duo.hip (round 6): the incremental step of a two-mode mixture with TWO lanes per walker
extern "C" hipError_t mcmc_hip_launch_inc_duo_1(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_inc_duo_9(const mcmc::IncStepArgs*, hipStream_t) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_whiten_directions_planes(const mcmc::IncDirArgs*, int,
                                                               hipStream_t) __attribute__((weak));
extern "C" int mcmc_hip_inc_any_fits(int d, int n_modes, int n_periodic, int n_walkers,
                                     int group_size) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_whiten_state(const double* x, double* y, const double* mean,
                                                   const double* Lrow, int d, int W, int K,
                                                   hipStream_t st) __attribute__((weak));
extern "C" hipError_t mcmc_hip_launch_whiten_directions(const mcmc::IncDirArgs* a, int n_groups,
                                                        hipStream_t st) __attribute__((weak));

// checkpoint_kernels.hip
extern "C" hipError_t mcmc_hip_launch_ckpt_window(const mcmc::CkptWindowArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_ckpt_payload(const mcmc::CkptPayloadArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_ckpt_solve(const mcmc::CkptSolveArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_ckpt_bounds(const mcmc::CkptBoundsArgs* a, int G, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_ckpt_bounds_reduce(const mcmc::CkptBoundsReduceArgs* a, hipStream_t st);
// pliklite_kernels.hip
extern "C" hipError_t mcmc_hip_launch_pl_walker(const mcmc::PlWalkerArgs* a, int accept, int propose,
                                                hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_prior(const double* t, int n, int d, const double* C,
                                               uint32_t norm_mask, double uniform_logp, double* lp,
                                               hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_residual(const mcmc::PlResidualArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_bin(const mcmc::PlBinArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_residual_mfma(const mcmc::PlResidualMfmaArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_fused(const mcmc::PlFusedArgs* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_chi2(const mcmc::PlChi2Args* a, hipStream_t st);
extern "C" hipError_t mcmc_hip_launch_pl_combine(const double* psum, double* chi2, int n, hipStream_t st);

struct mcmc_hip_ctx {
    mcmc_hip_config cfg{};
    const DimKernels* k = nullptr;    // d <= 32: lane-per-walker kernels of that dimension
    const BigKernels* kb = nullptr;   // 32 < d <= 128: column-sweep / matrix-core kernels
    const mcmc::PairKernels* kp = nullptr;   // 32 < d <= 56: the two-wave step kernel, if it fits
    hipStream_t stream = nullptr;
    std::string err;
    int d = 0, W = 0, G = 0, gs = 0, K = -1;
    bool have_prior = false, have_target = false, have_cov = false, have_state = false;
    // host copies of the problem
    std::vector<int32_t> kind, periodic;
    std::vector<double> lo, hi, loc, scale, mls;
    double uniform_logp = 0.0;
    uint32_t norm_mask = 0, periodic_mask = 0;
    uint32_t norm_mask4[4] = {0, 0, 0, 0};
    bool any_periodic = false;
    std::vector<double> mean, Linv, cnorm, weight;  // Linv: [K][d*d] row-major
    std::vector<double> cov, T;                     // proposal
    // parameter blocks (proposal.py:96-196); blocked == false: one block, identity order
    bool blocked = false;
    std::vector<int32_t> blk_size, blk_over, i_of_j;
    int drag_last_slow = -1, drag_steps = 0;
    DevBuf<int> dblk, vflag, vflag_f;               // dblk: size | oversample | i_of_j
    DevBuf<double> Vf;                              // dragging: directions of the fast blocks
    DevBuf<double> drag_cs;                         // drag_general_kernel: start points [d][W]
    std::vector<double> shift;                      // moment shift
    // device
    DevBuf<double> x, logpost, logprior, loglike, cblock, dT, V, rows, gsum, Sg, pooled, dshift;
    DevBuf<double> ex, elp, ell, eder, escratch, dLrow, dLcol;
    // incremental evaluation (MCMC_HIP_FLAG_INCREMENTAL): carried y, per-step (v, u) pairs,
    // padded prior constants, row-major L^-1 and the mean of the one mode
    bool incremental = false;
    bool y_valid = false;
    bool own_basis = false;     // MCMC_HIP_FLAG_OWN_BASIS (d > 1): a Haar basis per walker
    // incremental mode: walkers sharing one Haar basis (a multiple of group_size, flags bits
    // 8..11 = log2 of the multiple); the R-1 groups (moments) stay group_size wide
    int bgs = 0, BG = 0;
    DevBuf<double> y, inc_prior, inc_Lrow, inc_mean;
    // mixtures on the kernels that carry the log-density of every mode (inc_carries_modes):
    // amode[K][W]; valid = written by a launch (or set) since y was; else re-anchored on y
    DevBuf<double> amode;
    bool amode_valid = false;
    // The directions of a launch -- Haar columns V (Vf: the fast blocks' when dragging) and their
    // whitened pairs VU -- do not depend on the walkers' state, so the set of the NEXT launch is
    // computed on a second stream while the step kernel of this one runs (two sets, used in
    // turn).  A set computed ahead is used only if the launch that comes is the one predicted
    // (same first step, same length) and nothing the directions depend on has been set since
    // (dir_epoch); otherwise it is recomputed on the main stream.
    struct DirSet {
        DevBuf<double> V, Vf, VU;
        DevBuf<int> vflag, vflag_f, colflag;   // colflag: 1-D columns of the launch, in VU order
        bool has_flags = false;
        DevBuf<double> UU;                   // |u|^2 of the columns (step_inc_kernel: one mode, no periodic parameter)
        DevBuf<double> VW, NL;               // the carried log-prior's stream and (v.w, loc.w) of the columns
        hipEvent_t ready = nullptr;          // recorded on the stream that filled the set
        bool ahead = false;                  // filled ahead of its launch (on stream2)
        unsigned long long step0 = ~0ull, epoch = 0;
        int n = 0;
    } dirs[2];
    int dir_cur = 0;
    unsigned long long dir_epoch = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t mark = nullptr;               // main stream: behind the last step kernel
    bool mark_valid = false;
    bool prefetch = true;
    // the directions of the launch a call BEGINS with are formed at that call, not at the end of
    // the previous one (step_incremental): a proposal refreshed in between is then in them at once
    bool lazy_dirs = true;
    // step_inc_kernel: calls whose directions are formed together (MCMC_HIP_LOOKAHEAD, default 4)
    int lookahead = 4;
    // incremental_duo.hip (two lanes per walker): -1 = where the ensemble fills the chip with it
    // (kDuoMinWalkers), 0 = never, 1 = wherever the kernel serves the model (MCMC_HIP_DUO)
    int duo = -1;
    hipEvent_t T_event = nullptr;            // main stream: behind the last write of dT
    bool T_fresh = false;                    // ... which no direction set has been ordered behind yet
    // asynchronous checkpoint (mcmc_hip_request_moments / mcmc_hip_fetch_moments) and
    // stream-ordered proposal refresh: pinned host staging
    double* pin_mom = nullptr;                      // [G*d + d(d+1)/2 + 2]
    double* pin_T = nullptr;                        // ring of 4 transforms [4][d*d]
    int pin_T_slot = 0;
    hipEvent_t pin_T_done[4] = {nullptr, nullptr, nullptr, nullptr};   // the copy out of slot k has run
    hipEvent_t mom_event = nullptr;
    bool mom_pending = false;
    int64_t mom_n = 0;
    unsigned long long mom_step = 0;
    // drain_samples_pinned: ring of pinned host slots the packed rows are copied into (PCIe at
    // full rate, and the caller reads them in place)
    struct HostSlot { double* p = nullptr; size_t cap_rows = 0; };
    std::vector<HostSlot> slots = std::vector<HostSlot>(4);
    int slot_next = 0;
    DevBuf<double> pack_out;                        // drain_samples: packed rows
    DevBuf<long long> pack_off;
    DevBuf<int> weight_i, prej, burn, stuck, nrows;
    DevBuf<int> thin_acc;   // thinned emission (mcmc_hip_set_emit_thin): the weight a walker has added up
    int emit_thin = 1;
    DevBuf<long long> nacc;
    DevBuf<unsigned long long> acc_total;
    unsigned long long step = 0;
    int64_t n_snapshots = 0;
    // timing
    bool timing = false;
    struct Ev {
        hipEvent_t a, b;
        int kind;
    };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    // kinds 0..2: step kernels / directions / moment snapshots; 3..5: the three kernels of a
    // step on the binned target (pl_walker, pl_residual, pl_chi2), each launch timed
    double ms[6] = {0, 0, 0, 0, 0, 0};
    int64_t n_seen[6] = {0, 0, 0, 0, 0, 0}, n_timed[6] = {0, 0, 0, 0, 0, 0};   // timed regions per kind (Timed)
    int64_t n_step_launches = 0;
    std::string last_step_kernel;     // what the last step launcher said it launched
    // device-side learn / convergence checkpoint (checkpoint_kernels.hip)
    struct Ckpt {
        DevBuf<double> ring, wsum, payload, ws, out;
        DevBuf<unsigned long long> acc_prev;
        int cap = 0;               // ring slots
        long long n_done = 0;      // checkpoints taken so far (the next one goes to slot n_done % cap)
        double* pin_out = nullptr; // [8 + 2 d^2 + d]: the solve's outcome, or the reduced payload
        hipEvent_t ev = nullptr;
        bool begun = false, pending = false;
        bool payload_only = false; // the pending read-out is the payload (checkpoint_request_payload)
    } ck;
    // R-1 of the confidence bounds (mcmc.py:918-1002): ring of ensemble snapshots [slot][d][W]
    struct Bounds {
        DevBuf<double> ring, bounds, payload;
        int n_slots = 0;
        double* pin = nullptr;     // [1 + 4 d + G d 2]
    } bd;
    // the walker shards' communicator (comm.hip; not owned): the device checkpoint all-reduces
    // its payload over it in stream order
    mcmc_hip_comm* comm = nullptr;
    // binned-bandpower Gaussian target (planck_pliklite.py:143-155; pliklite_kernels.hip)
    struct Binned {
        bool on = false;
        int n_bins = 0, KT = 0, ntw = 0, n_lin = 0, nlp = 0, calib = 0, lmax = 0;
        std::vector<int32_t> bins;                       // [n_bins][3]
        std::vector<double> Linv, Bc0, BJ;               // host copies (tests hand them to the oracle)
        DevBuf<double> resp, theta0, Astream, weights, X, bjs, es;   // bjs, es: pl_residual_mfma_kernel
        DevBuf<double> Afused;                           // pl_fused_kernel: half-tile streams of L^-1
        unsigned long long f_off[8][5][2];
        int f_pairs[8][5][2];
        int f_shift = 0, f_ng = 0;
        DevBuf<int> dbins;
        DevBuf<double> delta, trial, lp_t, Ea, psum;     // step scratch, W walkers
        DevBuf<double> edelta, etrial, elp, echi2, epsum, ecl, eA;   // evaluate scratch
        unsigned long long tile_off[8][5];
        int nk[8][5];
    } bg;
};

namespace {

thread_local const char* g_noted_kernel = nullptr;

int fail(mcmc_hip_ctx* h, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

int block_slots(const mcmc_hip_ctx* h, int which)
{
    if (!h->blocked) return h->d;
    int L = 0;
    for (size_t b = 0; b < h->blk_size.size(); ++b) {
        if (which == 0) L += h->blk_over[b] * h->blk_size[b];
        else if (which == 1) L += ((int)b <= h->drag_last_slow) ? h->blk_size[b] : 0;
        else L += ((int)b > h->drag_last_slow) ? h->blk_size[b] : 0;
    }
    return L;
}

#define HIP_TRY(h, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, MCMC_HIP_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

hipEvent_t get_event(mcmc_hip_ctx* h)
{
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Timed {
    mcmc_hip_ctx* h;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    bool on = false;
    // Every step kernel is timed; of the regions around it (kind 1: directions, kind 2: moment
    // snapshot) one in eight, scaled up in mcmc_hip_kernel_times: an event record is a packet
    // of its own between two dependent kernels (about 6 us each on the critical path).
    Timed(mcmc_hip_ctx* h_, int kind_, hipStream_t st_ = nullptr)
        : h(h_), kind(kind_), st(st_ ? st_ : h_->stream)
    {
        if (h->timing) {
            on = kind == 0 || kind >= 3 || (h->n_seen[kind] % 8) == 0;
            h->n_seen[kind] += 1;
        }
        if (on) {
            h->n_timed[kind] += 1;
            a = get_event(h);
            b = get_event(h);
            (void)hipEventRecord(a, st);
        }
    }
    ~Timed()
    {
        if (on) {
            (void)hipEventRecord(b, st);
            h->pending.push_back({a, b, kind});
        }
    }
};

void resolve_timing(mcmc_hip_ctx* h)
{
    for (auto& e : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) h->ms[e.kind] += ms;
        h->pool.push_back(e.a);
        h->pool.push_back(e.b);
    }
    h->pending.clear();
}

int upload_constants(mcmc_hip_ctx* h)
{
    if (!h->have_prior || !h->have_target) return MCMC_HIP_OK;
    const int d = h->d, K = h->K;
    const ConstLayout cl{d, K};
    std::vector<double> c((size_t)cl.size() + 16, 0.0);
    for (int i = 0; i < d; ++i) {
        c[cl.lo() + i] = h->lo[i];
        c[cl.hi() + i] = h->hi[i];
        c[cl.loc() + i] = h->loc[i];
        c[cl.scale() + i] = h->scale[i];
        c[cl.mls() + i] = h->mls[i];
        c[cl.elem() + 3 * i + 0] = h->lo[i];
        c[cl.elem() + 3 * i + 1] = h->hi[i];
        c[cl.elem() + 3 * i + 2] = K > 0 ? h->mean[i] : 0.0;
    }
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < d; ++i) c[cl.mean(k) + i] = h->mean[(size_t)k * d + i];
        c[cl.cnorm() + k] = h->cnorm[k];
        c[cl.weight() + k] = h->weight[k];
        const double* Lk = h->Linv.data() + (size_t)k * d * d;
        double* dst = c.data() + cl.linv(k);
        mcmc::tri_stream_for_each(d, [&](int idx, int j, int i) { dst[idx] = Lk[(size_t)j * d + i]; });
    }
    HIP_TRY(h, h->cblock.resize(c.size()));
    HIP_TRY(h, hipMemcpyAsync(h->cblock.p, c.data(), sizeof(double) * c.size(),
                              hipMemcpyHostToDevice, h->stream));
    std::vector<double> lcol;
    if (h->own_basis && !h->kb && K > 0) {   // the general step kernel reads row-major L^-1
        HIP_TRY(h, h->dLrow.resize(h->Linv.size()));
        HIP_TRY(h, hipMemcpyAsync(h->dLrow.p, h->Linv.data(), sizeof(double) * h->Linv.size(),
                                  hipMemcpyHostToDevice, h->stream));
    }
    if (h->kb && K > 0) {
        // row-major L^-1 per mode (evaluator) and the column-major, zero-padded [dp][dp]
        // copy of mode 0 that the column-sweep step kernel stages in LDS
        const int dp = h->kb->dp;
        HIP_TRY(h, h->dLrow.resize(h->Linv.size()));
        HIP_TRY(h, hipMemcpyAsync(h->dLrow.p, h->Linv.data(), sizeof(double) * h->Linv.size(),
                                  hipMemcpyHostToDevice, h->stream));
        // 4x4 tiles [column block][row block][col][row], zero above the diagonal and in the pad
        const int nb = dp / 4;
        lcol.assign((size_t)dp * dp, 0.0);
        for (int j = 0; j < d; ++j)
            for (int i = 0; i <= j; ++i)
                lcol[(((size_t)(i / 4) * nb + (j / 4)) * 4 + (i % 4)) * 4 + (j % 4)] =
                    h->Linv[(size_t)j * d + i];
        // then the 16 x 4 tiles (R, kk), kk <= 4R + 3 - shift/4, of the matrix-core kernel,
        // R-major, each in A-operand lane order: lane l holds
        // L^-1[16R - shift + (l & 15)][4kk + (l >> 4)]
        {
            const int sh = h->kb->row_shift;   // the partial row tile comes first
            const int RT = (dp + sh + 15) / 16, KT = (dp + 3) / 4;
            for (int R = 0; R < RT; ++R)
                for (int kk = 0; kk < std::min(4 * R + 4 - sh / 4, KT); ++kk)
                    for (int l = 0; l < 64; ++l) {
                        const int j = 16 * R - sh + (l & 15), i = 4 * kk + (l >> 4);
                        lcol.push_back((j >= 0 && j < d && i <= j) ? h->Linv[(size_t)j * d + i] : 0.0);
                    }
            if ((int)(lcol.size() - (size_t)dp * dp) != 64 * h->kb->n_tiles)
                return fail(h, MCMC_HIP_ERR_ARG, "internal: tile count mismatch");
        }
        HIP_TRY(h, h->dLcol.resize(lcol.size()));
        HIP_TRY(h, hipMemcpyAsync(h->dLcol.p, lcol.data(), sizeof(double) * lcol.size(),
                                  hipMemcpyHostToDevice, h->stream));
    }
    if (h->incremental && K >= 1 && K <= mcmc::kMaxModes) {
        const int dq = (d + 3) / 4, dpad = 4 * dq;
        HIP_TRY(h, h->y.resize((size_t)K * d * h->W));
        if (K > 1) HIP_TRY(h, h->amode.resize((size_t)K * h->W));
        h->amode_valid = false;
        std::vector<double> pr((size_t)5 * dpad, 0.0);
        for (int i = 0; i < dpad; ++i) {
            pr[i] = i < d ? h->lo[i] : -INFINITY;
            pr[dpad + i] = i < d ? h->hi[i] : INFINITY;
            pr[2 * dpad + i] = i < d ? h->loc[i] : 0.0;
            // 1/scale = 0 and mls = 0: "no normal prior here" (kind 0 and the padding)
            const bool nrm = i < d && h->kind[i] == 1;
            pr[3 * dpad + i] = nrm ? 1.0 / h->scale[i] : 0.0;
            pr[4 * dpad + i] = nrm ? h->mls[i] : 0.0;
        }
        HIP_TRY(h, h->inc_prior.resize(pr.size()));
        HIP_TRY(h, hipMemcpyAsync(h->inc_prior.p, pr.data(), sizeof(double) * pr.size(),
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, h->inc_Lrow.resize((size_t)K * d * d));
        HIP_TRY(h, hipMemcpyAsync(h->inc_Lrow.p, h->Linv.data(), sizeof(double) * K * d * d,
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, h->inc_mean.resize((size_t)K * d));
        HIP_TRY(h, hipMemcpyAsync(h->inc_mean.p, h->mean.data(), sizeof(double) * K * d,
                                  hipMemcpyHostToDevice, h->stream));
        h->y_valid = false; h->amode_valid = false;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MCMC_HIP_OK;
}

int lds_check(mcmc_hip_ctx* h)
{
    const ConstLayout cl{h->d, h->K};
    (void)cl;
    if (h->kb) return MCMC_HIP_OK;  // the big step kernel's LDS does not depend on K
    const size_t lds = sizeof(double) * ((h->K > 1 ? (size_t)h->K * 256 : 0) +
                                         2 * (size_t)(256 / h->gs) * mcmc::v_slab(h->d));
    if (lds > 160 * 1024)
        return fail(h, MCMC_HIP_ERR_ARG,
                    "the step kernel needs %zu bytes of LDS per workgroup (> 160 KiB): fewer "
                    "mixture modes or a smaller group_size are required", lds);
    return MCMC_HIP_OK;
}

int set_target_common(mcmc_hip_ctx* h, int K, const double* means, const double* covs,
                      const double* weights, bool normalized)
{
    const int d = h->d;
    if (K < 1 || K > mcmc::kMaxModes)
        return fail(h, MCMC_HIP_ERR_ARG, "n_modes must be in 1..%d, got %d", mcmc::kMaxModes, K);
    std::vector<double> L((size_t)d * d), Li((size_t)d * d);
    h->mean.assign(means, means + (size_t)K * d);
    h->Linv.assign((size_t)K * d * d, 0.0);
    h->cnorm.assign(K, 0.0);
    h->weight.assign(K, 1.0 / K);
    for (int k = 0; k < K; ++k) {
        const double* C = covs + (size_t)k * d * d;
        if (!is_symmetric(d, C) || !cholesky_lower(d, C, L.data()))
            return fail(h, MCMC_HIP_ERR_NOT_PD,
                        "covariance of mode %d is not a symmetric positive-definite matrix", k);
        tri_inverse_lower(d, L.data(), Li.data());
        std::copy(Li.begin(), Li.end(), h->Linv.begin() + (size_t)k * d * d);
        double logdet = 0.0;
        for (int i = 0; i < d; ++i) logdet += std::log(L[i * d + i]);
        logdet *= 2.0;
        h->cnorm[k] = normalized ? d * std::log(2.0 * M_PI) + logdet : 0.0;
    }
    if (weights) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            if (!(weights[k] >= 0.0)) return fail(h, MCMC_HIP_ERR_ARG, "negative mixture weight");
            s += weights[k];
        }
        if (!(s > 0.0)) return fail(h, MCMC_HIP_ERR_ARG, "mixture weights sum to zero");
        const bool renorm = !(std::fabs(s - 1.0) <= 1e-8 + 1e-5);  // np.isclose(sum, 1)
        for (int k = 0; k < K; ++k) h->weight[k] = renorm ? weights[k] / s : weights[k];
    }
    h->K = K;
    h->bg.on = false;
    h->have_target = true;
    ++h->dir_epoch;
    h->have_state = false;
    int rc = lds_check(h);
    if (rc) return rc;
    return upload_constants(h);
}


// ------------------------------------------------------------------ binned Gaussian target
// chain of the chi2 sum that row tile R of NT joins (oracle: binned_class): its position in its
// group of eight tiles, the groups counted down from the last tile.  pl_fused_kernel gives the
// tile at a position to one wave per pair of walker tiles; pl_chi2_kernel (explicit points) gives
// wave q the tiles of class q.
inline int binned_shift(int NT) { return (8 - NT % 8) % 8; }
inline int binned_class(int R, int NT) { return (R + binned_shift(NT)) & 7; }

// the 32 partial sums of chi2 per walker of the residuals held in `delta` (n walkers, a multiple
// of 64) -> psum[32][n]; chi2 (may be null): their combination, one value per walker
int binned_chi2(mcmc_hip_ctx* h, const double* delta, double* psum, double* chi2, int n)
{
    auto& B = h->bg;
    mcmc::PlChi2Args c{};
    c.delta = delta; c.Astream = B.Astream.p; c.psum = psum;
    std::memcpy(c.tile_off, B.tile_off, sizeof c.tile_off);
    std::memcpy(c.nk, B.nk, sizeof c.nk);
    c.KT = B.KT; c.ntw = B.ntw; c.n_walkers = n; c.n_sets = n / 64;
    HIP_TRY(h, mcmc_hip_launch_pl_chi2(&c, h->stream));
    if (chi2) HIP_TRY(h, mcmc_hip_launch_pl_combine(psum, chi2, n, h->stream));
    return MCMC_HIP_OK;
}

int binned_residual(mcmc_hip_ctx* h, const double* trial, double* delta, int n)
{
    auto& B = h->bg;
    // MCMC_HIP_PL_SCALAR_RESIDUAL (developer switch): the lane-per-walker kernel of round 3
    static const bool scalar = getenv("MCMC_HIP_PL_SCALAR_RESIDUAL") != nullptr;
    if (!scalar) {
        mcmc::PlResidualMfmaArgs m{};
        m.trial = trial; m.theta0 = B.theta0.p; m.bjs = B.bjs.p; m.es = B.es.p; m.delta = delta;
        m.W = n; m.KT = B.KT; m.n_lin = B.n_lin; m.np = (B.n_lin + 7) / 8; m.calib = B.calib;
        m.n_tiles = (B.KT + 3) / 4;
        HIP_TRY(h, mcmc_hip_launch_pl_residual_mfma(&m, h->stream));
        return MCMC_HIP_OK;
    }
    mcmc::PlResidualArgs r{};
    r.trial = trial; r.theta0 = B.theta0.p; r.resp = B.resp.p; r.delta = delta;
    r.W = n; r.n_bins = B.n_bins; r.KT = B.KT; r.n_lin = B.n_lin; r.nlp = B.nlp; r.calib = B.calib;
    HIP_TRY(h, mcmc_hip_launch_pl_residual(&r, h->stream));
    return MCMC_HIP_OK;
}

// Model.logposterior for n points on the binned target (mcmc_hip_evaluate)
int evaluate_binned_points(mcmc_hip_ctx* h, int n, const double* x, double* logprior, double* loglike)
{
    auto& B = h->bg;
    const size_t d = h->d, np = ((size_t)n + 63) & ~(size_t)63;
    std::vector<double> t(d * np);
    for (size_t w = 0; w < np; ++w)
        for (size_t i = 0; i < d; ++i) t[i * np + w] = x[(w < (size_t)n ? w : 0) * d + i];
    HIP_TRY(h, B.etrial.resize(d * np));
    HIP_TRY(h, B.elp.resize(np));
    HIP_TRY(h, B.echi2.resize(np));
    HIP_TRY(h, B.edelta.resize((np / 64) * (size_t)B.KT * 256 + (size_t)mcmc::kPlPad * 256));
    HIP_TRY(h, hipMemcpyAsync(B.etrial.p, t.data(), sizeof(double) * d * np, hipMemcpyHostToDevice,
                              h->stream));
    HIP_TRY(h, mcmc_hip_launch_pl_prior(B.etrial.p, (int)np, (int)d, h->cblock.p, h->norm_mask,
                                        h->uniform_logp, B.elp.p, h->stream));
    int rc = binned_residual(h, B.etrial.p, B.edelta.p, (int)np);
    if (rc) return rc;
    HIP_TRY(h, B.epsum.resize(32 * np));
    rc = binned_chi2(h, B.edelta.p, B.epsum.p, B.echi2.p, (int)np);
    if (rc) return rc;
    std::vector<double> c2(np), lp(np);
    HIP_TRY(h, hipMemcpyAsync(lp.data(), B.elp.p, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(c2.data(), B.echi2.p, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int w = 0; w < n; ++w) {
        logprior[w] = lp[w];
        // (the likelihood is skipped outside the prior support, model.py:650-653)
        loglike[w] = std::isinf(lp[w]) ? -INFINITY : -0.5 * c2[w];
    }
    return MCMC_HIP_OK;
}

// mcmc_hip_step on the binned target: per step  [accept of the previous trial +] proposal ->
// residuals -> chi2 on the matrix cores; a call ends with the accept of its last trial, so the
// state is complete between calls.
int step_binned(mcmc_hip_ctx* h, int n_steps)
{
    auto& B = h->bg;
    const int d = h->d, W = h->W;
    if (h->blocked || h->drag_last_slow >= 0 || h->own_basis || h->cfg.emit_capacity > 0 || h->any_periodic)
        return fail(h, MCMC_HIP_ERR_ARG,
                    "the binned Gaussian target serves one parameter block, the shared basis, "
                    "non-periodic priors and emit_capacity 0");
    HIP_TRY(h, B.trial.resize((size_t)d * W));
    HIP_TRY(h, B.lp_t.resize(W));
    HIP_TRY(h, B.Ea.resize(W));
    HIP_TRY(h, B.psum.resize((size_t)32 * W));
    // MCMC_HIP_PL_UNFUSED (developer switch): residuals and chi2 as two launches (round 3)
    static const bool unfused = getenv("MCMC_HIP_PL_UNFUSED") != nullptr;
    if (unfused)    // (fused: delta lives in LDS, 323 MB of HBM less at 65 536 walkers)
        HIP_TRY(h, B.delta.resize(((size_t)W / 64) * (size_t)B.KT * 256 + (size_t)mcmc::kPlPad * 256));
    const size_t dd = (size_t)mcmc::v_slab(d);
    const int max_cyc = (int)std::max<size_t>(1, (64u << 20) / (sizeof(double) * dd * (size_t)h->G));
    mcmc::PlWalkerArgs a{};
    a.s.x = h->x.p; a.s.logpost = h->logpost.p; a.s.logprior = h->logprior.p;
    a.s.loglike = h->loglike.p; a.s.weight = h->weight_i.p; a.s.prior_rej = h->prej.p;
    a.s.burn_left = h->burn.p; a.s.n_accept = h->nacc.p; a.s.stuck = h->stuck.p;
    a.s.accept_total = h->acc_total.p;
    a.s.cblock = h->cblock.p; a.s.W = W; a.s.group_size = h->gs; a.s.n_modes = 0;
    a.s.norm_mask = h->norm_mask; a.s.walker0 = h->cfg.walker_offset;
    a.s.key0 = (uint32_t)h->cfg.seed; a.s.key1 = (uint32_t)(h->cfg.seed >> 32);
    a.s.uniform_logp = h->uniform_logp; a.s.temperature = h->cfg.temperature;
    a.s.max_tries = h->cfg.max_tries; a.s.cps = d; a.s.slab = (int)dd;
    a.d = d; a.trial = B.trial.p; a.lp_t = B.lp_t.p; a.Ea = B.Ea.p; a.psum_t = B.psum.p;
    int left = n_steps;
    bool pending = false;   // a trial has been proposed and evaluated, not yet accepted / rejected
    while (left > 0) {
        const unsigned long long c0 = h->step / (unsigned long long)d;
        const unsigned long long room = (c0 + (unsigned long long)max_cyc) * d - h->step;
        const int n = (int)std::min<unsigned long long>((unsigned long long)left, room);
        const int ncyc = (int)((h->step + (unsigned long long)n - 1) / d - c0 + 1);
        {
            Timed t(h, 1);
            HIP_TRY(h, h->V.resize((size_t)h->G * ncyc * dd));
            mcmc::BasisArgs b{};
            b.T = h->dT.p; b.V = h->V.p;
            b.group0 = h->cfg.walker_offset / (uint32_t)h->gs;
            b.cycle0 = (uint32_t)c0;
            b.key0 = (uint32_t)h->cfg.seed; b.key1 = (uint32_t)(h->cfg.seed >> 32);
            b.ncyc = ncyc;
            HIP_TRY(h, h->k->basis(b, h->G, h->stream));
        }
        a.s.V = h->V.p; a.s.ncyc = ncyc;
        for (int s = 0; s < n; ++s) {
            a.s.step0 = h->step;
            a.cyc = (int)(h->step / (unsigned long long)d - c0);
            a.col = (int)(h->step % (unsigned long long)d);
            {
                Timed t(h, 3);
                HIP_TRY(h, mcmc_hip_launch_pl_walker(&a, pending ? 1 : 0, 1, h->stream));
            }
            if (unfused) {
                {
                    Timed t(h, 4);
                    const int rc = binned_residual(h, B.trial.p, B.delta.p, W);
                    if (rc) return rc;
                }
                Timed t(h, 5);
                const int rc = binned_chi2(h, B.delta.p, B.psum.p, nullptr, W);
                if (rc) return rc;
            } else {
                Timed t(h, 5);
                mcmc::PlFusedArgs f{};
                f.trial = B.trial.p; f.theta0 = B.theta0.p; f.bjs = B.bjs.p; f.es = B.es.p;
                f.Astream = B.Afused.p; f.psum = B.psum.p;
                std::memcpy(f.a_off, B.f_off, sizeof f.a_off);
                std::memcpy(f.a_pairs, B.f_pairs, sizeof f.a_pairs);
                f.W = W; f.KT = B.KT; f.n_lin = B.n_lin; f.np = (B.n_lin + 7) / 8; f.calib = B.calib;
                f.n_tiles = (B.KT + 3) / 4; f.shift = B.f_shift; f.ng = B.f_ng; f.n_sets = W / 64;
                HIP_TRY(h, mcmc_hip_launch_pl_fused(&f, h->stream));
            }
            h->n_step_launches += 1;
            pending = true;
            h->step += 1;
        }
        left -= n;
    }
    if (pending) {
        Timed t(h, 3);
        HIP_TRY(h, mcmc_hip_launch_pl_walker(&a, 1, 0, h->stream));
    }
    if (g_noted_kernel) {
        h->last_step_kernel = std::string(g_noted_kernel) + " (n_bins=" + std::to_string(B.n_bins) + ")";
        g_noted_kernel = nullptr;
    }
    return MCMC_HIP_OK;
}

}  // namespace

extern "C" {

const char* mcmc_hip_version(void) { return "mcmc_hip 0.1 (gfx950)"; }

const char* mcmc_hip_last_error(const mcmc_hip_ctx* h)
{
    return h ? h->err.c_str() : g_create_error.c_str();
}

int mcmc_hip_dim_supported(int d)
{
    return kernels_for_dim(d) != nullptr || big_for_dim(d) != nullptr;
}

int mcmc_hip_incremental_supported(int32_t d, int32_t n_modes, int32_t n_periodic, int32_t n_drag,
                                   int32_t n_walkers, int32_t basis_group_size)
{
    const int dq = (d + 3) / 4, K = n_modes;
    if (d < 2 || d > 128 || K < 1 || K > mcmc::kMaxModes || n_periodic < 0 || n_periodic > d ||
        n_drag < 0 || n_walkers <= 0 || basis_group_size <= 0 || basis_group_size % 64 != 0 ||
        n_walkers % basis_group_size != 0 || !mcmc_hip_launch_whiten_state)
        return 0;
    if (n_drag > 0) {   // (step_incremental: a step's 1 + n_drag columns fit the LDS twice over)
        const int chunk_steps = std::max(1, (1024 / (4 * dq)) / (1 + n_drag));
        const size_t drag_lds = sizeof(double) * 2 * 2 * (size_t)chunk_steps * (1 + n_drag) * 4 * dq;
        return K == 1 && n_periodic == 0 && drag_lds <= (128u << 10);
    }
    if ((K == 1 || mcmc::inc_mix_serves(K, dq)) && (n_periodic == 0 || (K == 1 && n_periodic <= mcmc::kIncMaxPeriodic)))
        return 1;
    return mcmc_hip_inc_any_fits &&
           mcmc_hip_inc_any_fits(d, K, n_periodic, n_walkers, basis_group_size) ? 1 : 0;
}

int mcmc_hip_create(const mcmc_hip_config* cfg, mcmc_hip_ctx** out)
{
    if (!cfg || !out) return fail(nullptr, MCMC_HIP_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->d < 1) return fail(nullptr, MCMC_HIP_ERR_ARG, "d must be >= 1, got %d", cfg->d);
    const DimKernels* k = kernels_for_dim(cfg->d);
    const BigKernels* kb = k ? nullptr : big_for_dim(cfg->d);
    if (!k && !kb)
        return fail(nullptr, MCMC_HIP_ERR_ARG,
                    "no kernels compiled for d=%d (this build covers d = 1..32 lane-per-walker "
                    "and 33..%d column-sweep, as selected at build time)", cfg->d, kMaxDimBig);
    if (cfg->group_size != 64 && cfg->group_size != 128 && cfg->group_size != 256)
        return fail(nullptr, MCMC_HIP_ERR_ARG, "group_size must be 64, 128 or 256, got %d",
                    cfg->group_size);
    if (cfg->n_walkers < cfg->group_size || cfg->n_walkers % cfg->group_size)
        return fail(nullptr, MCMC_HIP_ERR_ARG,
                    "n_walkers (%d) must be a positive multiple of group_size (%d)",
                    cfg->n_walkers, cfg->group_size);
    if (cfg->walker_offset % (uint32_t)cfg->group_size)
        return fail(nullptr, MCMC_HIP_ERR_ARG, "walker_offset must be a multiple of group_size");
    if (!(cfg->temperature > 0) || !(cfg->proposal_scale > 0))
        return fail(nullptr, MCMC_HIP_ERR_ARG, "temperature and proposal_scale must be > 0");
    if (cfg->flags & ~(MCMC_HIP_FLAG_INCREMENTAL | MCMC_HIP_FLAG_OWN_BASIS | MCMC_HIP_FLAG_BASIS_GROUP_MASK))
        return fail(nullptr, MCMC_HIP_ERR_ARG, "unknown flags 0x%x", (unsigned)cfg->flags);
    {
        const int m = (cfg->flags & MCMC_HIP_FLAG_BASIS_GROUP_MASK) >> 8;
        const long long bgs = (long long)cfg->group_size << m;
        if (m && (!(cfg->flags & MCMC_HIP_FLAG_INCREMENTAL) || m > 6 || cfg->n_walkers % bgs ||
                  cfg->walker_offset % bgs))
            return fail(nullptr, MCMC_HIP_ERR_ARG,
                        "a basis group wider than group_size needs incremental evaluation, and "
                        "n_walkers and walker_offset must be multiples of it (%lld)", bgs);
    }
    if ((cfg->flags & MCMC_HIP_FLAG_INCREMENTAL) && (cfg->flags & MCMC_HIP_FLAG_OWN_BASIS))
        return fail(nullptr, MCMC_HIP_ERR_ARG,
                    "incremental evaluation needs the shared basis (the whitened direction is "
                    "shared with it)");
    if ((cfg->flags & MCMC_HIP_FLAG_INCREMENTAL) &&
        (cfg->d < 2 || cfg->group_size % 64 != 0 || !mcmc_hip_launch_whiten_state))
        return fail(nullptr, MCMC_HIP_ERR_ARG,
                    "incremental evaluation needs d >= 2 and a group_size that is a multiple of 64");
    if (cfg->emit_capacity < 0 || cfg->burn_in < 0)
        return fail(nullptr, MCMC_HIP_ERR_ARG, "emit_capacity and burn_in must be >= 0");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, MCMC_HIP_ERR_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MCMC_HIP_ERR_ARG, "device %d out of range (0..%d)", cfg->device,
                    ndev - 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, MCMC_HIP_ERR_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MCMC_HIP_ERR_DEVICE,
                    "device %d is %s; this library is built for gfx950 only", cfg->device,
                    prop.gcnArchName);
    if (hipSetDevice(cfg->device) != hipSuccess)
        return fail(nullptr, MCMC_HIP_ERR_DEVICE, "hipSetDevice(%d) failed", cfg->device);
    mcmc_hip_ctx* h = new mcmc_hip_ctx();
    h->cfg = *cfg;
    h->k = k;
    h->kb = kb;
    // MCMC_HIP_NO_PAIR_BIG (developer switch): keep 32 < d <= 56 on the matrix-core kernel
    h->kp = (kb && !getenv("MCMC_HIP_NO_PAIR_BIG")) ? pair_for_dim(cfg->d) : nullptr;
    h->d = cfg->d;
    h->W = cfg->n_walkers;
    h->gs = cfg->group_size;
    h->G = h->W / h->gs;
    h->bgs = h->gs << ((cfg->flags & MCMC_HIP_FLAG_BASIS_GROUP_MASK) >> 8);
    h->BG = h->W / h->bgs;
    h->shift.assign(h->d, 0.0);
    h->incremental = (cfg->flags & MCMC_HIP_FLAG_INCREMENTAL) != 0;
    h->own_basis = (cfg->flags & MCMC_HIP_FLAG_OWN_BASIS) != 0 && cfg->d > 1;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(nullptr, MCMC_HIP_ERR_DEVICE, "hipStreamCreate failed");
    }
    const size_t W = h->W, d = h->d, G = h->G, np = d * (d + 1) / 2;
    hipError_t r = hipSuccess;
    auto acc = [&](hipError_t x) { if (r == hipSuccess) r = x; };
    acc(h->x.resize(W * d)); acc(h->logpost.resize(W)); acc(h->logprior.resize(W));
    acc(h->loglike.resize(W)); acc(h->weight_i.resize(W)); acc(h->prej.resize(W));
    acc(h->burn.resize(W)); acc(h->nacc.resize(W)); acc(h->stuck.resize(1));
    acc(h->dT.resize(d * d + 16)); /* +16: wide scalar loads at the end of T */ acc(h->Sg.resize(G * np));
    // the checkpoint's read-out in ONE block (one copy and one fill per checkpoint instead of
    // three and two): [group sums | pooled second moments | accept counter]; `pooled` and
    // `acc_total` are views into it (n = 0: not owned)
    acc(h->gsum.resize(G * d + np + 1));
    if (r == hipSuccess) {
        h->pooled.p = h->gsum.p + G * d;
        h->acc_total.p = reinterpret_cast<unsigned long long*>(h->gsum.p + G * d + np);
    }
    acc(h->dshift.resize(d));
    if (cfg->emit_capacity > 0) {
        acc(h->rows.resize(W * (size_t)cfg->emit_capacity * (d + 4)));
        acc(h->nrows.resize(W));
    }
    // (y is sized when the target is known: [K][d][W])
    acc(hipHostMalloc((void**)&h->pin_mom, sizeof(double) * (G * d + np + 2), hipHostMallocDefault));
    acc(hipHostMalloc((void**)&h->pin_T, sizeof(double) * 4 * d * d, hipHostMallocDefault));
    for (auto& e : h->pin_T_done) acc(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    acc(hipEventCreateWithFlags(&h->mom_event, hipEventDisableTiming));
    if (h->incremental) {
        // (LOWEST priority: the step kernel's 1024 workgroups are exactly what the chip holds
        // at once, four per compute unit; a direction kernel that took some of those places
        // first would push the displaced step workgroups into a second round -- 1.74 ms
        // instead of 1.22, measured with the order of the two reversed.  The direction kernels
        // run behind the step kernel, beside the moment snapshot: step_incremental.)
        int prio_least = 0, prio_greatest = 0;
        acc(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        acc(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_least));
        acc(hipEventCreateWithFlags(&h->mark, hipEventDisableTiming));
        acc(hipEventCreateWithFlags(&h->T_event, hipEventDisableTiming));
        if (const char* e = getenv("MCMC_HIP_EAGER_DIRECTIONS")) h->lazy_dirs = !(e[0] && e[0] != '0');
        // MCMC_HIP_LOOKAHEAD (developer switch): calls per direction set of step_inc_kernel; 1: a set per call
        if (const char* e = getenv("MCMC_HIP_LOOKAHEAD")) h->lookahead = std::max(1, atoi(e));
        // MCMC_HIP_DUO (developer switch): 0 = the four-lane kernels always, 1 = the two-lane ones wherever they serve
        if (const char* e = getenv("MCMC_HIP_DUO")) h->duo = (e[0] && e[0] != '0') ? 1 : 0;
        for (auto& D : h->dirs) acc(hipEventCreateWithFlags(&D.ready, hipEventDisableTiming));
        // MCMC_HIP_NO_PREFETCH (developer switch): directions on the main stream, in line
        h->prefetch = !getenv("MCMC_HIP_NO_PREFETCH");
    }
    if (r == hipSuccess) r = hipMemsetAsync(h->gsum.p, 0, sizeof(double) * G * d, h->stream);
    if (r == hipSuccess) r = hipMemsetAsync(h->pooled.p, 0, sizeof(double) * np, h->stream);
    if (r == hipSuccess) r = hipMemsetAsync(h->dshift.p, 0, sizeof(double) * d, h->stream);
    if (r == hipSuccess) r = hipMemsetAsync(h->stuck.p, 0, sizeof(int), h->stream);
    if (r == hipSuccess) r = hipMemsetAsync(h->acc_total.p, 0, sizeof(unsigned long long), h->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(h->stream);
    if (r != hipSuccess) {
        fail(nullptr, MCMC_HIP_ERR_DEVICE, "device allocation failed: %s", hipGetErrorString(r));
        mcmc_hip_destroy(h);
        return MCMC_HIP_ERR_DEVICE;
    }
    *out = h;
    return MCMC_HIP_OK;
}

void mcmc_hip_destroy(mcmc_hip_ctx* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    resolve_timing(h);
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (auto& D : h->dirs) {
        D.V.release(); D.Vf.release(); D.VU.release(); D.vflag.release(); D.vflag_f.release(); D.UU.release();
        D.colflag.release(); D.VW.release(); D.NL.release();
        if (D.ready) (void)hipEventDestroy(D.ready);
    }
    if (h->mark) (void)hipEventDestroy(h->mark);
    if (h->T_event) (void)hipEventDestroy(h->T_event);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    h->x.release(); h->logpost.release(); h->logprior.release(); h->loglike.release();
    h->cblock.release(); h->dT.release(); h->V.release(); h->rows.release(); h->gsum.release();
    h->Sg.release(); h->pooled.p = nullptr; /* (a view into gsum) */ h->dshift.release(); h->ex.release(); h->elp.release();
    h->ell.release(); h->eder.release(); h->escratch.release(); h->dLrow.release();
    h->dLcol.release(); h->weight_i.release(); h->prej.release();
    h->burn.release(); h->stuck.release(); h->nrows.release(); h->nacc.release();
    h->acc_total.p = nullptr;   // (a view into gsum)
    h->dblk.release(); h->vflag.release(); h->vflag_f.release(); h->Vf.release(); h->drag_cs.release();
    for (auto& sl : h->slots)
        if (sl.p) (void)hipHostFree(sl.p);
    h->ck.ring.release(); h->ck.wsum.release(); h->ck.payload.release(); h->ck.ws.release();
    h->ck.out.release(); h->ck.acc_prev.release();
    h->bd.ring.release(); h->bd.bounds.release(); h->bd.payload.release();
    if (h->bd.pin) (void)hipHostFree(h->bd.pin);
    if (h->ck.pin_out) (void)hipHostFree(h->ck.pin_out);
    if (h->ck.ev) (void)hipEventDestroy(h->ck.ev);
    if (h->pin_mom) (void)hipHostFree(h->pin_mom);
    if (h->pin_T) (void)hipHostFree(h->pin_T);
    for (auto& e : h->pin_T_done) if (e) (void)hipEventDestroy(e);
    if (h->mom_event) (void)hipEventDestroy(h->mom_event);
    h->pack_out.release(); h->pack_off.release();
    h->y.release(); h->amode.release(); h->inc_prior.release(); h->inc_Lrow.release();
    h->inc_mean.release();
    {
        auto& B = h->bg;
        B.bjs.release(); B.es.release(); B.Afused.release();
        B.resp.release(); B.theta0.release(); B.Astream.release(); B.weights.release();
        B.X.release(); B.dbins.release(); B.delta.release(); B.trial.release(); B.lp_t.release();
        B.Ea.release(); B.psum.release(); B.epsum.release(); B.edelta.release(); B.etrial.release(); B.elp.release();
        B.echi2.release(); B.ecl.release(); B.eA.release();
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int mcmc_hip_set_prior(mcmc_hip_ctx* h, const int32_t* kind, const double* a, const double* b,
                       const int32_t* periodic)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!kind || !a || !b) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const int d = h->d;
    const double inf = std::numeric_limits<double>::infinity();
    h->kind.assign(kind, kind + d);
    h->periodic.assign(d, 0);
    h->lo.assign(d, -inf); h->hi.assign(d, inf);
    h->loc.assign(d, 0.0); h->scale.assign(d, 1.0); h->mls.assign(d, 0.0);
    h->norm_mask = h->periodic_mask = 0;
    h->norm_mask4[0] = h->norm_mask4[1] = h->norm_mask4[2] = h->norm_mask4[3] = 0;
    h->any_periodic = false;
    double ulp = 0.0;
    for (int i = 0; i < d; ++i) {
        if (kind[i] == 0) {
            if (!(b[i] > a[i]) || !std::isfinite(a[i]) || !std::isfinite(b[i]))
                return fail(h, MCMC_HIP_ERR_ARG, "uniform prior %d needs finite min < max", i);
            h->lo[i] = a[i]; h->hi[i] = b[i];
            ulp += std::log(b[i] - a[i]);
            if (periodic && periodic[i]) {
                h->periodic[i] = 1;
                h->any_periodic = true;
                if (i < 32) h->periodic_mask |= 1u << i;
            }
        } else if (kind[i] == 1) {
            if (!(b[i] > 0) || !std::isfinite(a[i]) || !std::isfinite(b[i]))
                return fail(h, MCMC_HIP_ERR_ARG, "normal prior %d needs finite loc, scale > 0", i);
            if (periodic && periodic[i])
                return fail(h, MCMC_HIP_ERR_ARG,
                            "parameter %d cannot be periodic if it is not bounded", i);
            h->loc[i] = a[i]; h->scale[i] = b[i];
            h->mls[i] = -std::log(b[i]) - std::log(2.0 * M_PI) / 2.0;  // tools.py:723
            h->norm_mask4[i >> 5] |= 1u << (i & 31);
            if (i < 32) h->norm_mask |= 1u << i;
        } else {
            return fail(h, MCMC_HIP_ERR_ARG,
                        "prior kind %d of parameter %d is not supported (0 uniform, 1 norm)",
                        kind[i], i);
        }
    }
    h->uniform_logp = -ulp;  // prior.py:528-533
    h->have_prior = true;
    ++h->dir_epoch;
    h->have_state = false;
    return upload_constants(h);
}

int mcmc_hip_set_target_gaussian_mixture(mcmc_hip_ctx* h, int32_t n_modes, const double* means,
                                         const double* covs, const double* weights)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!means || !covs) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    return set_target_common(h, n_modes, means, covs, weights, true);
}

int mcmc_hip_set_target_gaussian(mcmc_hip_ctx* h, const double* mean, const double* cov,
                                 int32_t normalized)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!mean || !cov) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    return set_target_common(h, 1, mean, cov, nullptr, normalized != 0);
}

int mcmc_hip_set_target_one(mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    h->K = 0;
    h->bg.on = false;
    h->mean.clear(); h->Linv.clear(); h->cnorm.clear(); h->weight.clear();
    h->have_target = true;
    ++h->dir_epoch;
    h->have_state = false;
    return upload_constants(h);
}

int mcmc_hip_set_target_binned_gaussian(mcmc_hip_ctx* h, int32_t n_bins, const int32_t* bins,
                                        int32_t lmax, const double* weights, const double* X,
                                        const double* cov, int32_t n_lin, const double* theta0,
                                        const double* D0, const double* J, int32_t calib_index)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!bins || !weights || !X || !cov || !theta0 || !D0 || !J)
        return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const int d = h->d;
    if (!h->k || n_lin != d - 1 || n_lin < 1 || calib_index < 0 || calib_index >= d)
        return fail(h, MCMC_HIP_ERR_ARG,
                    "the binned Gaussian target takes d - 1 = %d emulator parameters and one "
                    "calibration parameter among 2 <= d <= 32 sampled ones (n_lin=%d, calib=%d)",
                    d - 1, n_lin, calib_index);
    if (n_bins < 1 || n_bins > 640 || lmax < 1)
        return fail(h, MCMC_HIP_ERR_ARG, "n_bins must be in 1..640 (got %d) and lmax >= 1", n_bins);
    if (h->incremental || h->own_basis || h->cfg.emit_capacity > 0)
        return fail(h, MCMC_HIP_ERR_ARG,
                    "the binned Gaussian target is evaluated from scratch with the shared basis "
                    "and emit_capacity 0 (it is not Gaussian in the calibration parameter)");
    for (int b = 0; b < n_bins; ++b) {
        const int tp = bins[3 * b], l0 = bins[3 * b + 1], l1 = bins[3 * b + 2];
        if (tp < 0 || tp > 2 || l0 < 0 || l1 < l0 || l1 > lmax)
            return fail(h, MCMC_HIP_ERR_ARG, "bin %d = (%d, %d, %d) is not inside 0..%d", b, tp, l0, l1, lmax);
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto& B = h->bg;
    const size_t n = n_bins, L1 = (size_t)lmax + 1;
    // cov = L L^T; chi2 = |L^-1 delta|^2 (the quadratic form of functions.py:64-78)
    std::vector<double> L(n * n);
    B.Linv.assign(n * n, 0.0);
    if (!is_symmetric(n_bins, cov) || !cholesky_lower(n_bins, cov, L.data()))
        return fail(h, MCMC_HIP_ERR_NOT_PD,
                    "the covariance of the binned data is not a symmetric positive-definite matrix");
    tri_inverse_lower(n_bins, L.data(), B.Linv.data());
    // binned response of the linear emulator (oracle: orc_binned_collapse)
    B.Bc0.assign(n, 0.0);
    B.BJ.assign(n * (size_t)n_lin, 0.0);
    for (size_t b = 0; b < n; ++b) {
        const size_t tp = bins[3 * b], l0 = bins[3 * b + 1], l1 = bins[3 * b + 2];
        double acc = 0.0;
        for (size_t l = l0; l <= l1; ++l) acc = std::fma(D0[tp * L1 + l], weights[l], acc);
        B.Bc0[b] = acc;
        for (int p = 0; p < n_lin; ++p) {
            double a = 0.0;
            for (size_t l = l0; l <= l1; ++l) a = std::fma(J[(tp * L1 + l) * n_lin + p], weights[l], a);
            B.BJ[b * n_lin + p] = a;
        }
    }
    B.n_bins = n_bins; B.lmax = lmax; B.n_lin = n_lin; B.calib = calib_index;
    B.nlp = (n_lin + 3) & ~3;
    // k-steps of four bins, an EVEN number of them: pl_chi2_kernel fetches the operands of two
    // k-steps with one 16-byte load (a padding k-step is zeros: exact no-ops at the end of a chain)
    B.KT = (((n_bins + 3) / 4) + 1) & ~1;
    B.bins.assign(bins, bins + 3 * n);
    const int NT = (n_bins + 15) / 16;
    B.ntw = (NT + 7) / 8;
    // records (Bc0_b, BJ_b0 .. BJ_b,nlp-1, X_b) and the padded fiducial point
    std::vector<double> resp(n * (size_t)(B.nlp + 2), 0.0), th((size_t)B.nlp, 0.0);
    for (size_t b = 0; b < n; ++b) {
        double* r = resp.data() + b * (size_t)(B.nlp + 2);
        r[0] = B.Bc0[b];
        for (int p = 0; p < n_lin; ++p) r[1 + p] = B.BJ[b * n_lin + p];
        r[1 + B.nlp] = X[b];
    }
    th.resize(32, 0.0);    // (pl_residual_mfma_kernel reads 8 np <= 32 entries)
    std::copy(theta0, theta0 + n_lin, th.begin());
    // the same response as matrix-core operands (PlResidualMfmaArgs)
    const int n_tiles = (B.KT + 3) / 4, npairs = (n_lin + 7) / 8;
    std::vector<double> bjs((size_t)n_tiles * npairs * 128, 0.0), es((size_t)n_tiles * 4 * 128, 0.0);
    for (int T = 0; T < n_tiles; ++T) {
        for (int jp = 0; jp < npairs; ++jp)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const size_t b = 16 * (size_t)T + (l & 15);
                    const int p = 4 * (2 * jp + e) + (l >> 4);
                    if (b < n && p < n_lin)
                        bjs[(((size_t)T * npairs + jp) * 64 + l) * 2 + e] = B.BJ[b * n_lin + p];
                }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const size_t b = 16 * (size_t)T + 4 * r + (l >> 4);
                if (b >= n) continue;
                es[(((size_t)T * 4 + r / 2) * 64 + l) * 2 + (r & 1)] = B.Bc0[b];
                es[(((size_t)T * 4 + 2 + r / 2) * 64 + l) * 2 + (r & 1)] = X[b];
            }
    }
    HIP_TRY(h, B.bjs.resize(bjs.size()));
    HIP_TRY(h, B.es.resize(es.size()));
    HIP_TRY(h, hipMemcpy(B.bjs.p, bjs.data(), sizeof(double) * bjs.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.es.p, es.data(), sizeof(double) * es.size(), hipMemcpyHostToDevice));
    // tiles of L^-1 per wave of pl_chi2_kernel: wave q owns the 16-row tiles of class q in
    // ascending order, absent tiles first; tile R has min(4 R + 4, KT) k-steps
    std::vector<double> As;
    for (int q = 0; q < 8; ++q) {
        std::vector<int> mine;
        for (int R = 0; R < NT; ++R)
            if (binned_class(R, NT) == q) mine.push_back(R);
        const int absent = B.ntw - (int)mine.size();
        for (int t = 0; t < 5; ++t) { B.nk[q][t] = 0; B.tile_off[q][t] = 0; }
        for (int t = 0; t < (int)mine.size(); ++t) {
            const int R = mine[t], nk = std::min(4 * R + 4, B.KT);   // (even)
            B.nk[q][absent + t] = nk;
            B.tile_off[q][absent + t] = As.size();
            // (A-operand lane order, the k-steps 2 m and 2 m + 1 of a lane side by side)
            for (int kk2 = 0; kk2 < nk / 2; ++kk2)
                for (int l = 0; l < 64; ++l)
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const size_t j = 16 * (size_t)R + (l & 15), i = 4 * (size_t)(2 * kk2 + h2) + (l >> 4);
                        As.push_back((j < n && i <= j) ? B.Linv[j * n + i] : 0.0);
                    }
        }
    }
    As.resize(As.size() + (size_t)mcmc::kPlPad * 64, 0.0);   // (operands are fetched ahead)
    // pl_fused_kernel: per wave q and group G of eight virtual tiles (virtual = real + shift) the
    // tile at position s = min(q, 7 - q) (half 0) and at 7 - s (half 1), each as a stream of its
    // k-step pairs from pair 0, in A-operand lane order; absent tiles point at a block of zeros
    {
        const int sh = binned_shift(NT), NG = (NT + sh) / 8;
        B.f_shift = sh; B.f_ng = NG;
        std::vector<double> Af(128, 0.0);       // [0, 128): the zero block
        for (int q = 0; q < 8; ++q) {
            const int s_pos = q < 4 ? q : 7 - q;
            for (int G = 0; G < 5; ++G)
                for (int hf = 0; hf < 2; ++hf) {
                    B.f_off[q][G][hf] = 0; B.f_pairs[q][G][hf] = 0;
                    const int R = 8 * G + (hf ? 7 - s_pos : s_pos) - sh;
                    if (G >= NG || R < 0 || R >= NT) continue;
                    const int np2 = std::min(2 * R + 2, B.KT / 2);
                    B.f_off[q][G][hf] = Af.size();
                    B.f_pairs[q][G][hf] = np2 + 2 * sh;
                    for (int P = 0; P < np2; ++P)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 2; ++e) {
                                const size_t j = 16 * (size_t)R + (l & 15), i = 4 * (size_t)(2 * P + e) + (l >> 4);
                                Af.push_back((j < n && i <= j) ? B.Linv[j * n + i] : 0.0);
                            }
                }
        }
        Af.resize(Af.size() + 256, 0.0);
        HIP_TRY(h, B.Afused.resize(Af.size()));
        HIP_TRY(h, hipMemcpy(B.Afused.p, Af.data(), sizeof(double) * Af.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(h, B.resp.resize(resp.size()));
    HIP_TRY(h, B.theta0.resize(th.size()));
    HIP_TRY(h, B.Astream.resize(As.size()));
    HIP_TRY(h, B.weights.resize(L1));
    HIP_TRY(h, B.X.resize(n));
    HIP_TRY(h, B.dbins.resize(3 * n));
    HIP_TRY(h, hipMemcpy(B.resp.p, resp.data(), sizeof(double) * resp.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.theta0.p, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.Astream.p, As.data(), sizeof(double) * As.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.weights.p, weights, sizeof(double) * L1, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.X.p, X, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(B.dbins.p, bins, sizeof(int32_t) * 3 * n, hipMemcpyHostToDevice));
    B.on = true;
    h->K = 0;
    h->mean.clear(); h->Linv.clear(); h->cnorm.clear(); h->weight.clear();
    h->have_target = true;
    ++h->dir_epoch;
    h->have_state = false;
    return upload_constants(h);
}

int mcmc_hip_get_binned_constants(const mcmc_hip_ctx* h, double* Linv, double* Bc0, double* BJ)
{
    if (!h || !h->bg.on) return MCMC_HIP_ERR_STATE;
    const auto& B = h->bg;
    if (Linv) std::copy(B.Linv.begin(), B.Linv.end(), Linv);
    if (Bc0) std::copy(B.Bc0.begin(), B.Bc0.end(), Bc0);
    if (BJ) std::copy(B.BJ.begin(), B.BJ.end(), BJ);
    return MCMC_HIP_OK;
}

int mcmc_hip_evaluate_binned(mcmc_hip_ctx* h, int32_t n_pts, int32_t L0, int32_t n_ell,
                             const double* cl, const double* A, double* chi2)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->bg.on) return fail(h, MCMC_HIP_ERR_STATE, "set_target_binned_gaussian must precede evaluate_binned");
    auto& B = h->bg;
    if (n_pts <= 0 || !cl || !A || !chi2 || L0 < 0 || n_ell <= 0)
        return fail(h, MCMC_HIP_ERR_ARG, "bad argument");
    for (int b = 0; b < B.n_bins; ++b)
        if (B.bins[3 * b + 1] < L0 || B.bins[3 * b + 2] - L0 >= n_ell)
            return fail(h, MCMC_HIP_ERR_ARG, "bin %d (l = %d..%d) is outside the spectra given (l = %d..%d)",
                        b, B.bins[3 * b + 1], B.bins[3 * b + 2], L0, L0 + n_ell - 1);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t np = ((size_t)n_pts + 63) & ~(size_t)63;
    HIP_TRY(h, B.ecl.resize((size_t)n_pts * 3 * n_ell));
    HIP_TRY(h, B.eA.resize(n_pts));
    HIP_TRY(h, B.echi2.resize(np));
    HIP_TRY(h, B.edelta.resize((np / 64) * (size_t)B.KT * 256 + (size_t)mcmc::kPlPad * 256));
    HIP_TRY(h, hipMemcpyAsync(B.ecl.p, cl, sizeof(double) * (size_t)n_pts * 3 * n_ell,
                              hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(B.eA.p, A, sizeof(double) * n_pts, hipMemcpyHostToDevice, h->stream));
    mcmc::PlBinArgs b{};
    b.cl = B.ecl.p; b.A = B.eA.p; b.bins = B.dbins.p; b.weights = B.weights.p; b.X = B.X.p;
    b.delta = B.edelta.p; b.n_pts = n_pts; b.n_bins = B.n_bins; b.KT = B.KT; b.L0 = L0; b.stride = n_ell;
    HIP_TRY(h, mcmc_hip_launch_pl_bin(&b, h->stream));
    HIP_TRY(h, B.epsum.resize(32 * np));
    const int rc = binned_chi2(h, B.edelta.p, B.epsum.p, B.echi2.p, (int)np);
    if (rc) return rc;
    std::vector<double> c2(np);
    HIP_TRY(h, hipMemcpyAsync(c2.data(), B.echi2.p, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::copy(c2.begin(), c2.begin() + n_pts, chi2);
    g_noted_kernel = nullptr;
    return MCMC_HIP_OK;
}

int mcmc_hip_get_derived_constants(const mcmc_hip_ctx* h, double* uniform_logp, double* mls,
                                   double* Linv, double* cnorm, double* weight)
{
    if (!h || !h->have_prior || !h->have_target) return MCMC_HIP_ERR_STATE;
    if (uniform_logp) *uniform_logp = h->uniform_logp;
    if (mls) std::copy(h->mls.begin(), h->mls.end(), mls);
    if (Linv) std::copy(h->Linv.begin(), h->Linv.end(), Linv);
    if (cnorm) std::copy(h->cnorm.begin(), h->cnorm.end(), cnorm);
    if (weight) std::copy(h->weight.begin(), h->weight.end(), weight);
    return MCMC_HIP_OK;
}

int mcmc_hip_set_blocking(mcmc_hip_ctx* h, int32_t n_blocks, const int32_t* block_size,
                          const int32_t* oversampling, const int32_t* i_of_j,
                          int32_t drag_last_slow, int32_t drag_steps)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!block_size || !oversampling || !i_of_j) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const int d = h->d;
    // (d > 32 from scratch: the general kernels -- step_general_kernel, drag_general_kernel)
    if (n_blocks < 1 || n_blocks > 32)
        return fail(h, MCMC_HIP_ERR_ARG, "n_blocks must be in 1..32, got %d", n_blocks);
    int total = 0;
    bool trivial = n_blocks == 1;
    for (int b = 0; b < n_blocks; ++b) {
        if (block_size[b] < 1) return fail(h, MCMC_HIP_ERR_ARG, "empty parameter block %d", b);
        if (oversampling[b] < 1)   // proposal.py:131-137
            return fail(h, MCMC_HIP_ERR_ARG, "Oversampling factors must be integer! Got %d.",
                        oversampling[b]);
        total += block_size[b];
        trivial = trivial && oversampling[b] == 1;
    }
    std::vector<char> seen(d, 0);
    if (total == d)
        for (int j = 0; j < d; ++j) {
            if (i_of_j[j] < 0 || i_of_j[j] >= d || seen[i_of_j[j]]) { total = -1; break; }
            seen[i_of_j[j]] = 1;
            trivial = trivial && i_of_j[j] == j;
        }
    if (total != d)   // proposal.py:153-156
        return fail(h, MCMC_HIP_ERR_ARG, "The blocks do not contain all the parameter indices.");
    if (drag_last_slow >= 0) {
        if (drag_last_slow > n_blocks - 2)   // proposal.py:143-150: a fast block must remain
            return fail(h, MCMC_HIP_ERR_ARG,
                        "The index given for the last slow block, %d, is not valid: there are "
                        "only %d blocks.", drag_last_slow, n_blocks);
        if (drag_steps < 1 || drag_steps > 256)
            return fail(h, MCMC_HIP_ERR_ARG, "drag_steps must be in 1..256, got %d", drag_steps);
        // (emitted rows, emit_capacity > 0: the from-scratch drag_kernel -- d <= 32 --; the
        // incremental dragging kernel does not emit and refuses at mcmc_hip_step)
        trivial = false;
    } else {
        drag_last_slow = -1;
        drag_steps = 0;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->blocked = !trivial;
    ++h->dir_epoch;
    h->blk_size.assign(block_size, block_size + n_blocks);
    h->blk_over.assign(oversampling, oversampling + n_blocks);
    h->i_of_j.assign(i_of_j, i_of_j + d);
    h->drag_last_slow = drag_last_slow;
    h->drag_steps = drag_steps;
    for (int w = 0; w < 3; ++w)
        if (block_slots(h, w) > 2048)
            return fail(h, MCMC_HIP_ERR_ARG, "a cycle of %d steps exceeds the supported 2048",
                        block_slots(h, w));
    std::vector<int> pack;
    pack.insert(pack.end(), h->blk_size.begin(), h->blk_size.end());
    pack.insert(pack.end(), h->blk_over.begin(), h->blk_over.end());
    pack.insert(pack.end(), h->i_of_j.begin(), h->i_of_j.end());
    HIP_TRY(h, h->dblk.resize(pack.size()));
    HIP_TRY(h, hipMemcpy(h->dblk.p, pack.data(), sizeof(int) * pack.size(), hipMemcpyHostToDevice));
    h->have_cov = false;  // the transform depends on the parameter order
    return MCMC_HIP_OK;
}

int mcmc_hip_cycle_length(const mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    return block_slots(h, h->drag_last_slow >= 0 ? 1 : 0);
}

int mcmc_hip_set_proposal_cov(mcmc_hip_ctx* h, const double* cov)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!cov) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const int d = h->d;
    std::vector<double> corr((size_t)d * d), L((size_t)d * d), sd(d);
    if (!is_symmetric(d, cov))
        return fail(h, MCMC_HIP_ERR_NOT_PD,
                    "The given covmat is not a positive-definite, symmetric square matrix.");
    const std::vector<double> cov_in(cov, cov + (size_t)d * d);
    std::vector<double> sorted_cov;
    if (h->blocked) {  // proposal.py:250-252: reorder by i_of_j before std / corr / Cholesky
        sorted_cov.resize((size_t)d * d);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j)
                sorted_cov[i * d + j] = cov_in[h->i_of_j[i] * d + h->i_of_j[j]];
        cov = sorted_cov.data();
    }
    for (int i = 0; i < d; ++i) {
        if (!(cov[i * d + i] > 0.0) || !std::isfinite(cov[i * d + i]))
            return fail(h, MCMC_HIP_ERR_NOT_PD,
                        "The given covmat is not a positive-definite, symmetric square matrix.");
        sd[i] = std::sqrt(cov[i * d + i]);
    }
    // tools.py:779-788: corr = cov / std / std^T with unit diagonal
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
            corr[i * d + j] = (i == j) ? 1.0 : (1.0 / sd[i]) * cov[i * d + j] * (1.0 / sd[j]);
    if (!cholesky_lower(d, corr.data(), L.data()))
        return fail(h, MCMC_HIP_ERR_NOT_PD,
                    "The given covmat is not a positive-definite, symmetric square matrix.");
    h->cov = cov_in;
    h->T.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) h->T[i * d + j] = h->cfg.proposal_scale * (sd[i] * L[i * d + j]);
    // Stream-ordered, no host synchronisation: launches already queued keep the old transform
    // (their basis kernels precede this copy in the stream), later ones see the new one.  The
    // source is a pinned ring slot that stays untouched for the next three refreshes: before it
    // is written again the host waits for the copy that last read it -- an event recorded four
    // refreshes ago, long complete.  (Rounds 1-5 synchronised the whole stream whenever the ring
    // wrapped: every fourth refresh the host lost its lead of several launches, and the device
    // then idled through the host's pass over the checkpoint -- 275 instead of 96 us between
    // two step kernels, tools/gpu.sh timeline, round 6.)
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int k_slot = h->pin_T_slot;
    double* slot = h->pin_T + (size_t)k_slot * d * d;
    h->pin_T_slot = (h->pin_T_slot + 1) & 3;
    HIP_TRY(h, hipEventSynchronize(h->pin_T_done[k_slot]));   // (never recorded: returns at once)
    std::copy(h->T.begin(), h->T.end(), slot);
    // a direction set being filled ahead on the second stream still reads dT: the copy waits for
    // it (that set is stale after ++dir_epoch and is recomputed, but it must not read a torn T)
    for (auto& D : h->dirs)
        if (D.ahead && D.ready) HIP_TRY(h, hipStreamWaitEvent(h->stream, D.ready, 0));
    HIP_TRY(h, hipMemcpyAsync(h->dT.p, slot, sizeof(double) * d * d, hipMemcpyHostToDevice,
                              h->stream));
    HIP_TRY(h, hipEventRecord(h->pin_T_done[k_slot], h->stream));
    if (h->T_event) {
        HIP_TRY(h, hipEventRecord(h->T_event, h->stream));
        h->T_fresh = true;
    }
    h->have_cov = true;
    ++h->dir_epoch;
    return MCMC_HIP_OK;
}

int mcmc_hip_get_proposal_cov(const mcmc_hip_ctx* h, double* cov)
{
    if (!h || !cov || !h->have_cov) return MCMC_HIP_ERR_STATE;
    std::copy(h->cov.begin(), h->cov.end(), cov);
    return MCMC_HIP_OK;
}

int mcmc_hip_get_proposal_transform(const mcmc_hip_ctx* h, double* T)
{
    if (!h || !T || !h->have_cov) return MCMC_HIP_ERR_STATE;
    std::copy(h->T.begin(), h->T.end(), T);
    return MCMC_HIP_OK;
}

int mcmc_hip_evaluate(mcmc_hip_ctx* h, int32_t n, const double* x, double* logprior,
                      double* loglike, double* derived)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_prior || !h->have_target)
        return fail(h, MCMC_HIP_ERR_STATE, "set_prior and set_target_* must precede evaluate");
    if (n <= 0 || !x || !logprior || !loglike) return fail(h, MCMC_HIP_ERR_ARG, "bad argument");
    const size_t d = h->d, Kd = (size_t)std::max(h->K, 1) * d;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->bg.on) {
        if (derived) return fail(h, MCMC_HIP_ERR_ARG, "the binned Gaussian target has no derived parameters");
        return evaluate_binned_points(h, n, x, logprior, loglike);
    }
    HIP_TRY(h, h->ex.resize((size_t)n * d));
    HIP_TRY(h, h->elp.resize(n));
    HIP_TRY(h, h->ell.resize(n));
    if (derived) HIP_TRY(h, h->eder.resize((size_t)n * Kd));
    HIP_TRY(h, hipMemcpyAsync(h->ex.p, x, sizeof(double) * n * d, hipMemcpyHostToDevice, h->stream));
    mcmc::EvalArgs a{};
    a.x = h->ex.p; a.logprior = h->elp.p; a.loglike = h->ell.p;
    a.derived = (derived && h->K > 0) ? h->eder.p : nullptr;
    a.cblock = h->cblock.p; a.n = n; a.n_modes = h->K;
    a.norm_mask = h->norm_mask; a.periodic_mask = h->periodic_mask;
    a.uniform_logp = h->uniform_logp;
    for (int q = 0; q < 4; ++q) a.norm_mask4[q] = h->norm_mask4[q];
    if (h->kb) {
        HIP_TRY(h, h->escratch.resize((size_t)n * std::max(h->K, 1)));
        HIP_TRY(h, h->kb->evaluate(a, h->dLrow.p, h->d, h->escratch.p, h->stream));
    } else {
        HIP_TRY(h, h->k->evaluate(a, h->stream));
    }
    HIP_TRY(h, hipMemcpyAsync(logprior, h->elp.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(loglike, h->ell.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    if (a.derived)
        HIP_TRY(h, hipMemcpyAsync(derived, h->eder.p, sizeof(double) * n * Kd, hipMemcpyDeviceToHost,
                                  h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MCMC_HIP_OK;
}

int mcmc_hip_set_state(mcmc_hip_ctx* h, const double* x, int32_t* n_bad)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_prior || !h->have_target)
        return fail(h, MCMC_HIP_ERR_STATE, "set_prior and set_target_* must precede set_state");
    if (!x) return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const size_t W = h->W, d = h->d;
    std::vector<double> lp(W), ll(W), lpost(W), xt(W * d);
    int rc = mcmc_hip_evaluate(h, (int)W, x, lp.data(), ll.data(), nullptr);
    if (rc) return rc;
    int bad = 0;
    for (size_t w = 0; w < W; ++w) {
        lpost[w] = lp[w] + ll[w];
        if (!std::isfinite(lpost[w])) ++bad;
        for (size_t i = 0; i < d; ++i) xt[i * W + w] = x[w * d + i];
    }
    if (n_bad) *n_bad = bad;
    if (bad)
        return fail(h, MCMC_HIP_ERR_ARG, "%d initial points have a non-finite log-posterior", bad);
    std::vector<int> ones(W, 1), zeros(W, 0), burn(W, h->cfg.burn_in + 1);  // mcmc.py:265
    std::vector<long long> z64(W, 0);
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->x.p, xt.data(), sizeof(double) * W * d, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->logpost.p, lpost.data(), sizeof(double) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->logprior.p, lp.data(), sizeof(double) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->loglike.p, ll.data(), sizeof(double) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->weight_i.p, ones.data(), sizeof(int) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->prej.p, zeros.data(), sizeof(int) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->burn.p, burn.data(), sizeof(int) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->nacc.p, z64.data(), sizeof(long long) * W, hipMemcpyHostToDevice, s));
    if (h->nrows.p)
        HIP_TRY(h, hipMemcpyAsync(h->nrows.p, zeros.data(), sizeof(int) * W, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemsetAsync(h->stuck.p, 0, sizeof(int), s));
    HIP_TRY(h, hipMemsetAsync(h->acc_total.p, 0, sizeof(unsigned long long), s));
    // a fresh start: no thinning remainders from an earlier run (the oracle's State starts at zero;
    // mcmc_hip_set_full_state leaves them alone -- mcmc_hip_set_thin_carry follows it)
    if (h->thin_acc.p) HIP_TRY(h, hipMemsetAsync(h->thin_acc.p, 0, sizeof(int) * W, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    h->step = 0;
    h->have_state = true;
    h->y_valid = false; h->amode_valid = false;   // incremental mode: y = L^-1 (x - mu) is formed before the next step
    return MCMC_HIP_OK;
}

int mcmc_hip_get_state(mcmc_hip_ctx* h, double* x, double* logpost, double* logprior,
                       double* loglike, int32_t* weight)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state: call set_state first");
    const size_t W = h->W, d = h->d;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    std::vector<double> xt;
    if (x) {
        xt.resize(W * d);
        HIP_TRY(h, hipMemcpyAsync(xt.data(), h->x.p, sizeof(double) * W * d, hipMemcpyDeviceToHost, s));
    }
    if (logpost) HIP_TRY(h, hipMemcpyAsync(logpost, h->logpost.p, sizeof(double) * W, hipMemcpyDeviceToHost, s));
    if (logprior) HIP_TRY(h, hipMemcpyAsync(logprior, h->logprior.p, sizeof(double) * W, hipMemcpyDeviceToHost, s));
    if (loglike) HIP_TRY(h, hipMemcpyAsync(loglike, h->loglike.p, sizeof(double) * W, hipMemcpyDeviceToHost, s));
    if (weight) HIP_TRY(h, hipMemcpyAsync(weight, h->weight_i.p, sizeof(int) * W, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (x)
        for (size_t w = 0; w < W; ++w)
            for (size_t i = 0; i < d; ++i) x[w * d + i] = xt[i * W + w];
    return MCMC_HIP_OK;
}

int mcmc_hip_get_full_state(mcmc_hip_ctx* h, double* x, double* logpost, double* logprior,
                            double* loglike, int32_t* weight, int32_t* prior_rej,
                            int32_t* burn_left, int64_t* n_accept, uint64_t* step)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    int rc = mcmc_hip_get_state(h, x, logpost, logprior, loglike, weight);
    if (rc) return rc;
    const size_t W = h->W;
    if (prior_rej) HIP_TRY(h, hipMemcpy(prior_rej, h->prej.p, sizeof(int) * W, hipMemcpyDeviceToHost));
    if (burn_left) HIP_TRY(h, hipMemcpy(burn_left, h->burn.p, sizeof(int) * W, hipMemcpyDeviceToHost));
    if (n_accept) {
        static_assert(sizeof(long long) == sizeof(int64_t), "n_accept layout");
        HIP_TRY(h, hipMemcpy(n_accept, h->nacc.p, sizeof(int64_t) * W, hipMemcpyDeviceToHost));
    }
    if (step) *step = h->step;
    return MCMC_HIP_OK;
}

int mcmc_hip_set_full_state(mcmc_hip_ctx* h, const double* x, const double* logpost,
                            const double* logprior, const double* loglike, const int32_t* weight,
                            const int32_t* prior_rej, const int32_t* burn_left,
                            const int64_t* n_accept, uint64_t step)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_prior || !h->have_target)
        return fail(h, MCMC_HIP_ERR_STATE, "set_prior and set_target_* must precede set_full_state");
    if (!x || !logpost || !logprior || !loglike || !weight || !prior_rej || !burn_left || !n_accept)
        return fail(h, MCMC_HIP_ERR_ARG, "null argument");
    const size_t W = h->W, d = h->d;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<double> xt(W * d);
    for (size_t w = 0; w < W; ++w) {
        if (!std::isfinite(logpost[w]))
            return fail(h, MCMC_HIP_ERR_ARG, "walker %zu has a non-finite log-posterior", w);
        for (size_t i = 0; i < d; ++i) xt[i * W + w] = x[w * d + i];
    }
    HIP_TRY(h, hipMemcpy(h->x.p, xt.data(), sizeof(double) * W * d, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->logpost.p, logpost, sizeof(double) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->logprior.p, logprior, sizeof(double) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->loglike.p, loglike, sizeof(double) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->weight_i.p, weight, sizeof(int) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->prej.p, prior_rej, sizeof(int) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->burn.p, burn_left, sizeof(int) * W, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->nacc.p, n_accept, sizeof(int64_t) * W, hipMemcpyHostToDevice));
    if (h->nrows.p) HIP_TRY(h, hipMemset(h->nrows.p, 0, sizeof(int) * W));
    HIP_TRY(h, hipMemset(h->stuck.p, 0, sizeof(int)));
    {
        unsigned long long tot = 0;
        for (size_t w = 0; w < W; ++w) tot += (unsigned long long)n_accept[w];
        HIP_TRY(h, hipMemcpy(h->acc_total.p, &tot, sizeof tot, hipMemcpyHostToDevice));
    }
    h->step = step;
    h->have_state = true;
    h->y_valid = false; h->amode_valid = false;   // incremental mode: mcmc_hip_set_whitened must follow (bit-exact resume)
    return MCMC_HIP_OK;
}

namespace {

// fills `V` (and `flag` when the sequence has one-parameter blocks) with the directions of
// cycles [c0, c0 + ncyc) of sequence `which` of the blocked proposer
int blocked_basis(mcmc_hip_ctx* h, int which, unsigned long long c0, int ncyc, int L, size_t slab,
                  DevBuf<double>& V, DevBuf<int>& flag, bool& any_1d, hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const int nb = (int)h->blk_size.size();
    any_1d = false;
    for (int b = 0; b < nb; ++b) {
        const bool in_seq = which == 0 || (which == 1) == (b <= h->drag_last_slow);
        any_1d = any_1d || (in_seq && h->blk_size[b] == 1);
    }
    HIP_TRY(h, V.resize((size_t)h->BG * ncyc * slab));
    if (any_1d) HIP_TRY(h, flag.resize((size_t)h->BG * ncyc * L));
    mcmc::BlockedBasisArgs b{};
    b.T = h->dT.p; b.V = V.p; b.vflag = any_1d ? flag.p : nullptr;
    b.block_size = h->dblk.p; b.oversample = h->dblk.p + nb; b.i_of_j = h->dblk.p + 2 * nb;
    b.n_blocks = nb; b.d = h->d; b.which = which; b.drag_last_slow = h->drag_last_slow;
    b.L = L; b.slab = (int)slab;
    b.ld = h->d;
    b.nmax = *std::max_element(h->blk_size.begin(), h->blk_size.end());
    b.group0 = h->cfg.walker_offset / (uint32_t)h->bgs;   // (bgs == gs outside incremental mode)
    b.cycle0 = (uint32_t)c0;
    b.key0 = (uint32_t)h->cfg.seed; b.key1 = (uint32_t)(h->cfg.seed >> 32);
    b.ncyc = ncyc;
    HIP_TRY(h, mcmc_hip_launch_blocked_basis(&b, h->BG, st));
    return MCMC_HIP_OK;
}


// Does the kernel that serves this engine's incremental steps carry the log-density of every mode
// (round 5)?  step_inc_mix_kernel: 2..4 modes at d <= 64, 5 and 6 at d <= 32 (kernels.h:
// inc_mix_serves), no periodic parameter, Metropolis steps, no emitted rows.
bool inc_carries_modes(const mcmc_hip_ctx* h)
{
    if (!h->incremental || h->K < 2 || h->drag_last_slow >= 0) return false;
    for (int i = 0; i < h->d; ++i)
        if (h->periodic[i]) return false;
    // step_inc_mix_kernel without emitted rows.  Everything else goes to
    // the general kernels (incremental_any.hip), which sum every chi2_k from the trial's residual:
    // the carried form was built for the register-plane kernel as well and measured SLOWER there
    // (K = 5 / 8 / 16 at d = 30: 9.66 -> 8.26, 7.37 -> 6.46, 2.09 -> 1.85e9 evals/s,
    // profiles/r05_carried_modes.txt) -- those kernels wait on latency at one or two waves per
    // SIMD, and the extra K registers cost more than the d / 4 fewer FMAs per mode bought
    return mcmc::inc_mix_serves(h->K, (h->d + 3) / 4) && h->cfg.emit_capacity == 0;
}

// Does the kernel that serves this engine's incremental steps carry the log-prior (round 5)?
// step_inc_kernel (one mode, Metropolis steps; up to 16 periodic parameters without emitted
// rows; with emitted rows: no periodic parameter, no block of one parameter) with some normal
// prior.  The oracle's rule is the same (carries_prior: where the log-likelihood is carried).
bool inc_carries_prior(const mcmc_hip_ctx* h)
{
    if (!h->incremental || h->K != 1 || h->drag_last_slow >= 0) return false;
    if (!(h->norm_mask4[0] | h->norm_mask4[1] | h->norm_mask4[2] | h->norm_mask4[3])) return false;
    int n_periodic = 0;
    for (int i = 0; i < h->d; ++i) n_periodic += h->periodic[i] ? 1 : 0;
    if (n_periodic > mcmc::kIncMaxPeriodic) return false;
    if (h->cfg.emit_capacity > 0) {
        if (n_periodic > 0) return false;
        for (size_t b = 0; h->blocked && b < h->blk_size.size(); ++b)
            if (h->blk_size[b] == 1) return false;
    }
    return true;
}

// mcmc_hip_step in incremental mode (MCMC_HIP_FLAG_INCREMENTAL; incremental_kernels.hip).
// Launches are cut at the multiples of refresh_every = 40 cycle lengths, where y = L^-1 (x - mu)
// is recomputed from x (the specification: oracle/mcmc_oracle.c, orc_run).
// incremental_duo.hip (two lanes per walker) from this ensemble size on.  Measured (same box, d = 30,
// K = 2, step kernel ms per 1200 steps, four lanes / two; profiles/r06_duo.txt): 16 384 walkers 1.17 / 1.41,
// 32 768: 1.57 / 1.49 (K = 3, x in LDS: 1.85 / 2.13), 49 152: 2.06 / 1.77 (K = 3: 3.33 / 2.71; K = 4 at d = 24:
// 2.55 / 2.11), 65 536: 2.91 / 1.85, 98 304: 3.94 / 3.35, 131 072: 5.14 / 3.68 -- two lanes win once the
// four-lane kernel needs a second round of waves (49 152 walkers are its three waves per SIMD)
constexpr int kDuoMinWalkers = 49152;
struct IncPlan {   // what the cutting of launches depends on besides the step counter
    int d, dq, K, nd, chunk_steps, Lc, Lf, ld, max_cyc, max_cyc_f, max_steps_vu;
    size_t colb, dd, ddf;
    unsigned long long R;
    bool drag;
    bool any;   // the general kernel (incremental_any.hip): columns as planes (v, u_1 .. u_K)
    bool carry; // step_inc_kernel (one mode, no periodic parameter, Metropolis steps): the log-likelihood is carried
    bool carry_modes;   // step_inc_mix_kernel: the log-density of every mode is carried
    bool fold;          // step_inc_kernel: the refresh of y is the kernel's, a direction set spans a call
    bool carry_prior;   // step_inc_kernel with normal priors: the log-prior is carried (inc_carries_prior)
};
struct IncSeg {    // one launch: steps [step0, step0 + n)
    unsigned long long step0, c0, cyc0_f;
    int n, ncyc, ncyc_f;
};

IncSeg plan_segment(const IncPlan& P, unsigned long long step, int left)
{
    IncSeg s{};
    const unsigned long long Lc = (unsigned long long)P.Lc;
    s.step0 = step;
    s.c0 = step / Lc;
    unsigned long long room = P.R - step % P.R;
    room = std::min<unsigned long long>(room, (s.c0 + (unsigned long long)P.max_cyc) * Lc - step);
    room = std::min<unsigned long long>(room, (unsigned long long)P.max_steps_vu);
    int n = (int)std::min<unsigned long long>((unsigned long long)left, room);
    if (P.drag) {   // at most max_cyc_f cycles of fast directions per launch
        const unsigned long long und = (unsigned long long)P.nd, uLf = (unsigned long long)P.Lf;
        s.cyc0_f = step * und / uLf;
        const unsigned long long fend = (s.cyc0_f + (unsigned long long)P.max_cyc_f) * uLf;
        const unsigned long long room_f = (fend - step * und) / und;   // whole steps
        n = (int)std::min<unsigned long long>((unsigned long long)n, std::max<unsigned long long>(1, room_f));
        const unsigned long long f1 = (step + (unsigned long long)n) * und - 1;
        s.ncyc_f = (int)(f1 / uLf - s.cyc0_f + 1);
    }
    s.n = n;
    s.ncyc = (int)((step + (unsigned long long)n - 1) / Lc - s.c0 + 1);
    return s;
}

// step_inc_kernel (P.carry): the steps whose directions are formed TOGETHER -- a call's steps as
// far as the direction buffers hold them, NOT cut at the refresh of y: the launches inside (cut
// there by plan_segment) read their columns out of one set and follow each other directly
IncSeg plan_span(const IncPlan& P, unsigned long long step, int left)
{
    if (!P.fold) return plan_segment(P, step, left);
    IncSeg s{};
    const unsigned long long Lc = (unsigned long long)P.Lc;
    s.step0 = step;
    s.c0 = step / Lc;
    unsigned long long room = (s.c0 + (unsigned long long)P.max_cyc) * Lc - step;
    room = std::min<unsigned long long>(room, (unsigned long long)P.max_steps_vu);
    s.n = (int)std::min<unsigned long long>((unsigned long long)left, room);
    s.ncyc = (int)((step + (unsigned long long)s.n - 1) / Lc - s.c0 + 1);
    return s;
}

// fills the set D with the directions of launch `s`, on stream `st`
int make_directions(mcmc_hip_ctx* h, const IncPlan& P, const IncSeg& s, mcmc_hip_ctx::DirSet& D,
                    hipStream_t st)
{
    Timed t(h, 1, st);
    const int nd = P.nd;
    bool any_1d = false, any_1d_f = false;
    if (h->blocked) {
        int rc = blocked_basis(h, P.drag ? 1 : 0, s.c0, s.ncyc, P.Lc, P.dd, D.V, D.vflag, any_1d, st);
        if (rc != MCMC_HIP_OK) return rc;
        if (P.drag) {
            rc = blocked_basis(h, 2, s.cyc0_f, s.ncyc_f, P.Lf, P.ddf, D.Vf, D.vflag_f, any_1d_f, st);
            if (rc != MCMC_HIP_OK) return rc;
        }
    } else {
        HIP_TRY(h, D.V.resize((size_t)h->BG * s.ncyc * P.dd));
        mcmc::BasisArgs b{};
        b.T = h->dT.p; b.V = D.V.p;
        b.group0 = h->cfg.walker_offset / (uint32_t)h->bgs;
        b.cycle0 = (uint32_t)s.c0;
        b.key0 = (uint32_t)h->cfg.seed; b.key1 = (uint32_t)(h->cfg.seed >> 32);
        b.ncyc = s.ncyc;
        if (h->kb) HIP_TRY(h, h->kb->basis(b, h->BG, h->d, st));
        else HIP_TRY(h, h->k->basis(b, h->BG, st));
    }
    HIP_TRY(h, D.VU.resize((size_t)h->BG * s.n * (1 + nd) * P.colb));
    // one-parameter blocks: the columns that draw the RandProposer1D variates, in VU order
    D.has_flags = any_1d || any_1d_f;
    if (D.has_flags) HIP_TRY(h, D.colflag.resize((size_t)h->BG * s.n * (1 + nd)));
    mcmc::IncDirArgs w{};
    w.V = D.V.p; w.Lrow = h->inc_Lrow.p; w.VU = D.VU.p;
    w.step0 = s.step0; w.cycle0 = s.c0; w.n_steps = s.n; w.ncyc = s.ncyc;
    w.slab = (int)P.dd; w.ld = P.ld; w.d = P.d; w.dq = P.dq; w.n_modes = P.K; w.cps = P.Lc;
    w.out_total = s.n * (1 + nd);
    w.colflag = D.has_flags ? D.colflag.p : nullptr;
    w.vflag = any_1d ? D.vflag.p : nullptr;
    if (P.carry || P.carry_modes) {
        HIP_TRY(h, D.UU.resize((size_t)h->BG * s.n * (P.carry_modes ? (size_t)P.K : 1)));
        w.UU = D.UU.p;
    }
    if (P.carry_prior) {
        HIP_TRY(h, D.VW.resize((size_t)h->BG * s.n * 4 * (size_t)P.dq));
        HIP_TRY(h, D.NL.resize((size_t)h->BG * s.n * 2));
        w.prior = h->inc_prior.p; w.VW = D.VW.p; w.NL = D.NL.p;
    }

    if (P.drag) { w.out_div = 1; w.out_cols = 1 + nd; w.out_slot0 = 0; }
    if (P.any) HIP_TRY(h, mcmc_hip_launch_whiten_directions_planes(&w, h->BG, st));
    else HIP_TRY(h, mcmc_hip_launch_whiten_directions(&w, h->BG, st));
    if (P.drag) {   // the fast directions of the n * n_drag interpolation steps
        w.V = D.Vf.p;
        w.step0 = s.step0 * (unsigned long long)nd; w.cycle0 = s.cyc0_f;
        w.n_steps = s.n * nd; w.ncyc = s.ncyc_f; w.slab = (int)P.ddf; w.cps = P.Lf;
        w.out_div = nd; w.out_cols = 1 + nd; w.out_slot0 = 1;
        w.vflag = any_1d_f ? D.vflag_f.p : nullptr;
        HIP_TRY(h, mcmc_hip_launch_whiten_directions(&w, h->BG, st));
    }
    D.step0 = s.step0; D.n = s.n; D.epoch = h->dir_epoch;
    HIP_TRY(h, hipEventRecord(D.ready, st));
    return MCMC_HIP_OK;
}

int step_incremental(mcmc_hip_ctx* h, int n_steps)
{
    const int d = h->d, dq = (d + 3) / 4;
    const int K = h->K;
    IncPlan P{};
    P.d = d; P.dq = dq; P.K = K;
    P.drag = h->drag_last_slow >= 0;
    const int nd = P.nd = P.drag ? h->drag_steps : 0;
    // dragging: a step's 1 + n_drag columns must fit the LDS twice over
    P.chunk_steps = std::max(1, (1024 / (4 * dq)) / (1 + nd));
    const size_t drag_lds = sizeof(double) * 2 * 2 * (size_t)P.chunk_steps * (1 + nd) * 4 * dq;
    int n_periodic = 0;
    for (int i = 0; i < d; ++i) n_periodic += h->periodic[i] ? 1 : 0;
    // what the tuned kernels leave out runs on the general one (incremental_any.hip): more than
    // four modes, mixtures above d = 64, periodic parameters with a mixture, more than 16 of
    // them -- Metropolis steps only
    P.any = !P.drag && ((K > 1 && !mcmc::inc_mix_serves(K, dq)) || (n_periodic > 0 && (K > 1 || n_periodic > mcmc::kIncMaxPeriodic)));
    P.carry = false;   // (set below, once the kernel is chosen)
    if (K < 1 || K > mcmc::kMaxModes || (P.drag && (K > 1 || n_periodic > 0)) ||
        (P.drag && drag_lds > (128u << 10)))
        return fail(h, MCMC_HIP_ERR_ARG,
                    "incremental evaluation with dragging serves one Gaussian mode with "
                    "non-periodic priors; use evaluation: full for this model");
    if (P.any && (!mcmc_hip_launch_inc_any || !mcmc_hip_inc_any_fits ||
                  !mcmc_hip_inc_any_fits(d, K, n_periodic, h->W, h->bgs)))
        return fail(h, MCMC_HIP_ERR_ARG,
                    "incremental evaluation: %d modes at d=%d with %d periodic parameters do not "
                    "fit the LDS of a CU; use evaluation: full for this model", K, d, n_periodic);
    const bool emit = h->cfg.emit_capacity > 0;
    // (round 6: the general incremental kernels thin too -- mixtures, periodic parameters, blocks of
    // one parameter; dragging emits on the from-scratch kernels, which do not)
    if (emit && h->emit_thin > 1 && P.drag)
        return fail(h, MCMC_HIP_ERR_ARG,
                    "emit_thin: rows are thinned on the device by the incremental Metropolis kernels; "
                    "thin on the host");
    if (emit) {
        bool one_d = false;   // (a block of one parameter: its columns draw other variates)
        for (size_t b = 0; h->blocked && b < h->blk_size.size(); ++b) one_d = one_d || h->blk_size[b] == 1;
        if (P.drag)
            return fail(h, MCMC_HIP_ERR_ARG,
                        "incremental evaluation emits rows (emit_capacity > 0) with Metropolis "
                        "steps; use evaluation: full for dragging with emitted rows");
        // step_inc_kernel<.., EMIT> emits for one mode with non-periodic priors and blocks of at
        // least two parameters; every other shape on the general kernels, which emit at run time
        if (K != 1 || n_periodic > 0 || one_d) P.any = true;
        if (P.any && (!mcmc_hip_launch_inc_any || !mcmc_hip_inc_any_fits ||
                      !mcmc_hip_inc_any_fits(d, K, n_periodic, h->W, h->bgs)))
            return fail(h, MCMC_HIP_ERR_ARG,
                        "incremental evaluation: %d modes at d=%d with %d periodic parameters do "
                        "not fit the LDS of a CU; use evaluation: full for this model", K, d, n_periodic);
    }
    // one mode, Metropolis steps: step_inc_kernel / step_inc_periodic_kernel, which carry the
    // log-likelihood along the whitened direction and need |u|^2 of every column
    P.carry = !P.any && !P.drag && K == 1;   // (round 5: with up to 16 periodic parameters too)
    P.fold = P.carry && n_periodic == 0;     // step_inc_kernel: y refreshed in the kernel, sets of several launches
    // ... with normal priors: the log-prior is carried as well (inc_carries_prior says the same to
    // the caller); from d = 113 on its chunks leave no room for the refresh inside the kernel
    P.carry_prior = inc_carries_prior(h);
    if (P.carry_prior && dq >= 29) P.fold = false;
    // mixtures on step_inc_mix_kernel (2..4 modes, d <= 64, no periodic parameter): the log-density
    // of every mode is carried; |u_k|^2 of every column and mode (inc_carries_modes says the same
    // to the caller: the oracle takes the rule from there)
    P.carry_modes = inc_carries_modes(h);
    auto launch = P.any ? mcmc_hip_launch_inc_any
                  : emit ? (dq <= 8 ? mcmc_hip_launch_inc_emit_1 : dq <= 16 ? mcmc_hip_launch_inc_emit_9
                          : dq <= 24 ? mcmc_hip_launch_inc_emit_17 : mcmc_hip_launch_inc_emit_25)
                       : (dq <= 8 ? mcmc_hip_launch_inc_step_1 : dq <= 16 ? mcmc_hip_launch_inc_step_9
                          : dq <= 24 ? mcmc_hip_launch_inc_step_17 : mcmc_hip_launch_inc_step_25);
    // Two lanes per walker (incremental_duo.hip, round 6): two and three modes at d <= 32, four at
    // d <= 24 (kernels.h: duo_serves) with carried mode log-densities, no block of one parameter -- where the ensemble gives every SIMD its two waves
    // of 32 walkers (65 536 walkers per device); smaller ensembles keep the four-lane kernel, whose
    // twice as many waves cover their latencies
    {
        bool duo = h->duo != 0 && !P.any && !emit && !P.drag && n_periodic == 0 && mcmc::duo_serves(K, dq) &&
                   P.carry_modes && h->W % 128 == 0 && h->bgs % 128 == 0 &&
                   (h->duo == 1 || h->W >= kDuoMinWalkers);
        for (size_t b = 0; h->blocked && b < h->blk_size.size() && duo; ++b) duo = h->blk_size[b] != 1;
        auto duo_launch = dq <= 8 ? mcmc_hip_launch_inc_duo_1 : mcmc_hip_launch_inc_duo_9;
        if (duo && duo_launch) launch = duo_launch;
    }
    if (!launch || !mcmc_hip_launch_whiten_directions)
        return fail(h, MCMC_HIP_ERR_DEVICE, "the incremental kernels for d=%d are not linked in", d);
    // columns (= steps) per cycle: d for one block, sum_b oversample_b n_b with blocks, the slow
    // blocks' parameters when dragging (+ the fast sequence of the interpolation steps)
    P.Lc = block_slots(h, P.drag ? 1 : 0);
    P.Lf = P.drag ? block_slots(h, 2) : 0;
    P.R = 40ull * (unsigned long long)P.Lc;
    // doubles per column: (v, u) pairs, or the planes v, u_1 .. u_K of a mixture
    P.colb = ((K == 1 && !P.any) ? 8 : 4 * (size_t)(1 + K)) * (size_t)dq;
    P.max_steps_vu = (int)std::max<size_t>(
        4, ((size_t)512 << 20) / (sizeof(double) * P.colb * (size_t)(1 + nd) * (size_t)h->BG));
    // (blocked directions are written with column stride d at every d)
    P.dd = (h->kb && !h->blocked) ? (size_t)mcmc::v_slab_big(d) : (size_t)mcmc::v_slab_cols(P.Lc, d);
    P.ddf = P.drag ? (size_t)mcmc::v_slab_cols(P.Lf, d) : 0;
    P.ld = (h->kb && !h->blocked) ? mcmc::v_ld(d) : d;
    P.max_cyc = (int)std::max<size_t>(1, (256u << 20) / (sizeof(double) * P.dd * (size_t)h->BG));
    P.max_cyc_f =
        P.drag ? (int)std::max<size_t>(2, (256u << 20) / (sizeof(double) * P.ddf * (size_t)h->BG)) : 0;
    int left = n_steps;
    while (left > 0) {
        // step_inc_kernel (P.fold), round 5 late: a set of directions reaches over SEVERAL calls --
        // `lookahead` calls like this one -- and the calls that find their columns in it start
        // with nothing but the moment snapshot between them and the previous step kernel (the
        // direction kernels are latency-bound: 80 us for one launch's columns at config 2, hardly
        // more for four).  Directions are pure functions of (group, cycle, transform): a set
        // formed under another transform (dir_epoch) is dropped, never used.
        bool covers = false;
        if (P.fold) {
            const auto& C0 = h->dirs[h->dir_cur];
            covers = C0.n > 0 && C0.epoch == h->dir_epoch && C0.step0 <= h->step &&
                     h->step < C0.step0 + (unsigned long long)C0.n;
        }
        // the steps whose directions form one set: one launch (cut at the refresh of y), or --
        // step_inc_kernel -- the calls ahead as far as the buffers hold them
        IncSeg span = plan_span(P, h->step, P.fold ? std::max(left, std::min(h->lookahead, 16) * n_steps) : left);
        auto& D = h->dirs[h->dir_cur];
        if (covers) { span.step0 = D.step0; span.n = D.n; }
        const bool hit = covers ||
            (D.ahead && D.step0 == span.step0 && D.n == span.n && D.epoch == h->dir_epoch);
        // (a set filled ahead on stream2 -- hit or not -- must have been written before it is
        // read or overwritten here)
        if (D.ahead) HIP_TRY(h, hipStreamWaitEvent(h->stream, D.ready, 0));
        D.ahead = false;
        bool wait_ready = false;   // the set is being formed on the second stream
        if (!hit) {
            // Not prepared (the first launch of a call, see below): formed on the SECOND stream
            // behind the previous step kernel (`mark`) -- beside the moment snapshot and the y
            // refresh the main stream still holds, like a set prepared ahead -- and behind the
            // last write of the transform: a proposal refreshed since the previous call is in
            // them at once, nothing stale is computed and thrown away.
            if (h->prefetch && h->lazy_dirs && h->mark_valid && h->stream2) {
                HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->mark, 0));
                if (h->T_fresh) HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->T_event, 0));
                const int rc = make_directions(h, P, span, D, h->stream2);
                if (rc != MCMC_HIP_OK) return rc;
                // (the main stream waits for the set where it needs it: in front of the step
                // kernel, BEHIND the refresh of y -- which does not read the directions and ran
                // 24 us late behind this wait: timeline of round 5, 101 -> 77 us between the
                // step kernels of a call that forms its set)
                wait_ready = true;
            } else {
                const int rc = make_directions(h, P, span, D, h->stream);
                if (rc != MCMC_HIP_OK) return rc;
            }
            h->T_fresh = false;
        }
        // the steps of THIS call the set holds
        const int take = P.fold
            ? (int)std::min<unsigned long long>((unsigned long long)left,
                                                D.step0 + (unsigned long long)D.n - h->step)
            : span.n;
        for (int done = 0; done < take;) {
        bool anchor = false;   // y is refreshed from x before (or, step_inc_kernel: in) this launch
        bool refresh_in_kernel = false;
        if (!h->y_valid || h->step % P.R == 0) {
            if (P.fold && done > 0) {
                // (round 5) a launch INSIDE a call refreshes y itself: nothing stands between it
                // and the launch before.  The first launch of a call keeps the separate kernel:
                // the refresh inside the step kernel -- two barriers and a memory round trip per
                // eight dimensions before the first chunk can be staged -- costs that launch 22 us
                // (timeline: 912 against 890 us), whiten_state_kernel 14 beside the moment
                // snapshot's host gap
                refresh_in_kernel = true;
            } else {
                HIP_TRY(h, mcmc_hip_launch_whiten_state(h->x.p, h->y.p, h->inc_mean.p, h->inc_Lrow.p,
                                                        d, h->W, K, h->stream));
            }
            h->y_valid = true;
            anchor = true;
        }
        // (carried mode log-densities that no launch has written since y was set are re-anchored
        // on y: after set_state always; after a resume only if the state file did not hold them)
        if (P.carry_modes && !h->amode_valid) anchor = true;
        const IncSeg seg = plan_segment(P, h->step, take - done);
        const int n = seg.n;
        if (wait_ready) {
            HIP_TRY(h, hipStreamWaitEvent(h->stream, D.ready, 0));
            wait_ready = false;
        }
        {
            Timed t(h, 0);
            mcmc::IncStepArgs a{};
            a.s.x = h->x.p; a.s.logpost = h->logpost.p; a.s.logprior = h->logprior.p;
            a.s.loglike = h->loglike.p; a.s.weight = h->weight_i.p; a.s.prior_rej = h->prej.p;
            a.s.burn_left = h->burn.p; a.s.n_accept = h->nacc.p; a.s.stuck = h->stuck.p;
            a.s.accept_total = h->acc_total.p;
            a.s.rows = h->rows.p; a.s.n_rows = h->nrows.p; a.s.row_cap = h->cfg.emit_capacity;
            a.s.thin = h->emit_thin; a.s.thin_acc = h->thin_acc.p;
            a.s.W = h->W; a.s.n_modes = K; a.s.group_size = h->bgs;   // the walkers that share a column of VU
            a.s.cblock = h->cblock.p;
            {
                const ConstLayout cl{d, K};
                a.n_modes = K; a.cnorm_off = cl.cnorm(); a.weight_off = cl.weight();
            }
            a.s.walker0 = h->cfg.walker_offset;
            a.s.key0 = (uint32_t)h->cfg.seed; a.s.key1 = (uint32_t)(h->cfg.seed >> 32);
            a.s.step0 = h->step; a.s.n_steps = n;
            a.s.uniform_logp = h->uniform_logp; a.s.temperature = h->cfg.temperature;
            a.s.max_tries = h->cfg.max_tries;
            a.s.cnorm0 = h->cnorm[0];
            a.y = h->y.p; a.VU = D.VU.p; a.prior = h->inc_prior.p;
            a.d = d; a.dq = dq;
            a.has_norm = (h->norm_mask4[0] | h->norm_mask4[1] | h->norm_mask4[2] | h->norm_mask4[3]) != 0u;
            a.box = !a.has_norm;
            for (int i = 1; i < d && a.box; ++i)
                a.box = h->lo[i] == h->lo[0] && h->hi[i] == h->hi[0];
            a.box_lo = h->lo[0]; a.box_hi = h->hi[0];
            a.n_drag = nd; a.chunk_steps = P.chunk_steps;
            a.colflag = D.has_flags ? D.colflag.p : nullptr;
            a.Lrow = h->inc_Lrow.p;
            a.UU = (P.carry || P.carry_modes) ? D.UU.p : nullptr;
            a.anchor = (anchor ? 1 : 0) | (refresh_in_kernel ? 2 : 0);
            a.amode = P.carry_modes ? h->amode.p : nullptr;
            if (P.carry_modes) h->amode_valid = true;
            // (the launch's columns inside the set; 0 / 0: the set is this launch's own)
            a.vu_cols = P.fold ? D.n : 0;
            a.col0 = P.fold ? (int)(h->step - D.step0) : 0;
            a.mean = h->inc_mean.p;
            a.VW = P.carry_prior ? D.VW.p : nullptr;
            a.NL = P.carry_prior ? D.NL.p : nullptr;
            for (int i = 0; i < d; ++i)
                if (h->periodic[i]) a.periodic_mask4[i >> 5] |= 1u << (i & 31);
            HIP_TRY(h, launch(&a, h->stream));
            h->n_step_launches += 1;
            if (g_noted_kernel) {
                h->last_step_kernel = std::string(g_noted_kernel) + " (d=" + std::to_string(d) + ")";
                g_noted_kernel = nullptr;
            }
        }
        h->step += (unsigned long long)n;
        done += n;
        }   // launches of the span
        if (P.fold) {
            // the set is kept while it has columns left; the next one is formed by the call that
            // needs it (see lazy_dirs below), behind this step kernel
            HIP_TRY(h, hipEventRecord(h->mark, h->stream));
            h->mark_valid = true;
            left -= take;
            if (h->step >= D.step0 + (unsigned long long)D.n) {
                if (h->prefetch && h->stream2 && (left > 0 || !h->lazy_dirs)) {
                    auto& N = h->dirs[h->dir_cur ^ 1];
                    const IncSeg nxt = plan_span(
                        P, h->step, std::max(left, std::min(h->lookahead, 16) * n_steps));
                    HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->mark, 0));
                    const int rc = make_directions(h, P, nxt, N, h->stream2);
                    if (rc != MCMC_HIP_OK) return rc;
                    N.ahead = true;
                }
                h->dir_cur ^= 1;
            }
            continue;
        }
        if (h->prefetch) {
            // the launch expected next: the rest of this call, or a call like this one.  Its
            // directions are computed on the second stream BEHIND this step kernel (the event
            // is recorded after it), beside the moment snapshot and the refresh of y that the
            // main stream runs between two step kernels.  Never beside the step kernel: its
            // 1024 workgroups are exactly what the chip holds at once, and a direction kernel
            // that takes a few of those places first -- it happened once in a hundred launches
            // when both became runnable together -- costs the displaced workgroups a second
            // round (1.78 ms instead of 1.04; with the event recorded BEFORE the step kernel
            // d = 64 ran 6.13 ms per launch instead of 4.24, d = 48 and d = 100 unchanged).
            auto& N = h->dirs[h->dir_cur ^ 1];
            const IncSeg nxt = plan_span(P, h->step, left > span.n ? left - span.n : n_steps);
            HIP_TRY(h, hipEventRecord(h->mark, h->stream));
            h->mark_valid = true;
            // (round 4) the launch a LATER call begins with is left to that call (above): the
            // host is a launch ahead of the device, so its directions still run in the same
            // place -- behind this step kernel, beside the main stream's work -- but see a
            // transform that set_proposal_cov / the device checkpoint writes in between.
            // Before, a refreshed proposal made the set prepared here stale and the next call
            // recomputed it on the MAIN stream: 141 us instead of 72 between two step kernels
            // after every learn checkpoint (tools/gpu.sh timeline, round 4).
            if (left > span.n || !h->lazy_dirs) {
                HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->mark, 0));
                const int rc = make_directions(h, P, nxt, N, h->stream2);
                if (rc != MCMC_HIP_OK) return rc;
                N.ahead = true;
            }
        }
        h->dir_cur ^= 1;
        left -= span.n;
    }
    return MCMC_HIP_OK;
}

}  // namespace

int mcmc_hip_step(mcmc_hip_ctx* h, int32_t n_steps)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_state || !h->have_cov)
        return fail(h, MCMC_HIP_ERR_STATE, "set_state and set_proposal_cov must precede step");
    if (n_steps <= 0) return fail(h, MCMC_HIP_ERR_ARG, "n_steps must be > 0");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->bg.on) return step_binned(h, n_steps);
    if (h->incremental) return step_incremental(h, n_steps);
    if (h->emit_thin > 1)
        return fail(h, MCMC_HIP_ERR_ARG, "emit_thin needs incremental evaluation; thin on the host");
    const bool drag = h->drag_last_slow >= 0;
    // steps (= direction columns) per cycle, doubles per (group, cycle) slab of directions
    const int Lc = block_slots(h, drag ? 1 : 0);
    const unsigned long long d = (unsigned long long)Lc;
    // (parameter blocks and dragging at d > 32: blocked directions in the d <= 32 layout --
    // column stride d -- read by the general kernels)
    const bool big_blocked = h->kb && (h->blocked || drag);
    const size_t dd = (h->kb && !big_blocked) ? (size_t)mcmc::v_slab_big(h->d)
                                              : (size_t)mcmc::v_slab_cols(Lc, h->d);
    const bool big_norm = h->norm_mask4[0] || h->norm_mask4[1] || h->norm_mask4[2] || h->norm_mask4[3];
    // d > 32: what the matrix-core / two-wave / column-sweep kernels leave out (mixtures, `one`,
    // periodic parameters, emitted rows; odd ensemble sizes with normal priors or d > 112) runs
    // on the general kernel
    if (h->own_basis && (h->blocked || drag))
        return fail(h, MCMC_HIP_ERR_ARG,
                    "shared_basis: False serves a single parameter block without dragging");
    // (round 6: own bases at d <= 32 run on the tuned step kernel, step_kernel<.., OWN>)
    const bool general_big =
        (h->own_basis && h->kb) || big_blocked ||
        (h->kb && (h->K != 1 || h->any_periodic || h->cfg.emit_capacity > 0 ||
                   (h->W % 256 != 0 && (big_norm || h->d > 112))));
    // basis "groups": the walker groups, or every walker on its own
    const int n_basis = h->own_basis ? h->W : h->G;
    const uint32_t basis0 = h->own_basis ? h->cfg.walker_offset
                                         : h->cfg.walker_offset / (uint32_t)h->gs;
    if (general_big && sizeof(double) * 64 * (size_t)(2 * h->d + std::max(1, h->K)) > (160u << 10))
        return fail(h, MCMC_HIP_ERR_ARG,
                    "d=%d with %d modes does not fit the general d > 32 kernel (LDS)", h->d, h->K);
    // two slabs per group of a workgroup (workgroups are 256, 128 or 64 walkers wide)
    const int wg = (h->W % 256 == 0) ? 256 : (h->W % 128 == 0) ? 128 : 64;
    if (!h->kb && !drag && !h->own_basis &&
        (2 * (size_t)std::max(1, wg / h->gs) * dd + (h->K > 1 ? (size_t)h->K * wg : 0)) * sizeof(double) >
            (160u << 10))
        return fail(h, MCMC_HIP_ERR_ARG,
                    "a cycle of %d steps needs %zu KiB of LDS per group: use group_size 256 or "
                    "smaller oversampling factors", Lc, dd * sizeof(double) / 1024);
    // directions buffer: at most ~256 MiB of cycles per launch
    const int max_cyc = (int)std::max<size_t>(1, (256u << 20) / (sizeof(double) * dd * (size_t)n_basis));
    // dragging: the fast blocks' directions, n_drag columns per step
    const int Lf = drag ? block_slots(h, 2) : 0;
    const size_t ddf = drag ? (size_t)mcmc::v_slab_cols(Lf, h->d) : 0;
    const unsigned long long nd = (unsigned long long)h->drag_steps;
    const int max_cyc_f =
        drag ? (int)std::max<size_t>(2, (256u << 20) / (sizeof(double) * ddf * (size_t)h->G)) : 0;
    int left = n_steps;
    while (left > 0) {
        const unsigned long long c0 = h->step / d;
        const unsigned long long room = (c0 + (unsigned long long)max_cyc) * d - h->step;
        int n = (int)std::min<unsigned long long>((unsigned long long)left, room);
        if (drag) {  // at most max_cyc_f cycles of fast directions per launch
            const unsigned long long fc0 = h->step * nd / (unsigned long long)Lf;
            const unsigned long long fend = (fc0 + (unsigned long long)max_cyc_f) * (unsigned long long)Lf;
            const unsigned long long room_f = (fend - h->step * nd) / nd;   // whole steps
            n = (int)std::min<unsigned long long>((unsigned long long)n, std::max<unsigned long long>(1, room_f));
        }
        const unsigned long long c1 = (h->step + (unsigned long long)n - 1) / d;
        const int ncyc = (int)(c1 - c0 + 1);
        bool any_1d = false;
        {
            Timed t(h, 1);
            if (h->blocked) {
                const int rc = blocked_basis(h, drag ? 1 : 0, c0, ncyc, Lc, dd, h->V, h->vflag, any_1d);
                if (rc != MCMC_HIP_OK) return rc;
            } else {
                HIP_TRY(h, h->V.resize((size_t)n_basis * ncyc * dd));
                mcmc::BasisArgs b{};
                b.T = h->dT.p; b.V = h->V.p;
                b.group0 = basis0;
                b.cycle0 = (uint32_t)c0;
                b.key0 = (uint32_t)h->cfg.seed; b.key1 = (uint32_t)(h->cfg.seed >> 32);
                b.ncyc = ncyc;
                if (h->kb) HIP_TRY(h, h->kb->basis(b, n_basis, h->d, h->stream));
                else HIP_TRY(h, h->k->basis(b, n_basis, h->stream));
            }
        }
        {
            Timed t(h, 0);
            mcmc::StepArgs a{};
            a.x = h->x.p; a.logpost = h->logpost.p; a.logprior = h->logprior.p;
            a.loglike = h->loglike.p; a.weight = h->weight_i.p; a.prior_rej = h->prej.p;
            a.burn_left = h->burn.p; a.n_accept = h->nacc.p; a.stuck = h->stuck.p;
            a.accept_total = h->acc_total.p;
            a.rows = h->rows.p; a.n_rows = h->nrows.p; a.row_cap = h->cfg.emit_capacity;
            a.cblock = h->cblock.p; a.V = h->V.p; a.W = h->W; a.n_modes = h->K;
            a.group_size = h->gs;
            a.norm_mask = h->norm_mask; a.periodic_mask = h->periodic_mask;
            a.walker0 = h->cfg.walker_offset;
            a.key0 = (uint32_t)h->cfg.seed; a.key1 = (uint32_t)(h->cfg.seed >> 32);
            a.step0 = h->step; a.n_steps = n; a.ncyc = ncyc;
            a.uniform_logp = h->uniform_logp; a.temperature = h->cfg.temperature;
            a.max_tries = h->cfg.max_tries;
            a.cnorm0 = h->K > 0 ? h->cnorm[0] : 0.0;
            a.cps = Lc; a.slab = (int)dd;
            a.vflag = any_1d ? h->vflag.p : nullptr;
            a.own_basis = (h->own_basis && !h->kb) ? 1 : 0;
            if (drag) {
                mcmc::DragArgs g{};
                g.s = a;
                const unsigned long long f0 = h->step * nd;
                const unsigned long long f1 = (h->step + (unsigned long long)n) * nd - 1;
                g.cyc0 = c0;
                g.cyc0_f = f0 / (unsigned long long)Lf;
                g.ncyc_f = (int)(f1 / (unsigned long long)Lf - g.cyc0_f + 1);
                g.cps_f = Lf; g.slab_f = (int)ddf; g.n_drag = h->drag_steps;
                bool any_1d_f = false;
                const int rc = blocked_basis(h, 2, g.cyc0_f, g.ncyc_f, Lf, ddf, h->Vf, h->vflag_f,
                                             any_1d_f);
                if (rc != MCMC_HIP_OK) return rc;
                g.Vf = h->Vf.p;
                g.vflag_f = any_1d_f ? h->vflag_f.p : nullptr;
                if (h->kb) {   // 32 < d <= 128: the general dragging kernel
                    mcmc::GeneralDragArgs q{};
                    q.g.s = a;
                    q.g.Lrow = h->dLrow.p; q.g.d = h->d; q.g.ld = h->d; q.g.own_basis = 0;
                    for (int m = 0; m < 4; ++m) q.g.norm_mask4[m] = h->norm_mask4[m];
                    for (int i = 0; i < h->d; ++i)
                        if (h->periodic[i]) q.g.periodic_mask4[i >> 5] |= 1u << (i & 31);
                    HIP_TRY(h, h->drag_cs.resize((size_t)h->d * h->W));
                    q.Vf = g.Vf; q.vflag_f = g.vflag_f; q.cs = h->drag_cs.p;
                    q.cyc0 = g.cyc0; q.cyc0_f = g.cyc0_f; q.cps_f = g.cps_f; q.slab_f = g.slab_f;
                    q.ncyc_f = g.ncyc_f; q.n_drag = g.n_drag;
                    HIP_TRY(h, mcmc_hip_launch_general_drag(&q, h->stream));
                } else {
                    HIP_TRY(h, h->k->drag(g, h->stream));
                }
            } else if (general_big) {
                mcmc::GeneralStepArgs g{};
                g.s = a;
                g.Lrow = h->dLrow.p;
                g.d = h->d;
                g.ld = (h->kb && !big_blocked) ? mcmc::v_ld(h->d) : h->d;
                g.own_basis = h->own_basis ? 1 : 0;
                for (int q = 0; q < 4; ++q) g.norm_mask4[q] = h->norm_mask4[q];
                for (int i = 0; i < h->d; ++i)
                    if (h->periodic[i]) g.periodic_mask4[i >> 5] |= 1u << (i & 31);
                HIP_TRY(h, mcmc_hip_launch_general_step(&g, h->stream));
            } else if (h->kb) {
                a.norm_mask = h->norm_mask4[0];
                a.norm_mask_hi = h->norm_mask4[1];
                if (h->kp && h->kp->fits(a)) HIP_TRY(h, h->kp->step(a, h->stream));
                else HIP_TRY(h, h->kb->step(a, h->dLcol.p, h->d, h->norm_mask4, h->stream));
            }
            else HIP_TRY(h, h->k->step(a, h->gs, h->stream));
            h->n_step_launches += 1;
            if (g_noted_kernel) {
                h->last_step_kernel = std::string(g_noted_kernel) + " (d=" + std::to_string(h->d) + ")";
                g_noted_kernel = nullptr;
            }
        }
        h->step += (unsigned long long)n;
        left -= n;
    }
    return MCMC_HIP_OK;
}

int mcmc_hip_sync(mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2));   // directions computed ahead
    resolve_timing(h);
    int stuck = 0;
    HIP_TRY(h, hipMemcpy(&stuck, h->stuck.p, sizeof(int), hipMemcpyDeviceToHost));
    if (stuck)
        return fail(h, MCMC_HIP_ERR_STUCK,
                    "The chain has been stuck for %g attempts (walker %d), stopping sampling.",
                    h->cfg.max_tries, stuck - 1);
    return MCMC_HIP_OK;
}

int mcmc_hip_get_counters(mcmc_hip_ctx* h, int64_t counters[4])
{
    if (!h || !counters) return MCMC_HIP_ERR_ARG;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long tot = 0;
    HIP_TRY(h, hipMemcpy(&tot, h->acc_total.p, sizeof tot, hipMemcpyDeviceToHost));
    int stuck = 0;
    HIP_TRY(h, hipMemcpy(&stuck, h->stuck.p, sizeof(int), hipMemcpyDeviceToHost));
    int64_t dropped = 0;
    if (h->nrows.p) {
        std::vector<int> nr(h->W);
        HIP_TRY(h, hipMemcpy(nr.data(), h->nrows.p, sizeof(int) * h->W, hipMemcpyDeviceToHost));
        for (auto v : nr) dropped += std::max(0, v - h->cfg.emit_capacity);
    }
    counters[0] = (int64_t)h->step;
    counters[1] = (int64_t)tot;
    counters[2] = stuck;
    counters[3] = dropped;
    return MCMC_HIP_OK;
}

int mcmc_hip_drain_samples(mcmc_hip_ctx* h, double* rows, int64_t cap_rows, int64_t* n_rows)
{
    if (!h || !n_rows) return MCMC_HIP_ERR_ARG;
    *n_rows = 0;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    if (h->cfg.emit_capacity <= 0) return MCMC_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, d = h->d, cap = h->cfg.emit_capacity;
    std::vector<int> nr(W);
    HIP_TRY(h, hipMemcpy(nr.data(), h->nrows.p, sizeof(int) * W, hipMemcpyDeviceToHost));
    int64_t total = 0;
    for (size_t w = 0; w < W; ++w) total += std::min<int>(nr[w], (int)cap);
    *n_rows = total;
    if (!rows) return MCMC_HIP_OK;  // size query
    if (cap_rows < total)
        return fail(h, MCMC_HIP_ERR_ARG, "drain buffer holds %lld rows, %lld are pending",
                    (long long)cap_rows, (long long)total);
    // pack on the device, then move only the rows that exist (they are ~ acceptance x steps
    // of the buffer) straight into the caller's array
    std::vector<long long> off(W);
    long long run = 0;
    for (size_t w = 0; w < W; ++w) { off[w] = run; run += std::min<int>(nr[w], (int)cap); }
    if (total > 0) {
        HIP_TRY(h, h->pack_off.resize(W));
        HIP_TRY(h, h->pack_out.resize(std::min<size_t>(W * cap, (size_t)total + (size_t)total / 4 + 1024) * (d + 5)));
        HIP_TRY(h, hipMemcpyAsync(h->pack_off.p, off.data(), sizeof(long long) * W,
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, mcmc_hip_launch_pack_rows(h->rows.p, h->nrows.p, h->pack_off.p, h->pack_out.p,
                                             (int)W, (int)cap, (int)d, h->cfg.walker_offset,
                                             h->stream));
        HIP_TRY(h, hipMemcpyAsync(rows, h->pack_out.p, sizeof(double) * (size_t)total * (d + 5),
                                  hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    HIP_TRY(h, hipMemset(h->nrows.p, 0, sizeof(int) * W));
    return MCMC_HIP_OK;
}

// Thinned emission on the device (round 5; collection.py:1373-1383, OneSamplePoint.add_to_collection
// with output_thin > 1): every incremental Metropolis kernel that emits rows (round 6) -- the
// from-scratch and dragging kernels refuse at their first step, and the caller thins on the host.
int mcmc_hip_set_emit_thin(mcmc_hip_ctx* h, int32_t thin)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (thin < 1) return fail(h, MCMC_HIP_ERR_ARG, "thin must be >= 1");
    if (thin > 1 && h->cfg.emit_capacity <= 0)
        return fail(h, MCMC_HIP_ERR_ARG, "emit_thin needs emitted rows (emit_capacity > 0)");
    if (thin > 1) {   // (the configuration as it stands now; mcmc_hip_step checks again)
        const bool ok = h->incremental && h->K >= 1 && h->drag_last_slow < 0;
        if (!ok)
            return fail(h, MCMC_HIP_ERR_ARG,
                        "emit_thin: rows are thinned on the device by the incremental kernels (Gaussian "
                        "mixtures with Metropolis steps: step_inc_kernel<.., emit> for one mode, the "
                        "general incremental kernels for mixtures, periodic parameters and blocks of "
                        "one parameter); thin on the host");
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (thin > 1 && !h->thin_acc.p) HIP_TRY(h, h->thin_acc.resize((size_t)h->W));
    // remainders are in units of the factor they were added up under: a new factor starts from zero
    if (thin > 1 && thin != h->emit_thin)
        HIP_TRY(h, hipMemsetAsync(h->thin_acc.p, 0, sizeof(int) * (size_t)h->W, h->stream));
    h->emit_thin = thin;
    return MCMC_HIP_OK;
}

int mcmc_hip_get_thin_carry(mcmc_hip_ctx* h, int32_t* carry)
{
    if (!h || !carry) return MCMC_HIP_ERR_ARG;
    if (!h->thin_acc.p) return fail(h, MCMC_HIP_ERR_STATE, "emit_thin is not set");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(carry, h->thin_acc.p, sizeof(int) * (size_t)h->W, hipMemcpyDeviceToHost));
    return MCMC_HIP_OK;
}

int mcmc_hip_set_thin_carry(mcmc_hip_ctx* h, const int32_t* carry)
{
    if (!h || !carry) return MCMC_HIP_ERR_ARG;
    if (!h->thin_acc.p) return fail(h, MCMC_HIP_ERR_STATE, "emit_thin is not set");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->thin_acc.p, carry, sizeof(int) * (size_t)h->W, hipMemcpyHostToDevice));
    return MCMC_HIP_OK;
}

int mcmc_hip_set_drain_slots(mcmc_hip_ctx* h, int32_t n_slots)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (n_slots < 2 || n_slots > 64) return fail(h, MCMC_HIP_ERR_ARG, "n_slots must be in 2..64");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (auto& sl : h->slots)
        if (sl.p) (void)hipHostFree(sl.p);
    h->slots.assign((size_t)n_slots, mcmc_hip_ctx::HostSlot{});
    h->slot_next = 0;
    return MCMC_HIP_OK;
}

int mcmc_hip_drain_samples_pinned(mcmc_hip_ctx* h, const double** rows, int64_t* n_rows)
{
    if (!h || !rows || !n_rows) return MCMC_HIP_ERR_ARG;
    *rows = nullptr;
    *n_rows = 0;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    if (h->cfg.emit_capacity <= 0) return MCMC_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, d = h->d, cap = h->cfg.emit_capacity;
    std::vector<int> nr(W);
    HIP_TRY(h, hipMemcpy(nr.data(), h->nrows.p, sizeof(int) * W, hipMemcpyDeviceToHost));
    std::vector<long long> off(W);
    long long total = 0;
    for (size_t w = 0; w < W; ++w) { off[w] = total; total += std::min<int>(nr[w], (int)cap); }
    auto& sl = h->slots[(size_t)h->slot_next];
    h->slot_next = (h->slot_next + 1) % (int)h->slots.size();
    if (total > 0) {
        if ((size_t)total > sl.cap_rows) {   // (grown with headroom: pinning memory is slow)
            if (sl.p) (void)hipHostFree(sl.p);
            sl.p = nullptr;
            sl.cap_rows = 0;
            const size_t want = std::min<size_t>(W * cap, (size_t)total + (size_t)total / 4 + 1024);
            HIP_TRY(h, hipHostMalloc((void**)&sl.p, sizeof(double) * want * (d + 5), hipHostMallocDefault));
            sl.cap_rows = want;
        }
        HIP_TRY(h, h->pack_off.resize(W));
        // (the packed rows that exist: ~ acceptance x steps of the device buffer; sized to what
        // is there, with headroom, since the device buffer itself may be many GiB)
        HIP_TRY(h, h->pack_out.resize(std::min<size_t>(W * cap, (size_t)total + (size_t)total / 4 + 1024) * (d + 5)));
        HIP_TRY(h, hipMemcpyAsync(h->pack_off.p, off.data(), sizeof(long long) * W,
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, mcmc_hip_launch_pack_rows(h->rows.p, h->nrows.p, h->pack_off.p, h->pack_out.p,
                                             (int)W, (int)cap, (int)d, h->cfg.walker_offset,
                                             h->stream));
        HIP_TRY(h, hipMemcpyAsync(sl.p, h->pack_out.p, sizeof(double) * (size_t)total * (d + 5),
                                  hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->nrows.p, 0, sizeof(int) * W, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        *rows = sl.p;
    } else {
        HIP_TRY(h, hipMemset(h->nrows.p, 0, sizeof(int) * W));
    }
    *n_rows = total;
    return MCMC_HIP_OK;
}

int mcmc_hip_set_moment_shift(mcmc_hip_ctx* h, const double* shift)
{
    if (!h || !shift) return MCMC_HIP_ERR_ARG;
    if (h->n_snapshots != 0)
        return fail(h, MCMC_HIP_ERR_STATE, "the moment shift can only change right after a reset");
    h->shift.assign(shift, shift + h->d);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpyAsync(h->dshift.p, h->shift.data(), sizeof(double) * h->d,
                              hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MCMC_HIP_OK;
}

int mcmc_hip_accumulate_moments(mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    Timed t(h, 2);
    mcmc::MomentArgs a{};
    a.x = h->x.p; a.shift = h->dshift.p; a.group_sum = h->gsum.p; a.Sg = h->Sg.p;
    a.pooled = h->pooled.p; a.W = h->W; a.G = h->G;
    if (h->kb) HIP_TRY(h, h->kb->moments(a, h->gs, h->d, h->stream));
    else HIP_TRY(h, h->k->moments(a, h->gs, h->stream));
    h->n_snapshots += 1;
    return MCMC_HIP_OK;
}

int mcmc_hip_read_moments(mcmc_hip_ctx* h, int64_t* n_snapshots, double* group_sum,
                          double* pooled_S, int32_t reset)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2));   // (events of the direction kernels)
    resolve_timing(h);
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2;
    if (n_snapshots) *n_snapshots = h->n_snapshots;
    if (group_sum)
        HIP_TRY(h, hipMemcpy(group_sum, h->gsum.p, sizeof(double) * G * d, hipMemcpyDeviceToHost));
    if (pooled_S) {
        std::vector<double> p(np);
        HIP_TRY(h, hipMemcpy(p.data(), h->pooled.p, sizeof(double) * np, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < d; ++i)
            for (size_t j = 0; j <= i; ++j)
                pooled_S[i * d + j] = pooled_S[j * d + i] = p[i * (i + 1) / 2 + j];
    }
    if (reset) {
        HIP_TRY(h, hipMemset(h->gsum.p, 0, sizeof(double) * G * d));
        HIP_TRY(h, hipMemset(h->pooled.p, 0, sizeof(double) * np));
        h->n_snapshots = 0;
    }
    return MCMC_HIP_OK;
}

int mcmc_hip_set_moments(mcmc_hip_ctx* h, int64_t n_snapshots, const double* group_sum,
                         const double* pooled_S)
{
    if (!h || !group_sum || !pooled_S || n_snapshots < 0) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2;
    std::vector<double> p(np);
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j <= i; ++j) p[i * (i + 1) / 2 + j] = pooled_S[i * d + j];
    HIP_TRY(h, hipMemcpy(h->gsum.p, group_sum, sizeof(double) * G * d, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->pooled.p, p.data(), sizeof(double) * np, hipMemcpyHostToDevice));
    h->n_snapshots = n_snapshots;
    return MCMC_HIP_OK;
}

int mcmc_hip_request_moments(mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    if (h->mom_pending) return fail(h, MCMC_HIP_ERR_STATE, "a moment request is already pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2;
    hipStream_t s = h->stream;
    // (gsum, pooled and the accept counter are one block: one copy, one fill)
    HIP_TRY(h, hipMemcpyAsync(h->pin_mom, h->gsum.p, sizeof(double) * (G * d + np + 1),
                              hipMemcpyDeviceToHost, s));
    // the stuck flag travels with the read-out (the spare word behind the accept counter): the
    // run loop never synchronises, so this is where a walker that tripped max_tries is seen
    // (mcmc.py:717-743 stops at once)
    HIP_TRY(h, hipMemcpyAsync(h->pin_mom + G * d + np + 1, h->stuck.p, sizeof(int),
                              hipMemcpyDeviceToHost, s));
    // (with the device-side checkpoint the accumulators are reset by ckpt_window_kernel, which
    // first files them in the ring: mcmc_hip_checkpoint_begin must follow)
    if (!h->ck.ring.p) HIP_TRY(h, hipMemsetAsync(h->gsum.p, 0, sizeof(double) * (G * d + np), s));
    HIP_TRY(h, hipEventRecord(h->mom_event, s));
    h->mom_n = h->n_snapshots;
    h->mom_step = h->step;
    h->n_snapshots = 0;
    h->mom_pending = true;
    return MCMC_HIP_OK;
}

int mcmc_hip_fetch_moments(mcmc_hip_ctx* h, int64_t* n_snapshots, double* group_sum,
                           double* pooled_S, int64_t counters[2])
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->mom_pending) return fail(h, MCMC_HIP_ERR_STATE, "no moment request is pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(h->mom_event));
    h->mom_pending = false;
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2;
    if (n_snapshots) *n_snapshots = h->mom_n;
    if (group_sum) std::copy(h->pin_mom, h->pin_mom + G * d, group_sum);
    if (pooled_S) {
        const double* p = h->pin_mom + G * d;
        for (size_t i = 0; i < d; ++i)
            for (size_t j = 0; j <= i; ++j)
                pooled_S[i * d + j] = pooled_S[j * d + i] = p[i * (i + 1) / 2 + j];
    }
    if (counters) {
        unsigned long long tot;
        std::memcpy(&tot, h->pin_mom + G * d + np, sizeof tot);
        counters[0] = (int64_t)h->mom_step;
        counters[1] = (int64_t)tot;
    }
    int stuck = 0;
    std::memcpy(&stuck, h->pin_mom + G * d + np + 1, sizeof stuck);
    if (stuck)
        return fail(h, MCMC_HIP_ERR_STUCK,
                    "The chain has been stuck for %g attempts (walker %d), stopping sampling.",
                    h->cfg.max_tries, stuck - 1);
    return MCMC_HIP_OK;
}

// ---- the checkpoint on the device -------------------------------------------------------------
int mcmc_hip_checkpoint_set_ring(mcmc_hip_ctx* h, int32_t n_intervals, const double* group_sum,
                                 const double* pooled_S, int32_t min_capacity)
{
    if (!h || n_intervals < 0 || (n_intervals > 0 && (!group_sum || !pooled_S))) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto& K = h->ck;
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2, ne = G * d + np;
    int cap = 16;
    while (cap < std::max(n_intervals + 2, (int)min_capacity)) cap *= 2;
    K.ring.release();
    HIP_TRY(h, K.ring.resize((size_t)cap * ne));
    K.cap = cap;
    K.n_done = n_intervals;      // (slot of interval k of the list = k)
    std::vector<double> buf((size_t)std::max(n_intervals, 1) * ne, 0.0);
    for (int k = 0; k < n_intervals; ++k) {
        double* dst = buf.data() + (size_t)k * ne;
        std::copy(group_sum + (size_t)k * G * d, group_sum + (size_t)(k + 1) * G * d, dst);
        const double* S = pooled_S + (size_t)k * d * d;
        for (size_t i = 0; i < d; ++i)
            for (size_t j = 0; j <= i; ++j) dst[G * d + i * (i + 1) / 2 + j] = S[i * d + j];
    }
    if (n_intervals > 0)
        HIP_TRY(h, hipMemcpy(K.ring.p, buf.data(), sizeof(double) * (size_t)n_intervals * ne,
                             hipMemcpyHostToDevice));
    HIP_TRY(h, K.wsum.resize(ne + G * d));   // window sums | chain means
    HIP_TRY(h, K.payload.resize(5 + 2 * d * d + d));
    HIP_TRY(h, K.ws.resize(7 * d * d + 5 * d + 16));
    HIP_TRY(h, K.out.resize(8 + 2 * d * d));
    if (!K.acc_prev.p) {   // (a reload of the ring keeps the counter of the last checkpoint)
        HIP_TRY(h, K.acc_prev.resize(1));
        HIP_TRY(h, hipMemset(K.acc_prev.p, 0, sizeof(unsigned long long)));
    }
    if (!K.pin_out)
        HIP_TRY(h, hipHostMalloc((void**)&K.pin_out, sizeof(double) * (8 + 2 * d * d + d), hipHostMallocDefault));
    if (!K.ev) HIP_TRY(h, hipEventCreateWithFlags(&K.ev, hipEventDisableTiming));
    return MCMC_HIP_OK;
}

int mcmc_hip_checkpoint_set_accepted(mcmc_hip_ctx* h, int64_t accepted_at_last_checkpoint)
{
    if (!h || !h->ck.acc_prev.p) return MCMC_HIP_ERR_STATE;
    const unsigned long long v = (unsigned long long)accepted_at_last_checkpoint;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipMemcpy(h->ck.acc_prev.p, &v, sizeof v, hipMemcpyHostToDevice));
    return MCMC_HIP_OK;
}

int mcmc_hip_checkpoint_begin(mcmc_hip_ctx* h, int32_t n_window_intervals, int64_t n_window_snapshots,
                              double steps_since, uint64_t* payload_device_ptr, int32_t* payload_len)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    auto& K = h->ck;
    if (!K.ring.p) return fail(h, MCMC_HIP_ERR_STATE, "checkpoint_set_ring must precede checkpoint_begin");
    if (K.begun || K.pending) return fail(h, MCMC_HIP_ERR_STATE, "a device checkpoint is already in flight");
    if (!h->mom_pending)
        return fail(h, MCMC_HIP_ERR_STATE, "request_moments (the read-out of this interval) must precede checkpoint_begin");
    if (n_window_intervals < 1 || n_window_intervals > K.cap || n_window_snapshots < 1)
        return fail(h, MCMC_HIP_ERR_ARG, "the window holds %d intervals (ring capacity %d)", n_window_intervals, K.cap);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t d = h->d, G = h->G, np = d * (d + 1) / 2, ne = G * d + np;
    // (mcmc_hip_request_moments copied the interval out for the host's books and, with a ring,
    // left the accumulators alone: ckpt_window_kernel files them in the ring and resets them)
    mcmc::CkptWindowArgs w{};
    w.acc = h->gsum.p; w.ring = K.ring.p; w.wsum = K.wsum.p; w.n_elem = ne;
    w.means = K.wsum.p + ne; w.n_mean = G * d;
    w.n_per_chain = (double)n_window_snapshots * (double)h->gs;
    w.cap = K.cap; w.slot = (int)(K.n_done % K.cap);
    w.n_slots = n_window_intervals;
    w.first = (int)(((K.n_done - (n_window_intervals - 1)) % K.cap + K.cap) % K.cap);
    HIP_TRY(h, mcmc_hip_launch_ckpt_window(&w, h->stream));
    K.n_done += 1;
    mcmc::CkptPayloadArgs p{};
    p.wsum = K.wsum.p; p.means = K.wsum.p + ne; p.payload = K.payload.p; p.accept_total = h->acc_total.p;
    p.accept_prev = K.acc_prev.p; p.d = (int)d; p.G = (int)G; p.W = h->W;
    p.n_per_chain = (double)n_window_snapshots * (double)h->gs;
    p.steps_since = steps_since;
    HIP_TRY(h, mcmc_hip_launch_ckpt_payload(&p, h->stream));
    if (h->comm) {   // (also a communicator of ONE rank: the same RCCL launch an 8-GPU job queues)
        // ONE all-reduce per checkpoint, in place, in stream order: RCCL over xGMI
        if (int rc = mcmc_comm_allreduce_on_stream(h->comm, K.payload.p, 5 + 2 * d * d + d, 0, h->stream))
            return fail(h, rc, "%s", mcmc_comm_error(h->comm));
    }
    K.begun = true;
    if (payload_device_ptr) *payload_device_ptr = (uint64_t)(uintptr_t)K.payload.p;
    if (payload_len) *payload_len = (int32_t)(5 + 2 * d * d + d);
    return MCMC_HIP_OK;
}

int mcmc_hip_checkpoint_solve(mcmc_hip_ctx* h, double learn_lo, double learn_hi)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    auto& K = h->ck;
    if (!K.begun) return fail(h, MCMC_HIP_ERR_STATE, "checkpoint_begin must precede checkpoint_solve");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t d = h->d;
    // a direction set being filled ahead still reads the transform (as in set_proposal_cov)
    for (auto& D : h->dirs)
        if (D.ahead && D.ready) HIP_TRY(h, hipStreamWaitEvent(h->stream, D.ready, 0));
    mcmc::CkptSolveArgs s{};
    s.payload = K.payload.p; s.ws = K.ws.p; s.out = K.out.p; s.T = h->dT.p;
    s.i_of_j = h->blocked ? h->dblk.p + 2 * (int)h->blk_size.size() : nullptr;
    s.d = (int)d; s.group_size = (double)h->gs; s.learn_lo = learn_lo; s.learn_hi = learn_hi;
    s.proposal_scale = h->cfg.proposal_scale;
    HIP_TRY(h, mcmc_hip_launch_ckpt_solve(&s, h->stream));
    if (h->T_event) {   // (the kernel may have refreshed dT)
        HIP_TRY(h, hipEventRecord(h->T_event, h->stream));
        h->T_fresh = true;
    }
    HIP_TRY(h, hipMemcpyAsync(K.pin_out, K.out.p, sizeof(double) * (8 + 2 * d * d),
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(K.ev, h->stream));
    ++h->dir_epoch;     // the transform may have changed: directions computed ahead are stale
    K.begun = false;
    K.pending = true;
    K.payload_only = false;
    return MCMC_HIP_OK;
}

// The other way to finish a checkpoint begun on the device: only the (all-reduced) payload comes
// back -- 15 KB behind the launch, one event -- and the host solves it (mcmc_hip_gelman_rubin,
// mcmc_hip_set_proposal_cov) while the next launch runs: the window sums and the collective stay
// in stream order on the device, the d^3 work of ONE workgroup leaves the stream.
int mcmc_hip_checkpoint_request_payload(mcmc_hip_ctx* h)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    auto& K = h->ck;
    if (!K.begun) return fail(h, MCMC_HIP_ERR_STATE, "checkpoint_begin must precede checkpoint_request_payload");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t d = h->d;
    HIP_TRY(h, hipMemcpyAsync(K.pin_out, K.payload.p, sizeof(double) * (5 + 2 * d * d + d),
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(K.ev, h->stream));
    K.begun = false;
    K.pending = true;
    K.payload_only = true;
    return MCMC_HIP_OK;
}

int mcmc_hip_checkpoint_fetch_payload(mcmc_hip_ctx* h, double* payload, int32_t n)
{
    if (!h || !payload) return MCMC_HIP_ERR_ARG;
    auto& K = h->ck;
    if (!K.pending || !K.payload_only)
        return fail(h, MCMC_HIP_ERR_STATE, "no payload read-out is pending (checkpoint_request_payload)");
    const size_t d = h->d;
    if ((size_t)n != 5 + 2 * d * d + d)
        return fail(h, MCMC_HIP_ERR_ARG, "the payload holds %zu doubles, not %d", 5 + 2 * d * d + d, n);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(K.ev));
    K.pending = false;
    K.payload_only = false;
    std::copy(K.pin_out, K.pin_out + n, payload);
    return MCMC_HIP_OK;
}

int mcmc_hip_checkpoint_fetch(mcmc_hip_ctx* h, double stats[8], double* mean_of_covs)
{
    if (!h || !stats) return MCMC_HIP_ERR_ARG;
    auto& K = h->ck;
    if (!K.pending || K.payload_only) return fail(h, MCMC_HIP_ERR_STATE, "no device checkpoint is pending");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventSynchronize(K.ev));
    K.pending = false;
    const size_t d = h->d, nn = d * d;
    std::copy(K.pin_out, K.pin_out + 8, stats);
    if (mean_of_covs) std::copy(K.pin_out + 8, K.pin_out + 8 + nn, mean_of_covs);
    if (K.pin_out[2] != 0.0) {    // the proposal was refreshed on the device: mirror it on the host
        h->cov.assign(K.pin_out + 8, K.pin_out + 8 + nn);
        h->T.assign(K.pin_out + 8 + nn, K.pin_out + 8 + 2 * nn);
        h->have_cov = true;
    }
    return MCMC_HIP_OK;
}

// ---- R-1 of the confidence-interval bounds on the device ----------------------------------------
int mcmc_hip_bounds_configure(mcmc_hip_ctx* h, int32_t n_slots)
{
    if (!h || n_slots < 0) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto& B = h->bd;
    const size_t d = h->d, W = h->W, G = h->G;
    B.ring.release();
    B.n_slots = 0;
    if (n_slots == 0) return MCMC_HIP_OK;
    HIP_TRY(h, B.ring.resize((size_t)n_slots * d * W));
    HIP_TRY(h, B.bounds.resize(G * d * 2));
    HIP_TRY(h, B.payload.resize(1 + 4 * d));
    if (!B.pin) HIP_TRY(h, hipHostMalloc((void**)&B.pin, sizeof(double) * (1 + 4 * d + G * d * 2), hipHostMallocDefault));
    B.n_slots = n_slots;
    return MCMC_HIP_OK;
}

int mcmc_hip_bounds_snapshot(mcmc_hip_ctx* h, int32_t slot)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    auto& B = h->bd;
    if (slot < 0 || slot >= B.n_slots) return fail(h, MCMC_HIP_ERR_ARG, "bounds slot %d of %d", slot, B.n_slots);
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)h->d * h->W;
    HIP_TRY(h, hipMemcpyAsync(B.ring.p + (size_t)slot * n, h->x.p, sizeof(double) * n,
                              hipMemcpyDeviceToDevice, h->stream));
    return MCMC_HIP_OK;
}

int mcmc_hip_bounds_get_slot(mcmc_hip_ctx* h, int32_t slot, double* x)
{
    if (!h || !x) return MCMC_HIP_ERR_ARG;
    auto& B = h->bd;
    if (slot < 0 || slot >= B.n_slots) return fail(h, MCMC_HIP_ERR_ARG, "bounds slot %d of %d", slot, B.n_slots);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t d = h->d, W = h->W;
    std::vector<double> t(d * W);
    HIP_TRY(h, hipMemcpy(t.data(), B.ring.p + (size_t)slot * d * W, sizeof(double) * d * W, hipMemcpyDeviceToHost));
    for (size_t w = 0; w < W; ++w)
        for (size_t i = 0; i < d; ++i) x[w * d + i] = t[i * W + w];
    return MCMC_HIP_OK;
}

int mcmc_hip_bounds_set_slot(mcmc_hip_ctx* h, int32_t slot, const double* x)
{
    if (!h || !x) return MCMC_HIP_ERR_ARG;
    auto& B = h->bd;
    if (slot < 0 || slot >= B.n_slots) return fail(h, MCMC_HIP_ERR_ARG, "bounds slot %d of %d", slot, B.n_slots);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t d = h->d, W = h->W;
    std::vector<double> t(d * W);
    for (size_t w = 0; w < W; ++w)
        for (size_t i = 0; i < d; ++i) t[i * W + w] = x[w * d + i];
    HIP_TRY(h, hipMemcpy(B.ring.p + (size_t)slot * d * W, t.data(), sizeof(double) * d * W, hipMemcpyHostToDevice));
    return MCMC_HIP_OK;
}

int mcmc_hip_bounds_statistics(mcmc_hip_ctx* h, int32_t n_window, const int32_t* slots, double limfrac,
                               double* stats, double* bounds)
{
    if (!h || !slots || !stats) return MCMC_HIP_ERR_ARG;
    auto& B = h->bd;
    if (B.n_slots == 0) return fail(h, MCMC_HIP_ERR_STATE, "bounds_configure must precede bounds_statistics");
    if (n_window < 1 || n_window > mcmc::kBoundsMaxSlots || n_window > B.n_slots)
        return fail(h, MCMC_HIP_ERR_ARG, "the window holds %d snapshots (at most %d)", n_window,
                    std::min(mcmc::kBoundsMaxSlots, B.n_slots));
    if (!(limfrac > 0.0 && limfrac < 1.0)) return fail(h, MCMC_HIP_ERR_ARG, "limfrac must lie in (0, 1)");
    const long long n = (long long)n_window * h->gs;
    if ((size_t)n * sizeof(double) > (size_t)mcmc::kBoundsLdsBytes)
        return fail(h, MCMC_HIP_ERR_ARG, "%d snapshots of %d walkers do not fit the LDS of a compute unit", n_window, h->gs);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    mcmc::CkptBoundsArgs a{};
    a.ring = B.ring.p; a.bounds = B.bounds.p; a.n_slots = n_window; a.d = h->d; a.W = h->W; a.gs = h->gs;
    for (int s = 0; s < n_window; ++s) {
        if (slots[s] < 0 || slots[s] >= B.n_slots) return fail(h, MCMC_HIP_ERR_ARG, "bounds slot %d of %d", slots[s], B.n_slots);
        a.slots[s] = slots[s];
    }
    // GetDist's `confidence` (chains.py): index = searchsorted(cumsum(weights), target), capped at
    // n - 1, target = norm * limfrac (lower) | norm * (1 - limfrac) (upper); unit weights:
    // cumsum = 1, 2, ..., n, so the index is ceil(target) - 1
    auto order = [n](double target) {
        long long k = (long long)std::ceil(target) - 1;
        return (int)std::min(std::max(k, 0ll), n - 1);
    };
    a.k_lo = order((double)n * limfrac);
    a.k_hi = order((double)n * (1.0 - limfrac));
    HIP_TRY(h, mcmc_hip_launch_ckpt_bounds(&a, h->G, h->stream));
    mcmc::CkptBoundsReduceArgs r{};
    r.bounds = B.bounds.p; r.shift = h->dshift.p; r.payload = B.payload.p; r.d = h->d; r.G = h->G;
    HIP_TRY(h, mcmc_hip_launch_ckpt_bounds_reduce(&r, h->stream));
    const size_t np_ = 1 + 4 * (size_t)h->d, nb = (size_t)h->G * h->d * 2;
    if (h->comm)     // std over the chains of ALL ranks (mcmc.py:957 `mpi.gather(bound)`): one all-reduce
        if (int rc = mcmc_comm_allreduce_on_stream(h->comm, B.payload.p, np_, 0, h->stream))
            return fail(h, rc, "%s", mcmc_comm_error(h->comm));
    HIP_TRY(h, hipMemcpyAsync(B.pin, B.payload.p, sizeof(double) * np_, hipMemcpyDeviceToHost, h->stream));
    if (bounds)
        HIP_TRY(h, hipMemcpyAsync(B.pin + np_, B.bounds.p, sizeof(double) * nb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::copy(B.pin, B.pin + np_, stats);
    if (bounds) std::copy(B.pin + np_, B.pin + np_ + nb, bounds);
    return MCMC_HIP_OK;
}

int mcmc_hip_set_comm(mcmc_hip_ctx* h, mcmc_hip_comm* c)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (c && mcmc_comm_device(c) != h->cfg.device)
        return fail(h, MCMC_HIP_ERR_ARG, "the communicator lives on device %d, the engine on device %d",
                    mcmc_comm_device(c), h->cfg.device);
    if (h->ck.begun) return fail(h, MCMC_HIP_ERR_STATE, "a device checkpoint is in flight");
    h->comm = c;
    return MCMC_HIP_OK;
}

int mcmc_hip_gelman_rubin(int32_t d, double n_chains, double sum_N, const double* sum_Ncov,
                          const double* sum_mean, const double* sum_mm, double* Rminus1,
                          double* mean_of_covs)
{
    if (d < 1 || !sum_Ncov || !sum_mean || !sum_mm || !Rminus1 || !mean_of_covs)
        return MCMC_HIP_ERR_ARG;
    if (!(n_chains >= 2) || !(sum_N > 0)) return MCMC_HIP_ERR_ARG;
    const size_t n = d;
    std::vector<double> W(n * n), B(n * n), sd(n), nW(n * n), cB(n * n), L(n * n), Li(n * n),
        M(n * n), tmp(n * n), ev(n);
    for (size_t i = 0; i < n * n; ++i) W[i] = mean_of_covs[i] = sum_Ncov[i] / sum_N;  // mcmc.py:856
    // np.cov(means.T): (sum m m^T - n mbar mbar^T) / (n - 1)                        mcmc.py:860
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j)
            B[i * n + j] = (sum_mm[i * n + j] - sum_mean[i] * sum_mean[j] / n_chains) /
                           (n_chains - 1.0);
    for (size_t i = 0; i < n; ++i) {
        if (!(B[i * n + i] > 0.0)) return MCMC_HIP_ERR_NOT_PD;
        sd[i] = std::sqrt(B[i * n + i]);
    }
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            cB[i * n + j] = B[i * n + j] / sd[i] / sd[j];   // mcmc.py:865
            nW[i * n + j] = W[i * n + j] / sd[i] / sd[j];   // mcmc.py:866
        }
    if (!cholesky_lower(d, nW.data(), L.data())) return MCMC_HIP_ERR_NOT_PD;  // mcmc.py:871
    tri_inverse_lower(d, L.data(), Li.data());
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (size_t k = 0; k < n; ++k) s += Li[i * n + k] * cB[k * n + j];
            tmp[i * n + j] = s;
        }
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (size_t k = 0; k < n; ++k) s += tmp[i * n + k] * Li[j * n + k];
            M[i * n + j] = s;
        }
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < i; ++j) M[i * n + j] = M[j * n + i] = 0.5 * (M[i * n + j] + M[j * n + i]);
    if (!symmetric_eigenvalues(d, M.data(), ev.data())) return MCMC_HIP_ERR_NOT_PD;  // mcmc.py:881-887
    double r = 0.0;
    for (size_t i = 0; i < n; ++i) r = std::max(r, std::fabs(ev[i]));
    if (!std::isfinite(r)) return MCMC_HIP_ERR_NOT_PD;
    *Rminus1 = r;  // mcmc.py:889
    return MCMC_HIP_OK;
}

int mcmc_hip_get_whitened(mcmc_hip_ctx* h, double* y)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->incremental || !y) return fail(h, MCMC_HIP_ERR_ARG, "not in incremental mode, or null");
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "no state");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!h->y_valid) {
        HIP_TRY(h, mcmc_hip_launch_whiten_state(h->x.p, h->y.p, h->inc_mean.p, h->inc_Lrow.p, h->d,
                                                h->W, h->K, h->stream));
        h->y_valid = true;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, d = (size_t)h->d * (size_t)h->K;   // device: [K d][W]
    std::vector<double> yt(W * d);
    HIP_TRY(h, hipMemcpy(yt.data(), h->y.p, sizeof(double) * W * d, hipMemcpyDeviceToHost));
    for (size_t w = 0; w < W; ++w)
        for (size_t i = 0; i < d; ++i) y[w * d + i] = yt[i * W + w];
    return MCMC_HIP_OK;
}

int mcmc_hip_set_whitened(mcmc_hip_ctx* h, const double* y)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!h->incremental || !y) return fail(h, MCMC_HIP_ERR_ARG, "not in incremental mode, or null");
    if (!h->have_state) return fail(h, MCMC_HIP_ERR_STATE, "set_full_state must precede set_whitened");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, d = (size_t)h->d * (size_t)h->K;
    std::vector<double> yt(W * d);
    for (size_t w = 0; w < W; ++w)
        for (size_t i = 0; i < d; ++i) yt[i * W + w] = y[w * d + i];
    HIP_TRY(h, hipMemcpy(h->y.p, yt.data(), sizeof(double) * W * d, hipMemcpyHostToDevice));
    h->y_valid = true;
    return MCMC_HIP_OK;
}

int mcmc_hip_incremental_carries_periodic(const mcmc_hip_ctx* h)
{
    if (!h || !h->incremental || h->K != 1 || h->drag_last_slow >= 0 || h->cfg.emit_capacity > 0)
        return 0;
    int n = 0;
    for (int i = 0; i < h->d; ++i) n += h->periodic[i] ? 1 : 0;
    return n >= 1 && n <= mcmc::kIncMaxPeriodic ? 1 : 0;
}

int mcmc_hip_incremental_carries_modes(const mcmc_hip_ctx* h)
{
    return h && inc_carries_modes(h) ? 1 : 0;
}

int mcmc_hip_get_mode_logdensities(mcmc_hip_ctx* h, double* a)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!a || !inc_carries_modes(h))
        return fail(h, MCMC_HIP_ERR_ARG, "this engine does not carry mode log-densities, or null");
    if (!h->have_state || !h->amode_valid)
        return fail(h, MCMC_HIP_ERR_STATE, "no carried mode log-densities yet (a step forms them)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, K = (size_t)h->K;
    std::vector<double> t(W * K);
    HIP_TRY(h, hipMemcpy(t.data(), h->amode.p, sizeof(double) * W * K, hipMemcpyDeviceToHost));
    for (size_t w = 0; w < W; ++w)
        for (size_t k = 0; k < K; ++k) a[w * K + k] = t[k * W + w];
    return MCMC_HIP_OK;
}

int mcmc_hip_set_mode_logdensities(mcmc_hip_ctx* h, const double* a)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    if (!a || !inc_carries_modes(h))
        return fail(h, MCMC_HIP_ERR_ARG, "this engine does not carry mode log-densities, or null");
    if (!h->have_state || !h->y_valid)
        return fail(h, MCMC_HIP_ERR_STATE, "set_full_state and set_whitened must precede set_mode_logdensities");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t W = h->W, K = (size_t)h->K;
    std::vector<double> t(W * K);
    for (size_t w = 0; w < W; ++w)
        for (size_t k = 0; k < K; ++k) t[k * W + w] = a[w * K + k];
    HIP_TRY(h, hipMemcpy(h->amode.p, t.data(), sizeof(double) * W * K, hipMemcpyHostToDevice));
    h->amode_valid = true;
    return MCMC_HIP_OK;
}

void mcmc_hip_note_step_kernel(const char* name) { g_noted_kernel = name; }

const char* mcmc_hip_last_step_kernel(const mcmc_hip_ctx* h)
{
    return h ? h->last_step_kernel.c_str() : "";
}

int mcmc_hip_enable_timing(mcmc_hip_ctx* h, int32_t on)
{
    if (!h) return MCMC_HIP_ERR_ARG;
    h->timing = on != 0;
    return MCMC_HIP_OK;
}

uint64_t mcmc_hip_stream_handle(const mcmc_hip_ctx* h) { return h ? (uint64_t)(uintptr_t)h->stream : 0; }

int mcmc_hip_kernel_times(mcmc_hip_ctx* h, double ms[3], int64_t* n_step_launches, int32_t reset)
{
    if (!h || !ms) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2));   // (events of the direction kernels)
    resolve_timing(h);
    for (int i = 0; i < 3; ++i)   // (kinds 1 and 2 are sampled: scaled to all their regions)
        ms[i] = h->n_timed[i] > 0 ? h->ms[i] * ((double)h->n_seen[i] / (double)h->n_timed[i]) : 0.0;
    if (h->bg.on) ms[0] = h->ms[3] + h->ms[4] + h->ms[5];   // the three kernels of a step
    if (n_step_launches) *n_step_launches = h->n_step_launches;
    if (reset) {
        for (int i = 0; i < 6; ++i) { h->ms[i] = 0.0; h->n_seen[i] = h->n_timed[i] = 0; }
        h->n_step_launches = 0;
    }
    return MCMC_HIP_OK;
}

int mcmc_hip_binned_kernel_times(mcmc_hip_ctx* h, double ms[3], int64_t n_launches[3], int32_t reset)
{
    if (!h || !ms || !n_launches) return MCMC_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    resolve_timing(h);
    for (int i = 0; i < 3; ++i) {
        ms[i] = h->ms[3 + i];
        n_launches[i] = h->n_timed[3 + i];
        if (reset) { h->ms[3 + i] = 0.0; h->n_seen[3 + i] = h->n_timed[3 + i] = 0; }
    }
    return MCMC_HIP_OK;
}

}  // extern "C"
