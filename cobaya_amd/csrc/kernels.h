// Host/device shared declarations for the mcmc_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmc {

constexpr int kMaxDimLane = 32;   // lane-per-walker kernels: d <= 32 (state in VGPRs)
constexpr int kMaxDimPair = 56;   // ... and the two-wave step kernel alone up to here
constexpr int kMaxModes = 64;   // (16 until round 5; the tuned incremental kernels: 16, incremental_any.hip)

// Whitening factor L_k^-1 (lower triangular) packed in the order the kernels consume it, so
// that the wave-uniform operand stream is read front to back with wide scalar loads:
//   for jb = 0, RB, 2RB, ... (row blocks of RB rows)
//     for i = 0 .. min(jb + RB, d) - 1
//       for r = 0 .. RB-1 with jb + r < d and i <= jb + r:   L[jb + r][i]
constexpr int kRowBlock = 4;
__host__ __device__ constexpr int tri_size(int d) { return d * (d + 1) / 2; }

template <typename F>
__host__ __device__ inline void tri_stream_for_each(int d, F&& f)
{
    int idx = 0;
    for (int jb = 0; jb < d; jb += kRowBlock)
        for (int i = 0; i < jb + kRowBlock && i < d; ++i)
            for (int r = 0; r < kRowBlock; ++r)
                if (jb + r < d && i <= jb + r) f(idx++, jb + r, i);
}

// Doubles per (group, cycle) slab of proposal directions V[col][i]: d*d rounded up to whole
// KiB so that the step kernel can move a slab into LDS with 1 KiB global->LDS DMA pieces.
// With parameter blocks a cycle has `ncols` = sum_b oversample_b * n_b columns instead of d.
__host__ __device__ constexpr int v_slab_cols(int ncols, int d)
{
    return ((ncols * d * 8 + 1023) / 1024) * 128;
}
__host__ __device__ constexpr int v_slab(int d) { return v_slab_cols(d, d); }

// d > 32 ("big" kernels): columns of V are padded to an even number of doubles (16-byte aligned
// for the per-step 1 KiB global->LDS DMA), and a slab keeps >= 1 KiB behind its last column.
__host__ __device__ constexpr int v_ld(int d) { return (d + 1) & ~1; }
__host__ __device__ constexpr int v_slab_big(int d)
{
    return ((d * v_ld(d) * 8 + 1024 + 1023) / 1024) * 128;
}

// Constant block (doubles) in HBM, read through the scalar data cache:
//   lo[d] hi[d] loc[d] scale[d] mls[d] | elem[d][3] = {lo_i, hi_i, mean0_i} (the interleaved
//   stream of the fused proposal pass) | per mode k: mean[d] | cnorm[K] weight[K] |
//   per mode k: Linv stream (tri_size(d))
struct ConstLayout {
    int d, K;
    __host__ __device__ int lo() const { return 0; }
    __host__ __device__ int hi() const { return d; }
    __host__ __device__ int loc() const { return 2 * d; }
    __host__ __device__ int scale() const { return 3 * d; }
    __host__ __device__ int mls() const { return 4 * d; }
    __host__ __device__ int elem() const { return 5 * d; }
    __host__ __device__ int mean(int k) const { return 8 * d + k * d; }
    __host__ __device__ int cnorm() const { return 8 * d + K * d; }
    __host__ __device__ int weight() const { return cnorm() + K; }
    __host__ __device__ int linv(int k) const
    {
        return ((weight() + K + 7) & ~7) + k * ((tri_size(d) + 7) & ~7);
    }
    __host__ __device__ int size() const { return linv(K); }
};

struct StepArgs {
    // walker state (HBM): x is dimension-major [d][W] so that lane w reads x[i*W + w]
    double* x;
    double* logpost;
    double* logprior;
    double* loglike;
    int* weight;
    int* prior_rej;
    int* burn_left;
    long long* n_accept;
    unsigned long long* accept_total;  // [1] accepted steps of all walkers (one atomic per wave)
    int* stuck;
    // optional emission of accepted rows: rows[W][row_cap][d+4], n_rows[W]
    double* rows;
    int* n_rows;
    int row_cap;
    // thinned emission (round 5; OneSamplePoint.add_to_collection with output_thin > 1,
    // collection.py:1373-1383; step_inc_kernel<.., EMIT> only): thin <= 1: off; else the weights of
    // a walker accumulate in thin_acc[W] and a row is written when the sum reaches `thin`, with
    // weight sum / thin, the remainder carried
    int thin;
    int* thin_acc;
    // problem
    const double* cblock;
    const double* V;  // [G][ncyc][v_slab(d)] direction vectors of the cycles this launch spans
    int W;
    int group_size;
    int n_modes;
    uint32_t norm_mask, periodic_mask;
    uint32_t walker0;
    uint32_t key0, key1;
    unsigned long long step0;
    int n_steps;
    int ncyc;
    double uniform_logp, temperature, max_tries;
    double cnorm0;  // d log 2pi + log|S_0| of mode 0 (kernarg copy for the hot variant)
    // cycle geometry: columns (= steps) per cycle and doubles per (group, cycle) slab of V;
    // d and v_slab(d) for one block, sum_b oversample_b n_b with parameter blocks
    int cps;
    int slab;
    // [G][ncyc][cps] 1 where the column belongs to a one-parameter block (its step draws the
    // RandProposer1D variates, proposal.py:85-93), or null
    const int* vflag;
    uint32_t norm_mask_hi;   // dimensions 32..63 (the two-wave kernel of 32 < d <= 56)
    // MCMC_HIP_FLAG_OWN_BASIS at d <= 32 (round 6, step_kernel<.., OWN>): V is [W][ncyc][slab] -- every
    // walker reads the columns of its OWN Haar basis (proposal.py:59-69 to the letter) from HBM
    int own_basis;
};

// Directions of the blocked proposer (blocked_kernels.hip; any d <= 32).
struct BlockedBasisArgs {
    const double* T;        // [d*d] transform of the covariance in SORTED order
    double* V;              // [G][ncyc][slab]
    int* vflag;             // [G][ncyc][L] or null
    const int* block_size;  // [n_blocks]
    const int* oversample;  // [n_blocks]
    const int* i_of_j;      // [d]
    int n_blocks, d;
    int which;              // 0 all blocks with oversampling; 1 slow; 2 fast (dragging)
    int drag_last_slow;
    int L;                  // slots per cycle of this sequence
    int slab;
    uint32_t group0, cycle0, key0, key1;
    int ncyc;
    int ld;                 // column stride of V (d <= 32: d)
    int nmax;               // largest block (the d > 32 kernel sizes its LDS by it)
};

struct BasisArgs {
    const double* T;  // [d*d] row-major lower-triangular proposal transform (scale folded in)
    double* V;        // [G][ncyc][v_slab(d)]
    uint32_t group0;
    uint32_t cycle0;
    uint32_t key0, key1;
    int ncyc;
};

struct EvalArgs {
    const double* x;  // [n][d] point-major
    double* logprior;
    double* loglike;
    double* derived;  // [n][K*d] or null
    const double* cblock;
    int n;
    int n_modes;
    uint32_t norm_mask, periodic_mask;
    uint32_t norm_mask4[4];  // d > 32: one bit per dimension
    double uniform_logp;
};

struct MomentArgs {
    const double* x;    // [d][W]
    const double* shift;  // [d] subtracted before accumulating (conditioning)
    double* group_sum;  // [G][d]   accumulated
    double* Sg;         // [G][npairs] scratch (this call)
    double* pooled;     // [npairs] accumulated (lower triangle, i>=j, index i(i+1)/2+j)
    int W;
    int G;
};

// The dragging step (mcmc.py:564-668): `s` carries the state, the slow directions (V, cps,
// slab, ncyc, vflag; cycles counted from cyc0) and everything else of a Metropolis launch.
struct DragArgs {
    StepArgs s;
    const double* Vf;      // [G][ncyc_f][slab_f] directions of the fast blocks
    const int* vflag_f;    // [G][ncyc_f][cps_f] or null
    unsigned long long cyc0, cyc0_f;   // first slow / fast cycle held in V / Vf
    int cps_f, slab_f, ncyc_f;
    int n_drag;            // interpolation steps per dragging step
};

// Per-dimension launchers (one translation unit per d, see walker_kernels.hip).
struct DimKernels {
    hipError_t (*step)(const StepArgs&, int group_size, hipStream_t);
    hipError_t (*basis)(const BasisArgs&, int n_groups, hipStream_t);
    hipError_t (*evaluate)(const EvalArgs&, hipStream_t);
    hipError_t (*moments)(const MomentArgs&, int group_size, hipStream_t);
    hipError_t (*drag)(const DragArgs&, hipStream_t);
};

// The two-wave step kernel of one dimension 32 < d <= kMaxDimPair (walker_kernels.hip compiled
// for that d; it reads the d > 32 layout of V).  `fits`: the launch geometry of `a` is one the
// kernel serves (whole 256-walker workgroups, LDS) -- otherwise the caller uses BigKernels.step.
struct PairKernels {
    bool (*fits)(const StepArgs&);
    hipError_t (*step)(const StepArgs&, hipStream_t);
};

// The general step of 32 < d <= 128 (general_kernels.hip): mixtures, `one`, periodic parameters,
// emitted rows, and ensembles the specialised kernels do not take.
struct GeneralStepArgs {
    StepArgs s;
    const double* Lrow;           // [K][d][d] row-major L^-1
    int d;
    uint32_t norm_mask4[4];       // one bit per dimension with a normal prior
    uint32_t periodic_mask4[4];   // ... with a periodic parameter
    int ld;                       // column stride of V: d (d <= 32 layout) or v_ld(d)
    int own_basis;                // every walker has its own (group, cycle) slabs of V
};

// The general dragging step (general_kernels.hip: drag_general_kernel)
struct GeneralDragArgs {
    GeneralStepArgs g;     // g.s: state, V = the slow blocks' directions (cps, slab, vflag), ...
    const double* Vf;      // [G][ncyc_f][slab_f] directions of the fast blocks
    const int* vflag_f;    // [G][ncyc_f][cps_f] or null
    double* cs;            // [d][W] scratch: the start points of the dragging step
    unsigned long long cyc0, cyc0_f;   // first slow / fast cycle held in V / Vf
    int cps_f, slab_f, ncyc_f;
    int n_drag;
};

// step_inc_mix_kernel (carried mode log-densities): 2..4 modes up to d = 64, 5 and 6 as far as
// the state -- dq (K + 1) doubles per lane -- leaves the body its registers at two waves per SIMD
// (K = 5: d <= 32, K = 6: d <= 28; measured at d = 30, ms per 1200 steps of 65 536 walkers:
// K = 5 7.71 on the register-plane kernel -> 5.84 here; K = 6 at dq = 8 spills: 13.3 against ~8.6);
// everything else: the general incremental kernels
constexpr int kIncMixWideDq = 8;
__host__ __device__ constexpr bool inc_mix_serves(int K, int dq)
{
    return K >= 2 && ((K <= 4 && dq <= 16) || (K <= 6 && dq <= kIncMixWideDq && dq * (K + 1) <= 50));
}

// step_duo_mix_kernel (incremental_duo.hip, round 6): the same step with TWO lanes per walker, each
// holding 2 dq dimensions -- 2..4 modes while the residuals y_1 .. y_K of a lane (2 dq K doubles) leave
// the body its registers at two waves per SIMD; x moves to LDS where it does not fit beside them
// (duo_x_in_lds: two modes from d = 33 on, three from d = 25 on, four from d = 21 on).  K = 2: d <= 48;
// K = 3: d <= 32; K = 4: d <= 24.
constexpr int kDuoStateDoubles = 48;
#ifndef MCMC_DUO_MAX_DQ
#define MCMC_DUO_MAX_DQ 12
#endif
constexpr int kDuoMaxDq = MCMC_DUO_MAX_DQ;   // two modes up to d = 48 (round 6, late)
__host__ __device__ constexpr bool duo_serves(int K, int dq)
{
    return K >= 2 && K <= 4 && dq <= kDuoMaxDq && 2 * dq * K <= kDuoStateDoubles;
}
// x, y_1 .. y_K of a lane in registers up to 50 doubles, x in LDS above -- measured (65 536 walkers, ms per
// 40 d steps, x in LDS / in registers): K = 4 at d = 20 (50 doubles) 1.64 / 1.53; K = 3 at d = 28 (56) 2.43 / 2.90
#ifndef MCMC_DUO_XLDS_ABOVE
#define MCMC_DUO_XLDS_ABOVE 50   // (experiment hook)
#endif
__host__ __device__ constexpr bool duo_x_in_lds(int K, int dq) { return 2 * dq * (K + 1) > MCMC_DUO_XLDS_ABOVE; }

// periodic parameters step_inc_kernel<.., PER> serves (one mode, Metropolis steps, no emitted rows);
// more: the general incremental kernels (incremental_any.hip)
constexpr int kIncMaxPeriodic = 16;

// Incremental evaluation (incremental_kernels.hip): one Gaussian mode, non-periodic priors,
// one block; 2 <= d <= 128 with dq = ceil(d / 4) dimensions per lane, four lanes per walker.
struct IncStepArgs {
    StepArgs s;            // state, keys, step0, n_steps, uniform_logp, temperature, cnorm0, ...
    double* y;             // [d][W] whitened residual L^-1 (x - mu) of the current points
    const double* VU;      // [G][n_steps][dq][4][2]: (v_i, u_i) of every step, zero beyond d
    const double* prior;   // [5][4 dq]: lo, hi, loc, 1/scale, mls; a dimension without a normal
                           // prior has 1/scale = 0 and mls = 0; beyond d: -inf, +inf, 0, 0, 0
    int d, dq;
    int has_norm;          // some prior is normal
    int box;               // every prior is uniform on the same interval [box_lo, box_hi]
    double box_lo, box_hi;
    // mixtures (2..4 modes, dq <= 16): y is [K][d][W], VU holds PLANES [G][n_steps][1 + K][4 dq]
    // (v, u_1 .. u_K), and cnorm[k] / weight[k] sit at these offsets of s.cblock
    int n_modes, cnorm_off, weight_off;
    // dragging (drag_inc_kernel): interpolation steps per dragging step (0: plain steps); VU then
    // holds 1 + n_drag columns per step; chunk_steps dragging steps per LDS chunk
    int n_drag, chunk_steps;
    // one-parameter blocks: colflag[G][columns of the launch] (written with VU) marks the columns
    // whose step draws the RandProposer1D variates (proposal.py:85-93); null if no block has one
    const int* colflag;
    // periodic parameters (step_inc_periodic_kernel): bit i of periodic_mask4[i / 32] marks
    // dimension i; Lrow = L^-1 row-major [d][d] (a wrap moves the residual by a column of it)
    unsigned periodic_mask4[4];
    const double* Lrow;
    // step_inc_kernel (one mode, no periodic parameter): |u|^2 of the launch's columns
    // [G][n_steps], and whether this launch starts where y has just been refreshed from x -- the
    // carried log-likelihood is then re-anchored on y (oracle: orc_anchor_loglike)
    const double* UU;
    int anchor;
    // mixtures with carried mode log-densities (round 5: step_inc_mix_kernel, step_inc_regs_kernel
    // without periodic parameters; oracle: carries_modes): amode[K][W] = -(c_k + chi2_k) / 2 of the
    // current point per mode, UU then holds |u_k|^2 as [G][n_steps][K]; re-anchored with `anchor`
    double* amode;
    // step_inc_kernel (round 5): a direction set may cover SEVERAL launches of one call (the steps
    // of a call are cut at the refresh of y; their directions are formed together, so that the
    // launches follow each other without the direction kernels in between): the launch reads
    // the columns [col0, col0 + s.n_steps) of a set of vu_cols columns per group (0: the set is
    // this launch's own, vu_cols = s.n_steps).  anchor & 2: y is refreshed from x IN the kernel
    // (whiten_state_kernel's arithmetic: one ascending fma chain per row from +0.0, orc_whiten)
    // before the carried log-likelihood is re-anchored on it; mean = the mode's mean [d]
    int vu_cols, col0;
    const double* mean;
    // step_inc_kernel with normal priors (round 5, `carried log-prior`; oracle: carries_prior): the
    // log-prior moves along the direction like the log-likelihood,
    //     lp(x + r v) = fma(-r / 2, fma(r, v.w, 2 (x.w - loc.w)), lp(x)),   w_i = (v_i / s_i) / s_i,
    // with VW[G][n_steps][4 dq] = w of every column (zero beyond d and where no normal prior is)
    // and NL[G][n_steps][2] = (v.w, loc.w) in the four-chain pattern; re-anchored with `anchor`
    const double* VW;
    const double* NL;
};

struct IncDirArgs {
    const double* V;       // the basis kernels' buffer [G][ncyc][slab], column stride ld
    const double* Lrow;    // [d][d] row-major L^-1
    double* VU;            // [G][n_steps][dq][4][2]; mixtures: [G][n_steps][1 + K][4 dq]
    unsigned long long step0, cycle0;   // first step of the launch, first cycle held in V
    int n_steps, ncyc, slab, ld, d, dq;
    int n_modes;           // Lrow is [K][d][d]
    int cps;               // columns (= steps) per cycle: d, or sum_b oversample_b n_b with blocks
    // where column sr goes: out_div == 0: column sr of out_total; else column
    // (sr / out_div) * out_cols + out_slot0 + sr % out_div (dragging: slow / fast interleaved)
    int out_div, out_cols, out_slot0, out_total;
    // one-parameter blocks: the basis kernel's flags [G][ncyc][cps] of this sequence (null: none)
    // are copied to colflag[G][out_total] (null: nothing to write) at the columns' places in VU
    const int* vflag;
    int* colflag;
    // |u|^2 of every column, [G][out_total], in the four-chain pattern of chi2 (null: not wanted):
    // step_inc_kernel carries the log-likelihood along the direction (oracle: orc_direction_norms);
    // mixtures (the plane kernels): [G][out_total][K], |u_k|^2 of every mode
    double* UU;
    // carried log-prior (IncStepArgs::VW, NL; null: not wanted): prior = the engine's table
    // [5][4 dq] (lo, hi, loc, 1/scale, mls)
    const double* prior;
    double* VW;
    double* NL;
};

// Launchers of the d > 32 kernels (walker_kernels_big.hip, one TU per accumulator count).
struct BigKernels {
    int dp;  // largest dimension this instantiation serves
    hipError_t (*step)(const StepArgs&, const double* Lcol, int d, const uint32_t* norm_mask4,
                       hipStream_t);
    hipError_t (*basis)(const BasisArgs&, int n_groups, int d, hipStream_t);
    hipError_t (*evaluate)(const EvalArgs&, const double* Lrow, int d, double* scratch, hipStream_t);
    hipError_t (*moments)(const MomentArgs&, int group_size, int d, hipStream_t);
    int n_tiles;  // 16 x 4 tiles of L^-1 the matrix-core step kernel reads (after Lcol's dp*dp)
    int row_shift;  // row tile R of those holds the rows 16 R - row_shift .. + 15 (a multiple of 4)
};

}  // namespace mcmc

// Every step launcher names the kernel it is about to launch (a string literal); mcmc_hip_step
// keeps the last one for mcmc_hip_last_step_kernel, so that reports quote the kernel that ran
// instead of guessing it from the problem shape.
extern "C" void mcmc_hip_note_step_kernel(const char* name);

#define MCMC_DECLARE_BIG(DP) extern "C" const mcmc::BigKernels* mcmc_hip_big_##DP() __attribute__((weak));
#define MCMC_DECLARE_PAIR(D) extern "C" const mcmc::PairKernels* mcmc_hip_pair_##D() __attribute__((weak));
#define MCMC_DECLARE_DIM(D) extern "C" const mcmc::DimKernels* mcmc_hip_dim_##D() __attribute__((weak));
