// Kernel arguments of checkpoint_kernels.hip (shared with capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmc {

struct CkptWindowArgs {
    double* acc;        // [n_elem] interval accumulators: group sums [G][d] | pooled S (lower triangle)
    double* ring;       // [cap][n_elem] the checkpoint intervals of the run
    double* wsum;       // [n_elem] sums over the window
    double* means;      // [n_mean] group sums of the window / n_per_chain (the chain means)
    size_t n_elem, n_mean;
    double n_per_chain; // snapshots of the window x group_size
    int slot;           // ring slot this interval goes to
    int first, n_slots; // the window: slots first .. first + n_slots - 1 (mod cap), ending at `slot`
    int cap;
};

struct CkptPayloadArgs {
    const double* wsum;            // window sums (CkptWindowArgs::wsum)
    const double* means;           // chain means (CkptWindowArgs::means)
    double* payload;               // [5 + 2 d^2 + d]
    const unsigned long long* accept_total;
    unsigned long long* accept_prev;   // accepted steps at the previous checkpoint (updated)
    int d, G, W;
    double n_per_chain;            // snapshots of the window x group_size
    double steps_since;            // Metropolis steps per walker since the previous checkpoint
};

struct CkptSolveArgs {
    const double* payload;         // (all-reduced over the ranks)
    double* ws;                    // workspace: 7 d^2 + 5 d doubles
    double* out;                   // [8 + 2 d^2]: see ckpt_solve_kernel
    double* T;                     // the proposal transform in force, refreshed in place
    const int* i_of_j;             // parameter order of the blocked proposer, or null
    int d;
    double group_size;             // R-1 is quoted per walker: x group_size (sampler.py)
    double learn_lo, learn_hi;     // refresh iff learn_lo <= R-1 x group_size <= learn_hi
    double proposal_scale;
};

// R-1 of the confidence-interval bounds (mcmc.py:918-1002): ckpt_bounds_kernel, ckpt_bounds_reduce_kernel
constexpr int kBoundsMaxSlots = 64;          // snapshots one window may hold
constexpr int kBoundsLdsBytes = 128 * 1024;  // the keys of one (chain, parameter) live in LDS

struct CkptBoundsArgs {
    const double* ring;            // [n_ring_slots][d][W] snapshots of the ensemble (x, dimension-major)
    double* bounds;                // [G][d][2] lower / upper bound of every chain (= walker group)
    int slots[kBoundsMaxSlots];    // the window's ring slots, oldest first
    int n_slots;
    int d, W, gs;
    int k_lo, k_hi;                // 0-based order statistics of the n_slots * gs samples
};

struct CkptBoundsReduceArgs {
    const double* bounds;          // [G][d][2]
    const double* shift;           // [d] subtracted before squaring (conditioning only)
    double* payload;               // [1 + 4 d]: chains | sum lo | sum hi | sum lo^2 | sum hi^2 (shifted)
    int d, G;
};

}  // namespace mcmc
