// Kernel arguments of pliklite_kernels.hip (shared with capi.hip).
#pragma once
#include "kernels.h"

namespace mcmc {

constexpr int kPlPad = 8;   // k-steps of padding behind delta and the tile streams (>= the prefetch depth)

// ---- binned-bandpower Gaussian likelihood (pliklite_kernels.hip; planck_pliklite.py:143-155)
struct PlWalkerArgs {
    StepArgs s;            // state, prior constants (cblock, ConstLayout{d, 0}), V, keys;
                           // s.step0 = the step PROPOSED by this launch, s.ncyc cycles in V
    int d;
    int cyc, col;          // cycle (relative to the first one held in V) and column of that step
    double* trial;         // [d][W] trial points
    double* lp_t;          // [W] log-prior of the trial, -inf outside the support
    double* Ea;            // [W] Exp(1) variate of the accept test
    const double* psum_t;  // [32][W] partial sums of the trial's chi2 (pl_chi2_kernel)
};
struct PlResidualArgs {
    const double* trial;   // [d][W]
    const double* theta0;  // [nlp] fiducial parameters (zero beyond n_lin)
    const double* resp;    // [n_bins][nlp + 2] records (Bc0_b, BJ_b0 .. BJ_b,nlp-1, X_b)
    double* delta;         // [W / 64][KT / 2][4][64][2]: B-operand order of pl_chi2_kernel, k-steps in pairs
    int W, n_bins, KT, n_lin, nlp, calib;
};
// pl_residual_mfma_kernel: the same residuals as a [bins x (n_lin padded)] . [params x walkers]
// product on the matrix cores
struct PlResidualMfmaArgs {
    const double* trial;   // [d][W]
    const double* theta0;  // [>= 8 np] fiducial parameters (zero beyond n_lin)
    const double* bjs;     // [tiles][np][64][2]: A operands of bin tile T, k-step pair jp: lane l holds
                           // BJ[16 T + (l & 15)][4 (2 jp + e) + (l >> 4)], e = 0, 1 (zero outside)
    const double* es;      // [tiles][4][64][2]: per lane 16 c + n the rows 16 T + 4 r + c: (Bc0 r = 0, 1),
                           // (Bc0 r = 2, 3), (X r = 0, 1), (X r = 2, 3)
    double* delta;         // as PlResidualArgs
    int W, KT, n_lin, np, calib, n_tiles;
};
// pl_fused_kernel: residuals AND the triangular product in one launch; delta never leaves the CU
constexpr int kPlChunkPairs = 16;   // k-step pairs (= 8 bin tiles = 128 bins) per LDS chunk
// LDS of a workgroup: two chunks of residuals [pairs][4 walker tiles][64 lanes] x 16 B, then the
// producers' (dtheta, 1 / A^2) of the set: [4 waves][np + 1][64 lanes] x 16 B (np <= 4)
constexpr size_t kPlFusedChunkBytes = 2 * (size_t)kPlChunkPairs * 256 * 16;
constexpr size_t kPlFusedLdsBytes = kPlFusedChunkBytes + 4 * 5 * 1024;
struct PlFusedArgs {
    const double* trial;   // [d][W]
    const double* theta0;  // [32]
    const double* bjs;     // as PlResidualMfmaArgs
    const double* es;
    const double* Astream; // half-tile (wave q, group G, half h) at a_off[q][G][h]: [pairs][64][2] doubles in
                           // A-operand lane order from the tile's FIRST k-step pair, + padding
    double* psum;          // [8][4][n_walkers]: p[pos][c] of every walker
    unsigned long long a_off[8][5][2];
    int a_pairs[8][5][2];  // VIRTUAL pair count of the half-tile (0: absent): pairs [2 shift, a_pairs)
    int W, KT, n_lin, np, calib, n_tiles;   // n_tiles: real bin tiles (ceil(KT / 4))
    int shift;             // virtual tile index = real + shift (the tile groups end at the LAST tile)
    int ng;                // groups of 8 virtual tiles (= chunks)
    int n_sets, batches;
};
struct PlBinArgs {
    const double* cl;      // [n_pts][3][stride], element l - L0 of a row is D_l
    const double* A;       // [n_pts] calibration
    const int* bins;       // [n_bins][3] (spectrum, first l, last l)
    const double* weights; // [lmax + 1]
    const double* X;       // [n_bins]
    double* delta;         // as PlResidualArgs
    int n_pts, n_bins, KT, L0, stride;
};
struct PlChi2Args {
    const double* delta;   // [W / 64][KT / 2][4][64][2] (KT even: the k-steps 2 m, 2 m + 1 of a lane
                           // side by side) + kPlPad k-steps (256 doubles each) of padding
    const double* Astream; // tile t of wave q at tile_off[q][t]: [nk[q][t] / 2][64][2] doubles,
                           // k-step kk of lane l = L^-1[16 R + (l & 15)][4 kk + (l >> 4)]; kPlPad
                           // k-steps of padding behind the last tile (operands are fetched ahead)
    double* psum;          // [8][4][n_walkers]: p[q][c] of every walker (chains of its chi2)
    unsigned long long tile_off[8][5];   // (absent tiles: any valid offset)
    int nk[8][5];          // k-steps of tile t of wave q: ascending in t, absent tiles first (0)
    int KT, ntw;
    int n_walkers, n_sets; // walkers = 64 n_sets
    int batches;           // (set by the launcher) sets of 64 walkers per workgroup
};

}  // namespace mcmc
