/*
 * mcmc_hip.h -- C ABI of libmcmc_hip.so, the MI355X (gfx950) walker-ensemble Metropolis
 * engine behind Cobaya's sampler plugin surface.
 *
 * The reference (CobayaSampler/cobaya v3.6.2) is pure Python and has no FFI on this path;
 * each entry point below names the reference interface it stands in for (paths relative to
 * the reference checkout).  The Python class cobaya_amd.sampler.MCMCHip (registered as
 * sampler `mcmc_hip`) is the only caller: initialize() -> create/set_*, run() -> loop of
 * step/accumulate_moments/read_moments/gelman_rubin/set_proposal_cov/drain_samples,
 * products() -> get_state/drain_samples.  See INTEGRATION.md for the ctypes binding.
 *
 * Conventions: opaque handle; all buffers are caller-owned C-contiguous host arrays
 * (the library copies in/out and never retains a pointer); every function returns 0 on
 * success and a negative code on error, with a message available from
 * mcmc_hip_last_error(); nothing throws across the boundary; a handle is not thread-safe;
 * all calls may be made with the GIL released.  There is NO CPU fallback: create() fails
 * if no gfx950 device is usable.
 */
#ifndef MCMC_HIP_H
#define MCMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcmc_hip_ctx mcmc_hip_ctx;

/* The library is built with -fvisibility=hidden: exactly the functions declared in this header
 * are exported (tests/test_host_logic.py compares `nm -D` with it). */
#define MCMC_HIP_API __attribute__((visibility("default")))

enum {
    MCMC_HIP_OK = 0,
    MCMC_HIP_ERR_ARG = -1,       /* invalid argument / unsupported configuration */
    MCMC_HIP_ERR_DEVICE = -2,    /* HIP runtime error or no usable gfx950 device */
    MCMC_HIP_ERR_NOT_PD = -3,    /* matrix not symmetric positive definite */
    MCMC_HIP_ERR_STATE = -4,     /* call order violated (e.g. step before set_state) */
    MCMC_HIP_ERR_STUCK = -5      /* a walker exceeded max_tries (mcmc.py:717-743) */
};

/* Options fixed at creation.  Mirrors the attributes cobaya/samplers/mcmc/mcmc.py:111-271
 * (MCMC.initialize) reads from mcmc.yaml, plus the ensemble geometry. */
typedef struct mcmc_hip_config {
    int32_t d;               /* number of sampled parameters, 1..128 (mcmc_hip_dim_supported) */
    int32_t n_walkers;       /* walkers on this device; multiple of group_size */
    int32_t group_size;      /* walkers sharing one Haar basis: 64, 128 or 256 */
    int32_t device;          /* HIP device ordinal */
    uint64_t seed;           /* mcmc.yaml:75 `seed` (Philox key; sampler.py:369-384) */
    uint32_t walker_offset;  /* global id of this device's first walker (multi-GPU shard) */
    int32_t burn_in;         /* accepted steps discarded per walker (mcmc.py:265, 691-707) */
    double temperature;      /* mcmc.yaml:24 (mcmc.py:127-130, 438-440, 682) */
    double proposal_scale;   /* mcmc.yaml:17 (proposal.py:223) */
    double max_tries;        /* mcmc.yaml:9, already multiplied by d (mcmc.py:717-743) */
    int32_t emit_capacity;   /* accepted rows kept per walker between drains; 0 = none */
    int32_t flags;           /* MCMC_HIP_FLAG_* (0: the default ensemble) */
} mcmc_hip_config;

/* every walker draws its OWN Haar basis per cycle (proposal.py:59-69 to the letter) instead
 * of sharing the group's: the reference-faithful control, much slower */
#define MCMC_HIP_FLAG_OWN_BASIS 1
/* incremental evaluation (Gaussian mixtures of up to 64 modes with uniform / normal priors and
 * any number of periodic parameters, parameter blocks of any size and oversampling, Metropolis
 * steps -- mcmc_hip_incremental_supported says whether a shape fits; dragging for one mode with
 * non-periodic priors; 2 <= d <= 128; emitted rows, emit_capacity > 0, with Metropolis steps):
 * every walker carries y_k = L_k^-1 (x - mu_k) and a trial moves it along the whitened shared
 * direction, y_k' = y_k + r L_k^-1 v -- the same log-posterior (gaussian_mixture.py:158-163) in
 * O(d) per step and mode; y is recomputed from x every 40 cycle lengths (40 d steps for one
 * block).  With ONE mode and no periodic parameter the log-likelihood is carried as well:
 * loglike' = loglike - r/2 (2 y.u + r |u|^2), u = L^-1 v, re-anchored on y wherever y is
 * recomputed.  What the mode does not serve is refused by mcmc_hip_step with MCMC_HIP_ERR_ARG.
 * Specified in oracle/mcmc_oracle.c. */
#define MCMC_HIP_FLAG_INCREMENTAL 2
/* incremental mode: the walkers that share one Haar basis = group_size << ((flags >> 8) & 15)
 * (the R-1 groups of the moments stay group_size wide): fewer bases and whitened columns per
 * launch */
#define MCMC_HIP_FLAG_BASIS_GROUP_MASK 0x0F00

MCMC_HIP_API const char* mcmc_hip_version(void);
/* message of the last error on this handle (or of the last failed create if h == NULL) */
MCMC_HIP_API const char* mcmc_hip_last_error(const mcmc_hip_ctx* h);
/* 1 if the lane-per-walker kernels for dimension d were compiled into this library */
MCMC_HIP_API int mcmc_hip_dim_supported(int d);
/* 1 if MCMC_HIP_FLAG_INCREMENTAL serves a Gaussian mixture of n_modes (>= 1) modes in d dimensions
 * of which n_periodic are periodic (prior.py:658-676), with n_drag interpolation steps per dragging
 * step (0: Metropolis steps), for n_walkers walkers of which basis_group_size share a proposal
 * direction: the tuned kernels (one mode; up to four at d <= 64, five and six at d <= 32 / 28; up to 16 periodic parameters
 * of one mode; dragging of one non-periodic mode) or the general one (anything else without
 * dragging whose per-walker state -- n_modes * d doubles -- fits the LDS of a CU).  0: such a model
 * is sampled from scratch (no flag).  A pure function: no device is touched. */
MCMC_HIP_API int mcmc_hip_incremental_supported(int32_t d, int32_t n_modes, int32_t n_periodic, int32_t n_drag,
                                   int32_t n_walkers, int32_t basis_group_size);

/* Sampler.__init__ + MCMC.initialize (cobaya/sampler.py:257-322, mcmc.py:111-271) */
MCMC_HIP_API int mcmc_hip_create(const mcmc_hip_config* cfg, mcmc_hip_ctx** out);
MCMC_HIP_API void mcmc_hip_destroy(mcmc_hip_ctx* h);

/* Prior.__init__ constants (cobaya/prior.py:464-533): kind[i] 0 = uniform on [a,b],
 * 1 = normal(loc=a, scale=b); periodic[i] != 0 wraps into [a,b) (prior.py:658-676). */
MCMC_HIP_API int mcmc_hip_set_prior(mcmc_hip_ctx* h, const int32_t* kind, const double* a, const double* b,
                       const int32_t* periodic);

/* GaussianMixture.initialize_with_params (gaussian_mixture.py:45-136): means[K*d],
 * covs[K*d*d] row-major, weights[K] (NULL = equal; renormalised if they do not sum to 1). */
MCMC_HIP_API int mcmc_hip_set_target_gaussian_mixture(mcmc_hip_ctx* h, int32_t n_modes, const double* means,
                                         const double* covs, const double* weights);
/* Gaussian.initialize_with_params (gaussian/gaussian.py:30-94) */
MCMC_HIP_API int mcmc_hip_set_target_gaussian(mcmc_hip_ctx* h, const double* mean, const double* cov,
                                 int32_t normalized);
/* likelihoods/one/one.py:27-29: loglike = 0 (prior-only sampling) */
MCMC_HIP_API int mcmc_hip_set_target_one(mcmc_hip_ctx* h);

/* PlanckPlikLite.init_params (cobaya/likelihoods/base_classes/planck_pliklite.py:32-141) for the
 * used bins, + the Cl provider of `logp` (planck_pliklite.py:170-178) as a LINEAR emulator:
 *   bins[n_bins*3] = (spectrum 0 tt / 1 te / 2 ee, first l, last l) of every used bin in
 *   data-vector order (`used_indices`); weights[lmax+1] in D_l space (planck_pliklite.py:52-56);
 *   X[n_bins] = X_data; cov[n_bins*n_bins] row-major (the reference inverts it, :141; here
 *   chi2 = |L^-1 delta|^2 with cov = L L^T, the same quadratic form);
 *   D_l(theta) = D0[tp*(lmax+1)+l] + sum_p J[(tp*(lmax+1)+l)*n_lin+p] (theta_p - theta0_p), theta =
 *   the sampled parameters without the calibration parameter (`calibration_param`, A_planck) at
 *   position calib_index, in order; n_lin = d - 1.
 * loglike = -chi2/2, chi2 = get_chi_squared(0, D_tt, D_te, D_ee, A) (planck_pliklite.py:143-155).
 * Binning commutes with the linear emulator: the library bins D0 and J once and evaluates
 * cl_b(theta) from that binned response (specification: oracle/mcmc_oracle.c, orc_binned).
 * n_bins <= 640, 2 <= d <= 32, evaluation from scratch (no MCMC_HIP_FLAG_INCREMENTAL: the
 * posterior is not Gaussian in the calibration parameter), shared basis, emit_capacity 0. */
MCMC_HIP_API int mcmc_hip_set_target_binned_gaussian(mcmc_hip_ctx* h, int32_t n_bins, const int32_t* bins,
                                        int32_t lmax, const double* weights, const double* X,
                                        const double* cov, int32_t n_lin, const double* theta0,
                                        const double* D0, const double* J, int32_t calib_index);
/* PlanckPlikLite.get_chi_squared(L0, ctt, cte, cee, A_planck) (planck_pliklite.py:143-155) for
 * n_pts sets of EXPLICIT spectra: cl[n_pts*3*n_ell], element l - L0 of row (pt, spectrum) is
 * D_l; A[n_pts]; chi2[n_pts] out.  Binning on the device as the banded product it is, then the
 * same matrix-core kernel as the sampler's steps. */
MCMC_HIP_API int mcmc_hip_evaluate_binned(mcmc_hip_ctx* h, int32_t n_pts, int32_t L0, int32_t n_ell,
                             const double* cl, const double* A, double* chi2);
/* Constants derived by set_target_binned_gaussian: Linv[n_bins*n_bins] (functions.py:81-89 of
 * cov), the binned response Bc0[n_bins], BJ[n_bins*n_lin]; any pointer may be NULL.  Lets tests
 * hand the CPU oracle exactly the problem the kernels evaluate. */
MCMC_HIP_API int mcmc_hip_get_binned_constants(const mcmc_hip_ctx* h, double* Linv, double* Bc0, double* BJ);

/* BlockedProposer.__init__ (proposal.py:96-196) + MCMC.set_proposer_blocking (mcmc.py:320-410):
 * n_blocks parameter blocks sorted slow -> fast, block b holding block_size[b] consecutive
 * entries of i_of_j[d] (sampler indices in sorted order) and visited oversampling[b] *
 * block_size[b] times per cycle.  drag_last_slow >= 0 switches mcmc_hip_step to the dragging
 * step (mcmc.py:564-668) with blocks 0..drag_last_slow slow and drag_steps interpolation
 * steps; -1 keeps Metropolis steps.  From scratch: the tuned kernels for d <= 32, the general
 * ones (step_general_kernel, drag_general_kernel) for 32 < d <= 128; with emit_capacity > 0 a
 * dragging step emits the point it leaves (mcmc.py:656-668) from the from-scratch kernels.  Must
 * precede set_proposal_cov (a previous covariance is forgotten).  One block with factor 1 and the identity order is the default. */
MCMC_HIP_API int mcmc_hip_set_blocking(mcmc_hip_ctx* h, int32_t n_blocks, const int32_t* block_size,
                          const int32_t* oversampling, const int32_t* i_of_j,
                          int32_t drag_last_slow, int32_t drag_steps);
/* steps per cycle: d for one block, sum_b oversampling_b n_b with blocks, the number of slow
 * parameters when dragging (mcmc.py:400-407) */
MCMC_HIP_API int mcmc_hip_cycle_length(const mcmc_hip_ctx* h);

/* BlockedProposer.set_covariance (proposal.py:226-260); with blocks the covariance is
 * reordered by i_of_j first and get_proposal_transform returns T in that sorted order:
 * checks symmetric positive definite, builds T = scale * diag(std) * chol(corr).  `cov`
 * must already carry the temperature factor (mcmc.py:438-440), as in the reference. */
MCMC_HIP_API int mcmc_hip_set_proposal_cov(mcmc_hip_ctx* h, const double* cov);
MCMC_HIP_API int mcmc_hip_get_proposal_cov(const mcmc_hip_ctx* h, double* cov);          /* proposal.py:262 */
MCMC_HIP_API int mcmc_hip_get_proposal_transform(const mcmc_hip_ctx* h, double* T);      /* transform[0] * scale */

/* Model.logposterior for a batch (cobaya/model.py:579-678): x[n*d] point-major ->
 * logprior[n], loglike[n] (-inf outside the prior support), derived[n*K*d] or NULL =
 * L_k^-1 (x - mu_k) (gaussian_mixture.py:146-156). */
MCMC_HIP_API int mcmc_hip_evaluate(mcmc_hip_ctx* h, int32_t n, const double* x, double* logprior,
                      double* loglike, double* derived);

/* OneSamplePoint.add of the initial points (mcmc.py:219-222): x[n_walkers*d] walker-major.
 * Evaluates their log-posterior on the device; n_bad (may be NULL) receives the number of
 * walkers with a non-finite posterior (an error, as in model.py:707-754). */
MCMC_HIP_API int mcmc_hip_set_state(mcmc_hip_ctx* h, const double* x, int32_t* n_bad);
/* current point of every walker: x[W*d], logpost[W], logprior[W], loglike[W], weight[W];
 * any pointer may be NULL */
MCMC_HIP_API int mcmc_hip_get_state(mcmc_hip_ctx* h, double* x, double* logpost, double* logprior,
                       double* loglike, int32_t* weight);

/* Complete per-walker state for checkpoint/resume (mcmc.py:189-214, 1045-1078 -- the reference
 * restarts from the last stored row and cannot save its RNG state, sampler.py:373; here the
 * Philox counter is just `step`, so a resumed run continues bit-identically): x[W*d]
 * walker-major, logpost/logprior/loglike[W], weight/prior_rej/burn_left[W] (int32),
 * n_accept[W] (int64), *step = Metropolis steps taken per walker.  set_ restores all of it
 * without re-evaluating anything (set_prior, a set_target call and set_proposal_cov must precede). */
MCMC_HIP_API int mcmc_hip_get_full_state(mcmc_hip_ctx* h, double* x, double* logpost, double* logprior,
                            double* loglike, int32_t* weight, int32_t* prior_rej,
                            int32_t* burn_left, int64_t* n_accept, uint64_t* step);
MCMC_HIP_API int mcmc_hip_set_full_state(mcmc_hip_ctx* h, const double* x, const double* logpost,
                            const double* logprior, const double* loglike, const int32_t* weight,
                            const int32_t* prior_rej, const int32_t* burn_left,
                            const int64_t* n_accept, uint64_t step);

/* n_steps iterations of MCMC.get_new_sample_metropolis (mcmc.py:545-562) for every walker,
 * asynchronously on the engine's stream; generates the Haar bases the steps need. */
MCMC_HIP_API int mcmc_hip_step(mcmc_hip_ctx* h, int32_t n_steps);
/* wait for queued work; returns MCMC_HIP_ERR_STUCK if a walker tripped max_tries */
MCMC_HIP_API int mcmc_hip_sync(mcmc_hip_ctx* h);

/* counters[0] steps per walker so far, [1] accepted steps summed over walkers (n_steps_raw /
 * acceptance of mcmc.py:311-318, 472), [2] id+1 of a stuck walker or 0, [3] rows dropped
 * because emit_capacity was exceeded */
MCMC_HIP_API int mcmc_hip_get_counters(mcmc_hip_ctx* h, int64_t counters[4]);

/* SampleCollection.add rows accumulated since the last drain (mcmc.py:691-707,
 * collection.py:402-427): rows[n][d+5] = (walker id, weight, logpost, logprior, loglike,
 * x[0..d)), walker-major then in chain order.  cap_rows = capacity of `rows` in rows. */
MCMC_HIP_API int mcmc_hip_drain_samples(mcmc_hip_ctx* h, double* rows, int64_t cap_rows, int64_t* n_rows);

/* The same drain at PCIe speed and without a host-side copy: the packed rows are copied into a
 * pinned host slot owned by the library and `*rows` points at them ([*n_rows][d+5], as
 * drain_samples); the slot is reused after `n_slots - 1` further calls (ring of 4 slots unless
 * mcmc_hip_set_drain_slots changed it; 2..64), so the caller may keep reading the rows of the
 * last n_slots - 1 drains in place.  (The exception to "the library never hands out memory":
 * stated here.) */
MCMC_HIP_API int mcmc_hip_drain_samples_pinned(mcmc_hip_ctx* h, const double** rows, int64_t* n_rows);
MCMC_HIP_API int mcmc_hip_set_drain_slots(mcmc_hip_ctx* h, int32_t n_slots);
/* Thinned emission ON THE DEVICE (round 5): OneSamplePoint.add_to_collection with output_thin > 1
 * (collection.py:1373-1383) -- the weights of a walker's accepted rows add up, a row is emitted when
 * the sum reaches `thin`, with weight sum / thin, the remainder carried to its next rows.  `emit:
 * chains` is bound by PCIe (54 GB/s of rows): thinned by T it moves T times fewer.  Served by every
 * incremental Metropolis kernel (round 6: mixtures, periodic parameters and blocks of one parameter
 * too, on the general incremental kernels); the from-scratch and dragging kernels refuse at their
 * first step (thin on the host as before).  thin = 1: off.
 * get / set_thin_carry: the per-walker remainders [W] (part of the state of a resumed run). */
MCMC_HIP_API int mcmc_hip_set_emit_thin(mcmc_hip_ctx* h, int32_t thin);
MCMC_HIP_API int mcmc_hip_get_thin_carry(mcmc_hip_ctx* h, int32_t* carry);
MCMC_HIP_API int mcmc_hip_set_thin_carry(mcmc_hip_ctx* h, const int32_t* carry);

/* Constants the engine derived from set_prior / set_target_* (uniform_logp of prior.py:528-533,
 * mls[d] of tools.py:723, Linv[K*d*d] row-major of functions.py:81-89, cnorm[K] =
 * d log 2pi + log|S_k|, weight[K]); any pointer may be NULL.  Lets tests hand the CPU oracle
 * exactly the problem the kernels evaluate. */
MCMC_HIP_API int mcmc_hip_get_derived_constants(const mcmc_hip_ctx* h, double* uniform_logp, double* mls,
                                   double* Linv, double* cnorm, double* weight);

/* Vector subtracted from every walker before its moments are accumulated (numerical
 * conditioning only; R-1 and covariances are shift invariant).  Only right after a reset. */
MCMC_HIP_API int mcmc_hip_set_moment_shift(mcmc_hip_ctx* h, const double* shift);

/* Streaming replacement of SampleCollection.mean/cov (collection.py:893-981): adds the
 * current state of every walker to the interval accumulators (one "snapshot"). */
MCMC_HIP_API int mcmc_hip_accumulate_moments(mcmc_hip_ctx* h);
/* n_snapshots since the last reset; group_sum[G*d] = sum over snapshots and the group's
 * walkers of x; pooled_S[d*d] = sum over everything of x x^T.  reset != 0 clears them. */
MCMC_HIP_API int mcmc_hip_read_moments(mcmc_hip_ctx* h, int64_t* n_snapshots, double* group_sum,
                          double* pooled_S, int32_t reset);
/* restores accumulators read with reset == 0 (resume: the snapshots taken since the last
 * read-out are part of the state) */
MCMC_HIP_API int mcmc_hip_set_moments(mcmc_hip_ctx* h, int64_t n_snapshots, const double* group_sum,
                         const double* pooled_S);
/* The same read-out without stalling the host (the learn/convergence checkpoint off the
 * critical path): `request` queues the device->host copies of the accumulators and of the
 * accept counter behind the work already in the stream, resets the accumulators in stream
 * order and returns at once; `fetch` waits for those copies only -- launches queued AFTER the
 * request keep running meanwhile.  counters[2] = (steps per walker, accepted steps of all
 * walkers) at the time of the request.  One request may be pending at a time.  The stuck flag
 * travels with the read-out: `fetch` returns MCMC_HIP_ERR_STUCK (after filling its outputs) if
 * a walker had tripped max_tries when the request was served (mcmc.py:717-743) -- the run loop
 * never calls mcmc_hip_sync, so this is where it learns of it. */
MCMC_HIP_API int mcmc_hip_request_moments(mcmc_hip_ctx* h);
MCMC_HIP_API int mcmc_hip_fetch_moments(mcmc_hip_ctx* h, int64_t* n_snapshots, double* group_sum,
                           double* pooled_S, int64_t counters[2]);

/* The learn / convergence checkpoint ON THE DEVICE (MCMC.check_convergence_and_learn_proposal,
 * mcmc.py:773-1032; checkpoint_kernels.hip): the intervals between checkpoints are kept in a
 * device ring, the statistics of the window (the later half of the run, mcmc.py:787-790) are
 * summed there, R-1 of the means is formed (mcmc.py:856-889, functions.py:81-89) and -- when
 * learn_lo <= R-1 x group_size <= learn_hi (mcmc.py:1009-1023) -- the proposal transform is
 * refreshed IN PLACE (proposal.py:226-260) with the operations of mcmc_hip_set_proposal_cov, all
 * in stream order: launches queued after checkpoint_solve use the new proposal, no host round
 * trip.  Call order per checkpoint: request_moments (the host's copy of the interval) ->
 * checkpoint_begin (window sums + the buffer an all-reduce carries: *payload_device_ptr is a
 * device pointer to *payload_len doubles = [chains, sum N, accepted since the last checkpoint,
 * steps x walkers since, accepted | sum N cov (d*d) | sum of chain means (d) | sum of m m^T (d*d)];
 * with several processes it is all-reduced on the engine's stream before the solve -- by the
 * library itself when a communicator is attached (mcmc_hip_set_comm: ncclAllReduce in place,
 * queued by checkpoint_begin), else by the caller, see mcmc_hip_stream_handle) -> checkpoint_solve -> [more launches] -> checkpoint_fetch:
 * stats = {R-1 of the chain (= group) means, status (0 ok; 1, 2, 3: the LinAlgError cases of
 * mcmc.py:870-887), 1 if the proposal was refreshed, chains, sum N, accepted since the last
 * checkpoint, steps x walkers since, accepted}, mean_of_covs[d*d].
 * checkpoint_set_ring (re)loads the ring with the intervals the caller still holds (start: none;
 * resume; growth of the window beyond the capacity): group_sum[n][G*d], pooled_S[n][d*d]. */
MCMC_HIP_API int mcmc_hip_checkpoint_set_ring(mcmc_hip_ctx* h, int32_t n_intervals, const double* group_sum,
                                 const double* pooled_S, int32_t min_capacity);
MCMC_HIP_API int mcmc_hip_checkpoint_set_accepted(mcmc_hip_ctx* h, int64_t accepted_at_last_checkpoint);
MCMC_HIP_API int mcmc_hip_checkpoint_begin(mcmc_hip_ctx* h, int32_t n_window_intervals, int64_t n_window_snapshots,
                              double steps_since, uint64_t* payload_device_ptr, int32_t* payload_len);
MCMC_HIP_API int mcmc_hip_checkpoint_solve(mcmc_hip_ctx* h, double learn_lo, double learn_hi);
MCMC_HIP_API int mcmc_hip_checkpoint_fetch(mcmc_hip_ctx* h, double stats[8], double* mean_of_covs);
/* Instead of checkpoint_solve: only the (all-reduced) payload is read out behind the launch --
 * checkpoint_begin -> checkpoint_request_payload -> [the next launch] -> checkpoint_fetch_payload
 * (waits for the copy; n = the payload_len of checkpoint_begin) -- and the caller solves it on the
 * host (mcmc_hip_gelman_rubin, mcmc_hip_set_proposal_cov) while that launch runs.  The window sums
 * and the collective stay on the device in stream order; the single-workgroup d^3 solve leaves
 * the stream (replaces the gather + host arithmetic of mcmc.py:791-793, 856-889). */
MCMC_HIP_API int mcmc_hip_checkpoint_request_payload(mcmc_hip_ctx* h);
MCMC_HIP_API int mcmc_hip_checkpoint_fetch_payload(mcmc_hip_ctx* h, double* payload, int32_t n);
/* R-1 of the confidence-interval bounds ON THE DEVICE (mcmc.py:918-1002), in every emit mode: a ring
 * of n_slots ensemble snapshots [slot][d][n_walkers] (bounds_configure; 0 frees it);
 * bounds_snapshot(slot) copies the current points into a slot in stream order (the caller decides
 * which: a thinned record of the later half of the run, sampler.py `_bounds_take`);
 * bounds_statistics forms, per chain (= walker group) and parameter, the lower and the upper
 * bound as GetDist's `MCSamples.confidence(i, limfrac, upper)` does (mcmc.py:927-929: the sample
 * at which the cumulative weight first reaches limfrac * norm, resp. (1 - limfrac) * norm) from
 * the chain's samples in the n_window listed slots (unit weights: exact order statistics,
 * selected in LDS) and returns stats[1 + 4 d] = {chains, sum_c lo_i, sum_c hi_i, sum_c lo_i^2,
 * sum_c hi_i^2} with the moment shift (mcmc_hip_set_moment_shift) subtracted from the bounds --
 * summed over the chains of ALL ranks when a communicator is attached (mcmc.py:957
 * `mpi.gather(bound)`); np.std(bounds, axis=0) of mcmc.py:977 follows from them.  `bounds`
 * (may be NULL) receives this rank's [G][d][2].  Synchronous.  n_window * group_size <= 16384.
 * get_slot / set_slot: a slot as x[n_walkers][d] (checkpoint / resume). */
MCMC_HIP_API int mcmc_hip_bounds_configure(mcmc_hip_ctx* h, int32_t n_slots);
MCMC_HIP_API int mcmc_hip_bounds_snapshot(mcmc_hip_ctx* h, int32_t slot);
MCMC_HIP_API int mcmc_hip_bounds_statistics(mcmc_hip_ctx* h, int32_t n_window, const int32_t* slots, double limfrac,
                               double* stats, double* bounds);
MCMC_HIP_API int mcmc_hip_bounds_get_slot(mcmc_hip_ctx* h, int32_t slot, double* x);
MCMC_HIP_API int mcmc_hip_bounds_set_slot(mcmc_hip_ctx* h, int32_t slot, const double* x);
/* the engine's HIP stream (a hipStream_t as an integer), for callers that queue their own work
 * -- the all-reduce of a multi-process checkpoint -- in order with the engine's */
MCMC_HIP_API uint64_t mcmc_hip_stream_handle(const mcmc_hip_ctx* h);

/* Multi-GPU: the communicator of the walker shards -- RCCL over xGMI, one process per GPU
 * (SURVEY 8e).  Stands in for the mpi4py calls of the reference's checkpoint: the gather of
 * (N, mean, cov, acceptance rate) to the root and the broadcasts back (mcmc.py:791-793
 * `mpi.array_gather`, :1005-1007 and :1021 `mpi.share`; mpi.py:178-191) become ONE all-reduce(sum)
 * of pooled sufficient statistics.  A communicator is a process-level object (created before
 * the first engine: the job's seed is agreed through it, sampler.py:369-384):
 *   rank 0: mcmc_hip_comm_unique_id(id) -> the host hands `id` to every rank out of band (any
 *   launcher's store; cobaya_amd/dist.py uses MASTER_ADDR/MASTER_PORT) -> every rank:
 *   mcmc_hip_comm_create(id, rank, n_ranks, device) [collective: ncclCommInitRank] ->
 *   mcmc_hip_set_comm(engine, comm).
 * With a communicator attached, mcmc_hip_checkpoint_begin queues `ncclAllReduce` of its payload
 * IN PLACE on the engine's stream, between the payload and the solve kernel: every rank solves
 * the same reduced statistics and refreshes its own proposal, no host bounce and no host
 * synchronisation.  mcmc_hip_comm_allreduce reduces a HOST buffer (staged through pinned memory,
 * synchronous; op 0 = sum, 1 = max): counters, the host-path checkpoint's payload, the bench
 * clock.  mcmc_hip_comm_allreduce_device reduces n doubles at a device pointer in order on a
 * caller's stream (0: the communicator's own).  RCCL is bound at run time (dlopen of
 * librccl.so.1, or $MCMC_HIP_RCCL_LIB): single-GPU runs never load it.
 * mcmc_hip_comm_last_error(NULL): the last failure before a communicator existed. */
typedef struct mcmc_hip_comm mcmc_hip_comm;
#define MCMC_HIP_COMM_ID_BYTES 128
MCMC_HIP_API const char* mcmc_hip_comm_version(void);      /* "RCCL 2.x.y", "" if it cannot be loaded */
MCMC_HIP_API const char* mcmc_hip_comm_last_error(const mcmc_hip_comm* c);
MCMC_HIP_API int mcmc_hip_comm_unique_id(uint8_t id[MCMC_HIP_COMM_ID_BYTES]);
MCMC_HIP_API int mcmc_hip_comm_create(const uint8_t id[MCMC_HIP_COMM_ID_BYTES], int32_t rank, int32_t n_ranks,
                         int32_t device, mcmc_hip_comm** out);
MCMC_HIP_API void mcmc_hip_comm_destroy(mcmc_hip_comm* c);
MCMC_HIP_API int mcmc_hip_comm_rank(const mcmc_hip_comm* c);
MCMC_HIP_API int mcmc_hip_comm_size(const mcmc_hip_comm* c);
MCMC_HIP_API int mcmc_hip_comm_allreduce(mcmc_hip_comm* c, double* buf, int64_t n, int32_t op);
MCMC_HIP_API int mcmc_hip_comm_allreduce_device(mcmc_hip_comm* c, uint64_t device_ptr, int64_t n, int32_t op,
                                   uint64_t stream);
/* HIP-event time (microseconds per call) of `reps` back-to-back in-stream all-reduces of n doubles
 * on the communicator's own stream and scratch buffer -- what one device checkpoint's collective
 * costs the stream (bench.py reports it).  Collective: every rank must call it alike. */
MCMC_HIP_API int mcmc_hip_comm_time_allreduce(mcmc_hip_comm* c, int64_t n, int32_t reps, double* us_per_call);
/* attach (or, with NULL, detach) the communicator the device checkpoint reduces over; the
 * communicator must live on the engine's device and outlive the engine */
MCMC_HIP_API int mcmc_hip_set_comm(mcmc_hip_ctx* h, mcmc_hip_comm* c);

/* The R-1 arithmetic of MCMC.check_convergence_and_learn_proposal (mcmc.py:856-889) on
 * reduced sufficient statistics (what the RCCL all-reduce of SURVEY 8e carries):
 * n_chains, sum_N = sum_c N_c, sum_Ncov[d*d] = sum_c N_c cov_c, sum_mean[d] = sum_c m_c,
 * sum_mm[d*d] = sum_c m_c m_c^T.  Outputs Rminus1 and mean_of_covs[d*d].
 * Returns MCMC_HIP_ERR_NOT_PD where the reference catches LinAlgError (mcmc.py:870-887). */
MCMC_HIP_API int mcmc_hip_gelman_rubin(int32_t d, double n_chains, double sum_N, const double* sum_Ncov,
                          const double* sum_mean, const double* sum_mm, double* Rminus1,
                          double* mean_of_covs);

/* HIP-event time (ms) spent in step kernels / basis kernels / moment kernels since the last
 * call with reset != 0, and the number of step-kernel launches: the live measurement
 * bench.py's roofline block uses.  Timing is enabled by mcmc_hip_enable_timing(h, 1).  Every
 * step kernel is timed; of the direction and moment regions one in eight, scaled to all of
 * them (an event record costs the stream about 6 us between two dependent kernels). */
MCMC_HIP_API int mcmc_hip_enable_timing(mcmc_hip_ctx* h, int32_t on);
/* incremental mode: the carried y[n_walkers][n_modes * d] -- part of the state a bit-identical resume
 * needs (call mcmc_hip_set_whitened after mcmc_hip_set_full_state) */
MCMC_HIP_API int mcmc_hip_get_whitened(mcmc_hip_ctx* h, double* y);
MCMC_HIP_API int mcmc_hip_set_whitened(mcmc_hip_ctx* h, const double* y);
/* incremental mode, Gaussian mixtures (gaussian_mixture.py:138-163): 1 if the step kernel this
 * engine's configuration selects CARRIES the log-density a_k = -(c_k + chi2_k) / 2 of every mode
 * with the walker (round 5: step_inc_mix_kernel -- 2..4 modes at d <= 64, 5 at d <= 32, 6 at d <= 28, no periodic parameter,
 * Metropolis steps, emit_capacity 0), moved along the whitened direction like the carried
 * log-likelihood of a single mode; 0 if every chi2_k is summed from the trial's residual (the
 * general kernels: more modes, periodic parameters, emitted rows).  The
 * specification (oracle/mcmc_oracle.c: carries_modes) takes the rule from here.  The carried
 * a[n_walkers][n_modes] are part of the state a bit-identical resume needs (call
 * mcmc_hip_set_mode_logdensities after mcmc_hip_set_whitened; without it they are re-anchored on y
 * at the next step). */
MCMC_HIP_API int mcmc_hip_incremental_carries_modes(const mcmc_hip_ctx* h);
/* incremental mode, ONE mode with periodic parameters (prior.py:658-676): 1 if this configuration
 * runs on step_inc_kernel<.., periodic> (1..16 periodic parameters, Metropolis steps, emit_capacity 0),
 * whose rule since round 5 is: a periodic coordinate is wrapped only where the trial leaves
 * [lo, hi), and the log-likelihood is carried along the whitened direction, re-summed from the
 * moved residual at a step that wraps; 0: the coordinate passes through the wrap at every step
 * (the general kernels).  The specification (oracle: carry_periodic) takes the rule from here. */
MCMC_HIP_API int mcmc_hip_incremental_carries_periodic(const mcmc_hip_ctx* h);
MCMC_HIP_API int mcmc_hip_get_mode_logdensities(mcmc_hip_ctx* h, double* a);
MCMC_HIP_API int mcmc_hip_set_mode_logdensities(mcmc_hip_ctx* h, const double* a);

/* name of the step kernel the last mcmc_hip_step launched, e.g.
 * "mcmc::step_pair_kernel<true, false> (d=30)" -- reported by the launcher itself, so that
 * profiles and bench lines quote the kernel that ran ("" before the first step) */
MCMC_HIP_API const char* mcmc_hip_last_step_kernel(const mcmc_hip_ctx* h);
MCMC_HIP_API int mcmc_hip_kernel_times(mcmc_hip_ctx* h, double ms[3], int64_t* n_step_launches, int32_t reset);
/* binned Gaussian target: HIP-event time (ms) and number of launches of the three kernels of a
 * step -- pl_walker_kernel, pl_residual_kernel, pl_chi2_kernel -- since the last reset */
MCMC_HIP_API int mcmc_hip_binned_kernel_times(mcmc_hip_ctx* h, double ms[3], int64_t n_launches[3], int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* MCMC_HIP_H */
