/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Flavour (b) of the oracle: a plain-C CPU restatement of the walker-ensemble Metropolis
 * step that the HIP kernels implement (DESIGN.md "Ensemble specification"), on the same
 * counter-based Philox4x32-10 stream, written so that every floating-point operation is
 * fixed (explicit fma(), -ffp-contract=off): the HIP path must match it BIT FOR BIT
 * ("Tier B" of SURVEY.md 8c).  It is tied to the reference two ways:
 *   - orc_step_injected() replays a reference chain when fed the reference's own random
 *     draws (tests/test_oracle_c.py, against oracle/ref_numpy.py == golden G6);
 *   - orc_evaluate() reproduces the reference's logprior / loglike golden vectors G4/G5.
 * It is also the timed "cpu_baseline" (kind "port") of bench.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Reference restated (paths relative to /root/reference):
 *   proposal   cobaya/samplers/mcmc/proposal.py:59-82,222-224 ; cobaya/functions.py:35-61
 *   prior      cobaya/prior.py:658-676,733-763 ; cobaya/tools.py:720-729
 *   likelihood cobaya/likelihoods/gaussian_mixture/gaussian_mixture.py:138-163,
 *              cobaya/likelihoods/gaussian/gaussian.py:96-112
 *   accept     cobaya/samplers/mcmc/mcmc.py:670-683 ; bookkeeping mcmc.py:685-748
 *   moments    cobaya/collection.py:926-934,970-981 (as streaming sufficient statistics)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ Philox4x32-10 */
/* Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the generator
 * rocRAND/cuRAND ship as PHILOX4_32_10. */
static inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                 uint32_t c2, uint32_t c3, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                uint32_t c3, uint32_t out[4])
{
    philox4x32_10(k0, k1, c0, c1, c2, c3, out);
}

enum { STREAM_STEP = 0, STREAM_BASIS = 1, STREAM_PERM = 2 };
#define BRANCH_EXP_24 5536481u /* floor(0.33 * 2^24): proposal.py:79 */

/* u = (2k+1) 2^-53, k < 2^52: an odd multiple of 2^-53 in (0,1), exact in binary64 */
static inline double u52(uint64_t k) { return (double)(2 * k + 1) * 0x1p-53; }

/* ------------------------------------------------------------------ fixed-order math */
static inline double bits2d(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t d2bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* natural log of a positive normal double; argument reduction and minimax polynomial of
 * Sun's fdlibm e_log.c (public algorithm), every operation fixed. < 1 ulp. */
double orc_dlog(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    uint64_t b = d2bits(x);
    uint32_t hx = (uint32_t)(b >> 32);
    int k = (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;
    b = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (b & 0xffffffffu);
    k += (int)(i >> 20);
    double f = bits2d(b) - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - fma(s, hfsq + R, dk * ln2_lo)) - f);
}

/* -log(n 2^-b) for an ODD integer n < 2^29 (b <= 29): the logarithm of the paired variates
 * (walker_variates_pair), whose arguments are short -- 25 or 29 significant bits -- so that a
 * table step is exact.  n = m 2^e, m in [1/2, 1); j = top seven fraction bits of m;
 * f = fma(m, RC_j, -1) is EXACT (29 + 24 bits) with |f| <= 2^-8; log1p(f) by its Taylor
 * polynomial to f^6 (next term < 2^-58); result = (b - e) ln 2 + log(RC_j) - log1p(f).
 * Table: oracle/short_log_table.h (tools/make_short_log_table.py).  Absolute error < 3e-16. */
#include "short_log_table.h"
static const double short_log_table[SHORT_LOG_TABLE_SIZE][2] = SHORT_LOG_TABLE;
double orc_neg_log_short(uint32_t n, int32_t b)
{
    static const double LN2 = 6.93147180559945286227e-01, C2 = -0.5,
        C3 = 3.33333333333333314830e-01, C4 = -0.25, C5 = 2.00000000000000011102e-01,
        C6 = -1.66666666666666657415e-01;
    int e;
    const double m = frexp((double)n, &e);
    const unsigned j = (unsigned)(d2bits(m) >> 45) & 0x7Fu;
    const double f = fma(m, short_log_table[j][0], -1.0);
    const double p = f * fma(f, fma(f, fma(f, fma(f, fma(f, C6, C5), C4), C3), C2), 1.0);
    return fma((double)(b - e), LN2, short_log_table[j][1]) - p;
}

/* Table-driven log and exp of the INCREMENTAL mixtures' log-sum-exp (round 5: dlog_tab / dexp_tab in
 * det_math.h; the from-scratch evaluator keeps orc_dlog / orc_dexp, pinned by golden G5).  No
 * division: a K = 2 step spent ~75 of its ~370 issue slots in the two fdlibm routines.
 *   log x, x > 0 normal: x = m 2^e, m in [1/2, 1); j = top seven fraction bits of m;
 *     f = fma(m, RC_j, -1), |f| <= 2^-8 (one rounding: 2^-61 absolute); log x =
 *     log1p(f) - fma(-e, ln 2, LRC_j), log1p by its Taylor polynomial to f^6 (as orc_neg_log_short).
 *     Absolute error < 3e-16 + 1 ulp: what a log-sum-exp of terms in [w_max, 1] needs.
 *   exp x, x >= -708 (else 0): k = rint(x 64 / ln 2), r = x - k ln 2 / 64 (two fmas, |r| <= 0.0055),
 *     exp r - 1 = r (1 + r (1/2 + r (1/6 + r (1/24 + r / 120)))) (next term 3e-17), T_j = 2^(j / 64)
 *     with j = k mod 64 from the table, result fma(T_j, p, T_j) 2^(k div 64).  Within 2 ulp. */
static const double exp64_table[64] = EXP64_TABLE;
double orc_dlog_tab(double x)
{
    static const double LN2 = 6.93147180559945286227e-01, C2 = -0.5,
        C3 = 3.33333333333333314830e-01, C4 = -0.25, C5 = 2.00000000000000011102e-01,
        C6 = -1.66666666666666657415e-01;
    int e;
    const double m = frexp(x, &e);
    const unsigned j = (unsigned)(d2bits(m) >> 45) & 0x7Fu;
    const double f = fma(m, short_log_table[j][0], -1.0);
    const double p = f * fma(f, fma(f, fma(f, fma(f, fma(f, C6, C5), C4), C3), C2), 1.0);
    return p - fma((double)(-e), LN2, short_log_table[j][1]);
}

double orc_dexp_tab(double x)
{
    static const double C2 = 0.5, C3 = 1.66666666666666657415e-01, C4 = 4.16666666666666643537e-02,
        C5 = 8.33333333333333321769e-03;
    if (!(x >= -708.0)) return 0.0;
    const double kf = rint(x * EXP64_INV_LN2);
    double r = fma(-kf, EXP64_LN2_HI, x);
    r = fma(-kf, EXP64_LN2_LO, r);
    const double p = r * fma(r, fma(r, fma(r, fma(r, C5, C4), C3), C2), 1.0);
    const int k = (int)kf;
    const double T = exp64_table[k & 63];
    const double y = fma(T, p, T);
    return bits2d(d2bits(y) + ((uint64_t)(int64_t)(k >> 6) << 52));
}

/* exp(x) for x <= 0 (mixture log-sum-exp terms); fdlibm e_exp.c reduction/polynomial.
 * x < -708 returns 0 (the term is below 1e-307 of the leading one). */
double orc_dexp(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (!(x >= -708.0)) return 0.0;
    double kf = rint(x * invln2);
    double hi = fma(-kf, ln2_hi, x);
    double lo = kf * ln2_lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * fma(t, fma(t, fma(t, fma(t, P5, P4), P3), P2), P1);
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    int k = (int)kf; /* -1022 <= k <= 0 here */
    return bits2d(d2bits(y) + ((uint64_t)(int64_t)k << 52));
}

/* sin/cos kernels on [0, pi/4] (fdlibm k_sin.c / k_cos.c polynomials, y = 0) */
static inline double ksin(double x)
{
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    return fma(v, fma(z, r, S1), x);
}
static inline double kcos(double x)
{
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    return 1.0 - (0.5 * z - z * r);
}

/* (sin, cos)(2 pi u) for u = (2k+1) 2^-53: octant from the top 3 bits of k, exact
 * in-octant fraction phi = (2 rem + 1) 2^-50, reflected (1 - phi, exact) in odd octants */
void orc_sincos2pi(uint64_t k, double* sn, double* cs)
{
    static const double PIO4 = 7.85398163397448278999e-01;
    unsigned o = (unsigned)(k >> 49);
    uint64_t rem = k & ((1ull << 49) - 1);
    double phi = (double)(2 * rem + 1) * 0x1p-50;
    if (o & 1) phi = 1.0 - phi;
    double a = phi * PIO4;
    double s = ksin(a), c = kcos(a);
    double ss, cc;
    switch (o) {
    case 0: ss = s; cc = c; break;
    case 1: ss = c; cc = s; break;
    case 2: ss = c; cc = -s; break;
    case 3: ss = s; cc = -c; break;
    case 4: ss = -s; cc = -c; break;
    case 5: ss = -c; cc = -s; break;
    case 6: ss = -c; cc = s; break;
    default: ss = -s; cc = c; break;
    }
    *sn = ss; *cs = cc;
}

/* ------------------------------------------------------------------ problem description */
typedef struct {
    int32_t d;           /* sampled dimension */
    int32_t n_modes;     /* 0 = `one` likelihood (loglike 0), >=1 Gaussian mixture */
    int32_t group_size;  /* walkers sharing one Haar basis */
    int32_t has_periodic;
    uint64_t seed;
    double temperature;
    double max_tries;    /* mcmc.yaml:9 (already scaled by d) */
    /* prior (prior.py:514-533): kind 0 uniform [lo,hi], kind 1 normal(loc,scale) with
     * lo/hi = -/+inf; mls = -log(scale) - log(2pi)/2 */
    const int32_t* kind; const double* lo; const double* hi; const double* loc;
    const double* scale; const double* mls; const int32_t* periodic;
    double uniform_logp;
    /* target: per mode k: mean[k*d+i]; Linv row-major lower-triangular [k][j][i] (d*d per
     * mode, upper part ignored); cnorm[k] = d log 2pi + log|S_k| (0 if unnormalised);
     * weight[k] */
    const double* mean; const double* Linv; const double* cnorm; const double* weight;
    /* proposal transform T = scale * diag(std) * chol(corr), row-major lower-tri d*d; with
     * blocks it is built from the covariance in SORTED order (j-indexed, proposal.py:252) */
    const double* T;
    const struct orc_blocking* blocking; /* NULL: one block holding every parameter */
    /* evaluation mode (DESIGN.md "Incremental evaluation"): 0 = every trial is evaluated from
     * scratch (eval_point); 1 = the whitened residual y = L^-1 (x - mu) of every walker is
     * CARRIED and moved along the whitened direction, y' = y + r L^-1 v -- O(d) per step for
     * the same log-posterior -- and recomputed from x every `refresh_every` steps */
    int32_t incremental;
    int32_t refresh_every;
    /* incremental mode, plain steps: 1 = the variates of steps 2P and 2P+1 come from ONE Philox
     * block (walker_variates_pair) -- what the incremental kernels do; 0 = one block per step as
     * everywhere else (kept so that tests can run both modes on the same proposal stream) */
    int32_t paired_variates;
    /* incremental mode, mixtures (round 5): 1 = the log-density a_k of every MODE is carried with
     * the walker like the log-likelihood of a single mode (step_core_inc, `carry_modes`) -- what
     * step_inc_mix_kernel and the register-plane kernel without periodic parameters do; 0 = every
     * chi2_k is summed from the trial's residual (the general LDS kernel, periodic parameters).
     * The engine says which (mcmc_hip_incremental_carries_modes). */
    int32_t carry_modes;
    /* incremental mode, ONE mode with periodic parameters (round 5): 1 = step_inc_periodic_kernel's
     * rule -- a periodic coordinate is wrapped only where the trial LEAVES [lo, hi) (inside, the
     * reference's ((x - a) / (b - a)) % 1 * (b - a) + a, prior.py:675, returns x up to its own
     * rounding; here it returns x), and the log-likelihood is carried as without periodic
     * parameters, re-summed from the moved residual only at a step that wraps; 0 = the coordinate
     * passes through the wrap at every step and chi2 is summed from the trial's residual (the
     * general kernels).  The engine says which (mcmc_hip_incremental_carries_periodic). */
    int32_t carry_periodic;
    /* binned-bandpower Gaussian likelihood (planck_pliklite.py:143-155) instead of the mixture
     * (n_modes must be 0): see orc_binned below */
    const struct orc_binned* binned;
} orc_problem;

/* The plik-lite arithmetic (cobaya/likelihoods/base_classes/planck_pliklite.py:143-155 +
 * functions.py:64-78) with the operation order of the device kernels (pliklite_kernels.hip):
 *   response  the stand-in for provider.get_Cl (planck_pliklite.py:170-178) is LINEAR,
 *             D_l(theta) = D0[tp][l] + sum_p J[tp][l][p] (theta_p - theta0_p), theta = the sampled
 *             parameters without the calibration parameter, in order; binning (np.dot of
 *             planck_pliklite.py:148-151) commutes with it, so the binned response is formed ONCE
 *             (orc_binned_collapse; mcmc_hip_set_target_binned_gaussian does the same):
 *             Bc0_b = fma chain over l = first..last ascending of D0[tp][l] * weights[l] from +0,
 *             BJ_bp likewise with J[tp][l][p];
 *   binning   per point cl_b = Bc0_b then fma(BJ_bp, theta_p - theta0_p, .) for p ascending.
 *             For EXPLICIT spectra (orc_binned_chi2_of_cl = get_chi_squared's own signature)
 *             cl_b = fma chain over l ascending of D_l * weights[l] from +0;
 *   residual  delta_b = fma(-cl_b, 1 / (A A), X_b)        (cl /= A_planck**2; diff = X - cl);
 *   chi2      y_j = fma chain over i = 0..j ascending of Linv[j][i] delta_i from +0, with
 *             cov = L L^T (the same quadratic form as invcov.dot(diff).dot(diff));  the squares
 *             are summed in 32 interleaved chains p[q][c] over the rows with j mod 4 = c and
 *             class(j div 16) = q; with NT = ceil(n_bins / 16) tiles, class(R) = (R + shift) mod 8,
 *             shift = (8 - NT mod 8) mod 8: the position of tile R in its group of eight, the
 *             groups counted down from the last tile (the fused kernel gives the tile at a
 *             position to one wave per pair of walker tiles, and lane class c its rows 4r + c),
 *             s_q = (p[q][0] + p[q][1]) + (p[q][2] + p[q][3]),
 *             chi2 = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
 *   loglike   -chi2 / 2. */
typedef struct orc_binned {
    int32_t n_bins;        /* used bins, in data-vector order (planck_pliklite.py:126-141) */
    int32_t lmax;
    int32_t n_lin;         /* emulator parameters (= d - 1 when sampling) */
    int32_t calib;         /* index of the calibration parameter among the sampled ones */
    const int32_t* bins;   /* [n_bins][3] = (spectrum 0 tt / 1 te / 2 ee, first l, last l) */
    const double* weights; /* [lmax + 1], D_l space (planck_pliklite.py:52-56) */
    const double* X;       /* [n_bins] */
    const double* Linv;    /* [n_bins][n_bins] row-major, lower triangle used */
    const double* theta0;  /* [n_lin] */
    const double* D0;      /* [3][lmax + 1] */
    const double* J;       /* [3][lmax + 1][n_lin] */
    const double* Bc0;     /* [n_bins]          binned response, filled by orc_binned_collapse */
    const double* BJ;      /* [n_bins][n_lin] */
} orc_binned;

/* Blocked proposal (proposal.py:96-224): blocks sorted slow -> fast; parameter j of the
 * sorted order is sampler parameter i_of_j[j]; block b covers n_b consecutive j from
 * j_start(b).  drag_last_slow >= 0 selects the dragging step (mcmc.py:564-668) with the
 * blocks up to that index slow and drag_steps interpolation steps. */
typedef struct orc_blocking {
    int32_t n_blocks;
    const int32_t* size;
    const int32_t* oversample;
    const int32_t* i_of_j;
    int32_t drag_last_slow;
    int32_t drag_steps;
} orc_blocking;

/* ------------------------------------------------------------------ Haar basis (a3) */
/* Householder construction of functions.py:45-61 with a fixed operation order.
 * z: (d+2)(d-1)/2 standard normals; H: d*d row-major output (rows scaled by D). */
void orc_haar_from_normals(int d, const double* z, double* H)
{
    double* x = (double*)malloc(sizeof(double) * (size_t)d);
    double* D = (double*)malloc(sizeof(double) * (size_t)d);
    for (int i = 0; i < d * d; ++i) H[i] = 0.0;
    for (int i = 0; i < d; ++i) H[i * d + i] = 1.0;
    int ix = 0;
    double dprod = 1.0;
    for (int n = 0; n < d - 1; ++n) {
        int m = d - n;
        double norm2 = 0.0;
        for (int k = 0; k < m; ++k) { x[k] = z[ix + k]; norm2 = fma(x[k], x[k], norm2); }
        ix += m;
        double x0 = x[0];
        double Dn = (x0 < 0.0) ? -1.0 : 1.0;
        D[n] = Dn; dprod *= Dn;
        x[0] = x0 + Dn * sqrt(norm2);
        double t = norm2 - x0 * x0;
        t = t + x[0] * x[0];
        double den = sqrt(0.5 * t);
        for (int k = 0; k < m; ++k) x[k] = x[k] / den;
        for (int i = 0; i < d; ++i) {
            /* projection of row i on the reflector: one ascending chain for d <= 32; for d > 32
             * four interleaved chains over the columns j = c (mod 4), combined
             * (t0 + t1) + (t2 + t3) -- four lanes of the d > 32 basis kernel serve one row */
            double tc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < m; ++k) {
                const int c = d > 32 ? ((n + k) & 3) : 0;
                tc[c] = fma(H[i * d + n + k], x[k], tc[c]);
            }
            const double tmp = d > 32 ? (tc[0] + tc[1]) + (tc[2] + tc[3]) : tc[0];
            for (int k = 0; k < m; ++k) H[i * d + n + k] = fma(-tmp, x[k], H[i * d + n + k]);
        }
    }
    D[d - 1] = (((d - 1) & 1) ? -1.0 : 1.0) * dprod;
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) H[i * d + k] = D[i] * H[i * d + k];
    free(x); free(D);
}

/* normals of the basis stream for (group, cycle): Box-Muller on Philox pairs */
static void basis_normals(uint64_t seed, uint32_t group, uint32_t cycle, int n, double* z)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int j = 0; 2 * j < n; ++j) {
        uint32_t w[4];
        philox4x32_10(k0, k1, group, STREAM_BASIS, cycle, (uint32_t)j, w);
        uint64_t ka = ((uint64_t)w[0] << 20) | (w[1] >> 12);
        uint64_t kb = ((uint64_t)w[2] << 20) | (w[3] >> 12);
        double rad = sqrt(-2.0 * orc_dlog(u52(ka)));
        double sn, cs;
        orc_sincos2pi(kb, &sn, &cs);
        z[2 * j] = rad * cs;
        if (2 * j + 1 < n) z[2 * j + 1] = rad * sn;
    }
}

/* V[c*d + i] = sum_{k<=i} T[i][k] R[k][c]: the d proposal direction vectors of one cycle
 * (proposal.py:222-224 with transform of 256-260); column c is used at step cycle*d+c. */
void orc_basis(const orc_problem* p, uint32_t group, uint32_t cycle, double* V)
{
    int d = p->d;
    if (d == 1) { V[0] = p->T[0]; return; }
    int nz = (d + 2) * (d - 1) / 2;
    double* z = (double*)malloc(sizeof(double) * (size_t)(nz + 1));
    double* H = (double*)malloc(sizeof(double) * (size_t)d * d);
    basis_normals(p->seed, group, cycle, nz, z);
    orc_haar_from_normals(d, z, H);
    for (int c = 0; c < d; ++c)
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int k = 0; k <= i; ++k) s = fma(p->T[i * d + k], H[k * d + c], s);
            V[c * d + i] = s;
        }
    free(z); free(H);
}

/* ------------------------------------------------------------------ blocked schedule */
/* The three slot sequences of the blocked proposer (proposal.py:187-196):
 *   which 0: every block b listed oversample[b] * n_b times (block_cycler),
 *   which 1: the slow blocks, one slot per parameter (block_cycler_slow),
 *   which 2: the fast blocks, one slot per parameter (block_cycler_fast).
 * Returns the cycle length L and the unshuffled slot list in blk[]. */
int orc_block_slots(const orc_problem* p, int which, int32_t* blk)
{
    const orc_blocking* B = p->blocking;
    if (!B) { if (blk) for (int i = 0; i < p->d; ++i) blk[i] = 0; return p->d; }
    int L = 0;
    for (int b = 0; b < B->n_blocks; ++b) {
        int reps;
        if (which == 0) reps = B->oversample[b] * B->size[b];
        else if (which == 1) reps = (b <= B->drag_last_slow) ? B->size[b] : 0;
        else reps = (b > B->drag_last_slow) ? B->size[b] : 0;
        for (int r = 0; r < reps; ++r) { if (blk) blk[L] = b; ++L; }
    }
    return L;
}

/* Schedule of one (group, cycle): block, basis number and basis column of every slot.
 * CyclicIndexRandomizer.next (proposal.py:46-55) reshuffles the slot list once per cycle
 * when it is longer than 2; here a Fisher-Yates shuffle on the Philox stream
 * (group, STREAM_PERM | which << 8, cycle, i), j = floor(u32 * (i + 1) / 2^32), i descending.
 * The k-th use of block b in the cycle takes column k % n_b of its basis number k / n_b
 * (RandDirectionProposer.propose_vec, proposal.py:66-69: a new basis every n_b uses). */
int orc_block_schedule(const orc_problem* p, uint32_t group, uint32_t cycle, int which,
                       int32_t* blk, int32_t* basis, int32_t* col)
{
    int L = orc_block_slots(p, which, blk);
    uint32_t k0 = (uint32_t)p->seed, k1 = (uint32_t)(p->seed >> 32);
    if (p->blocking && L > 2)
        for (int i = L - 1; i >= 1; --i) {
            uint32_t w[4];
            philox4x32_10(k0, k1, group, STREAM_PERM | ((uint32_t)which << 8), cycle,
                          (uint32_t)i, w);
            int j = (int)(((uint64_t)w[0] * (uint64_t)(i + 1)) >> 32);
            int32_t t = blk[i]; blk[i] = blk[j]; blk[j] = t;
        }
    int used[64];
    for (int b = 0; b < 64; ++b) used[b] = 0;
    for (int s = 0; s < L; ++s) {
        int b = blk[s];
        int n = p->blocking ? p->blocking->size[b] : p->d;
        basis[s] = used[b] / n;
        col[s] = used[b] % n;
        ++used[b];
    }
    return L;
}

/* V[s*d + i] for the L slots of one (group, cycle) of sequence `which`:
 * v_sorted[j] = sum_{k <= min(j - j_b, n_b - 1)} T[j][j_b + k] u[k] for j >= j_b (0 above),
 * u = column `col` of the block's Haar basis number `basis` (u = 1 for a 1-d block,
 * proposal.py:85-93), scattered to sampler order through i_of_j (proposal.py:222-224).
 * The normals of basis q of block b come from the Philox stream
 * (group, STREAM_BASIS | which << 4 | b << 8, cycle, q << 16 | pair). flag1d[s] = 1 when the
 * slot's block has one parameter. */
int orc_basis_blocked(const orc_problem* p, uint32_t group, uint32_t cycle, int which,
                      double* V, int32_t* flag1d)
{
    const orc_blocking* B = p->blocking;
    int d = p->d;
    if (!B) { orc_basis(p, group, cycle, V); if (flag1d) for (int i = 0; i < d; ++i) flag1d[i] = (d == 1); return d; }
    int32_t* blk = (int32_t*)malloc(sizeof(int32_t) * 3 * 4096);
    int32_t* bas = blk + 4096; int32_t* col = bas + 4096;
    int L = orc_block_schedule(p, group, cycle, which, blk, bas, col);
    uint32_t k0 = (uint32_t)p->seed, k1 = (uint32_t)(p->seed >> 32);
    double* H = (double*)malloc(sizeof(double) * (size_t)d * d);
    double* z = (double*)malloc(sizeof(double) * (size_t)((d + 2) * (d - 1) / 2 + 2));
    int jb = 0;
    for (int b = 0; b < B->n_blocks; jb += B->size[b], ++b) {
        int n = B->size[b];
        int have = -1;
        for (int s = 0; s < L; ++s) {
            if (blk[s] != b) continue;
            double* v = V + (size_t)s * d;
            for (int i = 0; i < d; ++i) v[i] = 0.0;
            if (flag1d) flag1d[s] = (n == 1);
            if (n == 1) {
                for (int j = jb; j < d; ++j) v[B->i_of_j[j]] = p->T[j * d + jb];
                continue;
            }
            if (bas[s] != have) {
                int nz = (n + 2) * (n - 1) / 2;
                for (int j = 0; 2 * j < nz; ++j) {
                    uint32_t w[4];
                    philox4x32_10(k0, k1, group,
                                  STREAM_BASIS | ((uint32_t)which << 4) | ((uint32_t)b << 8), cycle,
                                  ((uint32_t)bas[s] << 16) | (uint32_t)j, w);
                    uint64_t ka = ((uint64_t)w[0] << 20) | (w[1] >> 12);
                    uint64_t kb = ((uint64_t)w[2] << 20) | (w[3] >> 12);
                    double rad = sqrt(-2.0 * orc_dlog(u52(ka)));
                    double sn, cs;
                    orc_sincos2pi(kb, &sn, &cs);
                    z[2 * j] = rad * cs;
                    if (2 * j + 1 < nz) z[2 * j + 1] = rad * sn;
                }
                orc_haar_from_normals(n, z, H);
                have = bas[s];
            }
            for (int j = jb; j < d; ++j) {
                int kmax = j - jb < n - 1 ? j - jb : n - 1;
                double acc = 0.0;
                for (int k = 0; k <= kmax; ++k)
                    acc = fma(p->T[j * d + jb + k], H[k * n + col[s]], acc);
                v[B->i_of_j[j]] = acc;
            }
        }
    }
    free(z); free(H); free(blk);
    return L;
}

/* ------------------------------------------------------------------ log-posterior (a6-a10) */
static inline double wrap_periodic(double t, double lo, double hi)
{
    /* prior.py:675: ((x - a) / (b - a)) % 1 * (b - a) + a, Python float modulo */
    double w = hi - lo;
    double y = (t - lo) / w;
    double m = y - floor(y);
    return m * w + lo;
}

/* ------------------------------------------------------------------ binned Gaussian (plik-lite) */
/* position of 16-row tile R in its group of eight, the groups counted down from the LAST tile */
static inline int binned_class(int R, int NT) { return (R + ((8 - NT % 8) % 8)) & 7; }

/* chi2 of a residual vector (order: see orc_binned) */
double orc_binned_chi2_of_delta(const orc_binned* b, const double* delta)
{
    const int n = b->n_bins, NT = (n + 15) / 16;
    double p[8][4];
    for (int q = 0; q < 8; ++q) for (int c = 0; c < 4; ++c) p[q][c] = 0.0;
    for (int j = 0; j < n; ++j) {
        const double* Lj = b->Linv + (size_t)j * n;
        double y = 0.0;
        for (int i = 0; i <= j; ++i) y = fma(Lj[i], delta[i], y);
        double* pc = &p[binned_class(j >> 4, NT)][j & 3];
        *pc = fma(y, y, *pc);
    }
    double s[8];
    for (int q = 0; q < 8; ++q) s[q] = (p[q][0] + p[q][1]) + (p[q][2] + p[q][3]);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

/* binned response of the linear emulator (see orc_binned): Bc0[n_bins], BJ[n_bins][n_lin] */
void orc_binned_collapse(const orc_binned* b, double* Bc0, double* BJ)
{
    const int n = b->n_lin, L1 = b->lmax + 1;
    for (int ib = 0; ib < b->n_bins; ++ib) {
        const int tp = b->bins[3 * ib], l0 = b->bins[3 * ib + 1], l1 = b->bins[3 * ib + 2];
        double acc = 0.0;
        for (int l = l0; l <= l1; ++l) acc = fma(b->D0[(size_t)tp * L1 + l], b->weights[l], acc);
        Bc0[ib] = acc;
        for (int p = 0; p < n; ++p) {
            double a = 0.0;
            for (int l = l0; l <= l1; ++l)
                a = fma(b->J[((size_t)tp * L1 + l) * n + p], b->weights[l], a);
            BJ[(size_t)ib * n + p] = a;
        }
    }
}

/* residual of the sampled point x[d] through the binned response */
void orc_binned_delta(const orc_binned* b, const double* x, double* delta)
{
    double dth[32];
    const int n = b->n_lin;
    for (int p = 0, i = 0; p < n; ++p, ++i) {
        if (i == b->calib) ++i;
        dth[p] = x[i] - b->theta0[p];
    }
    const double A = x[b->calib];
    const double iA2 = 1.0 / (A * A);
    for (int ib = 0; ib < b->n_bins; ++ib) {
        double cl = b->Bc0[ib];
        const double* BJb = b->BJ + (size_t)ib * n;
        for (int p = 0; p < n; ++p) cl = fma(BJb[p], dth[p], cl);
        delta[ib] = fma(-cl, iA2, b->X[ib]);
    }
}

/* PlanckPlikLite.get_chi_squared(L0, ctt, cte, cee, A_planck) (planck_pliklite.py:143-155) for
 * n_pts sets of explicit spectra: cl[pt][3][stride], element l - L0 of a row is D_l */
void orc_binned_chi2_of_cl(const orc_binned* b, int n_pts, int L0, int stride, const double* cl,
                           const double* A, double* chi2)
{
    double* delta = (double*)malloc(sizeof(double) * (size_t)b->n_bins);
    for (int k = 0; k < n_pts; ++k) {
        const double iA2 = 1.0 / (A[k] * A[k]);
        for (int ib = 0; ib < b->n_bins; ++ib) {
            const int tp = b->bins[3 * ib], l0 = b->bins[3 * ib + 1], l1 = b->bins[3 * ib + 2];
            const double* cell = cl + ((size_t)k * 3 + tp) * stride;
            double acc = 0.0;
            for (int l = l0; l <= l1; ++l) acc = fma(cell[l - L0], b->weights[l], acc);
            delta[ib] = fma(-acc, iA2, b->X[ib]);
        }
        chi2[k] = orc_binned_chi2_of_delta(b, delta);
    }
    free(delta);
}

static double binned_loglike(const orc_binned* b, const double* t)
{
    double* delta = (double*)malloc(sizeof(double) * (size_t)b->n_bins);
    orc_binned_delta(b, t, delta);
    const double chi2 = orc_binned_chi2_of_delta(b, delta);
    free(delta);
    return -0.5 * chi2;
}

/* returns 1 if inside the prior support; fills lp, ll (ll only if inside) and optionally
 * derived[k*d + j] = (L_k^-1 (t - mu_k))_j */
static int eval_point(const orc_problem* p, const double* t, double* lp_out, double* ll_out,
                      double* derived)
{
    int d = p->d;
    int inb = 1;
    for (int i = 0; i < d; ++i) inb &= (t[i] <= p->hi[i]) & (t[i] >= p->lo[i]);
    if (!inb) { *lp_out = -INFINITY; *ll_out = -INFINITY; return 0; }
    /* sum of the normal priors' terms: one ascending chain for d <= 32; for d > 32 four
     * interleaved chains s_c over the dimensions i = c (mod 4), combined (s0 + s1) + (s2 + s3)
     * -- the matrix-core kernel holds dimension i in lane class i mod 4 (see chi2 below) */
    double sc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < d; ++i)
        if (p->kind[i] == 1) {
            double q = (t[i] - p->loc[i]) / p->scale[i];
            const int c = d > 32 ? (i & 3) : 0;
            sc[c] = sc[c] + fma(-0.5 * q, q, p->mls[i]);
        }
    const double s = d > 32 ? (sc[0] + sc[1]) + (sc[2] + sc[3]) : sc[0];
    *lp_out = p->uniform_logp + s;
    int K = p->n_modes;
    if (p->binned) { *ll_out = binned_loglike(p->binned, t); return 1; }
    if (K == 0) { *ll_out = 0.0; return 1; }
    double a[64];
    double amax = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const double* Li = p->Linv + (size_t)k * d * d;
        const double* mu = p->mean + (size_t)k * d;
        /* chi2 = sum_j y_j^2: one ascending chain for d <= 32; for d > 32 four interleaved
         * chains p_c over the rows j = c (mod 4), combined as (p0 + p1) + (p2 + p3) -- the
         * order in which the matrix-core (MFMA) kernel holds the y_j, four rows per lane */
        double pc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = 0; j < d; ++j) {
            double y = 0.0;
            for (int i = 0; i <= j; ++i) y = fma(Li[j * d + i], t[i] - mu[i], y);
            if (derived) derived[k * d + j] = y;
            const int c = d > 32 ? (j & 3) : 0;
            pc[c] = fma(y, y, pc[c]);
        }
        const double chi2 = d > 32 ? (pc[0] + pc[1]) + (pc[2] + pc[3]) : pc[0];
        a[k] = -0.5 * (p->cnorm[k] + chi2);
        if (a[k] > amax) amax = a[k];
    }
    if (K == 1) { *ll_out = a[0]; return 1; }
    double S = 0.0;
    for (int k = 0; k < K; ++k) S = fma(p->weight[k], orc_dexp(a[k] - amax), S);
    *ll_out = orc_dlog(S) + amax;
    return 1;
}

void orc_evaluate(const orc_problem* p, int n, const double* x, double* logprior,
                  double* loglike, double* derived)
{
    int d = p->d, K = p->n_modes;
    for (int w = 0; w < n; ++w)
        eval_point(p, x + (size_t)w * d, logprior + w, loglike + w,
                   derived ? derived + (size_t)w * K * d : NULL);
}

/* ------------------------------------------------------------------ walker state */
typedef struct {
    double* x;         /* [W][d] walker-major */
    double* logprior;  /* [W] */
    double* loglike;   /* [W] */
    double* logpost;   /* [W] */
    int32_t* weight;   /* [W] multiplicity of the current point (collection.py:1353-1383) */
    int32_t* prior_rej;/* [W] mcmc.py:712-713 */
    int32_t* burn_left;/* [W] mcmc.py:265 */
    int64_t* n_accept; /* [W] accepted steps */
    int32_t* stuck;    /* [1] set to 1 + walker if a walker trips max_tries (mcmc.py:717-743) */
    /* optional emission of accepted rows (mcmc.py:691-707): rows[W][cap][d+4] =
     * (weight, logpost, logprior, loglike, x...), n_rows[W] */
    double* rows; int32_t* n_rows; int32_t row_cap;
    double* y;         /* [W][K][d] incremental mode: L_k^-1 (x - mu_k) of the current point */
    double* amode;     /* [W][K] carry_modes: the log-density -(c_k + chi2_k) / 2 of every mode */
    /* thinned emission (collection.py:1373-1383, OneSamplePoint.add_to_collection with
     * output_thin > 1): thin <= 1: every accepted row is written with its weight; else the weights
     * of a walker accumulate in thin_acc[W] and a row is written when the sum reaches `thin`, with
     * weight sum / thin, the remainder carried */
    int32_t thin; int32_t* thin_acc;
} orc_state;

/* the Metropolis bookkeeping shared by the Philox and the injected drivers */
static inline void commit(const orc_problem* p, orc_state* st, int w, const double* t,
                          int inb, double lp, double ll, double lt, int accept)
{
    int d = p->d;
    double* x = st->x + (size_t)w * d;
    if (accept) {
        if (st->burn_left[w] <= 0) {
            if (st->rows) {
                int32_t ew = st->weight[w];
                if (st->thin > 1 && st->thin_acc) {
                    const int32_t tot = st->thin_acc[w] + st->weight[w];
                    ew = tot / st->thin;
                    st->thin_acc[w] = tot % st->thin;
                }
                if (ew > 0) {
                    if (st->n_rows[w] < st->row_cap) {
                        double* row = st->rows + ((size_t)w * st->row_cap + st->n_rows[w]) * (d + 4);
                        row[0] = (double)ew; row[1] = st->logpost[w];
                        row[2] = st->logprior[w]; row[3] = st->loglike[w];
                        for (int i = 0; i < d; ++i) row[4 + i] = x[i];
                    }
                    st->n_rows[w] += 1; /* rows beyond the capacity are counted as dropped */
                }
            }
        } else {
            st->burn_left[w] -= 1;
        }
        for (int i = 0; i < d; ++i) x[i] = t[i];
        st->logprior[w] = lp; st->loglike[w] = ll; st->logpost[w] = lt;
        st->weight[w] = 1; st->prior_rej[w] = 0; st->n_accept[w] += 1;
    } else {
        st->weight[w] += 1;
        if (!inb) st->prior_rej[w] += 1;
        double max_now = p->max_tries * (st->burn_left[w] > 0 ? 10.0 : 1.0);
        if ((double)(st->weight[w] - st->prior_rej[w]) > max_now && st->stuck && !*st->stuck)
            *st->stuck = 1 + w;
    }
}

/* one step of walker w with the proposal increment `delta` and the Exp(1) variate
 * `exp_draw` for the accept test supplied by the caller */
static inline int step_core(const orc_problem* p, orc_state* st, int w, const double* delta,
                            double r, double exp_draw)
{
    int d = p->d;
    double t[128];
    const double* x = st->x + (size_t)w * d;
    for (int i = 0; i < d; ++i) t[i] = fma(r, delta[i], x[i]);
    if (p->has_periodic)
        for (int i = 0; i < d; ++i)
            if (p->periodic[i]) t[i] = wrap_periodic(t[i], p->lo[i], p->hi[i]);
    double lp, ll;
    int inb = eval_point(p, t, &lp, &ll, NULL);
    double lt = inb ? lp + ll : -INFINITY;
    int accept;
    if (!inb || lt == -INFINITY) accept = 0;
    else if (lt > st->logpost[w]) accept = 1;
    else accept = exp_draw > (st->logpost[w] - lt) / p->temperature;
    commit(p, st, w, t, inb, lp, ll, lt, accept);
    return accept;
}

/* ---- incremental evaluation (Gaussian modes; periodic parameters: step_core_inc) ---------
 * y[k*d + j] = sum_{i<=j} Linv_k[j][i] (x_i - mu_k,i), ascending fma chain from +0.0 (==
 * `derived` of eval_point): what a walker carries per mode, recomputed at every step s with
 * s % refresh_every == 0 */
void orc_whiten(const orc_problem* p, const double* x, double* y)
{
    int d = p->d, K = p->n_modes;
    for (int k = 0; k < K; ++k) {
        const double* Li = p->Linv + (size_t)k * d * d;
        const double* mu = p->mean + (size_t)k * d;
        for (int j = 0; j < d; ++j) {
            double a = 0.0;
            for (int i = 0; i <= j; ++i) a = fma(Li[j * d + i], x[i] - mu[i], a);
            y[k * d + j] = a;
        }
    }
}

/* U[(k*ncol + c)*d + j] = sum_{i<=j} Linv_k[j][i] V[c*d + i]: the whitened proposal directions
 * of a cycle, per mode */
void orc_whiten_directions(const orc_problem* p, int ncol, const double* V, double* U)
{
    int d = p->d, K = p->n_modes;
    for (int k = 0; k < K; ++k) {
        const double* Li = p->Linv + (size_t)k * d * d;
        for (int c = 0; c < ncol; ++c)
            for (int j = 0; j < d; ++j) {
                double a = 0.0;
                for (int i = 0; i <= j; ++i) a = fma(Li[j * d + i], V[c * d + i], a);
                U[((size_t)k * ncol + c) * d + j] = a;
            }
    }
}

/* sum of squares in the pattern of every chi2 here: four interleaved chains over i & 3, combined
 * (s0 + s1) + (s2 + s3) */
static inline double four_chain_squares(const double* a, int d)
{
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < d; ++i) s[i & 3] = fma(a[i], a[i], s[i & 3]);
    return (s[0] + s[1]) + (s[2] + s[3]);
}

/* |u_c|^2 of the whitened directions of a cycle (one mode), in that pattern */
void orc_direction_norms(const orc_problem* p, int ncol, const double* U, double* UU)
{
    for (int c = 0; c < ncol; ++c) UU[c] = four_chain_squares(U + (size_t)c * p->d, p->d);
}

/* ONE Gaussian mode, no periodic parameter (step_inc_kernel; round 4): the log-likelihood itself
 * is carried.  chi2(y + r u) - chi2(y) = r (2 y.u + r |u|^2), so the trial's
 *     ll_t = fma(-0.5 r, fma(r, |u|^2, y.u + y.u), ll)
 * with y.u in the four-chain pattern and |u|^2 formed once per direction (orc_direction_norms) --
 * one chain over the dimensions per trial instead of two (d fewer FP64 instructions of the ~ 60
 * a step issues per lane), and a difference that is formed directly instead of as the difference
 * of two sums.  Like y, ll follows the rounding of its own updates and is re-anchored where y is
 * refreshed: orc_anchor_loglike. */
static inline int carries_loglike(const orc_problem* p, const orc_state* st)
{
    if (!(p->incremental && p->n_modes == 1 && (!p->has_periodic || p->carry_periodic))) return 0;
    /* (emitted rows with a one-parameter block run on the general kernel, incremental_any.hip,
     * which forms every chi2 from the trial's residual) */
    if (st->rows && p->blocking)
        for (int b = 0; b < p->blocking->n_blocks; ++b)
            if (p->blocking->size[b] == 1) return 0;
    return 1;
}

/* ... and with NORMAL PRIORS the log-prior as well (step_inc_kernel MODE 2; round 5):
 *     lp(x + r v) - lp(x) = -1/2 sum_i ((t_i - loc_i)^2 - (x_i - loc_i)^2) / s_i^2
 *                         = -r/2 (r v.w + 2 (x.w - loc.w)),      w_i = (v_i / s_i) / s_i,
 * so the trial's lp_t = fma(-0.5 r, fma(r, v.w, xw + xw), lp), xw = x.w - loc.w, with w, v.w and
 * loc.w formed once per direction (orc_direction_prior) and x.w in the four-chain pattern: one fma
 * per dimension and trial instead of a subtraction, two products, an fma and an addition.  Like
 * the log-likelihood it is re-anchored (on x) wherever y is refreshed: orc_anchor_loglike.
 * A periodic parameter has a uniform prior, w_i = 0: a wrap of its coordinate moves nothing here.
 * The rule is the engine's (capi.hip: inc_carries_prior). */
static inline int carries_prior(const orc_problem* p, const orc_state* st)
{
    if (!carries_loglike(p, st)) return 0;   /* (with periodic parameters: carry_periodic) */
    if (p->blocking && p->blocking->drag_last_slow >= 0) return 0;
    for (int i = 0; i < p->d; ++i)
        if (p->kind[i] == 1) return 1;
    return 0;
}

/* the normal terms of the log-prior in incremental mode: four chains over i mod 4 */
static inline double inc_logprior(const orc_problem* p, const double* t)
{
    double sc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < p->d; ++i)
        if (p->kind[i] == 1) {
            /* (multiplication by the reciprocal of the scale, formed once: 1 ulp from the
             * division of eval_point, a fifth of its instructions) */
            double q = (t[i] - p->loc[i]) * (1.0 / p->scale[i]);
            sc[i & 3] = sc[i & 3] + fma(-0.5 * q, q, p->mls[i]);
        }
    return p->uniform_logp + ((sc[0] + sc[1]) + (sc[2] + sc[3]));
}

/* Wd[c*d + i] = w_i of column c, NL[2c] = v.w, NL[2c + 1] = loc.w (four chains over i mod 4) */
void orc_direction_prior(const orc_problem* p, int ncol, const double* V, double* Wd, double* NL)
{
    const int d = p->d;
    for (int c = 0; c < ncol; ++c) {
        double nn[4] = {0.0, 0.0, 0.0, 0.0}, lw[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < d; ++i) {
            const double inv = p->kind[i] == 1 ? 1.0 / p->scale[i] : 0.0;
            const double v = V[(size_t)c * d + i], w = (v * inv) * inv;
            Wd[(size_t)c * d + i] = w;
            nn[i & 3] = fma(v, w, nn[i & 3]);
            if (p->kind[i] == 1) lw[i & 3] = fma(p->loc[i], w, lw[i & 3]);
        }
        NL[2 * c] = (nn[0] + nn[1]) + (nn[2] + nn[3]);
        NL[2 * c + 1] = (lw[0] + lw[1]) + (lw[2] + lw[3]);
    }
}

void orc_anchor_loglike(const orc_problem* p, orc_state* st, int w)
{
    const double* y = st->y + (size_t)w * p->d;
    st->loglike[w] = -0.5 * (p->cnorm[0] + four_chain_squares(y, p->d));
    if (carries_prior(p, st)) st->logprior[w] = inc_logprior(p, st->x + (size_t)w * p->d);
    st->logpost[w] = st->logprior[w] + st->loglike[w];
}

/* MIXTURES with carried mode log-densities (round 5; step_inc_mix_kernel, step_inc_regs_kernel
 * without periodic parameters).  Per mode chi2_k(y_k + r u_k) - chi2_k(y_k) =
 * r (2 y_k.u_k + r |u_k|^2), so
 *     a_k' = fma(-0.5 r, fma(r, |u_k|^2, y_k.u_k + y_k.u_k), a_k),   a_k = -(c_k + chi2_k) / 2
 * with y_k.u_k in the four-chain pattern and |u_k|^2 formed once per (direction, mode)
 * (orc_direction_norms) -- one chain over the dimensions per mode and trial instead of two --, then
 * the log-sum-exp of eval_point on the a_k'.  The a_k follow the rounding of their own updates like
 * y_k and are re-anchored, with loglike and logpost, wherever y is refreshed (orc_anchor_modes). */
static inline int carries_modes(const orc_problem* p)
{
    return p->incremental && p->n_modes > 1 && p->carry_modes && !p->has_periodic;
}

static inline double mixture_lse(const orc_problem* p, const double* a)
{
    const int K = p->n_modes;
    double amax = -INFINITY, S = 0.0;
    for (int k = 0; k < K; ++k)
        if (a[k] > amax) amax = a[k];
    for (int k = 0; k < K; ++k) S = fma(p->weight[k], orc_dexp_tab(a[k] - amax), S);
    return orc_dlog_tab(S) + amax;
}

void orc_anchor_modes(const orc_problem* p, orc_state* st, int w)
{
    const int K = p->n_modes, d = p->d;
    double* a = st->amode + (size_t)w * K;
    for (int k = 0; k < K; ++k)
        a[k] = -0.5 * (p->cnorm[k] + four_chain_squares(st->y + ((size_t)w * K + k) * d, d));
    st->loglike[w] = mixture_lse(p, a);
    st->logpost[w] = st->logprior[w] + st->loglike[w];
}

/* One step in incremental mode.  Trial t = x + r v and, per mode, its whitened residual
 * yt_k = y_k + r u_k; prior terms and every chi2_k are summed as FOUR interleaved chains over the
 * dimensions i = c (mod 4) (the kernel keeps dimension i in lane i mod 4 of the walker's quad),
 * combined (s0 + s1) + (s2 + s3) -- for every d in this mode.  u: [K] pointers to the mode's
 * whitened direction of this step.  K > 1: log-sum-exp as in eval_point. */
static inline int step_core_inc(const orc_problem* p, orc_state* st, int w, const double* v,
                                const double* const* u, double uu, const double* uu_k,
                                const double* wp, const double* nl, double r, double exp_draw)
{
    int d = p->d, K = p->n_modes;
    const int carry = carries_loglike(p, st);
    const int carry_p = wp != NULL;   /* (carries_prior: the caller formed w, v.w and loc.w) */
    const int carry_k = carries_modes(p);
    double t[128], yt[64 * 128], sh[128];   /* (64: the engine's kMaxModes) */
    const double* x = st->x + (size_t)w * d;
    double* y = st->y + (size_t)w * K * d;
    int inb = 1, wound = 0;
    for (int i = 0; i < d; ++i) {
        t[i] = fma(r, v[i], x[i]);
        sh[i] = 0.0;
        if (p->has_periodic && p->periodic[i] &&
            !(carry && t[i] >= p->lo[i] && t[i] < p->hi[i])) {
            /* (carry_periodic: only a coordinate that left [lo, hi) is wrapped)
             * prior.py:675 (wrap_periodic, spelled out): the coordinate is the wrapped one; when
             * the winding number changes (floor != 0) the move of the coordinate, sh = t' - t, is
             * carried into the whitened residual below -- otherwise t' differs from t by the
             * rounding of the wrap alone, which y does not follow (as it does not follow the
             * rounding of its own updates: bounded by the refresh every refresh_every steps) */
            double wd = p->hi[i] - p->lo[i];
            double yv = (t[i] - p->lo[i]) / wd;
            double fl = floor(yv);
            double tw = (yv - fl) * wd + p->lo[i];
            if (fl != 0.0) { sh[i] = tw - t[i]; wound |= sh[i] != 0.0; }
            t[i] = tw;
        }
        inb &= (t[i] <= p->hi[i]) & (t[i] >= p->lo[i]);
    }
    double lp = -INFINITY, ll = -INFINITY, lt = -INFINITY;
    double a_new[64];   /* carry_modes: the trial's mode log-densities */
    if (inb) {
        if (carry_p) {   /* the carried log-prior moves along the direction */
            double sc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < d; ++i) sc[i & 3] = fma(x[i], wp[i], sc[i & 3]);
            const double xw = ((sc[0] + sc[1]) + (sc[2] + sc[3])) - nl[1];
            lp = fma(-0.5 * r, fma(r, nl[0], xw + xw), st->logprior[w]);
        } else {
            lp = inc_logprior(p, t);
        }
        double a[64], amax = -INFINITY;
        if (carry) {
            double q[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < d; ++i) q[i & 3] = fma(y[i], u[0][i], q[i & 3]);
            const double yu = (q[0] + q[1]) + (q[2] + q[3]);
            a[0] = fma(-0.5 * r, fma(r, uu, yu + yu), st->loglike[w]);
            for (int i = 0; i < d; ++i) yt[i] = fma(r, u[0][i], y[i]);
            if (wound) {   /* a wrap: the residual takes the moves, chi2 is summed from it */
                const double* Li = p->Linv;
                for (int i = 0; i < d; ++i)
                    if (sh[i] != 0.0)
                        for (int j = i; j < d; ++j) yt[j] = fma(sh[i], Li[j * d + i], yt[j]);
                a[0] = -0.5 * (p->cnorm[0] + four_chain_squares(yt, d));
            }
        }
        if (carry_k) {   /* (uu: |u_k|^2 of this step's direction, mode k at uu_k[k]) */
            const double* am = st->amode + (size_t)w * K;
            for (int k = 0; k < K; ++k) {
                double q[4] = {0.0, 0.0, 0.0, 0.0};
                for (int i = 0; i < d; ++i) q[i & 3] = fma(y[k * d + i], u[k][i], q[i & 3]);
                const double yu = (q[0] + q[1]) + (q[2] + q[3]);
                a[k] = fma(-0.5 * r, fma(r, uu_k[k], yu + yu), am[k]);
                for (int i = 0; i < d; ++i) yt[k * d + i] = fma(r, u[k][i], y[k * d + i]);
            }
            ll = mixture_lse(p, a);
            for (int k = 0; k < K; ++k) a_new[k] = a[k];
        }
        for (int k = 0; k < K && !carry && !carry_k; ++k) {
            double pc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < d; ++i) yt[k * d + i] = fma(r, u[k][i], y[k * d + i]);
            if (wound) {   /* a wrap by sh_i moves the residual by sh_i (column i of L^-1) */
                const double* Li = p->Linv + (size_t)k * d * d;
                for (int i = 0; i < d; ++i)
                    if (sh[i] != 0.0)
                        for (int j = i; j < d; ++j)
                            yt[k * d + j] = fma(sh[i], Li[j * d + i], yt[k * d + j]);
            }
            for (int i = 0; i < d; ++i)
                pc[i & 3] = fma(yt[k * d + i], yt[k * d + i], pc[i & 3]);
            a[k] = -0.5 * (p->cnorm[k] + ((pc[0] + pc[1]) + (pc[2] + pc[3])));
            if (a[k] > amax) amax = a[k];
        }
        if (carry_k) { /* (formed above) */ }
        else if (K == 1) ll = a[0];
        else {
            /* (incremental mode: the table-driven exp / log, as in mixture_lse) */
            double S = 0.0;
            for (int k = 0; k < K; ++k) S = fma(p->weight[k], orc_dexp_tab(a[k] - amax), S);
            ll = orc_dlog_tab(S) + amax;
        }
        lt = lp + ll;
    }
    int accept;
    if (!inb || lt == -INFINITY) accept = 0;
    else if (lt > st->logpost[w]) accept = 1;
    else accept = exp_draw > (st->logpost[w] - lt) / p->temperature;
    if (accept) {
        for (int i = 0; i < K * d; ++i) y[i] = yt[i];
        if (carry_k)
            for (int k = 0; k < K; ++k) st->amode[(size_t)w * K + k] = a_new[k];
    }
    commit(p, st, w, t, inb, lp, ll, lt, accept);
    return accept;
}

/* Tier-A link: walker 0 advances by x += T vec (vec = R[:,i] r scale as drawn by the
 * reference, proposal.py:69) with the reference's own accept variate (NaN if not drawn) */
int orc_step_injected(const orc_problem* p, orc_state* st, const double* vec, double exp_draw)
{
    int d = p->d;
    double delta[128];
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s = fma(p->T[i * d + k], vec[k], s);
        delta[i] = s;
    }
    return step_core(p, st, 0, delta, 1.0, exp_draw);
}

/* ------------------------------------------------------------------ the ensemble driver */
/* Random variates of (walker, step, sub): one Philox block on the counter
 * (walker, STREAM_STEP | sub << 16, step).  r = radial part of the proposal (proposal.py:71-82:
 * Exp(1) w.p. 0.33 else chi(min(n, 2))), Ea = Exp(1) variate of the accept test
 * (mcmc.py:683).  oned: the block has one parameter (proposal.py:85-93): chi(1) = sqrt(2E)
 * |cos| of a Box-Muller pair, random sign, and Ea from a second block (| 0x100).  sub = 0 is
 * the step itself, 1..n the interpolation steps of a dragging step. */
static inline void walker_variates(uint32_t k0, uint32_t k1, uint32_t gid, uint64_t step,
                                   uint32_t sub, int oned, double* r_out, double* Ea_out)
{
    uint32_t wd[4];
    uint32_t c1 = STREAM_STEP | (sub << 16);
    philox4x32_10(k0, k1, gid, c1, (uint32_t)step, (uint32_t)(step >> 32), wd);
    uint64_t kr = ((uint64_t)wd[1] << 20) | (wd[2] >> 12);
    uint64_t ka = ((uint64_t)wd[3] << 20) | ((uint64_t)(wd[2] & 0xFFFu) << 8) | (wd[0] & 0xFFu);
    double Er = -orc_dlog(u52(kr));
    if (oned) {
        double sn, cs;
        orc_sincos2pi(ka, &sn, &cs);
        double rr = ((wd[0] >> 8) < BRANCH_EXP_24) ? Er : sqrt(2.0 * Er) * fabs(cs);
        *r_out = (wd[0] & 0x80u) ? rr : -rr;
        uint32_t w2[4];
        philox4x32_10(k0, k1, gid, c1 | 0x100u, (uint32_t)step, (uint32_t)(step >> 32), w2);
        *Ea_out = -orc_dlog(u52(((uint64_t)w2[0] << 20) | (w2[1] >> 12)));
    } else {
        /* every walker takes its OWN sign (bit 7 of w0, as for one-parameter blocks): the
         * basis column v is shared by the walkers of a group, and for a FIXED v the move
         * x + r v with r > 0 is not a symmetric proposal -- only its average over v and -v
         * is.  With the private sign each walker's kernel is symmetric, hence pi-invariant,
         * for every fixed basis, so the walkers of a group are independent chains given the
         * bases (without it they drift together: a group was worth ~4 walkers, see DESIGN) */
        double rr = ((wd[0] >> 8) < BRANCH_EXP_24) ? Er : sqrt(2.0 * Er);
        *r_out = (wd[0] & 0x80u) ? rr : -rr;
        *Ea_out = -orc_dlog(u52(ka));
    }
}

/* Incremental mode, plain steps: the variates of the steps 2P and 2P + 1 share the Philox block
 * (walker, STREAM_STEP | 0x4000, P).  Half h = step & 1 uses the words a = w[2h], b = w[2h+1]:
 * sign = bit 31 of a (set = positive); exponential branch iff bits 30..20 of a < 676
 * (676 / 2048 = 0.33008, proposal.py:79); k_r = (a & 0xFFFFF) << 4 | b >> 28 (24 bits),
 * u_r = (2 k_r + 1) 2^-25; k_a = b & 0xFFFFFFF (28 bits), u_a = (2 k_a + 1) 2^-29.  The two
 * logarithms are short-argument ones (orc_neg_log_short). */
void orc_pair_variates(uint64_t seed, uint32_t gid, uint64_t step, double* r_out, double* Ea_out);
/* The far tail of a short uniform: a variate whose 24- / 28-bit uniform fell into the LOWEST bin
 * stands for u in (0, 2^-b) and is redrawn there at full width, u = 2^-b u', u' = u52 of the block
 * (walker, STREAM_STEP | 0x8000 | which << 13, step): -log u = fma(b, ln 2, -log u'). */
static inline double pair_tail(uint32_t k0, uint32_t k1, uint32_t gid, uint64_t step, uint32_t which,
                               double bits)
{
    static const double LN2 = 6.93147180559945286227e-01;
    uint32_t wd[4];
    philox4x32_10(k0, k1, gid, STREAM_STEP | 0x8000u | (which << 13), (uint32_t)step,
                  (uint32_t)(step >> 32), wd);
    return fma(bits, LN2, -orc_dlog(u52(((uint64_t)wd[0] << 20) | (wd[1] >> 12))));
}

static inline void walker_variates_pair(uint32_t k0, uint32_t k1, uint32_t gid, uint64_t step,
                                        double* r_out, double* Ea_out)
{
    uint32_t wd[4];
    const uint64_t P = step >> 1;
    philox4x32_10(k0, k1, gid, STREAM_STEP | 0x4000u, (uint32_t)P, (uint32_t)(P >> 32), wd);
    const uint32_t a = wd[2 * (step & 1)], b = wd[2 * (step & 1) + 1];
    const uint32_t kr = ((a & 0xFFFFFu) << 4) | (b >> 28);
    const uint32_t ka = b & 0x0FFFFFFFu;
    const double Er = kr ? orc_neg_log_short(2 * kr + 1, 25) : pair_tail(k0, k1, gid, step, 0, 24.0);
    const double rr = (((a >> 20) & 0x7FFu) < 676u) ? Er : sqrt(2.0 * Er);
    *r_out = (a & 0x80000000u) ? rr : -rr;
    *Ea_out = ka ? orc_neg_log_short(2 * ka + 1, 29) : pair_tail(k0, k1, gid, step, 1, 28.0);
}

void orc_pair_variates(uint64_t seed, uint32_t gid, uint64_t step, double* r_out, double* Ea_out)
{
    walker_variates_pair((uint32_t)seed, (uint32_t)(seed >> 32), gid, step, r_out, Ea_out);
}

/* Test helper: the first (walker, step) with walker in [gid0, gid0 + n_walkers) and step in
 * [step0, step0 + n_steps) whose short radial (which = 0: 24 bits) or accept (which = 1: 28 bits)
 * uniform fell into the lowest bin -- the draws pair_tail serves.  Returns 1 if found. */
int orc_find_short_tail(uint64_t seed, uint32_t gid0, uint32_t n_walkers, uint64_t step0,
                        uint64_t n_steps, int32_t which, uint32_t* gid_out, uint64_t* step_out)
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (uint64_t P = step0 >> 1; 2 * P < step0 + n_steps; ++P)
        for (uint32_t w = 0; w < n_walkers; ++w) {
            uint32_t wd[4];
            philox4x32_10(k0, k1, gid0 + w, STREAM_STEP | 0x4000u, (uint32_t)P, (uint32_t)(P >> 32), wd);
            for (int h = 0; h < 2; ++h) {
                const uint64_t step = 2 * P + (uint64_t)h;
                if (step < step0 || step >= step0 + n_steps) continue;
                const uint32_t a = wd[2 * h], b = wd[2 * h + 1];
                const uint32_t kr = ((a & 0xFFFFFu) << 4) | (b >> 28), ka = b & 0x0FFFFFFFu;
                if ((which == 0 && kr == 0) || (which == 1 && ka == 0)) {
                    *gid_out = gid0 + w; *step_out = step;
                    return 1;
                }
            }
        }
    return 0;
}

/* mcmc.py:670-683 with the Exp(1) variate supplied */
static inline int metropolis(double trial, double current, double T, double exp_draw)
{
    if (trial == -INFINITY) return 0;
    if (trial > current) return 1;
    return exp_draw > (current - trial) / T;
}

/* One dragging step of walker w (mcmc.py:564-668): vs = slow direction, vf[i] = the fast
 * directions of the n interpolation steps, variates r[0..n], Ea[0..n] (index 0: the slow
 * proposal and the final test). */
static int drag_core(const orc_problem* p, orc_state* st, int w, const double* vs,
                     const double* const* vf, const double* r, const double* Ea)
{
    int d = p->d, n = p->blocking->drag_steps;
    double cs[128], ce[128], t[128];
    const double* x = st->x + (size_t)w * d;
    for (int i = 0; i < d; ++i) { cs[i] = x[i]; ce[i] = fma(r[0], vs[i], x[i]); }
    if (p->has_periodic)
        for (int i = 0; i < d; ++i)
            if (p->periodic[i]) ce[i] = wrap_periodic(ce[i], p->lo[i], p->hi[i]);
    double cs_lt = st->logpost[w];
    double ce_lp, ce_ll;
    int inb = eval_point(p, ce, &ce_lp, &ce_ll, NULL);
    double ce_lt = inb ? ce_lp + ce_ll : -INFINITY;
    if (ce_lt == -INFINITY) { st->weight[w] += 1; return 0; }   /* mcmc.py:590-592 */
    double start_acc = cs_lt, end_acc = ce_lt;
    for (int i = 1; i <= n; ++i) {
        double delta[128];
        for (int k = 0; k < d; ++k) delta[k] = r[i] * vf[i - 1][k];
        if (p->has_periodic)   /* the reference wraps the DELTA (mcmc.py:606) */
            for (int k = 0; k < d; ++k)
                if (p->periodic[k]) delta[k] = wrap_periodic(delta[k], p->lo[k], p->hi[k]);
        for (int k = 0; k < d; ++k) t[k] = cs[k] + delta[k];
        double ps_lp, ps_ll;
        int in_s = eval_point(p, t, &ps_lp, &ps_ll, NULL);
        double ps_lt = in_s ? ps_lp + ps_ll : -INFINITY;
        if (ps_lt != -INFINITY) {
            double te[128];
            for (int k = 0; k < d; ++k) te[k] = ce[k] + delta[k];
            double pe_lp, pe_ll;
            int in_e = eval_point(p, te, &pe_lp, &pe_ll, NULL);
            double pe_lt = in_e ? pe_lp + pe_ll : -INFINITY;
            if (pe_lt != -INFINITY) {
                double frac = (double)i / (double)(1 + n);
                double pi = (1.0 - frac) * ps_lt + frac * pe_lt;
                double ci = (1.0 - frac) * cs_lt + frac * ce_lt;
                if (metropolis(pi, ci, p->temperature, Ea[i])) {
                    for (int k = 0; k < d; ++k) { cs[k] = t[k]; ce[k] = te[k]; }
                    cs_lt = ps_lt;
                    ce_lp = pe_lp; ce_ll = pe_ll; ce_lt = pe_lt;
                }
            }
        }
        start_acc += cs_lt;
        end_acc += ce_lt;
    }
    double navg = (double)(1 + n);
    int accept = metropolis(end_acc / navg, start_acc / navg, p->temperature, Ea[0]);
    commit(p, st, w, ce, 1, ce_lp, ce_ll, ce_lt, accept);
    return accept;
}

/* Advance walkers [0, W) (global ids walker0 + w, groups of p->group_size) by n_steps
 * steps starting at global step index step0.  Returns total accepts. */
/* log-posterior of a point given its whitened residual (incremental mode, one Gaussian mode):
 * prior support and normal terms from t, chi2 from yt, both as four interleaved chains */
static inline double eval_inc(const orc_problem* p, const double* t, const double* yt, double* lp_out,
                              double* ll_out)
{
    int d = p->d, inb = 1;
    for (int i = 0; i < d; ++i) inb &= (t[i] <= p->hi[i]) & (t[i] >= p->lo[i]);
    if (!inb) { *lp_out = -INFINITY; *ll_out = -INFINITY; return -INFINITY; }
    double sc[4] = {0.0, 0.0, 0.0, 0.0}, pc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < d; ++i) {
        if (p->kind[i] == 1) {
            double q = (t[i] - p->loc[i]) * (1.0 / p->scale[i]);
            sc[i & 3] = sc[i & 3] + fma(-0.5 * q, q, p->mls[i]);
        }
        pc[i & 3] = fma(yt[i], yt[i], pc[i & 3]);
    }
    *lp_out = p->uniform_logp + ((sc[0] + sc[1]) + (sc[2] + sc[3]));
    *ll_out = -0.5 * (p->cnorm[0] + ((pc[0] + pc[1]) + (pc[2] + pc[3])));
    return *lp_out + *ll_out;
}

/* One dragging step in incremental mode (one Gaussian mode, non-periodic): drag_core with the
 * whitened residuals of the start and end points carried along -- us / uf = L^-1 of the slow /
 * fast directions (U[0], U[1..n]).  A point moved by delta = r v has its residual moved by
 * fma(r, u, y). */
static int drag_core_inc(const orc_problem* p, orc_state* st, int w, const double* vs,
                         const double* const* vf, const double* U, const double* r,
                         const double* Ea)
{
    int d = p->d, n = p->blocking->drag_steps;
    double cs[128], ce[128], ys[128], ye[128], t[128], te[128], yst[128], yet[128];
    const double* x = st->x + (size_t)w * d;
    double* y = st->y + (size_t)w * d;
    for (int i = 0; i < d; ++i) {
        cs[i] = x[i]; ce[i] = fma(r[0], vs[i], x[i]);
        ys[i] = y[i]; ye[i] = fma(r[0], U[i], y[i]);
    }
    double cs_lt = st->logpost[w];
    double ce_lp, ce_ll;
    double ce_lt = eval_inc(p, ce, ye, &ce_lp, &ce_ll);
    if (ce_lt == -INFINITY) { st->weight[w] += 1; return 0; }   /* mcmc.py:590-592 */
    double start_acc = cs_lt, end_acc = ce_lt;
    for (int i = 1; i <= n; ++i) {
        const double* uf = U + (size_t)i * d;
        for (int k = 0; k < d; ++k) {
            double delta = r[i] * vf[i - 1][k];
            t[k] = cs[k] + delta;
            te[k] = ce[k] + delta;
            yst[k] = fma(r[i], uf[k], ys[k]);
            yet[k] = fma(r[i], uf[k], ye[k]);
        }
        double ps_lp, ps_ll, pe_lp, pe_ll;
        double ps_lt = eval_inc(p, t, yst, &ps_lp, &ps_ll);
        double pe_lt = eval_inc(p, te, yet, &pe_lp, &pe_ll);
        if (ps_lt != -INFINITY && pe_lt != -INFINITY) {
            double frac = (double)i / (double)(1 + n);
            double pi = (1.0 - frac) * ps_lt + frac * pe_lt;
            double ci = (1.0 - frac) * cs_lt + frac * ce_lt;
            if (metropolis(pi, ci, p->temperature, Ea[i])) {
                for (int k = 0; k < d; ++k) { cs[k] = t[k]; ce[k] = te[k]; ys[k] = yst[k]; ye[k] = yet[k]; }
                cs_lt = ps_lt;
                ce_lp = pe_lp; ce_ll = pe_ll; ce_lt = pe_lt;
            }
        }
        start_acc += cs_lt;
        end_acc += ce_lt;
    }
    double navg = (double)(1 + n);
    int accept = metropolis(end_acc / navg, start_acc / navg, p->temperature, Ea[0]);
    if (accept)
        for (int k = 0; k < d; ++k) y[k] = ye[k];
    commit(p, st, w, ce, 1, ce_lp, ce_ll, ce_lt, accept);
    return accept;
}

int64_t orc_run(const orc_problem* p, orc_state* st, int W, uint32_t walker0, uint64_t step0,
                int n_steps, int n_threads)
{
    int d = p->d, gs = p->group_size;
    int G = W / gs;
    uint32_t k0 = (uint32_t)p->seed, k1 = (uint32_t)(p->seed >> 32);
    const orc_blocking* B = p->blocking;
    const int drag = B && B->drag_last_slow >= 0;
    const int L0 = orc_block_slots(p, drag ? 1 : 0, NULL);    /* steps per cycle */
    const int Lf = drag ? orc_block_slots(p, 2, NULL) : 0;
    const int nd = drag ? B->drag_steps : 0;
    int64_t total = 0;
    /* Wide basis groups (4 096 walkers at the benchmark geometry) leave fewer groups than
     * threads: a group is then cut into `nsub` runs of walkers, each of which forms the group's
     * bases itself -- they are pure functions of (group, cycle), so the results are the same
     * bit for bit (tests/test_oracle_c.py::test_run_is_invariant_to_the_thread_split). */
    int nsub = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
    {
        int want = n_threads > 0 ? n_threads : omp_get_max_threads();
        while (G * nsub < want && gs % (2 * nsub) == 0 && gs / (2 * nsub) >= 64) nsub *= 2;
    }
#pragma omp parallel for schedule(static) reduction(+ : total)
#endif
    for (int gi = 0; gi < G * nsub; ++gi) {
        const int g = gi / nsub;
        const int l_lo = (gi % nsub) * (gs / nsub), l_hi = l_lo + gs / nsub;
        uint32_t group = walker0 / (uint32_t)gs + (uint32_t)g;
        double* V = (double*)malloc(sizeof(double) * (size_t)L0 * d);
        int32_t* f1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(L0 + Lf + 1));
        double* Vf = drag ? (double*)malloc(sizeof(double) * (size_t)Lf * d) : NULL;
        double* Vstep = drag ? (double*)malloc(sizeof(double) * (size_t)nd * d) : NULL;
        int32_t* f1f = f1 + L0;
        uint64_t have_cycle = UINT64_MAX, have_f[2] = {UINT64_MAX, UINT64_MAX};
        uint64_t have_u = UINT64_MAX;
        double* U = NULL;
        double* Wd = NULL;   /* carries_prior: w of the cycle's columns, then (v.w, loc.w) */
        for (int s = 0; s < n_steps; ++s) {
            uint64_t step = step0 + (uint64_t)s;
            uint64_t cycle = step / (uint64_t)L0;
            int col = (int)(step % (uint64_t)L0);
            if (cycle != have_cycle) {
                orc_basis_blocked(p, group, (uint32_t)cycle, drag ? 1 : 0, V, f1);
                have_cycle = cycle;
            }
            const double* v = V + (size_t)col * d;
            if (p->incremental && !drag) {
                const int K = p->n_modes;
                if (cycle != have_u) {
                    if (!U) U = (double*)malloc(sizeof(double) * ((size_t)K * L0 * d + (size_t)K * L0));
                    orc_whiten_directions(p, L0, V, U);
                    if (carries_loglike(p, st)) orc_direction_norms(p, L0, U, U + (size_t)K * L0 * d);
                    if (carries_modes(p))   /* |u_k|^2 of column c at [k * L0 + c] */
                        for (int k = 0; k < K; ++k)
                            orc_direction_norms(p, L0, U + (size_t)k * L0 * d,
                                                U + (size_t)K * L0 * d + (size_t)k * L0);
                    if (carries_prior(p, st)) {
                        if (!Wd) Wd = (double*)malloc(sizeof(double) * ((size_t)L0 * d + 2 * (size_t)L0));
                        orc_direction_prior(p, L0, V, Wd, Wd + (size_t)L0 * d);
                    }
                    have_u = cycle;
                }
                const double* wp = carries_prior(p, st) ? Wd + (size_t)col * d : NULL;
                const double* nl = carries_prior(p, st) ? Wd + (size_t)L0 * d + 2 * col : NULL;
                const double uu = carries_loglike(p, st) ? U[(size_t)K * L0 * d + col] : 0.0;
                const double* uk[64];
                double uuk[64];
                for (int k = 0; k < K; ++k) uk[k] = U + ((size_t)k * L0 + col) * d;
                if (carries_modes(p))
                    for (int k = 0; k < K; ++k) uuk[k] = U[(size_t)K * L0 * d + (size_t)k * L0 + col];
                for (int l = l_lo; l < l_hi; ++l) {
                    int w = g * gs + l;
                    double r, Ea;
                    if (step % (uint64_t)p->refresh_every == 0) {
                        orc_whiten(p, st->x + (size_t)w * d, st->y + (size_t)w * K * d);
                        if (carries_loglike(p, st)) orc_anchor_loglike(p, st, w);
                        if (carries_modes(p)) orc_anchor_modes(p, st, w);
                    }
                    /* a column of a one-parameter block draws the RandProposer1D variates of
                     * the un-paired stream, as in full evaluation (its half of the pair block
                     * stays unused) */
                    if (f1[col])
                        walker_variates(k0, k1, walker0 + (uint32_t)w, step, 0, 1, &r, &Ea);
                    else if (p->paired_variates)
                        walker_variates_pair(k0, k1, walker0 + (uint32_t)w, step, &r, &Ea);
                    else
                        walker_variates(k0, k1, walker0 + (uint32_t)w, step, 0, 0, &r, &Ea);
                    total += step_core_inc(p, st, w, v, uk, uu, uuk, wp, nl, r, Ea);
                }
                continue;
            }
            if (!drag) {
                for (int l = l_lo; l < l_hi; ++l) {
                    int w = g * gs + l;
                    double r, Ea;
                    walker_variates(k0, k1, walker0 + (uint32_t)w, step, 0, f1[col], &r, &Ea);
                    total += step_core(p, st, w, v, r, Ea);
                }
                continue;
            }
            /* dragging: interpolation step i of slow step `step` is fast step step*nd + i-1;
             * its direction is copied out of the (one-cycle) cache of fast directions */
            const double* vfp[256];
            int oned[257];
            oned[0] = f1[col];
            for (int i = 1; i <= nd; ++i) {
                uint64_t f = step * (uint64_t)nd + (uint64_t)(i - 1);
                uint64_t cf = f / (uint64_t)Lf;
                int fcol = (int)(f % (uint64_t)Lf);
                if (have_f[0] != cf) {
                    orc_basis_blocked(p, group, (uint32_t)cf, 2, Vf, f1f);
                    have_f[0] = cf;
                }
                memcpy(Vstep + (size_t)(i - 1) * d, Vf + (size_t)fcol * d, sizeof(double) * (size_t)d);
                vfp[i - 1] = Vstep + (size_t)(i - 1) * d;
                oned[i] = f1f[fcol];
            }
            if (p->incremental) {   /* one Gaussian mode: whitened images of the directions */
                if (!U) U = (double*)malloc(sizeof(double) * (size_t)(1 + nd) * d);
                orc_whiten_directions(p, 1, v, U);
                for (int i = 1; i <= nd; ++i) orc_whiten_directions(p, 1, vfp[i - 1], U + (size_t)i * d);
            }
            for (int l = l_lo; l < l_hi; ++l) {
                int w = g * gs + l;
                double r[257], Ea[257];
                for (int i = 0; i <= nd; ++i)
                    walker_variates(k0, k1, walker0 + (uint32_t)w, step, (uint32_t)i, oned[i],
                                    &r[i], &Ea[i]);
                if (p->incremental) {
                    if (step % (uint64_t)p->refresh_every == 0)
                        orc_whiten(p, st->x + (size_t)w * d, st->y + (size_t)w * d);
                    total += drag_core_inc(p, st, w, v, vfp, U, r, Ea);
                } else {
                    total += drag_core(p, st, w, v, vfp, r, Ea);
                }
            }
        }
        free(V); free(f1); free(Vf); free(Vstep); free(U); free(Wd);
    }
    return total;
}

/* Tier-A links for the blocked proposer: walker 0 advances with increments and accept
 * variates drawn by the reference (oracle/ref_numpy.py BlockedRefChain.draws): one Metropolis
 * step with the increment `delta` (sampler order), or one dragging step with the slow
 * increment, the n fast increments and the accept variates e[0] (final test), e[1..n]. */
int orc_step_injected_delta(const orc_problem* p, orc_state* st, const double* delta,
                            double exp_draw)
{
    return step_core(p, st, 0, delta, 1.0, exp_draw);
}

int orc_drag_injected(const orc_problem* p, orc_state* st, const double* slow,
                      const double* fast, const double* e)
{
    int n = p->blocking->drag_steps, d = p->d;
    const double* vf[256];
    double r[257];
    for (int i = 0; i <= n; ++i) r[i] = 1.0;
    for (int i = 0; i < n; ++i) vf[i] = fast + (size_t)i * d;
    return drag_core(p, st, 0, slow, vf, r, e);
}

/* ------------------------------------------------------------------ moments (a15) */
/* Sufficient statistics of the current ensemble state, in the fixed order the device
 * uses: per group g (walkers ascending) sum_x[g][i] = sum x_i, S_g[i][j] = sum x_i x_j
 * (fma chains from 0) of x - shift, then pooled S[i][j] += S_g[i][j] over groups ascending. */
void orc_moments(int d, int W, int gs, const double* x, const double* shift,
                 double* group_sum, double* pooled_S)
{
    int G = W / gs;
    double* Sg = (double*)malloc(sizeof(double) * (size_t)d * d);
    for (int g = 0; g < G; ++g) {
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int l = 0; l < gs; ++l) s = s + (x[((size_t)g * gs + l) * d + i] - shift[i]);
            group_sum[(size_t)g * d + i] += s;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
                for (int l = 0; l < gs; ++l) {
                    const double* xw = x + ((size_t)g * gs + l) * d;
                    s = fma(xw[i] - shift[i], xw[j] - shift[j], s);
                }
                Sg[i * d + j] = s;
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                pooled_S[i * d + j] += Sg[i * d + j];
                if (j != i) pooled_S[j * d + i] = pooled_S[i * d + j];
            }
    }
    free(Sg);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
