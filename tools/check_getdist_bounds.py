#!/usr/bin/env python3
"""One-command pin of `Rminus1_cl` for the day GetDist is at hand.

    python tools/check_getdist_bounds.py [--device]

The reference forms the R-1 of the confidence bounds (cobaya/samplers/mcmc/mcmc.py:918-1002) from
GetDist's `MCSamples.confidence(i, limfrac=Rminus1_cl_level / 2, upper=which)` (mcmc.py:925-930,
961-965; samples handed over by `SampleCollection._sampled_to_getdist`, collection.py:1139-1161).
GetDist (`GetDist>=1.3.1`, pyproject.toml) is NOT installed in the build container, so this repo's
statement of that routine -- `oracle/ref_numpy.py: confidence`, which the device kernels
(`ckpt_bounds_kernel`, `mcmc_hip_bounds_statistics`) equal exactly -- is restated from its published
source and nothing pins it ("parity unpinned").  This script is that pin:

1. the committed reference chains (golden G6: three single-chain traces with integer weights; golden
   G7: six chains of one multi-chain run; `tests/golden/*.npz`, written by the imported reference) are
   handed to `MCSamples(samples=, weights=, loglikes=, names=)` exactly as `_sampled_to_getdist` does,
   whole and in the split `first = i * cut, last = (i + 1) * cut - 1` of mcmc.py:944-950;
2. for every parameter, both bounds and the tail fractions 0.475 (the default `Rminus1_cl_level / 2`),
   0.025, 1/3 and 0.16, `mcsamples.confidence(i, limfrac, upper)` is compared with
   `oracle.ref_numpy.confidence` on the same rows and weights: the bounds are order statistics, so the
   bar is EXACT equality;
3. `Rminus1_cl` itself (`np.std(bounds, axis=0).T / sqrt(diag(mean_of_covs))`, max; mcmc.py:976-979)
   from GetDist's bounds against `oracle.ref_numpy.rminus1_of_bounds` of the restated ones (rtol 1e-12);
4. with `--device` (an MI355X and the built library): the six G7 chains, unweighted (256 rows each =
   one walker group per chain, the shape the engine's snapshot ring holds), through
   `mcmc_hip_bounds_statistics` against GetDist's bounds of the same rows -- exact again.

Without GetDist it says so, runs the restatement against itself on the same inputs (so that a broken
fixture or import shows today) and exits 0.  Exit code 1: a difference; 0: all equal (or dry run).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LIMFRACS = (0.475, 0.025, 1.0 / 3.0, 0.16)


def committed_chains():
    """-> [(name, samples [n][d], integer weights [n], minuslogpost [n])] from goldens G6 and G7."""
    out = []
    g6 = np.load(os.path.join(ROOT, "tests", "golden", "g6_traces.npz"), allow_pickle=True)
    for tag in sorted({k.split("__")[0] for k in g6.files}):
        cols = [str(c) for c in g6[f"{tag}__columns"]]
        data = g6[f"{tag}__data"]
        # weight, minuslogpost, the sampled parameters, then derived / priors / chi2's
        n_par = len(g6[f"{tag}__x0"])
        out.append((f"G6 {tag}", data[:, 2:2 + n_par], data[:, cols.index("weight")],
                    data[:, cols.index("minuslogpost")]))
    g7 = np.load(os.path.join(ROOT, "tests", "golden", "g7_multichain.npz"), allow_pickle=True)
    cols = [str(c) for c in g7["columns"]]
    n_par = g7["means"].shape[1]
    for c in range(len(g7["Ns"])):
        data = g7[f"chain{c}"]
        out.append((f"G7 chain{c}", data[:, 2:2 + n_par], data[:, cols.index("weight")],
                    data[:, cols.index("minuslogpost")]))
    return out


def pieces(n, m=4):
    """mcmc.py:944-950: the single chain's split -- `cut = n // m`, parts i = 1 .. m - 1 of rows
    [i * cut, (i + 1) * cut - 1) -- plus the whole chain."""
    cut = n // m
    return [(0, n)] + [(i * cut, (i + 1) * cut - 1) for i in range(1, m)]


def getdist_bounds(MCSamples, x, w, mlp, limfrac):
    """mcmc.py:925-930 to the letter, on an MCSamples built as collection.py:1156-1161 builds it."""
    names = [f"p{i}" for i in range(x.shape[1])]
    mcs = MCSamples(samples=np.array(x, dtype=np.float64), weights=np.array(w, dtype=np.float64),
                    loglikes=np.array(mlp, dtype=np.float64), names=names)
    return np.array([[mcs.confidence(i, limfrac=limfrac, upper=which) for i in range(x.shape[1])]
                     for which in [False, True]]).T


def restated_bounds(R, x, w, limfrac):
    return np.array([[R.confidence(x[:, i], w, limfrac, which) for i in range(x.shape[1])]
                     for which in (False, True)]).T


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--device", action="store_true",
                    help="also run mcmc_hip_bounds_statistics on cuda:0 against GetDist")
    a = ap.parse_args()
    from oracle import ref_numpy as R
    try:
        import logging
        logging.getLogger().setLevel(logging.ERROR)
        from getdist import MCSamples
        import getdist
        have = True
        print(f"GetDist {getattr(getdist, '__version__', '?')} found: pinning oracle/ref_numpy.py: confidence")
    except ImportError:
        MCSamples, have = None, False
        print("GetDist is not importable here: DRY RUN (the restatement against itself on the committed "
              "chains).  Install GetDist>=1.3.1 and run this again to pin Rminus1_cl.")
    chains = committed_chains()
    bad = n_cmp = 0
    for name, x, w, mlp in chains:
        for lo, hi in pieces(len(x)):
            xs, ws, ls = x[lo:hi], w[lo:hi], mlp[lo:hi]
            for limfrac in LIMFRACS:
                ours = restated_bounds(R, xs, ws, limfrac)
                ref = getdist_bounds(MCSamples, xs, ws, ls, limfrac) if have else ours
                n_cmp += ours.size
                if not np.array_equal(ours, ref):
                    bad += 1
                    print(f"DIFFERENT  {name} rows [{lo}:{hi}] limfrac {limfrac:.4g}: "
                          f"max |restated - GetDist| = {np.max(np.abs(ours - ref)):.3e}")
    print(f"{n_cmp} bounds of {len(chains)} committed chains (whole + split): "
          f"{'all equal' if not bad else f'{bad} sets differ'}"
          f"{'' if have else ' (restatement only)'}")
    # Rminus1_cl of the multi-chain golden (mcmc.py:976-979)
    g7 = [c for c in chains if c[0].startswith("G7")]
    covs = np.load(os.path.join(ROOT, "tests", "golden", "g7_multichain.npz"))
    mean_of_covs = np.average(covs["covs"], weights=covs["Ns"], axis=0)     # mcmc.py:856
    ours = np.array([restated_bounds(R, x, w, 0.475) for _, x, w, _ in g7])
    ref = np.array([getdist_bounds(MCSamples, x, w, l, 0.475) for _, x, w, l in g7]) if have else ours
    r_ours = R.rminus1_of_bounds(ours, mean_of_covs)
    r_ref = float(np.max(np.std(ref, axis=0).T / np.sqrt(np.diag(mean_of_covs))))
    print(f"Rminus1_cl of the G7 chains: restated {r_ours:.12g}, "
          f"{'GetDist' if have else 'restated'} {r_ref:.12g}")
    if abs(r_ours - r_ref) > 1e-12 * abs(r_ref):
        bad += 1
        print("DIFFERENT  Rminus1_cl")
    if a.device:
        from cobaya_amd.engine import Engine
        gs, d = 256, g7[0][1].shape[1]
        rows = np.vstack([x[:gs] for _, x, _, _ in g7])
        e = Engine(d, len(rows), group_size=gs, device=0, seed=3)
        e.set_prior([0] * d, [-1e3] * d, [1e3] * d)
        e.set_target_one()
        e.set_proposal_cov(np.eye(d))
        e.set_state(rows)
        e.bounds_configure(1)
        e.bounds_set_slot(0, rows)
        for limfrac in LIMFRACS:
            _, b = e.bounds_statistics([0], limfrac, want_bounds=True)
            ref = np.array([(getdist_bounds(MCSamples, x[:gs], np.ones(gs), l[:gs], limfrac) if have
                             else restated_bounds(R, x[:gs], np.ones(gs), limfrac)) for _, x, _, l in g7])
            same = np.array_equal(b, ref)
            bad += not same
            print(f"device bounds, limfrac {limfrac:.4g}: {'equal' if same else 'DIFFERENT'} "
                  f"({'GetDist' if have else 'restatement'})")
        e.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
