#!/usr/bin/env python3
"""One-command pin of the plik-lite arithmetic on the REAL Planck files, for the day they are at hand
(they cannot be downloaded into the build container: VERDICT r4 missing 5).

    python tools/check_real_pliklite.py /path/to/plik_lite_v22[_TTTEEE].dataset [--reference /root/reference]
                                        [--cls cls.txt] [--device]

What it does with the `.dataset` file of `plik_lite_v22` (planck_2018_highl_plik.TTTEEE_lite_native,
`cobaya/likelihoods/base_classes/planck_pliklite.py:22-27`):

1. reads it with this repo's reader (`cobaya_amd.pliklite.PlikLiteDataset.from_files`: the ini file,
   the text tables, the Fortran-binary covariance with its `tril` symmetrisation -- planck_pliklite.py:44-73);
2. forms chi2 = delta^T Sigma^-1 delta for a set of spectra with this repo's host arithmetic
   (`BinnedGaussian.chi_squared`, the restatement the HIP kernels are bit-exact against) and, with
   `--device`, on the GPU through the C ABI (`mcmc_hip_evaluate_binned`);
3. with `--reference DIR` (a checkout of CobayaSampler/cobaya), runs the reference's own
   `PlanckPlikLite.init_params` + `get_chi_squared` on the same files and spectra -- golden G13's
   comparison (tests/golden/make_golden.py: g13_pliklite), now on the real data -- and reports the
   largest relative difference (bar: 1e-10; golden G13 on the synthetic files agrees to 1e-12).

Spectra: `--cls FILE` = text columns `ell TT TE EE` in D_l = l (l + 1) C_l / 2 pi [muK^2] from l = 0 or 2
(a CAMB / CLASS output: with the Planck 2018 best fit this is the chi2 = 584.24 known answer of
tests/test_cosmo_planck_2018.py:123,145); without it, 8 smooth LCDM-shaped spectra of this repo's
generator (`fiducial_spectra`), scaled a few per cent, which pins the arithmetic though not the number.

Exit code 0: every comparison within the bar (or nothing to compare against); 1: a difference; 2: usage.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Ini:
    """The five accessors `PlanckPlikLite.init_params` uses of getdist's IniFile
    (planck_pliklite.py:32-76), over this repo's parse of the same file (no getdist needed)."""

    def __init__(self, folder, params):
        self.folder, self.params = folder, dict(params)

    def list(self, key):
        return str(self.params[key]).split()

    def int(self, key):
        return int(self.params[key])

    def int_list(self, key, default=None):
        return [int(x) for x in str(self.params[key]).split()] if key in self.params else default

    def string(self, key, default=None):
        return str(self.params.get(key, default))

    def relativeFileName(self, key):
        return os.path.join(self.folder, self.params[key])


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dataset", help="the .dataset file of plik_lite_v22")
    ap.add_argument("--reference", help="checkout of CobayaSampler/cobaya to compare with")
    ap.add_argument("--cls", help="text file: ell TT TE EE (D_l, muK^2)")
    ap.add_argument("--device", action="store_true", help="also evaluate on the GPU (C ABI)")
    a = ap.parse_args()
    if not os.path.isfile(a.dataset):
        print(f"{a.dataset}: not a file", file=sys.stderr)
        return 2
    from cobaya_amd import pliklite as P
    ds = P.PlikLiteDataset.from_files(a.dataset)
    tgt = P.BinnedGaussian.from_dataset(ds)
    print(f"read {a.dataset}: {ds.nbintt} TT + {ds.nbinte} TE + {ds.nbinee} EE bins, lmax {ds.lmax}, "
          f"{tgt.n_bins} bins used")
    # the spectra
    if a.cls:
        t = np.loadtxt(a.cls)
        l0 = int(t[0, 0])
        D = np.zeros((1, 3, ds.lmax + 1))
        n = min(len(t), ds.lmax + 1 - l0)
        D[0, :, l0:l0 + n] = t[:n, 1:4].T
        A = np.array([1.0])
    else:
        base = np.array(P.fiducial_spectra(ds.lmax))
        rs = np.random.default_rng(2018)
        D = base[None] * (1.0 + 0.03 * rs.standard_normal((8, 3, 1)))
        A = np.array([1.0, 1.0, 0.9975, 1.0025, 1.0, 1.005, 0.995, 1.0])
    ours = np.array([tgt.chi_squared(0, *D[k], A_planck=A[k]) for k in range(len(D))])
    for k, c in enumerate(ours):
        print(f"  spectra {k}: A_planck {A[k]:.4f}  chi2 = {c:.6f}")
    worst = 0.0
    if a.device:
        from cobaya_amd.engine import Engine
        emu = P.synthetic_emulator(26, ds.lmax)
        eng = Engine(27, 64, group_size=64)
        eng.set_target_binned_gaussian(tgt, emu, 26)
        dev = np.array([eng.evaluate_binned(0, D[k], A[k]) for k in range(len(D))]).ravel()
        eng.close()
        rel = np.max(np.abs(dev - ours) / np.abs(ours))
        worst = max(worst, rel)
        print(f"device (mcmc_hip_evaluate_binned) vs host arithmetic: max rel diff {rel:.2e}")
    if a.reference:
        sys.path.insert(0, a.reference)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        try:
            import getdist  # noqa: F401
        except ImportError:   # the reference imports getdist at module level; it is never called here
            import make_golden  # noqa: F401  (installs the 4-name stand-in used for the goldens)
        from cobaya.likelihoods.base_classes.planck_pliklite import PlanckPlikLite
        folder = os.path.dirname(os.path.abspath(a.dataset))
        like = object.__new__(PlanckPlikLite)   # no installer, no provider: the arithmetic alone
        like.init_params(_Ini(folder, P.read_dataset_ini(a.dataset)))
        ref = np.array([like.get_chi_squared(0, *D[k], A_planck=A[k]) for k in range(len(D))])
        rel = np.max(np.abs(ref - ours) / np.abs(ref))
        worst = max(worst, rel)
        print(f"reference get_chi_squared (planck_pliklite.py:143-155): {ref}")
        print(f"reference vs this repo: max rel diff {rel:.2e}   (bar 1e-10)")
        if a.cls:
            print("known answer of tests/test_cosmo_planck_2018.py for the Planck 2018 best fit: 584.24 "
                  "(TTTEEE lite_native)")
    ok = worst <= 1e-10
    print("OK" if ok else "DIFFERENCE above the bar")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
