"""Host side of the binned-bandpower Gaussian likelihood (`planck_pliklite`): the mirror of
`PlanckPlikLite.init_params` / `get_chi_squared` / `logp`
(cobaya/likelihoods/base_classes/planck_pliklite.py:32-178) that prepares what
`mcmc_hip_set_target_binned_gaussian` uploads, a plik-lite-SHAPED synthetic data set (the Planck
data -- `plik_lite_2018_AL.zip`, planck_pliklite.py:22-27 -- is not available offline), and the
linear `Cl(theta)` stand-in for the Boltzmann code (`provider.get_Cl`, planck_pliklite.py:170-178).

Nothing here evaluates the likelihood for the sampler: that is the device's job
(`cobaya_amd/csrc/pliklite_kernels.hip`).  `BinnedGaussian.chi_squared` below is the plain numpy
statement of planck_pliklite.py:143-155 used by tests as a cross-check of the fixtures.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

CL_NAMES = ("tt", "te", "ee")   # planck_pliklite.py:17 (the order of the bins in the data vector)


# ------------------------------------------------------------------------------------------
# The data set as the .dataset file and its companions describe it (planck_pliklite.py:32-76)
# ------------------------------------------------------------------------------------------
def read_dataset_ini(path, overrides=None):
    """The `.dataset` file of a DataSetLikelihood (base_classes/DataSetLikelihood.py:56-66 reads
    it with getdist's `IniFile`): CosmoMC-style plain text, one `key = value` per line, `#`
    starts a comment, `DEFAULT(file)` / `INCLUDE(file)` pull in another file of the same
    format (keys already set win over DEFAULTs, INCLUDEs override).  Returns a dict of strings;
    `overrides` (the likelihood's `dataset_params`) are applied last, as
    DataSetLikelihood.load_dataset_file does."""
    import os
    params, defaults = {}, []
    folder = os.path.dirname(os.path.abspath(path))
    with open(path, encoding="utf-8-sig") as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            for directive, bucket in (("DEFAULT(", defaults), ("INCLUDE(", None)):
                if line.upper().startswith(directive) and line.endswith(")"):
                    other = os.path.join(folder, line[len(directive):-1].strip())
                    if bucket is None:
                        params.update(read_dataset_ini(other))
                    else:
                        bucket.append(other)
                    break
            else:
                if "=" not in line:
                    raise ValueError(f"{path}: cannot parse the line {raw!r}")
                key, value = line.split("=", 1)
                params[key.strip()] = value.strip()
    for other in defaults:
        for k, v in read_dataset_ini(other).items():
            params.setdefault(k, v)
    for k, v in (overrides or {}).items():
        params[k] = " ".join(map(str, v)) if isinstance(v, (list, tuple)) else str(v)
    return params


def read_fortran_reals(path):
    """All reals of a Fortran sequential unformatted file (what `scipy.io.FortranFile(path,
    'r').read_reals(dtype=float)` returns for a one-record file, planck_pliklite.py:62-65):
    each record is <uint32 n_bytes> payload <uint32 n_bytes>; several records are
    concatenated.  float64, native byte order."""
    raw = np.fromfile(path, dtype=np.uint8)
    out, pos = [], 0
    while pos < len(raw):
        if pos + 4 > len(raw):
            raise ValueError(f"{path}: truncated record marker")
        n = int(raw[pos:pos + 4].view(np.uint32)[0])
        end = pos + 4 + n
        if end + 4 > len(raw) or int(raw[end:end + 4].view(np.uint32)[0]) != n or n % 8:
            raise ValueError(f"{path}: not a sequential unformatted file of 8-byte reals")
        out.append(raw[pos + 4:end].view(np.float64))
        pos = end + 4
    return np.concatenate(out) if out else np.zeros(0)


@dataclass
class PlikLiteDataset:
    """Contents of a plik-lite data set, as `init_params` reads them from files:
    `nbin{tt,te,ee}`, `lmax`, `bin_lmin_offset`; `blmin`/`blmax` [max nbin] RELATIVE to the
    offset; `weights` [lmax - offset + 1] in C_l units (planck_pliklite.py:52-56 converts them to
    D_l); `data` [nbins, 3] = (l_eff, C_b, sigma_b); `cov` [nbins, nbins].  `options`: what the
    .dataset file says about the selection (`use_cl`, `use_bins`, `bins_for_L_range`,
    `calibration_param`) -- `BinnedGaussian.from_dataset(ds, **ds.options)`."""
    nbintt: int
    nbinte: int
    nbinee: int
    lmax: int
    bin_lmin_offset: int
    blmin: np.ndarray
    blmax: np.ndarray
    weights: np.ndarray
    data: np.ndarray
    cov: np.ndarray
    options: dict = field(default_factory=dict)

    @property
    def nbins(self):
        return self.nbintt + self.nbinte + self.nbinee

    @classmethod
    def from_files(cls, dataset_file, dataset_params=None):
        """The real thing: `plik_lite_v22.dataset` and the files it names (the input side of
        planck_pliklite.py:32-76 -- the ini keys `nbintt nbinte nbinee lmax bin_lmin_offset data
        blmin blmax weights cov_file_binary cov_file use_cl use_bins bins_for_L_range
        calibration_param`, file names relative to the .dataset file, text files via
        `np.loadtxt` (gzip transparently), the covariance from the Fortran-binary file when it
        exists -- lower triangle authoritative, :62-65 -- else from the text `cov_file`).
        `dataset_params` override the file's keys (DataSetLikelihood.py:64)."""
        import os
        if ".dataset" not in os.path.basename(dataset_file):
            dataset_file += ".dataset"            # DataSetLikelihood.py:59-60
        ini = read_dataset_ini(dataset_file, dataset_params)
        folder = os.path.dirname(os.path.abspath(dataset_file))

        def rel(key):
            if key not in ini:
                raise ValueError(f"{dataset_file}: the key '{key}' is missing")
            return os.path.join(folder, ini[key])

        def ints(key):
            return [int(v) for v in ini.get(key, "").split()]

        n_tt, n_te, n_ee = (int(ini[k]) for k in ("nbintt", "nbinte", "nbinee"))
        n = n_tt + n_te + n_ee
        binary = os.path.join(folder, ini["cov_file_binary"]) if "cov_file_binary" in ini else None
        if binary and os.path.exists(binary):
            flat = read_fortran_reals(binary)
            if flat.size != n * n:
                raise ValueError(f"{binary}: {flat.size} reals, expected {n}^2")
            lower = np.tril(flat.reshape(n, n))
            cov = lower + np.tril(lower, -1).T
        else:
            cov = np.loadtxt(rel("cov_file"))
        options = {"use_cl": ini.get("use_cl", "").lower().split(),
                   "use_bins": ints("use_bins"), "bins_for_L_range": ints("bins_for_L_range"),
                   "calibration_param": ini.get("calibration_param", "A_planck")}
        return cls(nbintt=n_tt, nbinte=n_te, nbinee=n_ee, lmax=int(ini["lmax"]),
                   bin_lmin_offset=int(ini["bin_lmin_offset"]),
                   blmin=np.loadtxt(rel("blmin")).astype(int), blmax=np.loadtxt(rel("blmax")).astype(int),
                   weights=np.loadtxt(rel("weights")), data=np.loadtxt(rel("data")), cov=cov,
                   options=options)


@dataclass
class BinnedGaussian:
    """What `PlanckPlikLite.init_params` leaves on the likelihood object
    (planck_pliklite.py:32-141), for the selected spectra and bins: absolute `blmin`/`blmax`,
    `weights` [lmax + 1] in D_l units (zero below the offset), per spectrum the used bins, the
    data vector `X_data` and covariance `cov` of the used bins (`used_indices`)."""
    blmin: np.ndarray
    blmax: np.ndarray
    weights: np.ndarray
    used_bins: list            # per spectrum (tt, te, ee): indices of its used bins
    used_indices: np.ndarray
    X_data: np.ndarray
    cov: np.ndarray
    lmax: int
    calibration_param: str = "A_planck"

    @classmethod
    def from_dataset(cls, ds: PlikLiteDataset, use_cl=("tt", "te", "ee"), use_bins=(),
                     bins_for_L_range=(), calibration_param="A_planck"):
        """The selection logic of planck_pliklite.py:32-141 on the arrays of a data set: D_l
        weights, the bins each selected spectrum keeps (an explicit bin list, or the bins whose
        centre lies in an l range -- not both), their places in the data vector, and the data /
        covariance restricted to them."""
        wanted = {str(c).lower() for c in use_cl}
        if not wanted:
            raise ValueError("use_cl is empty")
        off = int(ds.bin_lmin_offset)
        lo, hi = np.asarray(ds.blmin).astype(int) + off, np.asarray(ds.blmax).astype(int) + off
        ell = off + np.arange(len(ds.weights), dtype=np.float64)
        # C_l-space weights -> D_l space (:52-56), padded with zeros below the first multipole
        w_dl = np.concatenate((np.zeros(off), np.asarray(ds.weights, dtype=np.float64)
                               * (2 * np.pi / ell / (ell + 1))))
        counts = {"tt": int(ds.nbintt), "te": int(ds.nbinte), "ee": int(ds.nbinee)}
        n_all, widest = sum(counts.values()), max(counts.values())
        cov, data = np.asarray(ds.cov, dtype=np.float64), np.asarray(ds.data, dtype=np.float64)
        if cov.shape != (n_all, n_all) or data.shape[0] != n_all:
            raise ValueError("data / covariance do not have nbintt + nbinte + nbinee rows")
        picked = [int(b) for b in use_bins]
        if picked and max(picked) >= widest:
            raise ValueError("use_bins has bin index out of range")
        if len(bins_for_L_range):
            if picked:
                raise ValueError("can only use one bin filter")
            if len(bins_for_L_range) != 2:
                raise ValueError("bins_for_L_range needs two values")
            centre = (lo[:widest] + hi[:widest]) / 2
            picked = np.flatnonzero((centre >= bins_for_L_range[0])
                                    & (centre <= bins_for_L_range[1])).tolist()
        per_spectrum, places, first = [], [], 0
        for name in CL_NAMES:
            n = counts[name]
            if name not in wanted:
                keep = np.zeros(0, dtype=int)
            elif picked:
                keep = np.array([b for b in picked if b < n], dtype=int)
            else:
                keep = np.arange(n, dtype=int)
            per_spectrum.append(keep)
            if name in wanted:
                places.append(first + keep)
            first += n
        places = np.concatenate(places)
        return cls(blmin=lo, blmax=hi, weights=w_dl, used_bins=per_spectrum, used_indices=places,
                   X_data=data[places, 1], cov=cov[np.ix_(places, places)], lmax=int(ds.lmax),
                   calibration_param=calibration_param)

    @property
    def n_bins(self):
        return len(self.used_indices)

    def bin_table(self):
        """(spectrum id, first l, last l) of every used bin, in data-vector order: what the
        device's binning kernel walks (planck_pliklite.py:146-152)."""
        rows = [(tp, int(self.blmin[i]), int(self.blmax[i]))
                for tp in range(3) for i in self.used_bins[tp]]
        return np.array(rows, dtype=np.int32).reshape(-1, 3)

    def chi_squared(self, L0, ctt, cte, cee, A_planck=1.0):
        """planck_pliklite.py:143-155 in numpy (a cross-check for tests; the sampler's
        evaluation is `mcmc_hip_evaluate_binned` / the step kernels)."""
        cl = np.empty(self.n_bins)
        ix = 0
        for tp, cell in enumerate((ctt, cte, cee)):
            for i in self.used_bins[tp]:
                cl[ix] = np.dot(cell[self.blmin[i] - L0:self.blmax[i] - L0 + 1],
                                self.weights[self.blmin[i]:self.blmax[i] + 1])
                ix += 1
        cl /= A_planck ** 2
        diff = self.X_data - cl
        return np.linalg.inv(self.cov).dot(diff).dot(diff)


# ------------------------------------------------------------------------------------------
# Cl(theta): the linear stand-in for the theory code
# ------------------------------------------------------------------------------------------
@dataclass
class LinearClEmulator:
    """D_l(theta) = D0_l + sum_p J[l][p] (theta_p - theta0_p) for the three spectra, l = 0..lmax
    (`ell_factor=True` units, as `provider.get_Cl` hands them to `logp`,
    planck_pliklite.py:170-178).  BASELINE config 5's "analytic LCDM, no CAMB"."""
    theta0: np.ndarray         # [n]
    D0: np.ndarray             # [3][lmax + 1]
    J: np.ndarray              # [3][lmax + 1][n]
    names: list = field(default_factory=list)

    @property
    def n(self):
        return len(self.theta0)

    @property
    def lmax(self):
        return self.D0.shape[1] - 1

    def cl(self, theta):
        """[3][lmax + 1] for one parameter vector (numpy; the device forms the same sums as
        fma chains over p ascending, oracle/mcmc_oracle.c `orc_binned`)."""
        return self.D0 + self.J @ (np.asarray(theta, dtype=np.float64) - self.theta0)


# ------------------------------------------------------------------------------------------
# A plik-lite-shaped synthetic data set
# ------------------------------------------------------------------------------------------
def plik_lite_bins(lmin=30, lmax=2508, nbin_pol=199):
    """Bin edges with the structure of plik_lite_v22: widths 5 (l < 100), 9 (l < 1504),
    17 (l < 2014), 33 above -> 215 TT bins over 30..2508; TE and EE keep the first 199
    (l <= 1996).  Returns (blmin, blmax) relative to `lmin`, and the pol bin count."""
    edges, l = [], lmin
    for width, stop in ((5, 100), (9, 1504), (17, 2014), (33, lmax + 1)):
        while l < min(stop, lmax + 1):
            edges.append((l, min(l + width - 1, lmax)))
            l += width
    edges = np.array(edges, dtype=int)
    return edges[:, 0] - lmin, edges[:, 1] - lmin, min(nbin_pol, len(edges))


def fiducial_spectra(lmax):
    """Smooth CMB-like D_l^{TT,TE,EE} in muK^2 for l = 0..lmax (zero below l = 2): acoustic
    peaks on a damping envelope.  Only the SHAPE matters (dynamic range, sign changes of TE)."""
    l = np.arange(lmax + 1, dtype=np.float64)
    x = l / 301.0
    damp = np.exp(-(l / 1350.0) ** 1.25)
    tt = 5500.0 * damp * (0.35 + 0.65 * np.cos(np.pi * (x - 0.73)) ** 2 * (1.0 + 0.25 * np.cos(np.pi * x)))
    tt += 900.0 / (1.0 + (l / 40.0) ** 2)
    ee = 42.0 * np.exp(-(l / 1500.0) ** 1.3) * (l / (l + 300.0)) ** 2 * (0.2 + 0.8 * np.sin(np.pi * (x - 0.73)) ** 2)
    te = 135.0 * np.exp(-(l / 1400.0) ** 1.2) * (l / (l + 150.0)) * np.sin(2.0 * np.pi * (x - 0.73))
    for a in (tt, te, ee):
        a[:2] = 0.0
    return np.array([tt, te, ee])


def synthetic_emulator(n_lin=26, lmax=2508):
    """Linear response of the fiducial spectra to `n_lin` parameters: six LCDM-like ones
    (log amplitude, tilt, peak position, damping scale, polarisation amplitude, odd/even peak
    contrast) followed by smooth multiplicative modes (a stand-in for an extended model).
    Deterministic: closed-form arrays, no random numbers."""
    if not 1 <= n_lin <= 31:
        raise ValueError("n_lin must be in 1..31")
    D0 = fiducial_spectra(lmax)
    l = np.arange(lmax + 1, dtype=np.float64)
    lp = np.maximum(l, 2.0)
    dDdl = np.gradient(D0, axis=1)
    pol = np.array([0.0, 0.5, 1.0])[:, None]
    modes = [D0,                                               # d/d ln A
             D0 * np.log(lp / 550.0),                          # tilt about l = 550
             -lp * dDdl * 0.3,                                 # peak position (theta_*-like)
             -D0 * (lp / 1350.0) ** 1.25 * 0.5,                # damping scale
             D0 * pol,                                         # polarisation amplitude (tau-like)
             D0 * 0.3 * np.cos(np.pi * lp / 301.0)]            # odd / even peak contrast
    k = 1
    while len(modes) < n_lin:                                  # smooth multiplicative modes
        ph = 0.37 * k
        modes.append(D0 * 0.2 * np.cos(np.pi * k * lp / 2508.0 + ph) * (1.0 + 0.5 * pol * (-1) ** k))
        k += 1
    J = np.stack(modes[:n_lin], axis=-1)                       # [3][lmax + 1][n]
    J[:, :2, :] = 0.0
    names = ["logA", "ns", "theta", "damp", "tau", "omb"][:n_lin]
    names += [f"ext{i}" for i in range(1, n_lin - len(names) + 1)]
    return LinearClEmulator(theta0=np.zeros(n_lin), D0=D0, J=J, names=names)


def synthetic_dataset(seed=0, lmin=30, lmax=2508, nbin_pol=199, band=8):
    """A data set with the layout of plik_lite_v22 (215 TT + 199 TE + 199 EE bins, l = 30..2508)
    and synthetic content: top-hat-like weights per bin, the binned fiducial spectra plus a
    noise realisation as data, and a dense-inverse SPD covariance `cov = B B^T` with `B` banded
    lower triangular (neighbouring bins and the same-l bins of the other spectra are
    correlated).  Entries of `cov` are rounded to float32-representable values so that the
    committed fixture stays small; the noise is drawn from N(0, cov)."""
    rng = np.random.default_rng(seed)
    bmin, bmax, npol = plik_lite_bins(lmin, lmax, nbin_pol)
    ntt = len(bmin)
    nb = ntt + 2 * npol
    lav = (bmin + bmax) // 2 + lmin
    # weights file: C_l-space weights, l = lmin..lmax; within a bin proportional to l(l+1)/2pi
    # times a gently varying positive profile, normalised to one per TT bin in D_l space
    ls = np.arange(lmin, lmax + 1, dtype=np.float64)
    prof = 1.0 + 0.2 * np.cos(0.7 * ls) + 0.1 * rng.random(len(ls))
    w = np.empty(len(ls))
    for a, b in zip(bmin, bmax):
        w[a:b + 1] = prof[a:b + 1] / prof[a:b + 1].sum()
    w *= ls * (ls + 1) / (2 * np.pi)
    D0 = fiducial_spectra(lmax)
    wD = np.hstack((np.zeros(lmin), w * 2 * np.pi / ls / (ls + 1)))
    binned = np.concatenate([
        [np.dot(D0[tp, a + lmin:b + lmin + 1], wD[a + lmin:b + lmin + 1])
         for a, b in zip(bmin[:n], bmax[:n])]
        for tp, n in ((0, ntt), (1, npol), (2, npol))])
    # noise amplitude per bin: cosmic variance + a white-noise floor, in binned C_l units
    lav_all = np.concatenate((lav, lav[:npol], lav[:npol])).astype(np.float64)
    width_all = np.concatenate((bmax - bmin + 1, (bmax - bmin + 1)[:npol], (bmax - bmin + 1)[:npol]))
    scale = 2 * np.pi / lav_all / (lav_all + 1)
    tt_b, te_b, ee_b = binned[:ntt], binned[ntt:ntt + npol], binned[ntt + npol:]
    cv = np.concatenate((np.abs(tt_b), np.sqrt(np.abs(tt_b[:npol] * ee_b) + te_b ** 2) / np.sqrt(2.0),
                         np.abs(ee_b)))
    floor = np.concatenate((np.full(ntt, 30.0), np.full(npol, 2.0), np.full(npol, 0.6))) * scale \
        * np.exp((lav_all / 1800.0) ** 2)
    sigma = np.sqrt(2.0 / ((2 * lav_all + 1) * width_all * 0.6)) * cv + floor
    # B: unit diagonal, decaying sub-diagonals within a spectrum, and couplings to the same-l bin
    # of the previous spectra; cov = diag(sigma) B B^T diag(sigma)
    B = np.eye(nb)
    starts = (0, ntt, ntt + npol)
    sizes = (ntt, npol, npol)
    for s0, n in zip(starts, sizes):
        for k in range(1, band + 1):
            idx = np.arange(k, n)
            B[s0 + idx, s0 + idx - k] = 0.35 * 0.6 ** (k - 1) * (1.0 + 0.2 * np.cos(0.3 * idx))
    for (s_to, n_to), s_from, amp in (((ntt, npol), 0, 0.30), ((ntt + npol, npol), 0, 0.10),
                                      ((ntt + npol, npol), ntt, 0.25)):
        for k in range(-2, 3):
            idx = np.arange(max(0, -k), min(n_to, n_to - k))
            B[s_to + idx, s_from + idx + k] = amp * 0.5 ** abs(k)
    cov = (sigma[:, None] * (B @ B.T)) * sigma[None, :]
    cov = cov.astype(np.float32).astype(np.float64)
    cov = np.tril(cov) + np.tril(cov, -1).T
    noise = np.linalg.cholesky(cov) @ rng.standard_normal(nb)
    data = np.column_stack((lav_all, binned + noise, np.sqrt(np.diag(cov))))
    return PlikLiteDataset(nbintt=ntt, nbinte=npol, nbinee=npol, lmax=lmax, bin_lmin_offset=lmin,
                           blmin=bmin, blmax=bmax, weights=w, data=data, cov=cov)


def fisher_covariance(target: BinnedGaussian, emu: LinearClEmulator, calib_prior_sigma=0.0025,
                      theta_prior_sigma=10.0):
    """Gaussian approximation of the posterior of (theta, A_planck) at the fiducial point: the
    Fisher matrix of the binned model (a proposal covariance for the sampler, sampler.py:485-685
    `covmat`; at A = 1, d binned / dA = -2 binned).  `theta_prior_sigma` bounds the directions
    the selected bins do not constrain (e.g. the polarisation amplitude with TT alone)."""
    L = target.lmax
    tab = target.bin_table()
    Bm = np.zeros((target.n_bins, emu.n + 1))
    for ib, (tp, a, b) in enumerate(tab):
        wv = target.weights[a:b + 1]
        Bm[ib, :emu.n] = wv @ emu.J[tp, a:b + 1, :]
        Bm[ib, emu.n] = -2.0 * (wv @ emu.D0[tp, a:b + 1])
    assert L == emu.lmax
    F = Bm.T @ np.linalg.solve(target.cov, Bm)
    F[emu.n, emu.n] += 1.0 / calib_prior_sigma ** 2
    F[np.arange(emu.n), np.arange(emu.n)] += 1.0 / theta_prior_sigma ** 2
    return np.linalg.inv(F)
