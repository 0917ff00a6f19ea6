"""Sample table with the reference's column contract (SURVEY.md 8b "Results contract").

Mirror of the parts of cobaya/collection.py the mcmc path touches: column naming
(collection.py:154-161, conventions.py:49-61), `add` (402-427), weighted `mean`/`cov`
(893-981: np.average / np.cov(ddof=0, fweights)), `len`, slicing by column, and the text
dump format (1290-1315: `#`-prefixed right-aligned header, `%{w}.8g` columns of width
max(15, len(name))).  `minuslogpost` is stored tempered (collection.py:530-532).
"""
from __future__ import annotations

import numpy as np
import pandas as pd


class SampleCollection:
    def __init__(self, sampled, derived=(), like_name="gaussian_mixture", temperature=1.0,
                 name=None):
        self.sampled_params = list(sampled)
        self.derived_params = list(derived)
        self.temperature = float(temperature)
        self.name = name
        self.minuslogprior_names = ["minuslogprior__0"]
        # one chi2__<name> column per likelihood (collection.py:154-161)
        like_names = [like_name] if isinstance(like_name, str) else list(like_name)
        self.chi2_names = ["chi2__" + n for n in like_names]
        self.columns = (["weight", "minuslogpost"] + self.sampled_params + self.derived_params
                        + ["minuslogprior"] + self.minuslogprior_names + ["chi2"]
                        + self.chi2_names)
        self._blocks = []
        self._data = None

    # ------------------------------------------------------------------ filling
    def add_rows(self, weight, logpost, x, logprior, loglike, derived=None, loglike_parts=None):
        """Vectorised SampleCollection.add (collection.py:402-427, 519-559).  loglike_parts
        [n][n_likelihoods]: the log-likelihood of every component (default: the only one)."""
        weight = np.atleast_1d(np.asarray(weight, dtype=np.float64))
        n = len(weight)
        x = np.asarray(x, dtype=np.float64).reshape(n, len(self.sampled_params))
        cols = [weight, -np.asarray(logpost, dtype=np.float64) / self.temperature]
        cols += [x[:, i] for i in range(x.shape[1])]
        if self.derived_params:
            derived = np.asarray(derived, dtype=np.float64).reshape(n, len(self.derived_params))
            cols += [derived[:, i] for i in range(derived.shape[1])]
        mlp = -np.asarray(logprior, dtype=np.float64)
        chi2 = -2 * np.asarray(loglike, dtype=np.float64)
        cols += [mlp, mlp, chi2]
        if loglike_parts is None:
            cols += [chi2] * len(self.chi2_names)
        else:
            parts = np.asarray(loglike_parts, dtype=np.float64).reshape(n, len(self.chi2_names))
            cols += [-2 * parts[:, i] for i in range(parts.shape[1])]
        self._blocks.append(np.column_stack(cols))
        self._data = None

    @property
    def data(self) -> pd.DataFrame:
        if self._data is None:
            arr = (np.vstack(self._blocks) if self._blocks
                   else np.zeros((0, len(self.columns))))
            self._blocks = [arr] if len(arr) else []
            self._data = pd.DataFrame(arr, columns=self.columns)
        return self._data

    def __len__(self):
        return sum(len(b) for b in self._blocks)

    def __getitem__(self, key):
        return self.data[key]

    # ------------------------------------------------------------------ statistics
    @property
    def is_tempered(self):
        return self.temperature != 1

    @staticmethod
    def _unit_temperature_factors(logposts, T):
        """exp(logp (T - 1) - max over the batch) per sample, for tempered log-posteriors `logposts`
        (a list of arrays, one per collection): what multiplies a tempered weight to give the
        weight under the unit-temperature posterior (collection.py:688-731; p**(1/T) was sampled,
        so the missing factor is p**(1 - 1/T) = exp(T logp_T - logp_T) with logp_T = logp / T).
        The largest factor of the batch is 1."""
        if not logposts or all(len(lp) == 0 for lp in logposts):
            return [np.ones(len(lp)) for lp in logposts]
        gain = [lp * T - lp for lp in logposts]          # log of the missing factor
        top = np.max(np.concatenate(logposts))
        ref = top * T - top
        return [np.exp(g - ref) for g in gain]

    def _detempered_weights(self, with_batch=None):
        """One weight vector per collection of the batch (this one first), as under the
        unit-temperature posterior; the batch must share one temperature."""
        colls = [self, *(with_batch or [])]
        T = float(self.temperature)
        others = [c.temperature for c in colls]
        if not np.allclose(others, T):
            raise ValueError(f"Temperature inconsistent across the batch: {others}.")
        w = [c.data["weight"].to_numpy(dtype=np.float64) for c in colls]
        if T == 1:
            return w
        lp_T = [-c.data["minuslogpost"].to_numpy(dtype=np.float64) for c in colls]
        return [wi * f for wi, f in zip(w, self._unit_temperature_factors(lp_T, T))]

    def _detempered_minuslogpost(self):
        """-log-posterior at unit temperature: the stored column is that of p**(1/T)."""
        col = self.data["minuslogpost"].to_numpy(dtype=np.float64)
        return col if self.temperature == 1 else col * float(self.temperature)

    def _set_data(self, arr):
        self._blocks, self._data = ([arr] if len(arr) else []), None

    def copy(self):
        """collection.py:850-857."""
        c = SampleCollection(self.sampled_params, self.derived_params,
                             [n[len("chi2__"):] for n in self.chi2_names], self.temperature,
                             self.name)
        c._set_data(self.data.to_numpy(dtype=np.float64).copy())
        return c

    def reset_temperature(self, with_batch=None):
        """collection.py:741-763: `weight` and `minuslogpost` become those of a
        unit-temperature sample (rows of zero weight dropped); irreversible."""
        weights_batch = self._detempered_weights(with_batch=with_batch)
        if self.temperature == 1:
            return
        for c, w in zip([self] + list(with_batch or []), weights_batch):
            arr = c.data.to_numpy(dtype=np.float64).copy()
            arr[:, c.columns.index("minuslogpost")] = c._detempered_minuslogpost()
            arr[:, c.columns.index("weight")] = w
            c.temperature = 1.0
            c._set_data(arr[arr[:, c.columns.index("weight")] > 0])

    def reweight(self, importance_weights, with_batch=None, check=True):
        """collection.py:988-1019: multiplies the (detempered) weights in place."""
        self.reset_temperature(with_batch=with_batch)
        if not hasattr(importance_weights[0], "__len__"):
            importance_weights = [importance_weights]
        batch = [self] + list(with_batch or [])
        for c, iw in zip(batch, importance_weights):
            iw = np.asarray(iw, dtype=np.float64)
            if check and (len(iw) != len(c) or np.any(iw < 0)):
                raise ValueError("importance weights must be non-negative, one per sample")
            arr = c.data.to_numpy(dtype=np.float64).copy()
            arr[:, c.columns.index("weight")] *= iw
            c._set_data(arr[arr[:, c.columns.index("weight")] > 0])

    def _weights_for_stats(self, first, last, weights, tempered):
        """collection.py:859-891: (weights, known to be integer)."""
        if weights is not None:
            weights = np.asarray(weights, dtype=np.float64)
            weights = weights / max(weights)
            return weights, bool(np.allclose(np.round(weights), weights))
        if self.is_tempered and not tempered:
            return self._detempered_weights()[0][first:last], False
        w = self.data["weight"][first:last].to_numpy(dtype=np.float64)
        return w, bool(np.allclose(np.round(w), w))

    def _x(self, first, last, derived):
        names = self.sampled_params + (self.derived_params if derived else [])
        return self.data[names][first:last].to_numpy(dtype=np.float64)

    def mean(self, first=None, last=None, weights=None, derived=False, tempered=False):
        """collection.py:893-934.  For a tempered sample the default is the mean of the
        unit-temperature posterior (weights detempered on the fly); `tempered=True` gives the
        mean of p**(1/T)."""
        if not len(self):
            raise ValueError("Collection is empty. Cannot compute mean.")
        w, _ = self._weights_for_stats(first, last, weights, tempered)
        return np.average(self._x(first, last, derived).T, weights=w, axis=-1)

    def cov(self, first=None, last=None, weights=None, derived=False, tempered=False):
        """collection.py:936-981 (ddof=0; integer weights as fweights, else aweights)."""
        if not len(self):
            raise ValueError("Collection is empty. Cannot compute cov.")
        w, are_int = self._weights_for_stats(first, last, weights, tempered)
        x = self._x(first, last, derived)
        if are_int:
            return np.atleast_2d(np.cov(x.T, ddof=0, fweights=np.round(w).astype(np.int64)))
        return np.atleast_2d(np.cov(x.T, ddof=0, aweights=w))

    # ------------------------------------------------------------------ output
    def to_txt(self, path):
        """collection.py:383-393, 1290-1315: GetDist-readable chain file
        (`prefix.<chain#>.txt`): '#' + right-aligned names, then `%{w}.8g` columns."""
        widths = [max(15, len(c)) for c in self.columns]
        with open(path, "w", encoding="utf-8") as out:
            out.write("#" + " ".join(f"{c:>{w}s}" for c, w in zip(self.columns, widths))[1:]
                      + "\n")
            np.savetxt(out, self.data.to_numpy(dtype=np.float64),
                       fmt=[f"%{w}.8g" for w in widths])
