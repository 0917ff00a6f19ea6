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
    def _xw(self, first, last, derived):
        df = self.data
        names = self.sampled_params + (self.derived_params if derived else [])
        return (df[names][first:last].to_numpy(dtype=np.float64),
                df["weight"][first:last].to_numpy(dtype=np.float64))

    def mean(self, first=None, last=None, derived=False):
        """collection.py:893-934."""
        if not len(self):
            raise ValueError("Collection is empty. Cannot compute mean.")
        x, w = self._xw(first, last, derived)
        return np.average(x.T, weights=w, axis=-1)

    def cov(self, first=None, last=None, derived=False):
        """collection.py:936-981 (ddof=0; integer weights as fweights)."""
        if not len(self):
            raise ValueError("Collection is empty. Cannot compute cov.")
        x, w = self._xw(first, last, derived)
        if np.allclose(np.round(w), w):
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
