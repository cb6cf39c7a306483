"""Synthetic Friedman / random-effects data (reference inst/common/friedmanData.R:1-127).

Same construction as the reference's test-data generator — ``set.seed(99)``, ``x ~ U(0,1)``,
``f = 10 round(sin(pi x1 x2), 14) + 20 (x3 - .5)^2 + 5 x5``, fixed part ``10 x4``, correlated
random intercept/slope on ``g.1`` and random intercept on ``g.2``, optional treatment ``z`` —
generalised to any ``n``, ``p >= 5`` predictors and group counts so that the BASELINE.json
configurations (n = 1e5 .. 1e7, p = 10 .. 100) can be produced.  Uses the R-compatible generator
of :mod:`stan4bart_amd.rcompat`; for ``n = 100, p = 10`` the ``x`` matrix equals R's
``matrix(runif(n * 10), n, 10)`` after ``set.seed(99)``.
"""
from __future__ import annotations

import numpy as np

from .rcompat import RRng, qnorm


def generate_friedman_data(n: int, ranef: bool = False, causal: bool = False, binary: bool = False,
                           p: int = 10, n_g1: int = 5, n_g2: int = 8, seed: int = 99) -> dict:
    rng = RRng(seed)
    sigma = 1.0
    x = rng.runif(n * p).reshape((n, p), order="F")

    def f(x):
        return 10 * np.round(np.sin(np.pi * x[:, 0] * x[:, 1]), 14) + 20 * (x[:, 2] - 0.5) ** 2 + 5 * x[:, 4]

    res = dict(x=x, sigma=sigma, mu_bart=f(x), mu_fixef=x[:, 3] * 10)
    if ranef:
        g1 = rng.sample_int(n_g1, n) if n <= 2000 else _fast_sample(rng, n_g1, n)
        Sigma_b1 = np.array([[1.5 ** 2, 0.2], [0.2, 1.0]])
        R_b = np.linalg.cholesky(Sigma_b1).T
        b1 = rng.rnorm(2 * n_g1).reshape((n_g1, 2), order="F") @ R_b
        g2 = rng.sample_int(n_g2, n) if n <= 2000 else _fast_sample(rng, n_g2, n)
        b2 = rng.rnorm(n_g2, 0.0, np.sqrt(1.2))
        res.update(g1=g1, g2=g2, b1=b1, b2=b2)
        res["mu_ranef"] = b1[g1 - 1, 0] + x[:, 3] * b1[g1 - 1, 1] + b2[g2 - 1]
        res["mu"] = res["mu_bart"] + res["mu_fixef"] + res["mu_ranef"]
    else:
        res["mu"] = res["mu_bart"] + res["mu_fixef"]

    if causal:
        tau = 5.0
        z = rng.rbinom1(n, 0.2)
        res.update(tau=tau, z=z)
        mu0 = res["mu"]
        mu1 = mu0 + tau
        if binary:
            both = np.concatenate([mu0, mu1])
            loc, scale = both.mean(), both.std(ddof=1) / float(qnorm(np.array([0.15]))[0])
            mu0, mu1 = (mu0 - loc) / scale, (mu1 - loc) / scale
            y0 = (rng.runif(n) < _pnorm(mu0)).astype(float)
            y1 = (rng.runif(n) < _pnorm(mu1)).astype(float)
        else:
            y0 = mu0 + rng.rnorm(n, 0.0, sigma)
            y1 = mu1 + rng.rnorm(n, 0.0, sigma)
        res.update(mu_0=mu0, mu_1=mu1, y_0=y0, y_1=y1, y=y1 * z + y0 * (1 - z))
    else:
        if binary:
            mu = res["mu"]
            loc, scale = mu.mean(), mu.std(ddof=1) / float(qnorm(np.array([0.15]))[0])
            res["mu"] = (mu - loc) / scale
            res["y"] = (rng.runif(n) < _pnorm(res["mu"])).astype(float)
        else:
            res["y"] = res["mu"] + rng.rnorm(n, 0.0, sigma)
    return res


def _fast_sample(rng: RRng, k: int, n: int) -> np.ndarray:
    """Vectorised ``sample(k, n, replace = TRUE)`` for large n (rejection on one 16-bit draw; k < 65536)."""
    bits = int(np.ceil(np.log2(k)))
    out = np.empty(n, dtype=np.int64)
    filled = 0
    while filled < n:
        m = int((n - filled) * 1.3) + 16
        v = np.floor(rng.runif(m) * 65536).astype(np.int64) & ((1 << bits) - 1)
        v = v[v < k][: n - filled]
        out[filled: filled + len(v)] = v + 1
        filled += len(v)
    return out


def _pnorm(x):
    from math import sqrt
    from scipy.special import erfc
    return 0.5 * erfc(-np.asarray(x) / sqrt(2.0))
