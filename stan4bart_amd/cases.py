"""Builders of the BASELINE configurations that need more than the Friedman generator: config 4 (IHDP shape, binary outcome) and
config 5 (n = 1e7, P = 100, 400 trees, 200 groups with random slopes).  Used by the tests (tests/conftest.py) and by bench.py's
`extra_configs`; the data are synthetic except the 747 x 25 IHDP covariates (a fixture made from the reference's data file by
tools/make_ihdp_fixture.py)."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ihdp_case(warmup=7, iter=13, T=50, seed=20260102):
    """BASELINE config 4 at its shape: the 747 x 25 IHDP covariates (6 continuous, 19 binary; tests/golden/ihdp_covariates.npz,
    made by tools/make_ihdp_fixture.py from the reference's ihdp/sim.data.gz following ihdp/data.R:1-22), treatment z, one
    26-level grouping factor (mother's age), formula of the reference's IHDP method (ihdp/methods/stan4bart.R:5):
    y ~ bart(x, z) + (1 + z | g1), treatment = z (test sample = counterfactual rows), binary outcome / probit link.
    The reference's simulation (ihdp/sim.R:55-85, response surface C) yields a continuous outcome; the binary variant
    BASELINE.json names is defined here as y = 1[y_c > median(y_c)] with y_c = main effects + sparse pairwise interactions +
    tau z + group intercept / slope + noise on the standardised covariates."""
    from stan4bart_amd import GroupTerm, make_sampler_args
    f = np.load(os.path.join(ROOT, "tests", "golden", "ihdp_covariates.npz"))
    x, z, g1 = f["x"], f["z"], f["g1"]
    n, p = x.shape
    assert (n, p) == (747, 25) and g1.max() == 26
    xz = x.copy()
    xz[:, :6] = (x[:, :6] - x[:, :6].mean(axis=0)) / x[:, :6].std(axis=0, ddof=1)
    g = np.random.default_rng(seed)
    beta = g.choice([0.0, 1.0, 2.0], size=p + 1, p=[0.6, 0.3, 0.1])
    pairs = [(i, j) for i in range(p) for j in range(i + 1, p)]
    sel = g.choice(len(pairs), size=20, replace=False)
    mu = beta[0] + xz @ beta[1:]
    for k in sel:
        i, j = pairs[k]
        mu = mu + g.choice([0.5, 1.0]) * xz[:, i] * xz[:, j]
    b = g.standard_normal((26, 2)) @ np.linalg.cholesky(np.array([[1.0, 0.2], [0.2, 0.5]])).T
    yc = mu + 4.0 * z + b[g1 - 1, 0] + b[g1 - 1, 1] * z + g.standard_normal(n)
    y = (yc > np.median(yc)).astype(np.float64)
    xb = np.column_stack([x, z])
    xt = np.column_stack([x, 1.0 - z])
    return make_sampler_args(y, xb, X=None, groups=[GroupTerm(g1, z, "g1")], family="binomial", iter=iter, warmup=warmup,
                             x_test=xt, bart_args={"n.trees": T})


def c5_case(n, P=100, T=400, n_groups=200, warmup=2, iter=4, n_test=0, seed=99, **kw):
    """BASELINE config 5 shape: P BART predictors, T trees, (1 + X4 | g.1) with n_groups groups (q = 2 n_groups).  The
    predictors come from numpy's generator column by column (the R-compatible stream would take minutes at n = 1e7; the
    path does not care which generator made x) and are built straight in the layout the C boundary takes, so that the
    n = 1e7 case needs one 8 GB matrix on the host and no second copy."""
    from stan4bart_amd import GroupTerm, make_sampler_args
    g = np.random.default_rng(seed)
    xb = np.empty((n, P), order="F")
    for j in range(P):
        xb[:, j] = g.random(n)
    x4 = g.random(n)
    z = (g.random(n) < 0.2).astype(np.float64)
    g1 = g.integers(1, n_groups + 1, size=n)
    b = g.standard_normal((n_groups, 2)) @ np.linalg.cholesky(np.array([[2.25, 0.2], [0.2, 1.0]])).T
    y = (10.0 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 20.0 * (xb[:, 2] - 0.5) ** 2 + 5.0 * xb[:, 3] + 10.0 * x4 + 5.0 * z
         + b[g1 - 1, 0] + b[g1 - 1, 1] * x4 + g.standard_normal(n))
    x_test = xb[:n_test].copy() if n_test else None
    return make_sampler_args(y, xb, X=np.column_stack([x4, z]), groups=[GroupTerm(g1, x4, "g.1")], iter=iter, warmup=warmup,
                             x_test=x_test, bart_args={"n.trees": T}, **kw), xb
