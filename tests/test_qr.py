"""stan_args = list(QR = TRUE) in the fit mirror (reference R/stan4bart_fit.R:239-258 forward, :560-570 back-transform;
tests/testthat/test-01-continuous.R:162-201 runs a fit with it)."""
import numpy as np
import pytest

from conftest import friedman_case


def _fit(emul_lib, **kw):
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd.fit import fit_worker
    args, d = friedman_case(n=120, ranef=True, warmup=8, iter=20, **kw)
    return args, d, fit_worker(lambda a, st: Sampler(emul_lib, "emu_", a, st), args, RRng(77))


def test_qr_design_and_back_transform(emul_lib):
    args, d, res = _fit(emul_lib, stan_args={"QR": True})
    plain, _ = friedman_case(n=120, ranef=True, warmup=8, iter=20)
    n, K = plain.X.shape
    R_inv = args.extras["R_inv"]
    # the sampler's design: orthogonal columns of norm sqrt(n - 1) (prior_autoscale is on by default), same column space, X_c = X_q R
    np.testing.assert_allclose(args.X.T @ args.X, (n - 1.0) * np.eye(K), atol=1e-9)
    np.testing.assert_allclose(args.X @ np.linalg.inv(R_inv), plain.X, atol=1e-10)
    np.testing.assert_allclose(args.extras["xbar_model"], args.extras["xbar"] @ R_inv)
    # no division of the prior scale by sd(x) with QR (R/stan4bart_fit.R:218)
    assert np.all(args.prior_scale == args.prior_scale[0]) and not np.all(plain.prior_scale == plain.prior_scale[0])
    # back-transform: the returned beta rows give the same linear predictor on the centred design as theta on Q * scale
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    raw_args, _ = friedman_case(n=120, ranef=True, warmup=8, iter=20, stan_args={"QR": True})
    raw_args.extras = dict(raw_args.extras, R_inv=None)            # the same chain without the back-transform
    from stan4bart_amd.fit import fit_worker
    raw = fit_worker(lambda a, st: Sampler(emul_lib, "emu_", a, st), raw_args, RRng(77))
    rows = [i for i, nm in enumerate(res["par_names"]) if nm.startswith("beta.")]
    assert len(rows) == K
    for ph in ("warmup", "sample"):
        beta, theta = res[ph]["stan"][rows, :], raw[ph]["stan"][rows, :]
        np.testing.assert_allclose(plain.X @ beta, args.X @ theta, rtol=1e-9, atol=1e-9)
        other = [i for i in range(len(res["par_names"])) if i not in rows]
        np.testing.assert_array_equal(res[ph]["stan"][other, :], raw[ph]["stan"][other, :])


def test_qr_needs_two_predictors():
    from stan4bart_amd import generate_friedman_data, make_sampler_args
    d = generate_friedman_data(60, ranef=False, causal=True)
    with pytest.raises(ValueError, match="multiple predictors"):
        make_sampler_args(d["y"], d["x"][:, :5], X=d["x"][:, 3:4], groups=[], iter=10, warmup=5, stan_args={"QR": True})
    a = make_sampler_args(d["y"], d["x"][:, :5], X=None, groups=[], iter=10, warmup=5, stan_args={"QR": True})   # no fixed effects: nothing to do
    assert a.extras["R_inv"] is None


def test_qr_fit_quality(emul_lib):
    """the QR fit explains the same data (reference test: the statistical thresholds of the plain fit, here: the two fits' posterior
    mean linear predictors agree closely on a short chain)"""
    argsq, d, rq = _fit(emul_lib, stan_args={"QR": True})
    argsp, _, rp = _fit(emul_lib)
    rows = [i for i, nm in enumerate(rq["par_names"]) if nm.startswith("beta.")]
    lp_q = (argsp.X @ rq["sample"]["stan"][rows, :]).mean(axis=1)
    lp_p = (argsp.X @ rp["sample"]["stan"][rows, :]).mean(axis=1)
    assert np.corrcoef(lp_q, lp_p)[0, 1] > 0.95


def test_stan4bart_qr_back_transforms_every_chain(emul_lib):
    """generics.stan4bart() runs its chains itself (not through fit_worker) and must map the beta rows back as well
    (reference R/stan4bart_fit.R:560-570: for every chain).  The same chains without the back-transform give the sampler's own
    coefficients theta on Q * scale; the fit's `indiv.fixef` must be the linear predictor of those (X_c R_inv theta), for the training and
    the test sample, and fitted / predict must agree with extract."""
    from stan4bart_amd import GroupTerm, generate_friedman_data
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd.generics import stan4bart
    import stan4bart_amd.generics as G
    d = generate_friedman_data(120, ranef=True, causal=True, p=10)
    x = d["x"]
    xb = x[:, [j for j in range(10) if j != 3]]
    X = np.column_stack([x[:, 3], d["z"]])
    groups = [GroupTerm(d["g1"], None, "g.1"), GroupTerm(d["g2"], None, "g.2")]
    rows = np.arange(11)
    gt = [GroupTerm(np.asarray(d["g1"])[rows], None, "g.1"), GroupTerm(np.asarray(d["g2"])[rows], None, "g.2")]
    kw = dict(X=X, groups=groups, x_bart_test=xb[rows], X_test=X[rows], groups_test=gt, chains=2, seed=5, iter=12, warmup=5,
              bart_args={"n.trees": 7, "keepTrees": True}, make_sampler=lambda a, st: Sampler(emul_lib, "emu_", a, st))
    fit = stan4bart(d["y"], xb, stan_args={"QR": True}, **kw)
    saved = G.qr_back_transform
    captured = {}
    try:
        G.qr_back_transform = lambda args, r: captured.setdefault("R_inv", args.extras["R_inv"])     # the same chains, theta left as sampled
        raw = stan4bart(d["y"], xb, stan_args={"QR": True}, **kw)
    finally:
        G.qr_back_transform = saved
    R_inv = captured["R_inv"]
    assert R_inv is not None
    theta = raw.extract("fixef", combine_chains=False)            # [K, S, chains] in Q space
    beta = fit.extract("fixef", combine_chains=False)
    np.testing.assert_allclose(beta, np.einsum("kj,jsc->ksc", R_inv, theta), rtol=1e-12, atol=1e-14)
    Xc = X - X.mean(axis=0)
    want = np.einsum("nk,ksc->nsc", Xc @ R_inv, theta)
    np.testing.assert_allclose(fit.extract("indiv.fixef", combine_chains=False) - fit.extract("indiv.fixef", combine_chains=False).mean(axis=0, keepdims=True),
                               want - want.mean(axis=0, keepdims=True), rtol=1e-9, atol=1e-9)
    # the pieces still add up, the test sample equals the duplicated training rows, predict equals extract
    parts = sum(fit.extract(t, combine_chains=False) for t in ("indiv.bart", "indiv.fixef", "indiv.ranef"))
    np.testing.assert_allclose(fit.extract("ev", combine_chains=False), parts, rtol=1e-12, atol=1e-12)
    for t in ("ev", "indiv.fixef"):
        np.testing.assert_allclose(fit.extract(t, sample="test", combine_chains=False), fit.extract(t, combine_chains=False)[rows], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(fit.predict(x_bart=xb[rows], X=X[rows], groups=gt, type=t, combine_chains=False),
                                   fit.extract(t, combine_chains=False)[rows], rtol=1e-9, atol=1e-9)
    for f in (fit, raw):
        f.close()
    # and the QR fit tells the same story as the plain one (chains long enough to have found the surface: the generator's
    # coefficients are 10 for X4 and 5 for z, inst/common/friedmanData.R)
    kw.update(iter=60, warmup=30, bart_args={"n.trees": 20})
    fitq, plain = stan4bart(d["y"], xb, stan_args={"QR": True}, **kw), stan4bart(d["y"], xb, **kw)
    assert np.corrcoef(fitq.fitted("ev"), plain.fitted("ev"))[0, 1] > 0.95
    assert np.corrcoef(fitq.fitted("indiv.fixef"), plain.fitted("indiv.fixef"))[0, 1] > 0.99
    np.testing.assert_allclose(fitq.extract("fixef").mean(axis=1), [10.0, 5.0], atol=2.0)
