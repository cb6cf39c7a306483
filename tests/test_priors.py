"""Coefficient prior families of continuous.stan beyond normal / student_t (reference src/stan_files/continuous.stan:
124-144 hs_prior / hsplus_prior, 298-322 beta transforms, 382-414 log-priors): hs, hs_plus, laplace, lasso,
product_normal.  The oracle is checked against scipy's densities and finite differences; the product (host model
behind the C-ABI) against the oracle."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

from conftest import assert_chain_parity, run_chain

dp = C.POINTER(C.c_double)
FAMILIES = [dict(dist="hs"), dict(dist="hs_plus", df1=3.0, df2=2.0), dict(dist="laplace"), dict(dist="lasso", df=2.0),
            dict(dist="product_normal", df=3), dict(dist="student_t", df=5.0)]


def _args(prior, n=100, **kw):
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=True, causal=True)
    x = d["x"]
    groups = [GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")]
    kw.setdefault("iter", 13); kw.setdefault("warmup", 7)
    return make_sampler_args(d["y"], x[:, [0, 1, 2, 4, 5]], X=np.column_stack([x[:, 3], d["z"], x[:, 6]]), groups=groups,
                             bart_args={"n.trees": 5}, stan_args={"prior": prior} if prior else None, **kw)


def _lp(lib, s, q):
    out, g = C.c_double(), np.zeros(len(q))
    assert lib.orc_test_log_prob_grad(s._h, q.ctypes.data_as(dp), C.byref(out), g.ctypes.data_as(dp)) == 0
    return out.value, g


@pytest.mark.parametrize("prior", FAMILIES, ids=lambda p: p["dist"])
def test_oracle_gradient_and_density(oracle_lib, prior):
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    a = _args(prior)
    s = Sampler(oracle_lib, "orc_", a, RRng(1).state)
    D = oracle_lib.orc_test_num_unconstrained(s._h)
    K = 3
    q = np.random.default_rng(3).uniform(-1, 1, D)
    lp, g = _lp(oracle_lib, s, q)
    fd = np.zeros(D)
    for i in range(D):
        h = 1e-6
        qp, qm = q.copy(), q.copy()
        qp[i] += h; qm[i] -= h
        fd[i] = (_lp(oracle_lib, s, qp)[0] - _lp(oracle_lib, s, qm)[0]) / (2 * h)
    np.testing.assert_allclose(g, fd, rtol=5e-6, atol=5e-6)

    # value: against the same model with a flat prior on beta (prior_dist 0) evaluated at the implied beta; the difference
    # is exactly the family's log-prior + the log-Jacobians of its positive parameters, computed here with scipy
    dist = prior["dist"]
    hs = {"hs": 2, "hs_plus": 4}.get(dist, 0)
    nzb = K * prior.get("df", 2) if dist == "product_normal" else K
    n_mix = K if dist in ("laplace", "lasso") else 0
    n_lam = 1 if dist == "lasso" else 0
    pos = 0
    zb = q[pos:pos + nzb]; pos += nzb
    gl = np.exp(q[pos:pos + hs]); jac = q[pos:pos + hs].sum(); pos += hs
    loc = np.exp(q[pos:pos + hs * K]).reshape(hs, K) if hs else None; jac += q[pos:pos + hs * K].sum(); pos += hs * K
    caux = np.exp(q[pos]) if hs else None
    if hs:
        jac += q[pos]; pos += 1
    mix = np.exp(q[pos:pos + n_mix]); jac += q[pos:pos + n_mix].sum(); pos += n_mix
    lam = np.exp(q[pos]) if n_lam else None
    if n_lam:
        jac += q[pos]; pos += 1
    rest = q[pos:]
    sigma = a.prior_scale_for_aux * np.exp(q[-1])
    ps, pm, pdf = np.asarray(a.prior_scale), np.asarray(a.prior_mean), np.asarray(a.prior_df)
    logp = stats.norm.logpdf(zb).sum()
    if hs:
        c2 = a.slab_scale ** 2 * caux
        tau = gl[0] * np.sqrt(gl[1]) * a.global_prior_scale * sigma
        lamb = loc[0] * np.sqrt(loc[1])
        if hs == 4:
            lamb = lamb * loc[2] * np.sqrt(loc[3])
        beta = zb * np.sqrt(c2 * lamb ** 2 / (c2 + tau ** 2 * lamb ** 2)) * tau
        logp += stats.halfnorm.logpdf(loc[0]).sum() - K * np.log(2) + np.log(2)      # normal_lpdf(v) - log_half, once
        logp += stats.invgamma.logpdf(loc[1], 0.5 * pdf, scale=0.5 * pdf).sum()
        if hs == 4:
            logp += stats.norm.logpdf(loc[2]).sum() + np.log(2)
            logp += stats.invgamma.logpdf(loc[3], 0.5 * ps, scale=0.5 * ps).sum()
        logp += stats.norm.logpdf(gl[0]) + np.log(2) + stats.invgamma.logpdf(gl[1], 0.5 * a.global_prior_df, scale=0.5 * a.global_prior_df)
        logp += stats.invgamma.logpdf(caux, 0.5 * a.slab_df, scale=0.5 * a.slab_df)
    elif dist == "laplace":
        beta = pm + ps * np.sqrt(2 * mix) * zb
        logp += stats.expon.logpdf(mix).sum()
    elif dist == "lasso":
        beta = pm + lam * ps * np.sqrt(2 * mix) * zb
        logp += stats.expon.logpdf(mix).sum() + stats.chi2.logpdf(lam, pdf[0])
    elif dist == "product_normal":
        m = prior["df"]
        beta = np.prod(zb.reshape(K, m), axis=1) * ps ** m + pm
    else:   # student_t via the Cornish-Fisher expansion: checked through the gradient only
        s.free()
        return
    a0 = _args({"dist": "none"})
    s0 = Sampler(oracle_lib, "orc_", a0, RRng(1).state)
    q0 = np.concatenate([beta, rest])
    lp0, _ = _lp(oracle_lib, s0, q0)
    np.testing.assert_allclose(lp - lp0, logp + jac, rtol=1e-10, atol=1e-8)
    # the emitted sample row carries the same beta
    names = s.stan_par_names()
    s.free(); s0.free()
    assert [n for n in names if n.startswith("beta.")] == ["beta.1", "beta.2", "beta.3"]
    if hs:
        assert "local.2.1" in names and names.index("local.1.1") + 1 == names.index("local.2.1") and "caux.1" in names
    if n_mix:
        assert "mix.1.3" in names
    if n_lam:
        assert "one_over_lambda.1" in names


@pytest.mark.parametrize("prior", FAMILIES, ids=lambda p: p["dist"])
def test_product_host_model_matches_oracle(oracle_lib, emul_lib, prior):
    args = _args(prior)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)
    names_ok = a["sample"]["stan"].shape == b["sample"]["stan"].shape
    assert names_ok


def test_hs_hand_off_uses_the_coefficients_not_the_local_scales(emul_lib):
    """The reference's get_parametric_mean skips `2 + K` slots for hs priors (src/stan_files/continuous.hpp:3673) where
    write_array emits `hs + hs K + 1` (global, local, caux; get_aux skips those correctly, :3640-3646): under hs / hs_plus it hands
    BART X times the wrong slice of the sample row.  This path deliberately FIXES that (beta_pos()): the parametric mean is
    X beta + Z b with the row's own `beta.*` and `b.*` entries."""
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    for dist in ("hs", "hs_plus"):
        args = _args({"dist": dist})
        s = Sampler(emul_lib, "emu_", args, RRng(3).state)
        row = s.run(3, True)["stan"][:, -1]
        names, pm = s.stan_par_names(), s.get_parametric_mean()
        s.free()
        beta = np.array([row[names.index(f"beta.{k + 1}")] for k in range(np.asarray(args.X).shape[1])])
        b = np.array([row[i] for i, nm in enumerate(names) if nm.startswith("b.")])
        eta = np.asarray(args.X) @ beta
        u, v, w = np.asarray(args.u), np.asarray(args.v), np.asarray(args.w)
        for i in range(len(eta)):
            eta[i] += np.dot(w[u[i]:u[i + 1]], b[v[u[i]:u[i + 1]]])
        np.testing.assert_allclose(pm, eta, rtol=1e-12, atol=1e-12)


def test_hs_rejected_for_binary(emul_lib):
    from stan4bart_amd import GroupTerm, RRng, generate_friedman_data, make_sampler_args
    from stan4bart_amd.abi import Sampler
    d = generate_friedman_data(60, ranef=False, causal=True, binary=True)
    a = make_sampler_args(d["y"], d["x"][:, :3], X=d["x"][:, 3:5], family="binomial", iter=4, warmup=2, bart_args={"n.trees": 3},
                          stan_args={"prior": {"dist": "hs"}})
    with pytest.raises(RuntimeError, match="binary"):
        Sampler(emul_lib, "emu_", a, RRng(1).state)


@pytest.mark.gpu
@pytest.mark.parametrize("prior", [dict(dist="hs"), dict(dist="lasso", df=2.0), dict(dist="product_normal", df=2)], ids=lambda p: p["dist"])
def test_hip_matches_oracle_with_prior_family(oracle_lib, hip_lib, prior):
    args = _args(prior)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)


CLOSED_FORM_CASES = [
    ("default_intercepts", dict(ranef=True), {}),
    ("slopes", dict(ranef=True, slopes=True), {}),
    ("no_ranef", dict(ranef=False), {}),
    ("flat_coefficients", dict(ranef=True, slopes=True, stan_args={"prior": {"dist": "none"}}), {}),
    ("aux_flat", dict(ranef=True, slopes=True), dict(prior_dist_for_aux=0)),
    ("aux_normal", dict(ranef=True, slopes=True), dict(prior_dist_for_aux=1, prior_mean_for_aux=0.3)),
    ("aux_student_t", dict(ranef=True), dict(prior_dist_for_aux=2, prior_df_for_aux=4.0, prior_mean_for_aux=0.1)),
    ("decov_shapes", dict(ranef=True, slopes=True, stan_args={"prior_covariance": dict(regularization=2.5, concentration=1.7, shape=1.4, scale=0.8)}), {}),
]


@pytest.mark.parametrize("name,kw,over", CLOSED_FORM_CASES, ids=[c[0] for c in CLOSED_FORM_CASES])
def test_closed_form_gradient_matches_tape_and_oracle(oracle_lib, emul_lib, monkeypatch, name, kw, over):
    """The default model family (normal / flat coefficient prior, at most two coefficients per grouping term, any aux prior) has its log
    density and gradient written out in closed form (stan_host.hpp closed_form_grad) — what a leapfrog costs on the host once the chain runs
    deep NUTS trees.  With S4B_GRADIENT_CHECK=2 every evaluation of the chain (initialisation, init_stepsize, every leapfrog) is repeated on
    the reverse-mode tape and compared; the chain itself is compared with the oracle's (reference continuous.stan:261-429)."""
    from conftest import friedman_case
    monkeypatch.setenv("S4B_GRADIENT_CHECK", "2")
    args, _ = friedman_case(n=150, warmup=7, iter=13, T=7, **kw)     # (free-running joint chains: the reference test horizon)
    for k, v in over.items():
        setattr(args, k, v)
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args))


def test_closed_form_gradient_binary_and_tape_only_chain(oracle_lib, emul_lib, monkeypatch):
    from conftest import binary_case, friedman_case
    monkeypatch.setenv("S4B_GRADIENT_CHECK", "2")
    args = binary_case(n=160, T=7, warmup=7, iter=13)
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args))
    # the tape alone (S4B_GRADIENT_CHECK=1) still carries a whole chain of the default family
    monkeypatch.setenv("S4B_GRADIENT_CHECK", "1")
    args, _ = friedman_case(n=150, ranef=True, slopes=True, warmup=7, iter=13, T=7)
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args))
