"""CPU oracle vs the known answers that exist for this path, plus the reference's own property tests.

The reference has no golden vectors for tree moves / HMC state (its tests are self-consistency and
statistical thresholds: tests/testthat/test-01-continuous.R, test-05-rng.R, test-06-no_ranef.R), so
the oracle is pinned on the published known answers of R's and Boost's generators and checked with
transcriptions of those property tests.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import friedman_case, run_chain

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "rng_kat.json")))
dp = C.POINTER(C.c_double)


def _r_draws(lib, seed, n_unif=0, n_norm=0, n_exp=0, n_idx=0, dn=10.0):
    u, z, e, ix = (np.zeros(max(1, k)) for k in (n_unif, n_norm, n_exp, n_idx))
    lib.orc_test_r_rng(C.c_uint32(seed), n_unif, u.ctypes.data_as(dp), n_norm, z.ctypes.data_as(dp), n_exp,
                       e.ctypes.data_as(dp), n_idx, C.c_double(dn), ix.ctypes.data_as(dp))
    return u[:n_unif], z[:n_norm], e[:n_exp], ix[:n_idx]


@pytest.mark.parametrize("seed", ["1", "42", "123"])
def test_r_runif_rnorm_known_answers(oracle_lib, seed):
    u, _, _, _ = _r_draws(oracle_lib, int(seed), n_unif=3)
    np.testing.assert_allclose(u, KAT["runif"][seed], atol=5e-8)
    _, z, _, _ = _r_draws(oracle_lib, int(seed), n_norm=3)
    np.testing.assert_allclose(z, KAT["rnorm"][seed], atol=5e-8)


def test_r_rexp_known_answer(oracle_lib):
    _, _, e, _ = _r_draws(oracle_lib, 1, n_exp=3)
    np.testing.assert_allclose(e, KAT["rexp"]["1"], atol=5e-8)


@pytest.mark.parametrize("seed", ["42", "123"])
def test_r_sample_known_answers(oracle_lib, seed):
    """sample(1:10) = partial Fisher-Yates over R_unif_index (rejection sampling)."""
    from stan4bart_amd.rcompat import RRng
    r = RRng(int(seed))
    x = list(range(1, 11))
    n, out = 10, []
    for _ in range(10):
        j = r.unif_index(n)
        out.append(x[j]); x[j] = x[n - 1]; n -= 1
    assert out == KAT["sample10"][seed]


def test_python_rng_mirror_equals_oracle(oracle_lib):
    from stan4bart_amd.rcompat import RRng, seed_state
    for seed in (1, 99, 12345, 2 ** 31 - 1):
        st = np.zeros(625, dtype=np.uint32)
        oracle_lib.orc_test_r_seed_state(C.c_uint32(seed), st.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert np.array_equal(st, seed_state(seed))
        u, z, _, _ = _r_draws(oracle_lib, seed, n_unif=700)
        np.testing.assert_array_equal(u, RRng(seed).runif(700))
        _, z, _, _ = _r_draws(oracle_lib, seed, n_norm=50)
        np.testing.assert_allclose(z, RRng(seed).rnorm(50), rtol=0, atol=1e-15)


def test_qnorm_against_scipy(oracle_lib):
    from scipy.special import ndtri
    oracle_lib.orc_test_qnorm.restype = C.c_double
    oracle_lib.orc_test_qnorm.argtypes = [C.c_double]
    ps = np.concatenate([np.linspace(1e-9, 1 - 1e-9, 501), 10.0 ** -np.arange(3, 18)])
    got = np.array([oracle_lib.orc_test_qnorm(float(p)) for p in ps])
    np.testing.assert_allclose(got, ndtri(ps), rtol=5e-15, atol=5e-15)


def test_ecuyer1988_boost_validation_constant(oracle_lib):
    oracle_lib.orc_test_ecuyer_nth.restype = C.c_uint32
    assert oracle_lib.orc_test_ecuyer_nth(10000) == KAT["ecuyer1988_default_10000th"]


def test_boost_normal_is_standard_normal(oracle_lib):
    from scipy import stats
    u, z = np.zeros(20000), np.zeros(400000)
    oracle_lib.orc_test_boost_draws(C.c_uint32(77), C.c_uint32(1), len(u), u.ctypes.data_as(dp), len(z), z.ctypes.data_as(dp))
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert abs((np.abs(z) > 3.4426).mean() - 2 * stats.norm.sf(3.4426)) < 1.5e-4   # exercises the ziggurat tail


@pytest.mark.parametrize("slopes", [None, 1, 2, 3])
def test_analytic_gradient_matches_finite_differences(oracle_lib, slopes):
    """log density of continuous.stan: forward-dual gradient vs central differences (SURVEY §8c iii)."""
    from stan4bart_amd import GroupTerm, RRng, generate_friedman_data, make_sampler_args
    from stan4bart_amd.abi import Sampler
    d = generate_friedman_data(100, ranef=True, causal=True)
    x = d["x"]
    sl = None if slopes is None else x[:, [3, 0, 1][:slopes]]
    groups = [GroupTerm(d["g1"], sl, "g.1"), GroupTerm(d["g2"], None, "g.2")]
    a = make_sampler_args(d["y"], x[:, [0, 1, 2, 4]], X=np.column_stack([x[:, 3], d["z"]]), groups=groups, iter=4,
                          warmup=2, bart_args={"n.trees": 3})
    s = Sampler(oracle_lib, "orc_", a, RRng(1).state)
    D = oracle_lib.orc_test_num_unconstrained(s._h)
    q = np.random.default_rng(0).uniform(-1, 1, D)

    def lp(qv):
        out, g = C.c_double(), np.zeros(D)
        oracle_lib.orc_test_log_prob_grad(s._h, qv.ctypes.data_as(dp), C.byref(out), g.ctypes.data_as(dp))
        return out.value, g
    _, g = lp(q)
    fd = np.zeros(D)
    for i in range(D):
        h = 1e-6
        qp, qm = q.copy(), q.copy()
        qp[i] += h; qm[i] -= h
        fd[i] = (lp(qp)[0] - lp(qm)[0]) / (2 * h)
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-6)
    s.free()


def test_reproducibility_same_seed(oracle_lib):
    """reference tests/testthat/test-05-rng.R:29-44: the same seed twice gives identical bart_train."""
    args, _ = friedman_case()
    a = run_chain(oracle_lib, "orc_", args, seed=12345)
    b = run_chain(oracle_lib, "orc_", args, seed=12345)
    c = run_chain(oracle_lib, "orc_", args, seed=12346)
    assert np.array_equal(a["sample"]["bart"]["train"], b["sample"]["bart"]["train"])
    assert np.array_equal(a["sample"]["stan"], b["sample"]["stan"])
    assert not np.array_equal(a["sample"]["bart"]["train"], c["sample"]["bart"]["train"])


def test_result_shapes_and_names(oracle_lib):
    """reference test-01-continuous.R:21-24,86-93: no intercept parameter; varcount is p x samples."""
    args, _ = friedman_case(n_test=10)
    r = run_chain(oracle_lib, "orc_", args)
    names = r["names"]
    assert not any(n.startswith("gamma") for n in names)
    assert names[:7] == ["lp__", "accept_stat__", "stepsize__", "treedepth__", "n_leapfrog__", "divergent__", "energy__"]
    assert r["sample"]["stan"].shape == (len(names), 6)
    assert r["sample"]["bart"]["varcount"].shape == (9, 6)
    assert r["sample"]["bart"]["test"].shape == (10, 6)
    # test rows are the first 10 training rows: identical fits (predict == extract, test-01:204-246)
    np.testing.assert_allclose(r["sample"]["bart"]["test"], r["sample"]["bart"]["train"][:10], rtol=1e-12)
    # sigma handed to BART is the Stan draw aux.1 (reference src/init.cpp:796-800)
    np.testing.assert_allclose(r["sample"]["bart"]["sigma"], r["sample"]["stan"][names.index("aux.1")], rtol=1e-12)
    # trees: counts of the root equal n
    t = r["trees"]
    assert np.all(t["n"][np.r_[True, t["tree"][1:] != t["tree"][:-1]]] == 100)


@pytest.mark.parametrize("ranef", [True, False])
def test_statistical_quality_thresholds(oracle_lib, ranef):
    """reference test-01-continuous.R:152-159 (cor(bart part, truth) >= .95, cor(fixef) via beta close to 10 / 5);
    shortened chains."""
    args, d = friedman_case(n=100, ranef=ranef, T=50, warmup=300, iter=600)
    r = run_chain(oracle_lib, "orc_", args, seed=0, trace=False)
    fit = r["sample"]["bart"]["train"].mean(axis=1)
    assert np.corrcoef(fit, d["mu_bart"])[0, 1] >= 0.95
    names = r["names"]
    beta = r["sample"]["stan"][[names.index("beta.1"), names.index("beta.2")]].mean(axis=1)
    assert abs(beta[0] - 10) < 2.5 and abs(beta[1] - 5) < 1.5
