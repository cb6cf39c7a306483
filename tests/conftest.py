import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _make(rel):
    subprocess.run(["make", "-C", os.path.join(ROOT, rel)], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (test infrastructure): oracle/_build/liboracle.so, symbols orc_*."""
    _make("oracle")
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


@pytest.fixture(scope="session")
def emul_lib():
    """Product host logic over a CPU emulation of the device layer (test infrastructure), symbols emu_*."""
    _make("tests/emul")
    return ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "_build", "libs4b_emul.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product: stan4bart_amd/csrc/libs4b.so, symbols s4b_* (needs the prebuilt .so or hipcc)."""
    from stan4bart_amd._lib import LIB_PATH, load_library
    if not os.path.exists(LIB_PATH):
        _make("stan4bart_amd/csrc")
    return load_library()


def friedman_case(n=100, ranef=True, slopes=False, p=10, T=11, warmup=7, iter=13, n_test=0, **kw):
    """Sampler arguments for the reference's test setting (tests/testthat/test-05-rng.R:11-27:
    warmup = 7, iter = 13, n.trees = 11; formula y ~ bart(. - g.1 - g.2 - X4 - z) + X4 + z + (..|g.1) + (1|g.2))."""
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=ranef, causal=True, p=p)
    x = d["x"]
    cols = [j for j in range(p) if j != 3]
    xb = x[:, cols]
    X = np.column_stack([x[:, 3], d["z"]])
    groups = []
    if ranef:
        groups = [GroupTerm(d["g1"], x[:, 3] if slopes else None, "g.1"), GroupTerm(d["g2"], None, "g.2")]
    x_test = xb[:n_test].copy() if n_test else None
    bart_args = dict(kw.pop("bart_args", {}))
    bart_args.setdefault("n.trees", T)
    args = make_sampler_args(d["y"], xb, X=X, groups=groups, iter=iter, warmup=warmup, x_test=x_test,
                             bart_args=bart_args, **kw)
    return args, d


def run_chain(lib, prefix, args, seed=12345, results_type=0, trace=True):
    """stan4bart_fit_worker (reference R/stan4bart_fit.R:33-60) with the diagnostics the parity tests compare."""
    import copy
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    args = copy.copy(args)
    rng = RRng(seed)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, prefix, args, rng.state)
    out = {}
    try:
        if trace:
            s.set_trace(True)
        traces = []
        if args.warmup > 0:
            out["warmup"] = s.run(args.warmup, True, results_type)
            if trace:
                traces.append(s.get_trace())
        s.disengage_adaptation()
        out["sample"] = s.run(args.iter - args.warmup, False, results_type)
        if trace:
            traces.append(s.get_trace())
            out["trace"] = np.concatenate(traces)
        out["trees"] = s.get_trees()
        out["rng"] = s.get_r_rng_state()
        out["leaf0"] = s.get_leaf_assignment(0)
        out["names"] = s.stan_par_names()
        out["range"] = s.get_bart_data_range()
        out["pm"] = s.get_parametric_mean()
        out["counters"] = s.get_counters()
    finally:
        s.free()
    return out


def assert_chain_parity(a, b, rtol=1e-6, atol=1e-9, stan=True):
    """Bit-exact on integer / tree state, rtol on floating-point state (BASELINE.json north_star)."""
    assert np.array_equal(a["trace"], b["trace"]), "tree-move trace differs"
    assert np.array_equal(a["rng"], b["rng"]), "R generator state differs"
    for k in ("tree", "n", "var", "split"):
        assert np.array_equal(a["trees"][k], b["trees"][k]), f"flattened trees differ in {k}"
    np.testing.assert_allclose(a["trees"]["value"], b["trees"]["value"], rtol=rtol, atol=atol)
    assert np.array_equal(a["leaf0"], b["leaf0"])
    np.testing.assert_allclose(a["range"], b["range"], rtol=rtol)
    for ph in ("warmup", "sample"):
        if ph not in a:
            continue
        if "bart" in a[ph]:
            assert np.array_equal(a[ph]["bart"]["varcount"], b[ph]["bart"]["varcount"])
            np.testing.assert_allclose(a[ph]["bart"]["train"], b[ph]["bart"]["train"], rtol=rtol, atol=atol)
            np.testing.assert_allclose(a[ph]["bart"]["test"], b[ph]["bart"]["test"], rtol=rtol, atol=atol)
            np.testing.assert_allclose(a[ph]["bart"]["sigma"], b[ph]["bart"]["sigma"], rtol=rtol)
        if stan and "stan" in a[ph]:
            sa, sb = a[ph]["stan"], b[ph]["stan"]
            # integer-valued sampler columns: treedepth__, n_leapfrog__, divergent__
            assert np.array_equal(sa[3:6], sb[3:6]), "NUTS depth / n_leapfrog / divergent differ"
            np.testing.assert_allclose(sa, sb, rtol=rtol, atol=atol)
