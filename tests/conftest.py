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


def binary_case(n=200, T=11, warmup=7, iter=13, ranef=True, n_test=0, **kw):
    """Friedman data with a binary response (probit link), same formula as friedman_case."""
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=ranef, causal=True, binary=True)
    x = d["x"]
    xb = x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]]
    groups = [GroupTerm(d["g1"], None), GroupTerm(d["g2"], None)] if ranef else []
    return make_sampler_args(d["y"], xb, X=np.column_stack([x[:, 3], d["z"]]), groups=groups, family="binomial", iter=iter,
                             warmup=warmup, x_test=xb[:n_test].copy() if n_test else None, bart_args={"n.trees": T}, **kw)


from stan4bart_amd.cases import c5_case, ihdp_case  # noqa: E402,F401  (BASELINE configs 4 and 5: shared with bench.py)


def make_sampler(lib, prefix, args, seed=12345):
    """One seeded sampler (chain set-up of stan4bart_fit_worker, reference R/stan4bart_fit.R:33-47)."""
    import copy
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    args = copy.copy(args)
    rng = RRng(seed)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    return Sampler(lib, prefix, args, rng.state)


class StateView:
    """Parsed view of a sampler-state blob (layout: include/stan4bart_amd.h, s4b_get_state) for the tests: field access and
    patching (forced step sizes) without going through either implementation."""

    def __init__(self, blob: bytes):
        self.b = bytearray(blob)
        magic, version, n, T, D, binary, p = np.frombuffer(self.b, dtype=np.uint32, count=2).tolist() + \
            [int(np.frombuffer(self.b, dtype=np.int64, count=1, offset=8)[0])] + np.frombuffer(self.b, dtype=np.int32, count=4, offset=16).tolist()
        assert magic == 0x53423453 and version == 1
        self.n, self.T, self.D, self.binary, self.p = n, T, D, binary, p
        self.k = float(np.frombuffer(self.b, dtype=np.float64, count=1, offset=32)[0])      # header.reserved[0]: the current value of a modeled k (0.0: k is fixed)
        o = 48
        self.off = {}
        for name, cnt in (("q", D), ("inv_metric", D), ("wm", D), ("wm2", D), ("nuts", 6), ("last", 7)):
            self.off[name] = (o, cnt, np.float64); o += 8 * cnt
        for name, cnt in (("win", 8), ("ecuyer", 2), ("r_rng", 626)):
            self.off[name] = (o, cnt, np.uint32); o += 4 * cnt
        for name, cnt in (("scale", 4), ("offset", n), ("total_fits", n)) + ((("latents", n),) if binary else ()):
            self.off[name] = (o, cnt, np.float64); o += 8 * cnt
        self.trees = []
        for _ in range(T):
            nn, nl = np.frombuffer(self.b, dtype=np.int32, count=2, offset=o).tolist(); o += 8
            nodes = np.frombuffer(self.b, dtype=np.int32, count=2 * nn, offset=o).reshape(nn, 2).copy(); o += 8 * nn
            mu = np.frombuffer(self.b, dtype=np.float64, count=nl, offset=o).copy(); o += 8 * nl
            self.trees.append((nodes, mu))
        assert o == len(self.b)

    def get(self, name):
        o, cnt, dt = self.off[name]
        return np.frombuffer(self.b, dtype=dt, count=cnt, offset=o).copy()

    def set(self, name, values):
        o, cnt, dt = self.off[name]
        v = np.ascontiguousarray(values, dtype=dt)
        assert v.shape == (cnt,)
        self.b[o:o + v.nbytes] = v.tobytes()

    def bytes(self):
        return bytes(self.b)


def assert_state_parity(a: "StateView", b: "StateView", rtol=1e-6, atol=1e-9):
    """Two state blobs describe the same chain state: integers / generator states / tree structure bit-exact, floats rtol."""
    for k in ("win", "ecuyer", "r_rng"):
        assert np.array_equal(a.get(k), b.get(k)), k
    np.testing.assert_allclose(a.k, b.k, rtol=rtol, err_msg="k")
    for k in ("q", "inv_metric", "wm", "wm2", "nuts", "last", "scale", "offset", "total_fits") + (("latents",) if a.binary else ()):
        np.testing.assert_allclose(a.get(k), b.get(k), rtol=rtol, atol=atol, err_msg=k)
    for (na, ma), (nb, mb) in zip(a.trees, b.trees):
        assert np.array_equal(na, nb)
        np.testing.assert_allclose(ma, mb, rtol=rtol, atol=atol)


def teacher_forced(oracle_lib, lib, prefix, args, seed=12345, patch=None, compare_states=True, rtol=1e-6, atol=1e-9):
    """Teacher-forced per-iteration parity (SURVEY.md §7 "Hard parts"): the oracle runs the whole chain; before EVERY Gibbs
    iteration its state is injected into the product sampler (`prefix`), both advance one iteration and are compared:
    bit-exact on the tree-move trace, R generator state, NUTS treedepth / n_leapfrog / divergent, rtol on the rest.
    `patch(it, StateView)` may edit the state both sides start iteration `it` from (returns True if it did).
    Returns the oracle's rows [num_pars x iter] and the list of iterations where a metric window ended."""
    so, sp = make_sampler(oracle_lib, "orc_", args, seed), make_sampler(lib, prefix, args, seed)
    rows, window_ends = [], []
    try:
        so.set_trace(True); sp.set_trace(True)
        for it in range(args.iter):
            warm = it < args.warmup
            if it == args.warmup:
                so.disengage_adaptation(); sp.disengage_adaptation()
            st = so.get_state()
            if patch is not None:
                sv = StateView(st)
                if patch(it, sv):
                    st = sv.bytes()
                    so.set_state(st)
            sp.set_state(st)
            ro, rp = so.run(1, warm), sp.run(1, warm)
            ctx = f"iteration {it}"
            assert np.array_equal(so.get_trace(), sp.get_trace()), ctx + ": tree-move trace differs"
            assert np.array_equal(ro["stan"][3:6], rp["stan"][3:6]), ctx + ": NUTS depth / n_leapfrog / divergent differ"
            np.testing.assert_allclose(ro["stan"], rp["stan"], rtol=rtol, atol=atol, err_msg=ctx)
            np.testing.assert_allclose(ro["bart"]["train"], rp["bart"]["train"], rtol=rtol, atol=atol, err_msg=ctx)
            assert np.array_equal(ro["bart"]["varcount"], rp["bart"]["varcount"]), ctx
            if "k" in ro["bart"]:
                np.testing.assert_allclose(ro["bart"]["k"], rp["bart"]["k"], rtol=rtol, err_msg=ctx + ": k")
            a, b = StateView(so.get_state()), StateView(sp.get_state())
            if compare_states:
                assert_state_parity(a, b, rtol=rtol, atol=atol)
            if not np.array_equal(StateView(st).get("inv_metric"), a.get("inv_metric")):
                window_ends.append(it)
            rows.append(ro["stan"][:, 0].copy())
    finally:
        so.free(); sp.free()
    return np.array(rows).T, window_ends


def run_chain(lib, prefix, args, seed=12345, results_type=0, trace=True, sharing=None, tree_path=None, test_hook=None):
    """stan4bart_fit_worker (reference R/stan4bart_fit.R:33-60) with the diagnostics the parity tests compare.
    ``sharing = (before_warmup, before_sampling)``: device-sharing hints given at those two points (None: not given)."""
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
        if sharing is not None and sharing[0] is not None:
            s.set_device_sharing(sharing[0])
        if tree_path is not None:
            s.set_tree_path(tree_path)
        if test_hook is not None:
            s.set_test_hook(*test_hook)
        traces = []
        if args.warmup > 0:
            out["warmup"] = s.run(args.warmup, True, results_type)
            if trace:
                traces.append(s.get_trace())
        s.disengage_adaptation()
        if sharing is not None and sharing[1] is not None:
            s.set_device_sharing(sharing[1])
        out["sample"] = s.run(args.iter - args.warmup, False, results_type)
        if trace:
            traces.append(s.get_trace())
            out["trace"] = np.concatenate(traces)
        out["trees"] = s.get_trees()
        if getattr(args, "keep_trees", False) and hasattr(s, "get_kept_trees"):
            out["kept_trees"] = s.get_kept_trees()
        out["rng"] = s.get_r_rng_state()
        out["leaf0"] = s.get_leaf_assignment(0)
        out["names"] = s.stan_par_names()
        out["range"] = s.get_bart_data_range()
        out["pm"] = s.get_parametric_mean()
        out["counters"] = s.get_counters()
        out["tree_path"] = s.get_tree_path()
        out["sweep_stats"] = s.get_sweep_stats()
        out["sweep_busy"] = s.get_sweep_busy()
        out["sweep_spec"] = s.get_sweep_spec()
        out["fused_stats"] = s.get_fused_stats()
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
            assert ("k" in a[ph]["bart"]) == ("k" in b[ph]["bart"]), "the k draws of a modeled k are missing on one side"
            if "k" in a[ph]["bart"]:
                np.testing.assert_allclose(a[ph]["bart"]["k"], b[ph]["bart"]["k"], rtol=rtol, err_msg="k draws")
        if stan and "stan" in a[ph]:
            sa, sb = a[ph]["stan"], b[ph]["stan"]
            # integer-valued sampler columns: treedepth__, n_leapfrog__, divergent__
            assert np.array_equal(sa[3:6], sb[3:6]), "NUTS depth / n_leapfrog / divergent differ"
            np.testing.assert_allclose(sa, sb, rtol=rtol, atol=atol)
