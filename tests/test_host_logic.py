"""Product host logic (Gibbs order, NUTS + tape gradient, propose/decide control code, C-ABI) run over a
CPU emulation of the device layer (tests/emul, test infrastructure) and compared with the CPU oracle.
Bit-exact on tree moves / indices / RNG state, 1e-6 relative on floating-point state.
No GPU involved: the HIP kernels themselves are covered by tests/test_gpu_parity.py (-m gpu)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_chain_parity, friedman_case, run_chain


@pytest.mark.parametrize("kw", [
    dict(),                                   # reference test setting: (1|g.1) + (1|g.2), warmup 7, iter 13, 11 trees
    dict(ranef=False),                        # reference test-06-no_ranef.R
    dict(slopes=True),                        # (1 + X4 | g.1) + (1 | g.2): BASELINE config 3 formula
    dict(n_test=17),
    dict(stan_args={"hmc_mode": 1}),          # per-leapfrog O(N) sums instead of sufficient statistics
    dict(skip=(2, 1)),                        # n.thin = 2 BART sweeps per iteration
    dict(skip=(1, 3), warmup=3, iter=6),      # 3 NUTS transitions per iteration (short: NUTS amplifies rounding)
    dict(n=1000, T=50, warmup=10, iter=20),
], ids=str)
def test_emulated_product_matches_oracle(oracle_lib, emul_lib, kw):
    kw = dict(kw)
    if "stan_args" in kw:
        args_o, _ = friedman_case(**{k: v for k, v in kw.items() if k != "stan_args"})
    else:
        args_o, _ = friedman_case(**kw)
    args_e, _ = friedman_case(**kw)
    a = run_chain(oracle_lib, "orc_", args_o)
    b = run_chain(emul_lib, "emu_", args_e)
    assert_chain_parity(a, b)
    np.testing.assert_allclose(a["pm"], b["pm"], rtol=1e-6, atol=1e-9)


def test_bart_only_long_run_matches_oracle(oracle_lib, emul_lib):
    """results_type = 1 (BART only, reference src/init.cpp:821): thousands of tree updates stay identical."""
    args, _ = friedman_case(n=500, T=30, warmup=40, iter=80)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(emul_lib, "emu_", args, results_type=1)
    assert len(a["trace"]) == 30 * 80
    assert set(np.unique(a["trace"][:, 0])) == {0, 1, 2, 3}
    assert_chain_parity(a, b, stan=False)


def test_user_offset_types(oracle_lib, emul_lib):
    """offset_type in {default, fixef, ranef, bart, parametric} (reference src/init.cpp:83-97,762-795,831-839)."""
    rng = np.random.default_rng(3)
    for ot in ("default", "fixef", "ranef", "bart", "parametric"):
        args, _ = friedman_case(offset=rng.normal(size=100), offset_type=ot, warmup=4, iter=8)
        a = run_chain(oracle_lib, "orc_", args)
        b = run_chain(emul_lib, "emu_", args)
        assert_chain_parity(a, b)


def test_keep_fits_false_and_callback(oracle_lib, emul_lib):
    """reference tests/testthat/test-11-callback.R:76-99: keep_fits = FALSE keeps one slot; callback sees every draw."""
    seen = {"o": [], "e": []}
    for key, lib, pfx in (("o", oracle_lib, "orc_"), ("e", emul_lib, "emu_")):
        args, _ = friedman_case(keep_fits=False, callback=lambda tr, te, sp, k=key: seen[k].append((tr.copy(), sp.copy())))
        r = run_chain(lib, pfx, args)
        assert r["sample"]["stan"].shape[1] == 1 and r["sample"]["bart"]["train"].shape[1] == 1
        np.testing.assert_allclose(r["sample"]["bart"]["train"][:, 0], seen[key][-1][0], rtol=0, atol=0)
    assert len(seen["o"]) == len(seen["e"]) == 13
    for (t1, s1), (t2, s2) in zip(seen["o"], seen["e"]):
        np.testing.assert_allclose(t1, t2, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(s1, s2, rtol=1e-6, atol=1e-9)


def test_multi_pass_bins_and_deep_trees(oracle_lib, emul_lib):
    """a flatter tree prior (base .99, power .45; still sub-critical) gives trees with > 16 leaves, the regime in
    which the stats kernel needs several bin passes; the control code must agree with the oracle there as well."""
    args, _ = friedman_case(n=2000, T=4, warmup=20, iter=40, ranef=False, bart_args={"base": 0.99, "power": 0.45, "k": 0.5})
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(emul_lib, "emu_", args, results_type=1)
    assert a["trace"][:, 4].max() > 16
    assert_chain_parity(a, b, stan=False)


def test_error_reporting(emul_lib):
    """errors come back as status + message (the R shim raises them with Rf_error; reference src/init.cpp:319,681)."""
    emul_lib.emu_last_error.restype = C.c_char_p
    assert emul_lib.emu_run(None, 1, 0, 0, None) != 0
    assert b"NULL sampler" in emul_lib.emu_last_error()
    args, _ = friedman_case()
    args.sigma_init = -1.0
    with pytest.raises(RuntimeError, match="sigma_init"):
        run_chain(emul_lib, "emu_", args)
    args, _ = friedman_case(bart_args={"n.trees": 4})
    args.node_capacity = 3
    with pytest.raises(RuntimeError, match="node_capacity|node capacity"):
        run_chain(emul_lib, "emu_", args)
    args, _ = friedman_case()
    args.weights = np.ones(100) * 2.0
    args.weights[3] = 0.0
    with pytest.raises(RuntimeError, match="weights must be positive"):
        run_chain(emul_lib, "emu_", args)


def test_interface_version_is_checked_at_create(emul_lib, oracle_lib):
    """s4b_bart_control and s4b_results have grown fields over the revisions and carry no size field: create() refuses a caller whose header is not
    this one's (the word was `reserved`, 0, in the older headers) before it reads any newer field."""
    from stan4bart_amd import abi
    hdr = open(os.path.join(ROOT, "include", "stan4bart_amd.h")).read()
    assert int(re.search(r"#define S4B_INTERFACE_VERSION (\d+)", hdr).group(1)) == abi.INTERFACE_VERSION
    args, _ = friedman_case()
    saved = abi.INTERFACE_VERSION
    try:
        for v in (0, saved - 1, saved + 1):
            abi.INTERFACE_VERSION = v
            for lib, pfx in ((emul_lib, "emu_"), (oracle_lib, "orc_")):
                with pytest.raises(RuntimeError, match="interface_version"):
                    run_chain(lib, pfx, args)
    finally:
        abi.INTERFACE_VERSION = saved
    run_chain(emul_lib, "emu_", args)


def test_round6_extensions_on_the_emulated_device_layer(emul_lib):
    """s4b_get_sweep_spec reports zeros where no persistent launch exists, and the busy test hook is refused there (include/stan4bart_amd.h)."""
    from conftest import make_sampler
    args, _ = friedman_case()
    s = make_sampler(emul_lib, "emu_", args)
    try:
        s.run(2, True)
        assert s.get_sweep_spec() == (0, 0, 0, 0) and s.get_sweep_stats() == (0, 0) and s.get_sweep_busy() == 0
        with pytest.raises(RuntimeError, match="set_test_hook"):
            s.set_test_hook(1, 2)
    finally:
        s.free()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "stan4bart_amd.h")).read()
    return sorted(set(re.findall(r"S4B_FN\((\w+)\)\s*\(", hdr)) - {"name"})


def test_product_library_exports_every_declared_symbol(hip_lib):
    """the C-ABI library loads on a CPU-only box and exports exactly what include/stan4bart_amd.h declares."""
    names = _declared_symbols()
    assert len(names) >= 26
    for n in names:
        assert hasattr(hip_lib, "s4b_" + n), n
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "stan4bart_amd", "csrc", "libs4b.so")],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set(re.findall(r" T (s4b_\w+)", out))
    assert exported == {"s4b_" + n for n in names}


def test_product_has_no_cpu_fallback(hip_lib):
    """without a HIP device the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    args, _ = friedman_case()
    with pytest.raises(RuntimeError, match="no HIP device|HIP error"):
        run_chain(hip_lib, "s4b_", args)


def test_product_sources_do_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "stan4bart_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".inc", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "oracle/" not in text and "import oracle" not in text, os.path.join(dirpath, f)
                assert "libs4b_emul" not in text and "dev_cpu" not in text, os.path.join(dirpath, f)


@pytest.mark.parametrize("ranef", [True, False])
def test_probit_emulated_product_matches_oracle(oracle_lib, emul_lib, ranef):
    """binary response / probit link: latents from R's sequential stream (dbarts sampleProbitLatentVariables)."""
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(200, ranef=ranef, causal=True, binary=True)
    x = d["x"]
    groups = [GroupTerm(d["g1"], None), GroupTerm(d["g2"], None)] if ranef else []
    args = make_sampler_args(d["y"], x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]], X=np.column_stack([x[:, 3], d["z"]]), groups=groups,
                             family="binomial", iter=13, warmup=7, bart_args={"n.trees": 11})
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)
    assert "aux.1" not in a["names"] and a["names"] == b["names"]


def _odd_predictors_case(n=400, T=8, warmup=5, iter=15):
    """predictors a CART sampler can trip over: a constant column, a 0/1 column, a 3-level column, a column with no
    cut points at all (n.cuts = 0), very few cuts, many cuts, duplicated values."""
    from stan4bart_amd import generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=False, causal=True)
    x = d["x"].copy()
    xb = np.column_stack([x[:, 0], np.full(n, 3.0), d["z"], np.floor(3 * x[:, 1]), x[:, 2], np.round(x[:, 4], 1), x[:, 5], x[:, 6]])
    n_cuts = np.array([100, 5, 1, 2, 0, 7, 1000, 3], dtype=np.int32)
    args = make_sampler_args(d["y"], xb, X=x[:, 3:4], iter=iter, warmup=warmup, bart_args={"n.trees": T, "n.cuts": n_cuts})
    return args


def test_odd_predictors_and_cut_counts(oracle_lib, emul_lib):
    args = _odd_predictors_case()
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)
    used = np.unique(a["trace"][a["trace"][:, 2] >= 0, 2])
    assert 4 not in used            # the predictor without cut points is never proposed


def test_node_capacity_overflow_is_reported(emul_lib):
    """a tree that outgrows node_capacity must surface as an error, not as silent truncation."""
    args, _ = friedman_case(n=2000, T=2, warmup=10, iter=30, ranef=False, bart_args={"base": 0.99, "power": 0.25, "k": 0.3})
    args.node_capacity = 40
    with pytest.raises(RuntimeError, match="node_capacity|node capacity"):
        run_chain(emul_lib, "emu_", args, results_type=1)


@pytest.mark.parametrize("family", ["gaussian", "binomial"])
def test_predict_equals_extract(oracle_lib, emul_lib, family):
    """reference tests/testthat/test-01-continuous.R:204-246 / test-02-binary.R:81-123: with keepTrees, predicting at the
    training rows reproduces the stored training fits, and at the test rows the stored test fits; oracle and product agree."""
    from stan4bart_amd import GroupTerm, RRng, generate_friedman_data, make_sampler_args
    from stan4bart_amd.abi import Sampler
    d = generate_friedman_data(150, ranef=True, causal=True, binary=family == "binomial")
    x = d["x"]
    xb = x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]]
    xt = xb[:20] + 0.01
    res = {}
    for key, lib, pfx in (("o", oracle_lib, "orc_"), ("e", emul_lib, "emu_")):
        args = make_sampler_args(d["y"], xb, X=np.column_stack([x[:, 3], d["z"]]), groups=[GroupTerm(d["g1"]), GroupTerm(d["g2"])],
                                 family=family, iter=13, warmup=7, x_test=xt, bart_args={"n.trees": 11, "keepTrees": True})
        rng = RRng(4242)
        args.seed = int(rng.sample_int(2147483647, 1)[0])
        s = Sampler(lib, pfx, args, rng.state)
        s.run(7, True)
        s.disengage_adaptation()
        r = s.run(6, False)
        ptrain, ptest = s.predict_bart(xb), s.predict_bart(xt)
        s.free()
        assert ptrain.shape == (150, 6) and ptest.shape == (20, 6)
        np.testing.assert_allclose(ptrain, r["bart"]["train"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ptest, r["bart"]["test"], rtol=1e-9, atol=1e-9)
        res[key] = (ptrain, ptest)
    np.testing.assert_allclose(res["o"][0], res["e"][0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(res["o"][1], res["e"][1], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("kw", [dict(), dict(stan_args={"hmc_mode": 1}), dict(ranef=False, n=300, T=20, warmup=10, iter=25)], ids=str)
def test_observation_weights_match_oracle(oracle_lib, emul_lib, kw):
    """weights reach both blocks: dbarts' weighted leaf statistics and continuous.stan's weighted likelihood (:358-366)."""
    kw_o = {k: v for k, v in kw.items() if k != "stan_args"}
    n = kw.get("n", 100)
    w = np.random.default_rng(5).uniform(0.3, 3.0, n)
    a = run_chain(oracle_lib, "orc_", friedman_case(weights=w, **kw_o)[0])
    b = run_chain(emul_lib, "emu_", friedman_case(weights=w, **kw)[0])
    assert_chain_parity(a, b)
    c = run_chain(emul_lib, "emu_", friedman_case(**kw)[0])
    assert not np.array_equal(b["sample"]["bart"]["train"], c["sample"]["bart"]["train"])     # the weights matter
    # constant weights c are the same model as sigma -> sigma / sqrt(c): unit weights reproduce the unweighted chain exactly
    d = run_chain(emul_lib, "emu_", friedman_case(weights=np.ones(n), **kw)[0])
    np.testing.assert_array_equal(d["trace"], c["trace"])
    np.testing.assert_allclose(d["sample"]["bart"]["train"], c["sample"]["bart"]["train"], rtol=1e-9, atol=1e-9)


def _c5_shape_case(n, P, T, n_g1, warmup, iter, **kw):
    """BASELINE config 5 shape: P BART predictors, (1 + X4 | g.1) with many groups (q = 2 n_g1)."""
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=True, causal=True, p=P + 1, n_g1=n_g1)
    x = d["x"]
    xb = x[:, [j for j in range(P + 1) if j != 3]]
    return make_sampler_args(d["y"], xb, X=np.column_stack([x[:, 3], d["z"]]), groups=[GroupTerm(d["g1"], x[:, 3], "g.1")],
                             iter=iter, warmup=warmup, bart_args={"n.trees": T}, **kw)


@pytest.mark.parametrize("P", [100, 140])
def test_config5_shape_many_predictors_and_groups(oracle_lib, emul_lib, P):
    """more than 64 / more than 128 BART predictors (register-table, second register and memory paths of the device's
    predictor tables) and 200 groups with random slopes (q = 400)."""
    args = _c5_shape_case(1500, P, 12, 200, 5, 10)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)
    assert a["sample"]["bart"]["varcount"].shape[0] == P


@pytest.mark.parametrize("T", [1, 2])
def test_one_and_two_trees(oracle_lib, emul_lib, T):
    args, _ = friedman_case(T=T, warmup=10, iter=30, ranef=False)
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args))


def test_interface_details_of_the_call_layer(emul_lib, capfd):
    """getTrees with index vectors, predictBART(offset_test), printTrees, progress / cancellation hook, verbose output
    (reference src/init.cpp:354-403, 448-671, 745-754; src/stan_sampler.hpp:44-48)."""
    from conftest import make_sampler
    args, d = friedman_case(n=150, T=6, warmup=4, iter=10, bart_args={"keepTrees": True})
    s = make_sampler(emul_lib, "emu_", args)
    s.run(4, True)
    s.disengage_adaptation()
    s.run(6, False)
    allt = s.get_kept_trees(-1)
    sel = s.get_kept_trees_indexed([4, 1], [5, 0, 2])
    want = np.concatenate([np.flatnonzero((allt["sample"] == k) & (allt["tree"] == t)) for k in (4, 1) for t in (5, 0, 2)])
    for k in allt:
        assert np.array_equal(sel[k], allt[k][want]), k
    assert np.array_equal(s.get_kept_trees_indexed(None, None)["value"], allt["value"])
    with pytest.raises(RuntimeError, match="samples specified but only 6"):
        s.get_kept_trees_indexed(list(range(7)), None)
    with pytest.raises(RuntimeError, match="tree index out of range"):
        s.get_kept_trees_indexed(None, [6])
    xt = np.asfortranarray(args.x_bart[:9])
    off = np.linspace(-1, 1, 9)
    np.testing.assert_allclose(s.predict_bart(xt, off), s.predict_bart(xt) + off[:, None], rtol=0, atol=1e-12)
    capfd.readouterr()
    s.print_trees([0], [1])
    out = capfd.readouterr().out
    assert out.startswith("sample 1 tree 2:") and ("mu = " in out)
    # progress hook: called at refresh multiples, a True return cancels the run
    seen = []
    s.set_progress(lambda it, n, w: (seen.append((it, n, w)), it >= 3)[1])
    with pytest.raises(RuntimeError, match="interrupted"):
        s.run(5, False)
    assert seen[-1] == (3, 5, False)
    # an exception inside the hook cancels the run and comes back to the caller (ctypes alone would swallow it)
    def boom(it, n, w):
        if it >= 2:
            raise KeyError("from the progress hook")
        return False
    s.set_progress(boom)
    with pytest.raises(KeyError, match="from the progress hook"):
        s.run(5, False)
    s.set_progress(None)
    s.run(1, False)              # the sampler is still usable
    s.free()
    # ... and so does an exception inside the per-iteration callback
    import copy
    a2 = copy.copy(args)
    calls = []

    def bad_callback(tr, te, sp):
        calls.append(1)
        raise ValueError("from the callback")
    a2.callback = bad_callback
    s = make_sampler(emul_lib, "emu_", a2)
    with pytest.raises(ValueError, match="from the callback"):
        s.run(5, True)
    assert len(calls) == 1       # the run stops at the failing iteration (non-zero return of s4b_callback_fn), it is not called again
    s.free()
    # verbose / refresh without a hook: the reference's lines
    args.verbose, args.refresh = 2, 2
    s = make_sampler(emul_lib, "emu_", args)
    capfd.readouterr()
    s.run(4, True)
    out = capfd.readouterr().out
    assert "starting warmup, 4 draws, both BART and Stan" in out and "iter 002 / 004" in out and "iter 004 / 004" in out
    s.free()
