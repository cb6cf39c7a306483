"""Arguments of bart_args beyond the tree count (reference tests/testthat/test-09-bartArgs.R): cgm(split.probs = ), dbartsControl(useQuantiles = ),
and the k hyperprior the path does not carry.  CPU side: the product's host logic over the emulated device layer against the oracle, the
closed-form properties of the two options, and the refusals."""
import numpy as np
import pytest

from conftest import assert_chain_parity, friedman_case, run_chain


def bart_args_cases():
    """(name, keyword arguments of friedman_case); shared with the `-m gpu` twin in test_gpu_parity.py."""
    return [
        # the reference's own call: test-09-bartArgs.R:20 `split.probs = c(X3 = 2, .default = 1)` (X3 is the third BART column)
        ("split_probs_reference_call", dict(n=120, ranef=False, bart_args={"split.probs": {"X3": 2, ".default": 1},
                                                                               "predictor.names": ["X1", "X2", "X3", "X5", "X6", "X7", "X8", "X9", "X10"]})),
        ("split_probs_vector", dict(n=300, ranef=True, warmup=9, iter=30, bart_args={"split.probs": [5, 1, 1, 0.2, 1, 1, 3, 1, 0.01], "n.trees": 7})),
        ("split_probs_deep", dict(n=500, ranef=False, warmup=10, iter=40, bart_args={"split.probs": {0: 4, 4: 0.5}, "base": 0.99, "power": 0.5, "n.trees": 5})),
        ("quantile_cuts", dict(n=250, ranef=True, bart_args={"useQuantiles": True, "n.cuts": 20})),
        ("quantile_cuts_few_distinct", dict(n=90, ranef=False, warmup=9, iter=25, bart_args={"useQuantiles": True, "n.cuts": 100, "n.trees": 6})),
        ("quantile_cuts_and_split_probs", dict(n=400, ranef=True, slopes=True, n_test=17,
                                               bart_args={"useQuantiles": True, "n.cuts": 7, "split.probs": {2: 3.0}, "n.trees": 9})),
    ]


@pytest.mark.parametrize("name,kw", bart_args_cases(), ids=[c[0] for c in bart_args_cases()])
def test_host_logic_matches_oracle(oracle_lib, emul_lib, name, kw):
    args, _ = friedman_case(**kw)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)


def test_split_probs_reach_the_sampler():
    from stan4bart_amd.fit import _split_probs
    w = _split_probs({"X3": 2, ".default": 1}, 4, ["X1", "X2", "X3", "X4"])
    np.testing.assert_allclose(w, np.array([1, 1, 2, 1]) / 5.0)
    np.testing.assert_allclose(_split_probs([1, 3], 2), [0.25, 0.75])
    np.testing.assert_allclose(_split_probs({1: 9.0}, 3), np.array([1, 9, 1]) / 11.0)
    assert _split_probs(None, 3) is None
    for bad in ([1, 2, 3], {7: 1.0}, [1.0, -1.0], [0.0, 1.0], [np.nan, 1.0], {"X9": 2}):
        with pytest.raises(ValueError):
            _split_probs(bad, 2, ["X1", "X2"])
    args, _ = friedman_case(n=50, ranef=False, bart_args={"split.probs": {1: 2.0}})
    np.testing.assert_allclose(args.split_probs, np.array([1, 2, 1, 1, 1, 1, 1, 1, 1]) / 10.0)
    assert args.use_quantiles is False


def test_split_probs_shift_the_rules(oracle_lib, emul_lib):
    """A predictor with (almost) all the weight takes (almost) all the rules; equal weights reproduce the unweighted prior's
    acceptance arithmetic up to the draw of the predictor (one uniform instead of one integer draw, so not the same chain)."""
    heavy = np.full(9, 1e-6); heavy[5] = 1.0
    args, _ = friedman_case(n=200, ranef=False, warmup=20, iter=60, bart_args={"split.probs": heavy, "n.trees": 10})
    for lib, prefix in ((oracle_lib, "orc_"), (emul_lib, "emu_")):
        out = run_chain(lib, prefix, args, results_type=1)
        vc = out["sample"]["bart"]["varcount"].sum(axis=1)
        assert vc.sum() > 0 and vc[5] >= 0.99 * vc.sum(), vc


def test_quantile_cut_points_closed_form(oracle_lib, emul_lib):
    """Few distinct values: a cut between every two neighbours; many: maxCuts cuts at ranks k * step + step / 2 of the sorted distinct
    values (oracle/bart_ref.hpp setCutPoints).  Read back through the rules of the sampled trees: every split value is one of them."""
    from stan4bart_amd import make_sampler_args
    g = np.random.default_rng(5)
    n = 300
    x0 = g.integers(0, 4, size=n).astype(np.float64)            # 4 distinct -> cuts 0.5, 1.5, 2.5
    x1 = np.round(g.normal(size=n), 1)                           # ~50 distinct, 10 cuts requested
    xb = np.column_stack([x0, x1])
    y = np.sin(x1) + 0.5 * x0 + 0.1 * g.normal(size=n)
    args = make_sampler_args(y, xb, X=g.random(n)[:, None], groups=[], iter=40, warmup=10,
                             bart_args={"n.trees": 8, "useQuantiles": True, "n.cuts": [10, 10]})
    u = np.unique(x1); nu = len(u); step = nu // 10; off = step // 2
    idx = np.minimum(np.arange(10) * step + off, nu - 2)
    expect = {0: np.array([0.5, 1.5, 2.5]), 1: 0.5 * (u[idx] + u[idx + 1])}
    outs = [run_chain(oracle_lib, "orc_", args, results_type=1), run_chain(emul_lib, "emu_", args, results_type=1)]
    assert_chain_parity(outs[0], outs[1], stan=False)
    for lib, prefix in ((oracle_lib, "orc_"), (emul_lib, "emu_")):
        cuts = exported_cut_points(lib, prefix, args)
        assert len(cuts) == 2
        for j in range(2):
            np.testing.assert_allclose(cuts[j], expect[j], rtol=0, atol=1e-14)
    for out in outs:            # and the rules of the sampled trees index into them
        tr = out["trees"]
        rules = tr["var"] >= 0
        assert rules.sum() > 0
        for v, k in zip(tr["var"][rules], tr["split"][rules]):
            assert 0 <= k < len(expect[int(v)])


def test_quantile_cuts_with_a_column_without_cuts(oracle_lib, emul_lib):
    """n.cuts = 0 for one column is inside the validated range: with useQuantiles = TRUE it used to divide by zero;
    the column simply takes no rule."""
    from stan4bart_amd import make_sampler_args
    g = np.random.default_rng(6)
    n = 200
    xb = np.column_stack([g.random(n), np.round(g.normal(size=n), 1), g.random(n)])
    y = np.sin(3 * xb[:, 0]) + xb[:, 1] + 0.1 * g.normal(size=n)
    args = make_sampler_args(y, xb, X=g.random(n)[:, None], groups=[], iter=30, warmup=10,
                             bart_args={"n.trees": 6, "useQuantiles": True, "n.cuts": [12, 0, 7]})
    outs = [run_chain(oracle_lib, "orc_", args, results_type=1), run_chain(emul_lib, "emu_", args, results_type=1)]
    assert_chain_parity(outs[0], outs[1], stan=False)
    for lib, prefix in ((oracle_lib, "orc_"), (emul_lib, "emu_")):
        cuts = exported_cut_points(lib, prefix, args)
        assert [len(c) for c in cuts] == [12, 0, 7]
    for out in outs:
        assert out["sample"]["bart"]["varcount"][1].sum() == 0 and out["sample"]["bart"]["varcount"].sum() > 0


def exported_cut_points(lib, prefix, args):
    """The cut points a sampler holds, read from its exported BART state (stan4bart_exportBARTState; each implementation has its own
    byte layout: oracle/gibbs_ref.cpp orc_export_bart_state, stan4bart_amd/csrc/sampler_core.hpp export_state)."""
    from conftest import make_sampler
    s = make_sampler(lib, prefix, args)
    try:
        blob = s.export_bart_state()
    finally:
        s.free()
    P = args.x_bart.shape[1] if hasattr(args, "x_bart") else None
    out = []
    if prefix == "orc_":
        assert np.frombuffer(blob, dtype=np.uint32, count=1)[0] == 0x5343524F
        P = int(np.frombuffer(blob, dtype=np.uint64, count=1, offset=4)[0]); o = 4 + 8 + 4 + 4 + 8
        for _ in range(P):
            nc = int(np.frombuffer(blob, dtype=np.int32, count=1, offset=o)[0]); o += 4
            out.append(np.frombuffer(blob, dtype=np.float64, count=nc, offset=o).copy() if nc else np.zeros(0)); o += 8 * nc
    else:
        head = np.frombuffer(blob, dtype=np.uint32, count=5)
        P = int(head[2]); o = 20 + 16
        ncs = np.frombuffer(blob, dtype=np.int32, count=P, offset=o).tolist(); o += 4 * P
        for nc in ncs:
            out.append(np.frombuffer(blob[o:o + 8 * nc], dtype=np.float64).copy()); o += 8 * nc
    return out


def test_k_prior_forms():
    """bart_args$k as dbarts takes it (reference R/stan4bart_fit.R:460-465; R/stan4bart.R:202 "allow calls like bart_args = list(k = chi(2, Inf))";
    tests/testthat/test-09-bartArgs.R:32 uses chi(1.25, Inf)): a number, or the chi(degreesOfFreedom, scale) hyperprior in several spellings."""
    from stan4bart_amd.fit import _k_prior
    assert _k_prior(3) == (3.0, None) and _k_prior(2.5) == (2.5, None)
    for form in ("chi(1.25, Inf)", "chi(degreesOfFreedom = 1.25, scale = Inf)", "chi(1.25)", "chi()", "chi", {"chi": (1.25, np.inf)}, {"chi": {"degreesOfFreedom": 1.25}},
                 ("chi", 1.25, np.inf), ("chi",)):
        assert _k_prior(form) == (2.0, (1.25, np.inf)), form
    assert _k_prior("chi(2, 3.5)") == (2.0, (2.0, 3.5)) and _k_prior({"chi": (4.0, 0.5)}) == (2.0, (4.0, 0.5)) and _k_prior("chi(scale = 2, df = 3)") == (2.0, (3.0, 2.0))
    for bad in ("gamma(1, 2)", None, {"chi": (0.0, 1.0)}, "chi(1, 0)", "chi(foo = 1)", [1, 2]):
        with pytest.raises(ValueError):
            _k_prior(bad)
    args, _ = friedman_case(n=50, ranef=False, bart_args={"k": 3})
    assert args.k == 3.0 and args.k_hyper is None
    args, _ = friedman_case(n=50, ranef=False, bart_args={"k": "chi(1.25, Inf)"})
    assert args.k == 2.0 and args.k_hyper == (1.25, np.inf)


K_CASES = [
    ("continuous_improper", dict(n=150, ranef=True, slopes=True, bart_args={"k": "chi(1.25, Inf)"}), None),
    ("continuous_proper_test_rows", dict(n=150, ranef=True, n_test=13, bart_args={"k": {"chi": (3.0, 1.5)}, "n.trees": 9}), None),
    ("thinned", dict(n=120, ranef=False, skip=(3, 1), bart_args={"k": ("chi", 2.0, 4.0), "n.trees": 6}), None),
    ("binary", dict(n=180, T=8), "binary"),
]


@pytest.mark.parametrize("name,kw,kind", K_CASES, ids=[c[0] for c in K_CASES])
def test_k_hyperprior_chain_matches_oracle(oracle_lib, emul_lib, name, kw, kind):
    """normal(k = chi(df, scale)): after every sweep (trees, then the latents of a binary response) k is drawn from its conditional given the
    leaf values — one rgamma from R's stream — and the leaf prior precision of the next sweep follows.  Product host logic vs the oracle's
    independent restatement: tree-move trace and R generator bit-exact, the k draws (the reference's fifth result element "k",
    src/bart_util.cpp:17-26,75-76) within 1e-6; the draws move and stay in a sane range."""
    if kind == "binary":
        from conftest import binary_case
        args = binary_case(**kw)
        args.k_hyper, args.k = (1.25, np.inf), 2.0
    else:
        args, _ = friedman_case(**kw)
    a, b = run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args)
    assert_chain_parity(a, b)
    kd = np.concatenate([b["warmup"]["bart"]["k"], b["sample"]["bart"]["k"]])
    assert kd.shape == (args.iter,) and np.all(np.isfinite(kd)) and np.all(kd > 0.05) and np.all(kd < 50) and np.std(kd) > 0
    # a fixed k returns no fifth element
    fixed, _ = friedman_case(n=60, ranef=False)
    assert "k" not in run_chain(emul_lib, "emu_", fixed)["sample"]["bart"]


def test_k_hyperprior_bart_block_long_run(oracle_lib, emul_lib):
    """the BART block alone (results_type 1: no NUTS sensitivity) over 1 500 tree updates with a modeled k"""
    args, _ = friedman_case(n=300, ranef=False, warmup=50, iter=100, T=15, bart_args={"k": "chi(1.25, Inf)"})
    a, b = run_chain(oracle_lib, "orc_", args, results_type=1), run_chain(emul_lib, "emu_", args, results_type=1)
    assert_chain_parity(a, b, stan=False)
    k = b["sample"]["bart"]["k"]
    assert 0.3 < np.median(k) < 10


def test_k_hyperprior_state_round_trip_and_teacher_forcing(oracle_lib, emul_lib):
    """the current k is part of the chain state (header.reserved[0] of the blob): teacher forcing through it, and a malformed value is refused"""
    from conftest import StateView, make_sampler, teacher_forced
    args, _ = friedman_case(n=120, ranef=True, warmup=8, iter=20, T=7, bart_args={"k": "chi(2, 3)"})
    teacher_forced(oracle_lib, emul_lib, "emu_", args)
    s = make_sampler(emul_lib, "emu_", args)
    try:
        s.run(3, True)
        sv = StateView(s.get_state())
        assert sv.k > 0 and sv.k != 2.0
        bad = bytearray(sv.bytes()); bad[32:40] = np.float64(-1.0).tobytes()
        with pytest.raises(RuntimeError, match="k must be positive"):
            s.set_state(bytes(bad))
        good = sv.bytes()
    finally:
        s.free()
    # the flag is checked in both directions: the state of a modeled-k chain is refused by a sampler whose k is fixed, and vice versa,
    # by the product's host logic and by the oracle alike
    fixed, _ = friedman_case(n=120, ranef=True, warmup=8, iter=20, T=7)
    for lib, pfx in ((emul_lib, "emu_"), (oracle_lib, "orc_")):
        f = make_sampler(lib, pfx, fixed)
        m = make_sampler(lib, pfx, args)
        try:
            with pytest.raises(RuntimeError, match="modeled k"):
                f.set_state(good)
            with pytest.raises(RuntimeError, match="k must be positive"):
                m.set_state(f.get_state())
            f.set_state(f.get_state()); m.set_state(good)
        finally:
            f.free(); m.free()


def test_r_gamma_is_a_gamma_sampler(oracle_lib):
    """rgamma from R's stream (nmath/rgamma.c restated twice: oracle/r_rng.hpp, stan4bart_amd/csrc/rrng_hd.hpp): both algorithms (GS for a < 1, GD
    for a >= 1, every branch of GD's quotient constants) produce the distribution — Kolmogorov-Smirnov against scipy's gamma."""
    import ctypes as C
    from scipy import stats
    if not hasattr(oracle_lib, "orc_test_rgamma"):
        pytest.skip("oracle built without the rgamma test hook")
    oracle_lib.orc_test_rgamma.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double)]
    for shape in (0.3, 0.95, 1.0, 2.2, 3.686, 7.5, 13.022, 40.0, 350.0):
        out = np.zeros(4000)
        assert oracle_lib.orc_test_rgamma(77, shape, 1.7, len(out), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert stats.kstest(out, stats.gamma(shape, scale=1.7).cdf).pvalue > 1e-3, shape
