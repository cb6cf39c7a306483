"""Parity of the HIP path (through the C-ABI, libs4b.so) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): bit-exact for tree structure / indexing moves / RNG state / NUTS integer
diagnostics, 1e-6 relative for leaf means, sigma, fits and HMC floating-point state.  NUTS trajectories
amplify rounding differences by ~10x per Gibbs iteration on this posterior (hundreds of leapfrogs per
transition), so joint (Stan + BART) chains are compared over the reference's own short test horizon
(warmup 7 / iter 13, reference tests/testthat/test-05-rng.R:11-27) and the BART block alone over long runs.
"""
import numpy as np
import pytest

from conftest import assert_chain_parity, friedman_case, run_chain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(),                                   # BASELINE config 1 shape: n = 100, (1|g.1) + (1|g.2)
    dict(T=50),
    dict(ranef=False),
    dict(slopes=True),
    dict(n_test=17),
    dict(stan_args={"hmc_mode": 1}),
    dict(skip=(2, 1)),
    dict(skip=(1, 3), warmup=3, iter=6),
    dict(n=1003, T=50),                       # n not a multiple of 4: ragged tail of the vector loads
    dict(n=1001, T=20, ranef=False, warmup=10, iter=20),
    dict(n=1002, T=20, ranef=False),
    dict(n=7, T=3, warmup=2, iter=4, ranef=False),
    dict(T=1, warmup=10, iter=30, ranef=False),      # a single tree: no next tree to propose for while deciding
    dict(T=2, warmup=10, iter=30),
], ids=str)
def test_hip_matches_oracle(oracle_lib, hip_lib, kw):
    kw_o = {k: v for k, v in kw.items() if k != "stan_args"}
    a = run_chain(oracle_lib, "orc_", friedman_case(**kw_o)[0])
    b = run_chain(hip_lib, "s4b_", friedman_case(**kw)[0])
    assert_chain_parity(a, b)
    np.testing.assert_allclose(a["pm"], b["pm"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("n", [5000, 1003])
@pytest.mark.parametrize("path", ["persistent", "fused", "two-kernel"])
def test_bart_block_long_run(oracle_lib, hip_lib, n, path):
    """2400 tree updates on each of the three tree-update paths of the device layer (s4b_set_tree_path): the persistent sweep
    (k_sweep: one launch per sweep, residual in registers, bin partials exchanged through integer atomics), one fused launch per
    tree (k_step), two kernels per tree (k_tree + k_control)."""
    args, _ = friedman_case(n=n, T=40, warmup=30, iter=60)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path=path)
    assert b["tree_path"] == (path, path)
    assert len(a["trace"]) == 40 * 60 and set(np.unique(a["trace"][:, 0])) == {0, 1, 2, 3}
    assert_chain_parity(a, b, stan=False)
    if path == "persistent":
        assert b["sweep_stats"] == (61, 0)        # the sweep at creation + every sweep of the run, each in one launch, none handed over


@pytest.mark.parametrize("kw", [dict(), dict(n=1003, T=50), dict(n=7, T=3, warmup=2, iter=4, ranef=False), dict(T=1, warmup=10, iter=30, ranef=False),
                                dict(T=2, warmup=10, iter=30), dict(n_test=17), dict(slopes=True)], ids=str)
def test_fused_path_joint_chain(oracle_lib, hip_lib, kw):
    """the joint (Stan + BART) chain on the fused tree update (the automatic choice is the persistent sweep: test_hip_matches_oracle
    covers that one): the same draws as the oracle"""
    args = friedman_case(**kw)[0]
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, tree_path="fused")
    assert b["tree_path"][1] == "fused"
    assert_chain_parity(a, b)


def test_automatic_path_is_the_persistent_sweep(hip_lib):
    """at most 16 observations per pass thread: the persistent sweep — k_sweep, with observation weights k_sweep_w (round 6) — also when chains share the
    device (round 5: they take turns); weights together with split.probs: two kernels per tree"""
    from conftest import make_sampler
    w = np.random.default_rng(1).random(3000) + 0.5
    sp = {"split.probs": {0: 2.0}}
    for kw, sharing, want in ((dict(), None, "persistent"), (dict(weights=w), None, "persistent"), (dict(), 3, "persistent"), (dict(weights=w), 3, "persistent"),
                              (dict(bart_args=sp), None, "persistent"), (dict(weights=w, bart_args=sp), None, "two-kernel")):
        args, _ = friedman_case(n=3000, T=5, warmup=2, iter=4, **kw)
        s = make_sampler(hip_lib, "s4b_", args)
        try:
            if sharing:
                s.set_device_sharing(sharing)
            assert s.get_tree_path() == ("auto", want)
        finally:
            s.free()


@pytest.mark.parametrize("paths", [("persistent", "fused"), ("two-kernel", "persistent"), ("fused", "persistent")], ids=str)
def test_tree_path_can_change_between_runs(oracle_lib, hip_lib, paths):
    """every path starts a sweep from the same state (main tree arrays, residual, generator slot 0): switching between warm-up and
    sampling gives the oracle's chain"""
    from conftest import make_sampler
    args, _ = friedman_case(n=3000, T=12, warmup=8, iter=16, ranef=True)
    a = run_chain(oracle_lib, "orc_", args)
    s = make_sampler(hip_lib, "s4b_", args)
    try:
        s.set_trace(True)
        s.set_tree_path(paths[0]); w = s.run(args.warmup, True); t1 = s.get_trace()
        s.disengage_adaptation()
        s.set_tree_path(paths[1]); r = s.run(args.iter - args.warmup, False); t2 = s.get_trace()
        assert s.get_tree_path() == (paths[1], paths[1])
        assert np.array_equal(np.concatenate([t1, t2]), a["trace"])
        np.testing.assert_allclose(r["bart"]["train"], a["sample"]["bart"]["train"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(w["stan"], a["warmup"]["stan"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(s.get_r_rng_state(), a["rng"])
    finally:
        s.free()


def test_config2_shape_fixed_effects_only(oracle_lib, hip_lib):
    """BASELINE config 2 shape (n = 1e5, p = 10, 200 trees, fixed effects only), a few iterations."""
    args, _ = friedman_case(n=100000, T=200, warmup=3, iter=6, ranef=False)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)


def test_config3_shape_random_slopes_p50(oracle_lib, hip_lib):
    """BASELINE config 3 shape at reduced n: p = 50 predictors, (1 + X4 | g.1) + (1 | g.2), 200 trees."""
    args, _ = friedman_case(n=20000, p=50, T=200, warmup=2, iter=4, slopes=True)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)


def test_user_offset_types(oracle_lib, hip_lib):
    rng = np.random.default_rng(3)
    for ot in ("default", "fixef", "ranef", "bart", "parametric"):
        args, _ = friedman_case(offset=rng.normal(size=100), offset_type=ot, warmup=4, iter=8)
        assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(hip_lib, "s4b_", args))


@pytest.mark.parametrize("path", ["persistent", "fused", "two-kernel"])
def test_multi_pass_bins_and_deep_trees(oracle_lib, hip_lib, path):
    """> 16 leaves: several bin passes (8 bins per pass in k_sweep, 16 in k_step / k_tree); trees that outgrow the 64 node slots of
    the wave-register control path: k_sweep hands the rest of such a sweep over to k_step launches"""
    args, _ = friedman_case(n=2000, T=4, warmup=20, iter=40, ranef=False, bart_args={"base": 0.99, "power": 0.45, "k": 0.5})
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path=path)
    assert a["trace"][:, 4].max() > 16
    assert_chain_parity(a, b, stan=False)
    if path == "persistent":
        assert b["tree_path"][1] == "persistent" and b["sweep_stats"][0] == 41


def test_large_node_capacity_global_fallback(oracle_lib, hip_lib):
    """node_capacity large enough that the control kernel cannot stage the tree in LDS (global-memory path)."""
    args, _ = friedman_case(n=300, T=5, warmup=5, iter=10)
    a = run_chain(oracle_lib, "orc_", args)
    args.node_capacity = 3000
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)


def test_reproducible_run_to_run(hip_lib):
    """reference tests/testthat/test-05-rng.R:29-44 on the device: same seed twice => bitwise identical draws
    (all reductions have a fixed order)."""
    args, _ = friedman_case(n=30000, T=50, warmup=5, iter=10)
    a = run_chain(hip_lib, "s4b_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert np.array_equal(a["sample"]["bart"]["train"], b["sample"]["bart"]["train"])
    assert np.array_equal(a["sample"]["stan"], b["sample"]["stan"])
    assert np.array_equal(a["trace"], b["trace"])


def test_keep_fits_false_and_callback(hip_lib):
    seen = []
    args, _ = friedman_case(keep_fits=False, callback=lambda tr, te, sp: seen.append((tr.copy(), sp.copy())))
    r = run_chain(hip_lib, "s4b_", args)
    assert len(seen) == 13 and r["sample"]["bart"]["train"].shape[1] == 1
    np.testing.assert_array_equal(r["sample"]["bart"]["train"][:, 0], seen[-1][0])


def test_size_independent_invariants_at_scale(hip_lib):
    """n = 1e6, p = 50 (BASELINE config 3 size for one chain): properties that need no oracle —
    every tree's leaf counts sum to n; test rows equal to training rows get identical fits; the fit returned
    by run() equals the fit re-assembled from the flattened trees (sum of leaf values, un-rescaled)."""
    n, n_test = 1_000_000, 64
    args, d = friedman_case(n=n, p=50, T=20, warmup=2, iter=4, slopes=True, n_test=n_test)
    r = run_chain(hip_lib, "s4b_", args, trace=False)
    tr = r["trees"]
    roots = np.r_[True, tr["tree"][1:] != tr["tree"][:-1]]
    assert np.all(tr["n"][roots] == n)
    leaf = tr["var"] < 0
    for t in range(20):
        m = (tr["tree"] == t) & leaf
        assert tr["n"][m].sum() == n
    np.testing.assert_allclose(r["sample"]["bart"]["test"][:, -1], r["sample"]["bart"]["train"][:n_test, -1], rtol=1e-9)
    # re-assemble the last fit of observation 0..63 from the flattened trees
    x = d["x"][:n_test][:, [j for j in range(50) if j != 3]]
    lo, hi = r["range"]
    fit = np.zeros(n_test)
    for t in range(20):
        idx = np.flatnonzero(tr["tree"] == t)
        for i in range(n_test):
            k = 0
            while tr["var"][idx[k]] >= 0:
                # preorder: left child is next; right child follows the whole left subtree
                if x[i, tr["var"][idx[k]]] <= tr["value"][idx[k]]:
                    k += 1
                else:
                    depth, k = 1, k + 1
                    while depth > 0:
                        depth += 1 if tr["var"][idx[k]] >= 0 else -1
                        k += 1
            fit[i] += tr["value"][idx[k]]
    np.testing.assert_allclose((fit + 0.5) * (hi - lo) + lo, r["sample"]["bart"]["train"][:n_test, -1], rtol=1e-8, atol=1e-8)


from conftest import binary_case as _binary_case  # noqa: E402


@pytest.mark.parametrize("kw", [dict(), dict(ranef=False), dict(n=747, T=50, n_test=25), dict(n=5001, T=20, warmup=3, iter=6)], ids=str)
def test_probit_matches_oracle(oracle_lib, hip_lib, kw):
    """binary response, probit link (BASELINE config 4 shape: n = 747): latents drawn on the device from R's stream."""
    args = _binary_case(**kw)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)
    assert np.all(a["sample"]["bart"]["sigma"] == 1.0) and "aux.1" not in b["names"]


@pytest.mark.parametrize("path", ["fused", "two-kernel"])
def test_probit_and_thinning_on_the_other_tree_paths(oracle_lib, hip_lib, path):
    """binary response (latents drawn between sweeps from the same generator the sweep leaves in slot 0) and skip = (2, 1): two BART
    sweeps per Gibbs iteration, on the fused and on the two-kernel tree update (the automatic choice — the persistent sweep — is
    covered by test_probit_matches_oracle and test_hip_matches_oracle)"""
    for args in (_binary_case(n=747, T=20, n_test=11), friedman_case(n=1200, T=9, skip=(2, 1), warmup=5, iter=10)[0]):
        a = run_chain(oracle_lib, "orc_", args)
        b = run_chain(hip_lib, "s4b_", args, tree_path=path)
        assert b["tree_path"][1] == path
        assert_chain_parity(a, b)


def test_trees_with_more_than_128_node_slots(oracle_lib, hip_lib):
    """regression: the LDS records of k_tree are carved by size; a tree that uses > 128 node slots exercises the
    upper half of every table (control code takes the global-memory path there: > 64 slots)."""
    args, _ = friedman_case(n=4000, T=2, warmup=10, iter=30, ranef=False, bart_args={"base": 0.99, "power": 0.25, "k": 0.3})
    args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    assert a["trace"][:, 4].max() > 128
    for path in ("persistent", "fused"):     # (persistent: sweeps are handed over to k_step launches where a tree outgrows the wave path)
        b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path=path)
        assert_chain_parity(a, b, stan=False)
        if path == "persistent" and b["tree_path"][1] == "persistent":
            assert b["sweep_stats"][1] > 0


@pytest.mark.parametrize("sharing", [(4, None), (None, 4), (4, 1)])
def test_device_sharing_hint_switches_the_tree_update_without_changing_the_draws(oracle_lib, hip_lib, sharing):
    """s4b_set_device_sharing (where the persistent sweep does not apply — here: more than 16 observations per pass thread, n = 1.1e6): three or more
    chains per GPU -> two-kernel tree update, fewer -> the fused launch; given before the warm-up, between warm-up and sampling, and back again — the
    chain is the oracle's either way."""
    n = 1_100_000
    args, _ = friedman_case(n=n, T=3, warmup=3, iter=6, ranef=True, weights=np.random.default_rng(7).random(n) + 0.5)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, sharing=sharing)
    assert b["tree_path"][1] in ("fused", "two-kernel")
    assert_chain_parity(a, b)


def test_odd_predictors_and_cut_counts(oracle_lib, hip_lib):
    from test_host_logic import _odd_predictors_case
    args = _odd_predictors_case()
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(hip_lib, "s4b_", args))


def test_node_capacity_overflow_is_reported(hip_lib):
    args, _ = friedman_case(n=2000, T=2, warmup=10, iter=30, ranef=False, bart_args={"base": 0.99, "power": 0.25, "k": 0.3})
    args.node_capacity = 40
    with pytest.raises(RuntimeError, match="node_capacity|node capacity"):
        run_chain(hip_lib, "s4b_", args, results_type=1)


@pytest.mark.parametrize("family", ["gaussian", "binomial"])
def test_predict_equals_extract(hip_lib, family):
    """keepTrees + predict on the device (reference test-01-continuous.R:204-246): predictions at the training / test rows
    equal the stored fits of the same draws."""
    from stan4bart_amd import GroupTerm, RRng, generate_friedman_data, make_sampler_args
    from stan4bart_amd.abi import Sampler
    d = generate_friedman_data(3000, ranef=True, causal=True, binary=family == "binomial")
    x = d["x"]
    xb = x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]]
    xt = xb[:50] + 0.01
    args = make_sampler_args(d["y"], xb, X=np.column_stack([x[:, 3], d["z"]]), groups=[GroupTerm(d["g1"]), GroupTerm(d["g2"])],
                             family=family, iter=13, warmup=7, x_test=xt, bart_args={"n.trees": 25, "keepTrees": True})
    rng = RRng(4242)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(hip_lib, "s4b_", args, rng.state)
    s.run(7, True)
    s.disengage_adaptation()
    r = s.run(6, False)
    ptrain, ptest = s.predict_bart(xb), s.predict_bart(xt)
    s.free()
    np.testing.assert_allclose(ptrain, r["bart"]["train"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ptest, r["bart"]["test"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("kw", [dict(), dict(stan_args={"hmc_mode": 1}), dict(n=1003, T=20, ranef=False, warmup=10, iter=25),
                                dict(n=3000, T=6, warmup=15, iter=30, ranef=False, bart_args={"base": 0.99, "power": 0.45, "k": 0.5})], ids=str)
def test_observation_weights(oracle_lib, hip_lib, kw):
    """weights in both blocks (weighted leaf statistics, weighted Stan likelihood); the last case has > 8 bins, i.e.
    several bin passes of the weighted kernel."""
    kw_o = {k: v for k, v in kw.items() if k != "stan_args"}
    n = kw.get("n", 100)
    w = np.random.default_rng(5).uniform(0.3, 3.0, n)
    a = run_chain(oracle_lib, "orc_", friedman_case(weights=w, **kw_o)[0])
    b = run_chain(hip_lib, "s4b_", friedman_case(weights=w, **kw)[0])
    assert b["tree_path"][1] == "persistent"          # (k_sweep_w, round 6; one workgroup at these sizes — the full grid: test_observation_weights_on_the_full_grid)
    assert_chain_parity(a, b)
    for path in ("fused", "two-kernel"):
        e = run_chain(hip_lib, "s4b_", friedman_case(weights=w, **kw)[0], tree_path=path)
        assert e["tree_path"][1] == path
        assert_chain_parity(a, e)
    c = run_chain(hip_lib, "s4b_", friedman_case(**kw)[0])
    d = run_chain(hip_lib, "s4b_", friedman_case(weights=np.ones(n), **kw)[0])
    np.testing.assert_array_equal(d["trace"], c["trace"])          # unit weights == no weights


@pytest.mark.parametrize("n,T,scale,deep", [(24_000, 9, 1.0, False), (100_000, 7, 1e6, False), (300_000, 5, 1e-7, False), (60_000, 3, 3.0, True), (1_000_000, 6, 1.0, False)])
def test_observation_weights_on_the_full_grid(oracle_lib, hip_lib, n, T, scale, deep):
    """k_sweep_w with 6 ... 255 pass workgroups exchanging (sum of w r, count, sum of w) per bin — the sums of the weights travel as bins of their own, everything
    times a power of two that brings the largest weight into (0.5, 1] —; weights of 1e6 and 1e-7 (the fixed-point words hold sums of O(1) terms whatever the
    user's scale), a few weights 1e-3 of the rest; a deep prior (sweeps end as k_step<weighted> launches; trees beyond 32 bins end the launch: 2 x 32 "bins")."""
    g = np.random.default_rng(n)
    w = g.uniform(0.3, 3.0, n) * scale
    w[g.integers(0, n, 50)] *= 1e-3
    kw = dict(n=n, T=T, warmup=6, iter=14, ranef=False)
    if deep:
        kw["bart_args"] = {"base": 0.99, "power": 0.3, "k": 0.3}
    args, _ = friedman_case(weights=w, **kw)
    if deep:
        args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="persistent")
    assert b["tree_path"] == ("persistent", "persistent")
    assert_chain_parity(a, b, stan=False)
    sweeps, handed = b["sweep_stats"]
    assert sweeps == 15 and (handed > 0) == deep, b["sweep_stats"]
    if not deep:
        launches, inside, early, ok = b["sweep_spec"]
        assert inside == sweeps * T and early > 0.3 * inside, b["sweep_spec"]


@pytest.mark.parametrize("P", [100, 140])
def test_config5_shape_many_predictors_and_groups(oracle_lib, hip_lib, P):
    """BASELINE config 5 shape at reduced n: P = 100 predictors (140: beyond the two register tables), 200 groups with
    random slopes (q = 400), T trees."""
    from test_host_logic import _c5_shape_case
    args = _c5_shape_case(1500, P, 12, 200, 5, 10)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)
    used = np.flatnonzero(a["sample"]["bart"]["varcount"].sum(axis=1))
    assert used.max() >= 64          # rules on predictors held in the second table / in memory were proposed and accepted


def test_probit_latents_long_stream(oracle_lib, hip_lib):
    """binary response at a size where the latent draws cross many generator blocks and windows (producer / consumer
    waves of k_latents) and both rejection branches occur; the R generator state must come out identical."""
    from stan4bart_amd import generate_friedman_data, make_sampler_args
    d = generate_friedman_data(6000, ranef=False, causal=True, binary=True, p=6)
    x = d["x"]
    args = make_sampler_args(d["y"], x[:, [0, 1, 2, 4, 5]], X=np.column_stack([x[:, 3], d["z"]]), family="binomial", iter=12, warmup=6,
                             bart_args={"n.trees": 15})
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1)
    assert_chain_parity(a, b, stan=False)


@pytest.mark.parametrize("scale", [1e5, 1e-6, 3e9, 1.0])
@pytest.mark.parametrize("hmc_mode", [0, 1])
def test_response_scale_extremes(oracle_lib, hip_lib, scale, hmc_mode):
    """The O(N) sums of the Stan block are accumulated in fixed point (k_stan_fused): the representation follows the magnitude of
    what is summed, and an evaluation whose range / resolution check fails is repeated in plain doubles.  Responses of order 1e5
    (sum of squares ~ 1e13 at n = 1.5e3: beyond the unscaled 2^43 range), 1e-6 (far below the unscaled 2^-56 resolution) and 3e9
    give the oracle's chain; once the chain has left its random initial values the fallback is the exception."""
    from conftest import make_sampler
    from stan4bart_amd import GroupTerm, make_sampler_args
    _, d = friedman_case(n=1500, T=10, warmup=7, iter=13, slopes=True)
    x = d["x"]
    xb = x[:, [j for j in range(10) if j != 3]]
    y = d["y"] * scale

    def mk(**extra):
        return make_sampler_args(y, xb, X=np.column_stack([x[:, 3], d["z"]]), groups=[GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")],
                                 iter=13, warmup=7, bart_args={"n.trees": 10}, **extra)
    a = run_chain(oracle_lib, "orc_", mk())
    b = run_chain(hip_lib, "s4b_", mk(stan_args={"hmc_mode": hmc_mode}))
    assert_chain_parity(a, b, rtol=1e-6, atol=1e-9 * scale)
    # a longer chain: the sampling phase runs on the fixed-point path
    args = mk(stan_args={"hmc_mode": hmc_mode})
    s = make_sampler(hip_lib, "s4b_", args)
    try:
        s.run(40, True, 0)
        e0, f0 = s.get_fused_stats()
        s.disengage_adaptation()
        s.run(40, False, 0)
        e1, f1 = s.get_fused_stats()
    finally:
        s.free()
    assert e1 - e0 >= 40 and f1 - f0 <= 2 + (e1 - e0) // 50, (e0, f0, e1, f1)


def _bart_args_cases():
    from test_bart_args import bart_args_cases
    big = [("split_probs_n20000", dict(n=20000, ranef=False, warmup=8, iter=20, bart_args={"split.probs": {0: 3.0, 4: 0.3}, "n.trees": 20})),
           # (150 predictors: the wave-register code answers for them in three groups of 64 lanes — counts in registers, in registers, in memory)
           ("split_probs_p150", dict(n=3000, p=150, ranef=False, warmup=8, iter=30, bart_args={"split.probs": {0: 6.0, 3: 4.0, 70: 5.0, 130: 7.0, 148: 0.1}, "n.trees": 10, "n.cuts": 3})),
           ("split_probs_p150_deep", dict(n=6000, p=150, ranef=False, warmup=8, iter=30, bart_args={"split.probs": {1: 6.0, 64: 9.0, 127: 5.0, 128: 7.0}, "n.trees": 6, "n.cuts": 2, "base": 0.99, "power": 0.6})),
           ("quantile_cuts_n20000", dict(n=20000, ranef=False, warmup=8, iter=20, bart_args={"useQuantiles": True, "n.cuts": 64, "n.trees": 20}))]
    return bart_args_cases() + big


@pytest.mark.parametrize("path", ["persistent", "fused", "two-kernel"])
@pytest.mark.parametrize("name,kw", _bart_args_cases(), ids=[c[0] for c in _bart_args_cases()])
def test_split_probs_and_quantile_cuts(oracle_lib, hip_lib, name, kw, path):
    """cgm(split.probs = ) (reference tests/testthat/test-09-bartArgs.R:20) and dbartsControl(useQuantiles = ) on every tree path.
    The weighted predictor draw exists in the persistent sweep (k_sweep_sp / k_sweep_few_sp: the wave-register control code compiled with it)
    and in the pointer-storage control code of the two-kernel path (k_control), which also serves a request for the fused path —
    s4b_get_tree_path reports both."""
    args, _ = friedman_case(**kw)
    joint = kw["n"] <= 1000
    a = run_chain(oracle_lib, "orc_", args, results_type=0 if joint else 1)
    b = run_chain(hip_lib, "s4b_", args, results_type=0 if joint else 1, tree_path=path)
    assert b["tree_path"][0] == path
    assert_chain_parity(a, b, stan=joint)
    assert b["tree_path"][1] == ("two-kernel" if "split.probs" in kw["bart_args"] and path != "persistent" else path)
    if path == "persistent":
        sweeps, handed_over = b["sweep_stats"]
        assert sweeps > 0 and handed_over == 0, b["sweep_stats"]


def test_test_row_fits_on_both_kernels(oracle_lib, hip_lib):
    """The fits of the test rows: up to 65 536 rows a thread per (row, tree) (k_test_fits_few), beyond that a thread per row (k_test_fits).  70 000 test rows
    that ARE the training rows: both against the oracle, and against the training fits of the same draw; the predictors' use counts (formed on the device
    since round 6) with them."""
    for n, n_test in ((70_000, 70_000), (70_000, 3_001)):
        args, _ = friedman_case(n=n, T=9, warmup=3, iter=6, ranef=False, n_test=n_test)
        a = run_chain(oracle_lib, "orc_", args)
        b = run_chain(hip_lib, "s4b_", args)
        assert_chain_parity(a, b)
        fits = b["sample"]["bart"]
        assert fits["test"].shape[0] == n_test
        np.testing.assert_allclose(fits["test"], fits["train"][:n_test], rtol=1e-12, atol=1e-12)
        assert np.array_equal(a["sample"]["bart"]["varcount"], fits["varcount"])


@pytest.mark.parametrize("hmc_mode", [0, 1])
def test_handed_over_sweeps_in_a_joint_chain(oracle_lib, hip_lib, hmc_mode):
    """The host queues the Stan inputs behind k_sweep without waiting for its status word (sweep_and_stan_inputs); a sweep that ends early —
    a tree outgrew the wave-register control path — is finished with k_step launches and the inputs are formed again.  Trees drawn from a
    deep prior in a joint chain: sweeps are handed over, and the Stan block still sees the finished sweep."""
    kw = dict(n=4000, T=2, warmup=5, iter=11, ranef=True, bart_args={"base": 0.99, "power": 0.25, "k": 0.3})
    args, _ = friedman_case(stan_args={"hmc_mode": hmc_mode}, **kw)
    oargs, _ = friedman_case(**kw)
    args.node_capacity = oargs.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", oargs)
    assert a["trace"][:, 4].max() > 32
    b = run_chain(hip_lib, "s4b_", args, tree_path="persistent")
    assert b["tree_path"][1] == "persistent"
    sweeps, handed_over = b["sweep_stats"]
    assert sweeps == 12 and handed_over > 0, b["sweep_stats"]
    assert_chain_parity(a, b)


def test_hand_over_with_every_workgroup_resident(oracle_lib, hip_lib):
    """n = 3e5: all 256 workgroups of k_sweep take part in the hand-over (tickets drawn across the 8 XCDs), trees of 12 ... 54 leaves drawn
    from a deep prior: every sweep starts persistent and ends as k_step launches"""
    args, _ = friedman_case(n=300000, T=6, warmup=2, iter=5, ranef=False, bart_args={"base": 0.99, "power": 0.3, "k": 0.3})
    args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    assert a["trace"][:, 4].max() > 32
    for path in ("persistent", "fused"):
        b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path=path)
        assert_chain_parity(a, b, stan=False)
        if path == "persistent":
            assert b["tree_path"][1] == "persistent" and b["sweep_stats"] == (6, 6)


def test_two_samplers_in_two_threads_take_turns_on_the_persistent_path(hip_lib):
    """Samplers of one process that share a device without saying so (no s4b_set_device_sharing): persistent sweeps need every workgroup
    resident, so the library serialises them per device.  Two chains driven from two threads finish and draw exactly what each draws alone."""
    import threading
    args, _ = friedman_case(n=60000, T=30, warmup=5, iter=12, ranef=True)
    alone = run_chain(hip_lib, "s4b_", args, seed=777, tree_path="persistent")
    assert alone["tree_path"][1] == "persistent"
    results, errors = {}, []

    def work(k):
        try:
            results[k] = run_chain(hip_lib, "s4b_", args, seed=777, tree_path="persistent")
        except Exception as e:          # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for k in range(2):
        assert results[k]["tree_path"][1] == "persistent"
        assert np.array_equal(results[k]["trace"], alone["trace"]) and np.array_equal(results[k]["rng"], alone["rng"])
        np.testing.assert_array_equal(results[k]["sample"]["stan"], alone["sample"]["stan"])


def test_hand_over_from_the_one_workgroup_sweep(oracle_lib, hip_lib):
    """At most 4096 observations: k_sweep runs as ONE workgroup (no exchange), while the k_step launches that finish a handed-over sweep use
    their own geometry (here two pass workgroups).  Regression (fuzz seed 3738): the hand-over summed the per-workgroup partial slots of
    k_step's geometry — the second one stale from an earlier sweep — instead of the one slot the single workgroup had written."""
    from stan4bart_amd import make_sampler_args
    g = np.random.default_rng(3738)
    n = 2317
    xb = np.column_stack([g.normal(size=n), g.random(n), (g.random(n) < 0.3).astype(np.float64)])
    x4 = g.random(n)
    y = 3.0 * np.sin(2.0 * xb[:, 0]) + 2.0 * (xb[:, 2] > 0.5) + 1.5 * x4 + g.normal(size=n)
    args = make_sampler_args(y, xb, X=x4[:, None], groups=[], iter=14, warmup=6, bart_args={"n.trees": 8, "n.cuts": 5, "k": 0.3, "base": 0.99, "power": 0.25})
    args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    assert a["trace"][:, 4].max() > 32
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="persistent")
    assert b["tree_path"][1] == "persistent" and b["sweep_stats"][1] >= 10
    assert_chain_parity(a, b, stan=False)


@pytest.mark.parametrize("path", ["persistent", "fused", "two-kernel"])
@pytest.mark.parametrize("kind", ["continuous", "binary"])
def test_k_hyperprior_on_every_tree_path(oracle_lib, hip_lib, kind, path):
    """normal(k = chi(1.25, Inf)) (reference R/stan4bart.R:202, tests/testthat/test-09-bartArgs.R:32): after every sweep — trees, then the
    latents of a binary response — k_draw_k draws k from its conditional given the leaf values, from R's stream on the device, and the leaf
    prior precision of the next sweep's launches follows.  BART block over 1 500 tree updates on each tree path, against the oracle's
    independent restatement: trace and generator bit-exact, k draws (result element "k", src/bart_util.cpp:17-26,75-76) within 1e-6."""
    if kind == "binary":
        from conftest import binary_case
        args = binary_case(n=3000, T=15, warmup=40, iter=100)
        args.k_hyper, args.k = (1.25, np.inf), 2.0
    else:
        args, _ = friedman_case(n=3000, T=15, warmup=40, iter=100, bart_args={"k": "chi(1.25, Inf)"})
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path=path)
    assert b["tree_path"][1] == path
    assert_chain_parity(a, b, stan=False)
    k = b["sample"]["bart"]["k"]
    assert k.shape == (60,) and np.std(k) > 0 and 0.2 < np.median(k) < 20


def test_k_hyperprior_joint_chain_and_teacher_forcing(oracle_lib, hip_lib):
    """the joint chain with a modeled k over the reference's test horizon, and teacher-forced through the state blob (the current k travels
    in header.reserved[0])"""
    from conftest import teacher_forced
    args, _ = friedman_case(n=500, ranef=True, slopes=True, n_test=11, bart_args={"k": "chi(2.5, 4)"})
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(hip_lib, "s4b_", args))
    args, _ = friedman_case(n=400, ranef=True, warmup=10, iter=30, T=9, bart_args={"k": "chi(1.25, Inf)"})
    teacher_forced(oracle_lib, hip_lib, "s4b_", args)
