"""The streaming variant of the persistent sweep (k_sweep_stream, dev_sweep.inc; tree path "stream"): same launch, same wave roles, same
exchange and redundant decisions as k_sweep, but the pass waves read and write the residual per tree (22 B per observation and tree update)
instead of holding it in registers — the variant VERDICT round 4 asked for beyond 16 observations per pass thread (n > 1 044 480).  It
reproduces the oracle's chain everywhere, and it is slower than k_step / k_tree + k_control at every size (four register-heavy pass waves per
compute unit stream at 3.1 TB/s: DESIGN.md 8, round 5), so the automatic choice never takes it; it stays as a tested, selectable path
(reference semantics of one sweep: src/init.cpp:824)."""
import numpy as np
import pytest

from conftest import assert_chain_parity, binary_case, friedman_case, run_chain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(), dict(T=50), dict(ranef=False), dict(slopes=True, n_test=17), dict(n=1003, T=50), dict(n=7, T=3, warmup=2, iter=4, ranef=False),
    dict(T=1, warmup=10, iter=30, ranef=False), dict(T=2, warmup=10, iter=30), dict(skip=(2, 1)), dict(stan_args={"hmc_mode": 1}),
], ids=str)
def test_streaming_sweep_joint_chain(oracle_lib, hip_lib, kw):
    kw_o = {k: v for k, v in kw.items() if k != "stan_args"}
    a = run_chain(oracle_lib, "orc_", friedman_case(**kw_o)[0])
    b = run_chain(hip_lib, "s4b_", friedman_case(**kw)[0], tree_path="stream")
    assert b["tree_path"] == ("stream", "stream")
    assert_chain_parity(a, b)


@pytest.mark.parametrize("n", [5000, 1003, 70000])
def test_streaming_sweep_bart_block_long_run(oracle_lib, hip_lib, n):
    """2 400 tree updates; n = 70 000: every pass thread works through several batches per tree"""
    args, _ = friedman_case(n=n, T=40, warmup=30, iter=60)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="stream")
    assert b["tree_path"] == ("stream", "stream") and b["sweep_stats"] == (61, 0)
    assert set(np.unique(a["trace"][:, 0])) == {0, 1, 2, 3}
    assert_chain_parity(a, b, stan=False)


def test_streaming_sweep_multi_pass_bins_deep_trees_and_hand_over(oracle_lib, hip_lib):
    """more than 8 bins: several passes over the observations per tree (the first one also finishes tree t-1); trees beyond the 64 node slots
    of the wave-register control path: the rest of the sweep is handed over to k_step launches"""
    args, _ = friedman_case(n=2000, T=4, warmup=20, iter=40, ranef=False, bart_args={"base": 0.99, "power": 0.45, "k": 0.5})
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    assert a["trace"][:, 4].max() > 16
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="stream")
    assert b["tree_path"][1] == "stream"
    assert_chain_parity(a, b, stan=False)
    args, _ = friedman_case(n=300000, T=6, warmup=2, iter=5, ranef=False, bart_args={"base": 0.99, "power": 0.3, "k": 0.3})
    args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    assert a["trace"][:, 4].max() > 32
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="stream")
    assert b["tree_path"][1] == "stream" and b["sweep_stats"] == (6, 6)
    assert_chain_parity(a, b, stan=False)


def test_streaming_sweep_binary_k_hyperprior_and_path_changes(oracle_lib, hip_lib):
    from conftest import make_sampler
    args = binary_case(n=3000, T=15, warmup=20, iter=50)
    args.k_hyper, args.k = (1.25, np.inf), 2.0
    assert_chain_parity(run_chain(oracle_lib, "orc_", args, results_type=1), run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="stream"), stan=False)
    # switching between the register and the streaming variant (and k_step) between runs: every path starts a sweep from the same state
    args, _ = friedman_case(n=3000, T=12, warmup=8, iter=16, ranef=True)
    a = run_chain(oracle_lib, "orc_", args)
    for first, second in (("persistent", "stream"), ("stream", "fused"), ("two-kernel", "stream")):
        s = make_sampler(hip_lib, "s4b_", args)
        try:
            s.set_trace(True)
            s.set_tree_path(first); s.run(args.warmup, True); t1 = s.get_trace()
            s.disengage_adaptation()
            s.set_tree_path(second); r = s.run(args.iter - args.warmup, False); t2 = s.get_trace()
            assert s.get_tree_path() == (second, second)
            assert np.array_equal(np.concatenate([t1, t2]), a["trace"]) and np.array_equal(s.get_r_rng_state(), a["rng"])
            np.testing.assert_allclose(r["bart"]["train"], a["sample"]["bart"]["train"], rtol=1e-6, atol=1e-9)
        finally:
            s.free()


def test_streaming_sweep_beyond_the_register_capacity(oracle_lib, hip_lib):
    """n = 2e6 (twice what k_sweep's registers hold): on request the streaming variant runs there, one launch per sweep, and the chain is the
    oracle's; the automatic choice at that size stays the fused launch, and "persistent" falls back to it"""
    from conftest import make_sampler
    args, _ = friedman_case(n=2_000_000, p=10, T=12, warmup=2, iter=5, ranef=False)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="stream")
    assert b["tree_path"] == ("stream", "stream") and b["sweep_stats"] == (5, 0)      # (the sweep at creation ran on the automatic path)
    assert_chain_parity(a, b, stan=False)
    s = make_sampler(hip_lib, "s4b_", args)
    try:
        assert s.get_tree_path() == ("auto", "fused")
        s.set_tree_path("persistent")
        assert s.get_tree_path() == ("persistent", "fused")
    finally:
        s.free()
