"""The busy fallback of the persistent tree path: a persistent launch whose roll call finds the device shared changes
nothing, its sweep is rerun as per-tree launches (k_step; the two-kernel update when the sharing hint says three or more chains), the next
16, 32, ... sweeps stay there, and in a joint chain the Stan inputs queued behind the launch are discarded and formed again.  Until round 6
only tools/two_process_probe.py went there, and it compared nothing.  The test hook of the C-ABI (s4b_set_test_hook 1: every k-th persistent
launch finds its roll call decided "busy") drives those branches inside one process; the chain must be the oracle's chain, bit for bit where
the suite demands it everywhere else."""
import numpy as np
import pytest

from conftest import assert_chain_parity, binary_case, friedman_case, run_chain

pytestmark = pytest.mark.gpu
N = 24_000      # 6 pass workgroups + the control workgroup: a launch with a roll call (a one-workgroup launch holds none)


def _check(a, b, min_busy, stan=True):
    assert b["tree_path"][1] == "persistent"
    assert b["sweep_busy"] >= min_busy, b["sweep_busy"]
    assert_chain_parity(a, b, stan=stan)


@pytest.mark.parametrize("every", [2, 3])
def test_busy_launches_in_a_joint_chain(oracle_lib, hip_lib, every):
    """the Stan inputs of an iteration are queued behind the persistent launch before anybody knows it was refused: discarded, formed again"""
    args, _ = friedman_case(n=N, T=11, warmup=6, iter=13)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, every))
    _check(a, b, 1)
    assert b["sweep_stats"][0] == 14          # (every sweep is counted, whoever ran it)
    assert b["sweep_spec"][0] < 14            # ... but fewer persistent launches ran than sweeps


def test_busy_launches_in_a_bart_only_chain_with_back_off(oracle_lib, hip_lib):
    """results_type 1, 70 iterations: busy launch, 16 sweeps of back-off, persistent again, busy again, 32 sweeps"""
    args, _ = friedman_case(n=N, T=5, warmup=10, iter=70)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, test_hook=(1, 3))
    _check(a, b, 2, stan=False)
    launches = b["sweep_spec"][0]
    assert 0 < launches < 71 - 16


def test_busy_on_the_last_sweep_of_a_thinned_iteration(oracle_lib, hip_lib):
    """skip.bart = 3: the first two sweeps of an iteration are launches of their own, the third carries the Stan inputs behind it.  Launch 1 is the
    hook counts the launches after it was set, i.e. after the sweep at creation: launches 1 - 3 are the first iteration, every third launch busy hits
    the one with the Stan inputs queued behind it, in every iteration that launches at all."""
    args, _ = friedman_case(n=N, T=7, warmup=5, iter=12, skip=(1, 3))
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, 3))
    _check(a, b, 1)


def test_busy_launches_with_a_binary_response(oracle_lib, hip_lib):
    args = binary_case(n=N, T=7, warmup=5, iter=11)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, 2))
    _check(a, b, 1)


def test_busy_launches_with_a_modeled_k(oracle_lib, hip_lib):
    args, _ = friedman_case(n=N, T=9, warmup=5, iter=12, bart_args={"k": ("chi", 1.25, float("inf"))})
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, 3))
    _check(a, b, 1)
    np.testing.assert_allclose(a["sample"]["bart"]["k"], b["sample"]["bart"]["k"], rtol=1e-6)


def test_busy_fallback_honours_the_sharing_hint(oracle_lib, hip_lib):
    """three or more chains on the device: the refused sweep and the back-off run on the two-kernel update, not on the fused launch"""
    args, _ = friedman_case(n=N, T=11, warmup=6, iter=13)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, 2), sharing=(4, 4))
    _check(a, b, 1)


def test_busy_launches_with_split_probs(oracle_lib, hip_lib):
    """cgm(split.probs = ): the persistent sweep is k_sweep_sp; a refused launch and its back-off run on the two-kernel update (k_control's
    pointer-storage control code knows the weights, k_step's wave-register code does not)"""
    args, _ = friedman_case(n=N, T=9, warmup=6, iter=14, bart_args={"split.probs": {0: 4.0, 3: 0.25, 7: 2.0}})
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args, test_hook=(1, 2))
    _check(a, b, 1)


def test_handed_over_sweeps_with_split_probs(oracle_lib, hip_lib):
    """a deep prior with cgm(split.probs = ): trees outgrow the 64 node slots of the wave-register control code, k_sweep_sp ends early and the sweep is
    finished with k_step launches — whose control steps then all run in the launch's sequential tail"""
    kw = dict(n=N, T=3, warmup=5, iter=12, ranef=True, bart_args={"base": 0.99, "power": 0.25, "k": 0.3, "split.probs": [3, 1, 1, 0.5, 1, 2, 1, 1, 0.1]})
    args, _ = friedman_case(**kw)
    args.node_capacity = 1024
    a = run_chain(oracle_lib, "orc_", args)
    assert a["trace"][:, 4].max() > 32
    b = run_chain(hip_lib, "s4b_", args, tree_path="persistent")
    assert b["tree_path"][1] == "persistent"
    sweeps, handed_over = b["sweep_stats"]
    assert sweeps == 13 and handed_over > 0, b["sweep_stats"]
    assert_chain_parity(a, b)


def test_the_hook_is_refused_for_anything_else(hip_lib):
    from conftest import make_sampler
    args, _ = friedman_case(n=200)
    s = make_sampler(hip_lib, "s4b_", args)
    try:
        with pytest.raises(RuntimeError, match="set_test_hook"):
            s.set_test_hook(2, 1)
        with pytest.raises(RuntimeError, match="set_test_hook"):
            s.set_test_hook(1, -1)
    finally:
        s.free()
