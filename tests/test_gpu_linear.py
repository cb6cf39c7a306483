"""k_sweep with the LINEAR statistics (`make -C stan4bart_amd/csrc linear` -> libs4b_linear.so; dev_sweep.inc "LINEARITY"): the O(N) statistics
pass of a tree runs one step ahead and gathers integer contingency counts against the leaves of the tree before it; the exchange wave completes the workgroup's
statistics from the new leaf values with leaves x bins multiply-adds and publishes them itself.  Built in round 6, the same chain as the oracle's on every test of the
GPU suite, and measured SLOWER than the default in the benchmark's stationary chain: it is not the default, and these tests keep it honest — the
persistent path against the oracle at sizes from six pass workgroups to the full grid, hand-overs, a modeled k, a joint chain, and the stationary regime."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT, StateView, assert_chain_parity, assert_state_parity, friedman_case, make_sampler, run_chain
from large_cases import large_case
from test_gpu_fuzz import random_case

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "stan4bart_amd", "csrc", "libs4b_linear.so")


@pytest.fixture(scope="module")
def linear_lib():
    if not os.path.exists(LIB):
        pytest.skip("libs4b_linear.so not built (make -C stan4bart_amd/csrc linear)")
    lib = ctypes.CDLL(LIB)
    lib.s4b_last_error.restype = ctypes.c_char_p
    return lib


def test_joint_chain(oracle_lib, linear_lib):
    args, _ = friedman_case(n=24_000, T=11, warmup=7, iter=13, slopes=True)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(linear_lib, "s4b_", args)
    assert b["tree_path"][1] == "persistent"
    assert_chain_parity(a, b)
    launches, inside, early, ok = b["sweep_spec"]
    assert launches == 14 and early > 0 and ok <= early <= inside


@pytest.mark.parametrize("seed,n,iters,deep,k_chi,trees", [(0, 50_000, (10, 40), False, None, 12), (2, 262_144, (8, 38), True, None, 5),
                                                           (4, 655_360, (6, 36), False, (2.0, 1.5), 6), (7, 1_044_480, (5, 35), False, None, 4)])
def test_large_configuration(oracle_lib, linear_lib, seed, n, iters, deep, k_chi, trees):
    args, what = large_case(seed, n=n, iters=iters, deep=deep, k_chi=k_chi, trees=trees)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(linear_lib, "s4b_", args, results_type=1, tree_path="persistent")
    assert b["tree_path"] == ("persistent", "persistent"), (what, b["tree_path"])
    assert_chain_parity(a, b, stan=False)
    if deep:
        assert b["sweep_stats"][1] > 0


@pytest.mark.parametrize("seed", range(0, 160, 8))
def test_random_configuration(oracle_lib, linear_lib, seed):
    args, joint, what = random_case(seed)
    rt = 0 if joint else 1
    a = run_chain(oracle_lib, "orc_", args, results_type=rt)
    b = run_chain(linear_lib, "s4b_", args, results_type=rt, tree_path="persistent")
    try:
        assert_chain_parity(a, b, stan=joint)
    except AssertionError as e:
        raise AssertionError(f"seed {seed} on the persistent path with the linear statistics, case {what}: {e}") from e


def test_full_size_young_chain_three_times(oracle_lib, linear_lib):
    """n = 1e6, p = 50, 200 trees: a young chain accepts a quarter of its moves — every other step falls back to the statistics done the old way, with the ahead pass
    behind it.  Three runs: the races this path had while it was built showed in one run of four."""
    args, _ = friedman_case(n=1_000_000, p=50, T=200, warmup=2, iter=3, slopes=True)
    a = run_chain(oracle_lib, "orc_", args)
    for _ in range(3):
        b = run_chain(linear_lib, "s4b_", args)
        assert_chain_parity(a, b)


def test_stationary_regime_teacher_forced(oracle_lib, linear_lib):
    """as tests/test_gpu_configs.py::test_config3_stationary_teacher_forced, one iteration: 95 % of the steps publish from the ahead pass here"""
    burn = 300
    args, _ = friedman_case(n=1_000_000, p=50, T=200, warmup=burn, iter=burn + 2, slopes=True, keep_fits=False)
    sp = make_sampler(linear_lib, "s4b_", args)
    so = make_sampler(oracle_lib, "orc_", args)
    try:
        sp.run(burn, True, 0)
        sp.disengage_adaptation(); so.disengage_adaptation()
        spec0 = sp.get_sweep_spec()
        st = sp.get_state()
        so.set_trace(True); sp.set_trace(True)
        so.set_state(st)
        ro = so.run(1, False, 0)
        rp = sp.run(1, False, 0)
        assert np.array_equal(so.get_trace(), sp.get_trace())
        assert np.array_equal(ro["stan"][3:6], rp["stan"][3:6])
        np.testing.assert_allclose(ro["stan"], rp["stan"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(ro["bart"]["train"], rp["bart"]["train"], rtol=1e-6, atol=1e-9)
        assert_state_parity(StateView(so.get_state()), StateView(sp.get_state()))
        ran, inside, early, ok = (b - a for a, b in zip(spec0, sp.get_sweep_spec()))
        assert ran == 1 and inside == 200 and early >= 160 and ok >= 0.9 * early, (ran, inside, early, ok)
    finally:
        so.free(); sp.free()
