"""A divergence from the reference that round 6's weighted fuzz campaign found and that is NOT fixed: dbarts — restated in oracle/bart_ref.hpp — gives a
branch with an EMPTY leaf the log-likelihood -1e7 and compares it with the FULL integrated likelihood of the other branch (within-leaf sum of squares included); the product
(tree_hd.hpp decide / leaf_loglik: every device path, and the CPU emulation of the device layer used here) carries only the terms that do not cancel between two non-empty
branches.  Once sum of w (r - mean)^2 / sigma^2 of a branch exceeds 2e7 the reference accepts a rule that leaves a leaf empty and the product rejects it.  Observation weights
of 9 ... 148 at n = 424 330 get there in the sweep at creation.  strict xfail: the day the product follows the reference here this test must be turned around."""
import numpy as np
import pytest

from conftest import make_sampler
from large_cases import large_case


def _weights_of_the_first_generator(seed, n):
    g = np.random.default_rng(700000 + seed)
    w = g.uniform(0.25, 4.0, n) * 37.0
    w[g.integers(0, n, 100)] *= 1e-3
    return w


@pytest.mark.xfail(strict=True, reason="empty-leaf rule at |log-likelihood| > 1e7: the reference accepts, the product rejects (DESIGN.md 7)")
def test_trees_after_creation_with_heavy_weights_match_the_oracle(oracle_lib, emul_lib):
    args, what = large_case(79, weights=True, iters=(1, 2))
    args.weights[:] = _weights_of_the_first_generator(79, what["n"])
    so, se = make_sampler(oracle_lib, "orc_", args), make_sampler(emul_lib, "emu_", args)
    try:
        to, te = so.get_trees(), se.get_trees()
        assert (to["n"][to["var"] < 0] == 0).any()            # the oracle's trees carry a leaf without observations ...
        assert not (te["n"][te["var"] < 0] == 0).any()        # ... the product's do not
        assert np.array_equal(to["var"], te["var"]) and np.array_equal(to["n"], te["n"])
    finally:
        so.free(); se.free()


def test_the_same_data_with_weights_of_order_one_match(oracle_lib, emul_lib):
    """(the same configuration in the regime both sides implement)"""
    args, what = large_case(79, weights=True, iters=(1, 2))
    so, se = make_sampler(oracle_lib, "orc_", args), make_sampler(emul_lib, "emu_", args)
    try:
        to, te = so.get_trees(), se.get_trees()
        assert np.array_equal(to["var"], te["var"]) and np.array_equal(to["n"], te["n"])
        np.testing.assert_allclose(to["value"], te["value"], rtol=1e-6, atol=1e-9)
        assert np.array_equal(so.get_r_rng_state(), se.get_r_rng_state())
    finally:
        so.free(); se.free()
