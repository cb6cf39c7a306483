"""The sizes at which the persistent sweep fills the device — up to 255 pass workgroups exchanging bin partials, both exchange rings, statistics
published before the verdict and redone when it says otherwise — against the oracle, in the driver-run suite (until
round 6 only tools/fuzz_large.py and tools/soak.py, which the driver never runs, went there; the suite's other fuzz generator stops at
n = 40 000 = 10 pass workgroups).  Eight configurations of tests/large_cases.py, BART block, 30 and more iterations each: n = 5e4 ... 1.044e6
(the largest n the persistent sweep takes), two with trees of tens of leaves (sweeps handed over to k_step part-way), two with a modeled k.
Four more with cgm(split.probs = ): the weighted predictor choice compiled into the wave-register control code (k_sweep_sp / k_sweep_few_sp), one of
them with a deep prior (the sweeps end as k_step launches whose control steps all run in the launch's tail, the only code of k_step that knows the weights).
Same bar as everywhere: tree-move trace, trees, generator state bit-exact; floating-point state to 1e-6."""
import pytest

from conftest import assert_chain_parity, run_chain
from large_cases import large_case

pytestmark = pytest.mark.gpu

# (seed, n, (warm-up, total iterations), deep prior, modeled k, trees)
LARGE = [
    (0, 50_000, (10, 40), False, None, 12),
    (1, 131_072, (10, 40), False, (1.25, float("inf")), 8),
    (2, 262_144, (8, 38), True, None, 5),
    (3, 400_003, (10, 40), False, None, 10),
    (4, 655_360, (6, 36), False, (2.0, 1.5), 6),
    (5, 820_001, (5, 35), True, None, 3),
    (6, 1_000_000, (10, 40), False, None, 6),
    (7, 1_044_480, (5, 35), False, None, 4),
]


# the same with cgm(split.probs = )
LARGE_SP = [
    (10, 60_000, (10, 40), False, None, 9),
    (11, 300_000, (8, 38), False, (1.25, float("inf")), 7),
    (12, 500_000, (6, 30), True, None, 4),
    (13, 1_000_000, (10, 40), False, None, 6),
]


@pytest.mark.parametrize("seed,n,iters,deep,k_chi,trees,split_probs", [c + (False,) for c in LARGE] + [c + (True,) for c in LARGE_SP])
def test_large_configuration_on_the_persistent_path(oracle_lib, hip_lib, seed, n, iters, deep, k_chi, trees, split_probs):
    args, what = large_case(seed, n=n, iters=iters, deep=deep, k_chi=k_chi, trees=trees, split_probs=split_probs)
    a = run_chain(oracle_lib, "orc_", args, results_type=1)
    b = run_chain(hip_lib, "s4b_", args, results_type=1, tree_path="persistent")
    assert b["tree_path"] == ("persistent", "persistent"), (what, b["tree_path"])
    try:
        assert_chain_parity(a, b, stan=False)
    except AssertionError as e:
        raise AssertionError(f"large case {seed}, {what}: {e}") from e
    sweeps, handed = b["sweep_stats"]
    assert sweeps == iters[1] + 1 and b["sweep_busy"] == 0, (what, b["sweep_stats"], b["sweep_busy"])
    if deep:
        assert handed > 0, (what, b["sweep_stats"])          # trees beyond the 64 node slots of the wave-register control code
    launches, inside, published_early, borne_out = b["sweep_spec"]
    assert launches == sweeps and inside <= sweeps * trees
    if not deep and what["binary"] is False:
        # every workgroup of these launches is resident and exchanges: most steps publish their statistics before the verdict
        assert borne_out <= published_early <= inside
        assert published_early > 0.4 * inside, (what, b["sweep_spec"])
