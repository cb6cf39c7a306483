"""The seeded random configurations of test_gpu_fuzz.py, CPU side: the product's host logic over the emulated device layer
(tests/emul) against the oracle — what `pytest -m "not gpu"` can check of them without a GPU."""
import pytest

from conftest import assert_chain_parity, run_chain
from test_gpu_fuzz import REGRESSION_SEEDS, random_case


@pytest.mark.parametrize("seed", list(range(160)) + REGRESSION_SEEDS)
def test_random_configuration_host_logic(oracle_lib, emul_lib, seed):
    args, joint, what = random_case(seed)
    rt = 0 if joint else 1
    a = run_chain(oracle_lib, "orc_", args, results_type=rt)
    b = run_chain(emul_lib, "emu_", args, results_type=rt)
    try:
        assert_chain_parity(a, b, stan=joint)
    except AssertionError as e:
        raise AssertionError(f"seed {seed}, case {what}: {e}") from e
