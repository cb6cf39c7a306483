"""Teacher-forced per-transition parity of the HIP path (through the C-ABI) against the CPU oracle: the oracle's state is
injected into the HIP sampler before every Gibbs iteration (s4b_set_state), both advance one iteration, and everything is
compared — bit-exact tree-move trace / R generator / NUTS integer diagnostics, 1e-6 on floats (BASELINE.json north_star).
Covers what free-running chains cannot reach before rounding differences decorrelate them: metric-window ends
(reference var_adaptation.hpp:17-46, adapt_diag_e_nuts.hpp:28-41), treedepth >= 6 (base_nuts.hpp:247-352), divergences
(base_nuts.hpp:262), the non-finite-energy path (base_hamiltonian.hpp:61-70).  Same harness as tests/test_teacher_forced.py.
"""
import numpy as np
import pytest

from conftest import StateView, assert_state_parity, binary_case, friedman_case, make_sampler, teacher_forced

pytestmark = pytest.mark.gpu


def test_state_round_trip_on_device(hip_lib):
    args, _ = friedman_case(n=1003, T=20, warmup=8, iter=14, slopes=True)
    a = make_sampler(hip_lib, "s4b_", args)
    a.run(5, True)
    blob = a.get_state()
    ra = a.run(3, True)
    b = make_sampler(hip_lib, "s4b_", args, seed=777)
    b.set_state(blob)
    assert_state_parity(StateView(blob), StateView(b.get_state()), rtol=1e-13, atol=1e-13)
    rb = b.run(3, True)
    np.testing.assert_allclose(ra["stan"], rb["stan"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ra["bart"]["train"], rb["bart"]["train"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(a.get_r_rng_state(), b.get_r_rng_state())
    a.free(); b.free()


@pytest.mark.parametrize("kw", [dict(), dict(stan_args={"hmc_mode": 1})], ids=["gram", "per-leapfrog-kernels"])
def test_forced_through_two_metric_windows_ranef(oracle_lib, hip_lib, kw):
    """the reference's window schedule (75 / 50 / 25) with warmup = 200 and random slopes: window ends at transitions 99 and
    149 (metric update, init_stepsize, dual-averaging restart), then the sampling phase after disengage."""
    args, _ = friedman_case(n=300, T=20, warmup=200, iter=212, slopes=True, **kw)
    rows, ends = teacher_forced(oracle_lib, hip_lib, "s4b_", args)
    assert ends == [99, 149], ends
    assert rows[3].max() >= 6          # deep trajectories occur on the way


def test_forced_deep_trajectories(oracle_lib, hip_lib):
    """every sampling transition at treedepth >= 6: the adapted step size divided by 16."""
    args, _ = friedman_case(n=500, T=10, warmup=40, iter=52, slopes=True)
    args.adapt_init_buffer, args.adapt_term_buffer, args.adapt_window = 10, 10, 10
    base = {}

    def patch(it, sv):
        if it < 40:
            return False
        nuts = sv.get("nuts"); base.setdefault("eps", nuts[0]); nuts[0] = base["eps"] / 16; sv.set("nuts", nuts)
        return True
    rows, _ = teacher_forced(oracle_lib, hip_lib, "s4b_", args, patch=patch)
    assert rows[3, 40:].min() >= 6, rows[3]


@pytest.mark.parametrize("mode", [0, 1])
def test_forced_divergence_and_nonfinite_energy(oracle_lib, hip_lib, mode):
    args, _ = friedman_case(n=400, T=10, warmup=3, iter=8, slopes=True, stan_args={"hmc_mode": mode})
    sizes = {1: 30.0, 2: 1e3, 4: 1e6, 6: 2.0}

    def patch(it, sv):
        if it not in sizes:
            return False
        nuts = sv.get("nuts"); nuts[0] = sizes[it]; sv.set("nuts", nuts)
        return True
    rows, _ = teacher_forced(oracle_lib, hip_lib, "s4b_", args, patch=patch)
    assert rows[5, 1] == 1 and rows[5, 2] == 1 and rows[5, 4] == 1, rows[5]


def test_forced_probit(oracle_lib, hip_lib):
    args = binary_case(n=747, T=20, warmup=30, iter=40)
    args.adapt_init_buffer, args.adapt_term_buffer, args.adapt_window = 5, 5, 10
    rows, ends = teacher_forced(oracle_lib, hip_lib, "s4b_", args)
    assert len(ends) >= 1


def test_forced_long_chain_weights_and_test_rows(oracle_lib, hip_lib):
    """300 forced iterations with observation weights and test rows: no horizon."""
    w = np.random.default_rng(11).uniform(0.5, 2.0, 250)
    args, _ = friedman_case(n=250, T=15, warmup=150, iter=300, weights=w, n_test=9)
    args.adapt_init_buffer, args.adapt_term_buffer, args.adapt_window = 20, 20, 15
    rows, ends = teacher_forced(oracle_lib, hip_lib, "s4b_", args, compare_states=False)
    assert len(ends) >= 2
