"""Teacher-forced per-transition parity on the CPU: the product's host logic (NUTS, adaptation, tape gradient, tree control
code, C-ABI) over the CPU emulation of the device layer, against the oracle, with the oracle's state injected before
every Gibbs iteration (s4b_set_state).  Free-running chains decorrelate after ~20 iterations (NUTS amplifies rounding
differences); forcing removes the horizon, so everything that only happens late in a chain is compared transition by
transition: metric-window ends (reference var_adaptation.hpp:17-46, adapt_diag_e_nuts.hpp:28-41), deep trajectories
(base_nuts.hpp:247-352), divergences (base_nuts.hpp:262), the non-finite-energy path (base_hamiltonian.hpp:61-70).
The same harness runs against the HIP path in test_gpu_teacher_forced.py.
"""
import numpy as np
import pytest

from conftest import StateView, assert_state_parity, friedman_case, make_sampler, teacher_forced

SMALL_WINDOWS = dict(adapt_init_buffer=10, adapt_term_buffer=10, adapt_window=10)   # windows end at transitions 19 and 49


def _small_windows(args):
    for k, v in SMALL_WINDOWS.items():
        setattr(args, k, v)
    return args


def test_state_round_trip(oracle_lib, emul_lib):
    """get_state -> set_state on a fresh sampler of the same implementation continues the chain bit for bit (checkpoint / resume),
    and the blob of one implementation is accepted by the other."""
    args, _ = friedman_case(n=150, T=7, warmup=8, iter=14, slopes=True)
    for lib, pfx in ((oracle_lib, "orc_"), (emul_lib, "emu_")):
        a = make_sampler(lib, pfx, args)
        a.run(5, True)
        blob = a.get_state()
        ra = a.run(3, True)
        b = make_sampler(lib, pfx, args, seed=777)          # different seed: everything must come from the blob
        b.set_state(blob)
        assert_state_parity(StateView(blob), StateView(b.get_state()), rtol=1e-13, atol=1e-13)
        rb = b.run(3, True)
        np.testing.assert_allclose(ra["stan"], rb["stan"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ra["bart"]["train"], rb["bart"]["train"], rtol=1e-9, atol=1e-12)
        assert np.array_equal(a.get_r_rng_state(), b.get_r_rng_state())
        a.free(); b.free()


def test_state_rejects_malformed(emul_lib):
    args, _ = friedman_case(n=60, T=3, warmup=2, iter=4)
    s = make_sampler(emul_lib, "emu_", args)
    blob = s.get_state()
    with pytest.raises(RuntimeError, match="truncated"):
        s.set_state(blob[:-9])
    with pytest.raises(RuntimeError, match="not a stan4bart sampler state"):
        s.set_state(b"\0" * len(blob))
    other, _ = friedman_case(n=61, T=3, warmup=2, iter=4)
    t = make_sampler(emul_lib, "emu_", other)
    with pytest.raises(RuntimeError, match="dimensions"):
        t.set_state(blob)
    s.free(); t.free()


@pytest.mark.parametrize("kw", [dict(n=120, T=9, slopes=True), dict(n=200, T=11, ranef=False), dict(n=150, T=7, slopes=True, stan_args={"hmc_mode": 1})],
                         ids=["ranef-slopes", "fixef-only", "hmc_mode1"])
def test_forced_through_metric_windows(oracle_lib, emul_lib, kw):
    """60 warm-up iterations with small adaptation windows (two window ends: metric update, init_stepsize, dual-averaging
    restart), then sampling after disengage; every iteration compared from the oracle's state."""
    args = _small_windows(friedman_case(warmup=60, iter=75, **kw)[0])
    rows, ends = teacher_forced(oracle_lib, emul_lib, "emu_", args)
    assert ends == [19, 49], ends
    assert rows[3].max() >= 4


def test_forced_default_windows_warmup200(oracle_lib, emul_lib):
    """the reference's own window schedule (init 75 / term 50 / base 25; R cannot change them: R/stan4bart_fit.R:482-488) with
    warmup = 200: window ends at transitions 99 and 149."""
    args, _ = friedman_case(n=100, T=5, warmup=200, iter=206)
    rows, ends = teacher_forced(oracle_lib, emul_lib, "emu_", args, compare_states=False)
    assert ends == [99, 149], ends


def test_free_running_fixef_through_window_end(oracle_lib, emul_lib):
    """un-forced: ranef = FALSE, n = 400, warmup = 24 stays in parity through the end of warm-up."""
    from conftest import assert_chain_parity, run_chain
    args, _ = friedman_case(n=400, ranef=False, warmup=24, iter=30)
    assert_chain_parity(run_chain(oracle_lib, "orc_", args), run_chain(emul_lib, "emu_", args))


def test_forced_deep_trajectories(oracle_lib, emul_lib):
    """transitions with treedepth >= 6 (64+ leapfrogs; the nested sub-tree u-turn checks of base_nuts.hpp:247-352): after
    warm-up the sampling phase is forced to 1/16 of the adapted step size, which makes every trajectory long."""
    args = _small_windows(friedman_case(n=120, T=5, warmup=40, iter=50, slopes=True)[0])
    base = {}

    def patch(it, sv):
        if it < 40:
            return False
        nuts = sv.get("nuts"); base.setdefault("eps", nuts[0]); nuts[0] = base["eps"] / 16; sv.set("nuts", nuts)
        return True
    rows, _ = teacher_forced(oracle_lib, emul_lib, "emu_", args, patch=patch)
    assert rows[3, 40:].min() >= 6 and rows[3].max() <= 10, rows[3]
    assert np.all(rows[4, 40:] >= 127)


def test_forced_divergence_and_nonfinite_energy(oracle_lib, emul_lib):
    """a huge forced step size: the first leapfrog leaves the typical set (divergent__ = 1, base_nuts.hpp:262) or makes the
    energy non-finite (V = +inf, base_hamiltonian.hpp:61-70); both implementations must agree on the whole row."""
    args, _ = friedman_case(n=100, T=5, warmup=3, iter=8, slopes=True)
    sizes = {1: 30.0, 2: 1e3, 4: 1e6, 6: 2.0}

    def patch(it, sv):
        if it not in sizes:
            return False
        nuts = sv.get("nuts"); nuts[0] = sizes[it]; sv.set("nuts", nuts)
        return True
    rows, _ = teacher_forced(oracle_lib, emul_lib, "emu_", args, patch=patch)
    assert rows[5, 1] == 1 and rows[5, 2] == 1 and rows[5, 4] == 1, rows[5]
    assert rows[5, 0] == 0


def test_forced_probit(oracle_lib, emul_lib):
    from conftest import binary_case
    args = binary_case(n=160, T=7, warmup=30, iter=36)
    args.adapt_init_buffer, args.adapt_term_buffer, args.adapt_window = 5, 5, 10
    rows, ends = teacher_forced(oracle_lib, emul_lib, "emu_", args)
    assert len(ends) >= 1
