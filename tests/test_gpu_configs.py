"""BASELINE.json's configurations at their stated shape and size on the HIP path (VERDICT r01 `configs_untested`):
  c3  Friedman n = 1e6, p = 50, ntree = 200, (1 + X4 | g.1) + (1 | g.2): oracle parity at full size, both gradient modes
  c4  IHDP 747 x 25 mixed covariates, one 26-level group, probit: oracle parity, free-running and teacher-forced
  c5  n = 1e7, P = 100, ntree = 400, 200 groups with random slopes: size-independent properties at full size,
      oracle parity at n = 1e6 for the same shape
"""
import numpy as np
import pytest

from conftest import StateView, assert_chain_parity, assert_state_parity, c5_case, friedman_case, ihdp_case, make_sampler, run_chain, teacher_forced

pytestmark = pytest.mark.gpu


def test_config3_full_size_both_gradient_modes(oracle_lib, hip_lib):
    """n = 1e6, p = 50, 200 trees, random slopes: 3 Gibbs iterations (600 tree updates, 3 NUTS transitions) against the oracle.
    hmc_mode 0 evaluates the likelihood from s0 - 2 theta'c + theta'G theta, whose cancellation grows with N: compared here at
    the full N against the oracle (which sums over the observations) and against hmc_mode 1 (per-leapfrog O(N) kernels)."""
    args, _ = friedman_case(n=1_000_000, p=50, T=200, warmup=2, iter=3, slopes=True)
    a = run_chain(oracle_lib, "orc_", args)
    b0 = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b0)
    args.hmc_mode = 1
    b1 = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b1)
    np.testing.assert_allclose(b0["sample"]["stan"], b1["sample"]["stan"], rtol=1e-6, atol=1e-9)
    assert np.array_equal(b0["trace"], b1["trace"])


def test_config3_stationary_teacher_forced(oracle_lib, hip_lib):
    """The regime the benchmark's headline is measured in, under the oracle: BASELINE config 3 at full size (n = 1e6, p = 50,
    200 trees, random slopes — 255 pass workgroups + the control workgroup, both exchange rings), the HIP chain burned in for 300 warm-up iterations
    (trees of 4 - 5 leaves, ~98 % of the moves rejected, hundreds of leapfrogs per transition), adaptation disengaged.  Then teacher forcing, as
    conftest.teacher_forced does it: the state is injected into the oracle, both advance ONE sampling iteration, everything is compared — tree-move
    trace, generator state, NUTS depth / leapfrogs / divergence bit-exact, floats to 1e-6, the whole state after the iteration —, the oracle's state
    goes back into the HIP sampler, and again; each iteration once per gradient mode on the HIP side.  And the sweeps ran where the headline runs:
    one persistent launch per sweep, none handed over, at least 80 % of the tree updates publishing their statistics before the verdict."""
    burn = 300
    args, _ = friedman_case(n=1_000_000, p=50, T=200, warmup=burn, iter=burn + 4, slopes=True, keep_fits=False)
    sp = make_sampler(hip_lib, "s4b_", args)
    so = make_sampler(oracle_lib, "orc_", args)
    try:
        sp.run(burn, True, 0)
        sp.disengage_adaptation()
        so.disengage_adaptation()
        assert sp.get_tree_path()[1] == "persistent"
        spec0, stats0 = sp.get_sweep_spec(), sp.get_sweep_stats()
        assert stats0 == (burn + 1, 0), stats0
        st = sp.get_state()
        so.set_trace(True); sp.set_trace(True)
        launches = 0
        for it in range(2):
            so.set_state(st)
            ro = so.run(1, False, 0)
            to = so.get_trace()
            after = StateView(so.get_state())
            for mode in (0, 1):
                sp.set_state(st)
                sp.set_hmc_mode(mode)
                rp = sp.run(1, False, 0)
                launches += 1
                ctx = f"stationary iteration {it}, hmc_mode {mode}"
                assert np.array_equal(to, sp.get_trace()), ctx + ": tree-move trace differs"
                assert np.array_equal(ro["stan"][3:6], rp["stan"][3:6]), ctx + f": NUTS depth / n_leapfrog / divergent differ: {ro['stan'][3:6].ravel()} vs {rp['stan'][3:6].ravel()}"
                np.testing.assert_allclose(ro["stan"], rp["stan"], rtol=1e-6, atol=1e-9, err_msg=ctx)
                np.testing.assert_allclose(ro["bart"]["train"], rp["bart"]["train"], rtol=1e-6, atol=1e-9, err_msg=ctx)
                assert np.array_equal(ro["bart"]["varcount"], rp["bart"]["varcount"]), ctx
                assert_state_parity(after, StateView(sp.get_state()))
            assert ro["stan"][4, 0] >= 63, f"not the stationary regime: {ro['stan'][4, 0]} leapfrogs in the transition"
            st = so.get_state()
        spec1, stats1 = sp.get_sweep_spec(), sp.get_sweep_stats()
        assert stats1 == (burn + 1 + launches, 0), stats1
        ran, inside, early, ok = (b - a for a, b in zip(spec0, spec1))
        assert ran == launches and inside == launches * 200
        assert early >= 0.8 * inside and ok >= 0.9 * early, (ran, inside, early, ok)
    finally:
        so.free(); sp.free()


def test_config4_ihdp_shape_probit(oracle_lib, hip_lib):
    args = ihdp_case()
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)
    assert a["sample"]["bart"]["test"].shape == (747, 6) and "aux.1" not in b["names"]
    assert sum(nm.startswith("b.") for nm in b["names"]) == 52          # 26 levels x (intercept, slope on z)
    used = np.flatnonzero(a["sample"]["bart"]["varcount"].sum(axis=1))
    assert (used >= 6).any()                                             # rules on binary covariates were accepted


def test_config4_ihdp_teacher_forced(oracle_lib, hip_lib):
    args = ihdp_case(warmup=40, iter=50, T=50)
    args.adapt_init_buffer, args.adapt_term_buffer, args.adapt_window = 10, 10, 10
    rows, ends = teacher_forced(oracle_lib, hip_lib, "s4b_", args)
    assert ends == [19, 29], ends


def test_config5_shape_at_1e6_matches_oracle(oracle_lib, hip_lib):
    """P = 100, 400 trees, 200 groups with slopes (q = 400) at n = 1e6: 2 Gibbs iterations against the oracle."""
    args, _ = c5_case(1_000_000, warmup=1, iter=2)
    a = run_chain(oracle_lib, "orc_", args)
    b = run_chain(hip_lib, "s4b_", args)
    assert_chain_parity(a, b)
    assert sum(nm.startswith("b.") for nm in b["names"]) == 400


def test_config5_full_size_properties(hip_lib):
    """n = 1e7, P = 100, 400 trees, q = 400 (BASELINE config 5 for one chain): properties that need no oracle — every tree's
    leaf counts sum to n; test rows equal to training rows get identical fits; the fit returned by run() equals the fit
    re-assembled from the flattened trees; a second chain from the same seed is bitwise identical."""
    n, n_test, T = 10_000_000, 64, 400
    args, xb = c5_case(n, warmup=2, iter=4, n_test=n_test, keep_fits=False)
    xt = xb[:n_test].copy()
    del xb
    s = make_sampler(hip_lib, "s4b_", args)
    s.run(2, True)
    s.disengage_adaptation()
    r = s.run(2, False)
    tr, rng_state, rg = s.get_trees(), s.get_r_rng_state(), s.get_bart_data_range()
    s.free()
    roots = np.r_[True, tr["tree"][1:] != tr["tree"][:-1]]
    assert roots.sum() == T and np.all(tr["n"][roots] == n)
    leaf = tr["var"] < 0
    assert np.array_equal(np.bincount(tr["tree"][leaf], weights=tr["n"][leaf], minlength=T), np.full(T, float(n)))
    train = r["bart"]["train"][:, -1]
    np.testing.assert_allclose(r["bart"]["test"][:, -1], train[:n_test], rtol=1e-9)
    lo, hi = rg
    fit = np.zeros(n_test)
    starts = np.flatnonzero(roots)
    for t in range(T):
        var, val = tr["var"][starts[t]:], tr["value"][starts[t]:]
        for i in range(n_test):
            k = 0
            while var[k] >= 0:
                if xt[i, var[k]] <= val[k]:
                    k += 1
                else:
                    depth, k = 1, k + 1
                    while depth > 0:
                        depth += 1 if var[k] >= 0 else -1
                        k += 1
            fit[i] += val[k]
    np.testing.assert_allclose((fit + 0.5) * (hi - lo) + lo, train[:n_test], rtol=1e-8, atol=1e-8)
    s2 = make_sampler(hip_lib, "s4b_", args)
    s2.run(2, True)
    s2.disengage_adaptation()
    r2 = s2.run(2, False)
    assert np.array_equal(rng_state, s2.get_r_rng_state())
    assert np.array_equal(r2["bart"]["train"], r["bart"]["train"]) and np.array_equal(r2["stan"], r["stan"])
    # the two gradient modes at the largest N: hmc_mode 0 evaluates s0 - 2 theta'c + theta'G theta from sums gathered once per
    # iteration (cancellation grows with N), hmc_mode 1 sums the N residuals at every leapfrog.  Same state, one iteration each.
    st = s2.get_state()
    s2.set_trace(True)
    a0 = s2.run(1, False)
    t0 = s2.get_trace()
    s2.set_state(st)
    s2.set_hmc_mode(1)
    a1 = s2.run(1, False)
    t1 = s2.get_trace()
    s2.free()
    assert np.array_equal(a0["stan"][3:6], a1["stan"][3:6]) and np.array_equal(t0, t1)
    np.testing.assert_allclose(a0["stan"], a1["stan"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(a0["bart"]["train"], a1["bart"]["train"], rtol=1e-6, atol=1e-9)
