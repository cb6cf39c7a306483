"""The user surface around the hot path (stan4bart_amd/generics.py: stan4bart(), extract, fitted, predict), following the
reference's own tests: predict == extract for duplicated rows (tests/testthat/test-01-continuous.R:204-246,
test-02-binary.R:81-123), component identities, shapes and argument errors (test-01:21-149)."""
import numpy as np
import pytest

from stan4bart_amd import GroupTerm, generate_friedman_data
from stan4bart_amd.abi import Sampler
from stan4bart_amd.generics import combine_chains_f, stan4bart


def _data(n=120, binary=False, seed_rows=17):
    d = generate_friedman_data(n, ranef=True, causal=True, binary=binary, p=10)
    x = d["x"]
    xb = x[:, [j for j in range(10) if j != 3]]
    X = np.column_stack([x[:, 3], d["z"]])
    groups = [GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")]
    rows = np.arange(seed_rows)
    groups_t = [GroupTerm(np.asarray(d["g1"])[rows], x[rows, 3], "g.1"), GroupTerm(np.asarray(d["g2"])[rows], None, "g.2")]
    return d, xb, X, groups, rows, groups_t


def _fit(lib, prefix, binary=False, chains=2, keep_trees=True):
    d, xb, X, groups, rows, groups_t = _data(binary=binary)
    fit = stan4bart(d["y"], xb, X=X, groups=groups, x_bart_test=xb[rows], X_test=X[rows], groups_test=groups_t,
                    family="binomial" if binary else "gaussian", chains=chains, seed=99, iter=14, warmup=6,
                    bart_args={"n.trees": 9, "keepTrees": keep_trees},
                    make_sampler=lambda a, st: Sampler(lib, prefix, a, st))
    return fit, (d, xb, X, groups, rows, groups_t)


def _check_surface(fit, data):
    d, xb, X, groups, rows, groups_t = data
    n, S, C = len(d["y"]), 8, 2
    ev = fit.extract("ev", combine_chains=False)
    assert ev.shape == (n, S, C)
    assert fit.extract("ev").shape == (n, S * C)
    np.testing.assert_array_equal(fit.extract("ev")[:, :S], ev[:, :, 0])          # chain 1's draws first
    assert fit.extract("ev", include_warmup=True, combine_chains=False).shape == (n, 6 + S, C)
    assert fit.extract("ev", include_warmup="only", combine_chains=False).shape == (n, 6, C)
    # components add up; fitted = posterior mean
    parts = sum(fit.extract(t, combine_chains=False) for t in ("indiv.bart", "indiv.fixef", "indiv.ranef"))
    if fit.family == "binomial":
        from stan4bart_amd.generics import _pnorm
        parts = _pnorm(parts)
    np.testing.assert_allclose(ev, parts, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fit.fitted("ev"), fit.extract("ev").mean(axis=1), rtol=1e-12)
    # test sample == the same rows of the training sample (duplicated rows)
    for t in ("ev", "indiv.bart", "indiv.fixef", "indiv.ranef"):
        np.testing.assert_allclose(fit.extract(t, sample="test", combine_chains=False), fit.extract(t, combine_chains=False)[rows],
                                   rtol=1e-9, atol=1e-9)
    # predict from the kept trees == extract (reference test-01:204-246)
    for t in ("ev", "indiv.bart", "indiv.fixef", "indiv.ranef"):
        p = fit.predict(x_bart=xb[rows], X=X[rows], groups=groups_t, type=t, combine_chains=False)
        np.testing.assert_allclose(p, fit.extract(t, combine_chains=False)[rows], rtol=1e-9, atol=1e-9)
    # parametric pieces
    assert fit.extract("fixef").shape == (2, S * C)
    re = fit.extract("ranef", combine_chains=False)
    assert re["g.1"].shape == (2, 5, S, C) and re["g.2"].shape == (1, 8, S, C)
    Sg = fit.extract("Sigma", combine_chains=False)
    assert Sg["g.1"].shape == (2, 2, S, C) and np.all(np.linalg.eigvalsh(Sg["g.1"][:, :, 0, 0]) >= -1e-12)
    assert fit.extract("varcount").shape == (9, S * C) and np.all(fit.extract("varcount").sum(axis=0) >= 0)
    if fit.family == "gaussian":
        assert fit.extract("sigma").shape == (S * C,) and np.all(fit.extract("sigma") > 0)
        ppd = fit.extract("ppd", seed=1)
        assert ppd.shape == (n, S * C) and np.std(ppd - fit.extract("ev")) > 0
    else:
        with pytest.raises(ValueError, match="sigma"):
            fit.extract("sigma")
        assert set(np.unique(fit.extract("ppd", seed=1))) <= {0.0, 1.0}
        assert np.all((ev >= 0) & (ev <= 1))
    # unseen grouping levels: zero effect, or a draw from Sigma
    g_new = [GroupTerm(np.full(3, 6), X[:3, 0], "g.1"), GroupTerm(np.asarray(d["g2"])[:3], None, "g.2")]
    r0 = fit.predict(x_bart=xb[:3], X=X[:3], groups=g_new, type="indiv.ranef", sample_new_levels=False, combine_chains=False)
    g2_only = fit.extract("ranef", combine_chains=False)["g.2"][0][np.asarray(d["g2"])[:3] - 1]
    np.testing.assert_allclose(r0, g2_only, rtol=1e-12)
    r1 = fit.predict(x_bart=xb[:3], X=X[:3], groups=g_new, type="indiv.ranef", sample_new_levels=True, seed=3, combine_chains=False)
    assert np.std(r1 - r0) > 0
    _check_trees(fit, xb, rows, n)
    with pytest.raises(ValueError):
        fit.extract("nonsense")
    fit.close()


def _check_trees(fit, xb, rows, n):
    """reference tests/testthat/test-07-extractedTrees.R (all trees == the draws extracted one by one) plus: walking the
    extracted trees reproduces predict's BART component."""
    S, C = fit.bart_train.shape[1:]
    T = 9
    allt = fit.extract("trees")
    assert set(allt) == {"chain", "sample", "tree", "n", "var", "split", "value"}
    combos = set(zip(allt["chain"].tolist(), allt["sample"].tolist(), allt["tree"].tolist()))
    assert combos == {(c, k, t) for c in range(C) for k in range(S) for t in range(T)}
    one_by_one = [fit.extract("trees", sampleNums=k, chainNums=c) for c in range(C) for k in range(S)]
    for key in allt:
        np.testing.assert_array_equal(allt[key], np.concatenate([o[key] for o in one_by_one]))
    sub = fit.extract("trees", treeNums=[2, 5])
    assert set(np.unique(sub["tree"])) == {2, 5}
    # root of every tree holds all observations; leaves of a tree partition them
    first = np.r_[True, (np.diff(allt["tree"]) != 0) | (np.diff(allt["sample"]) != 0) | (np.diff(allt["chain"]) != 0)]
    assert np.all(allt["n"][first] == n)
    # prediction by walking the flattened trees (preorder: left subtree follows its parent)
    x = xb[rows]
    pred = np.zeros((len(rows), S, C))
    pos = 0
    var, val, ns = allt["var"], allt["value"], len(allt["var"])

    def walk(i, xi):
        while var[i] >= 0:
            if xi[var[i]] <= val[i]:
                i += 1
            else:   # skip the left subtree
                j, open_ = i + 1, 1
                while open_ > 0:
                    open_ += 1 if var[j] >= 0 else -1
                    j += 1
                i = j
        return val[i], i
    starts = np.flatnonzero(first)
    for s_i, st in enumerate(starts):
        c, k = allt["chain"][st], allt["sample"][st]
        for r in range(len(rows)):
            pred[r, k, c] += walk(st, x[r])[0]
    bart = fit.predict(x_bart=x, type="indiv.bart", combine_chains=False)
    if fit.family == "gaussian":
        lo, hi = fit.range_bart[0], fit.range_bart[1]
        # the range of the last draw is only exact for that draw (the scale is updated during warmup only, so it is constant here)
        pred = (pred + 0.5) * (hi - lo)[None, None, :] + lo[None, None, :]
    np.testing.assert_allclose(pred, bart, rtol=1e-9, atol=1e-9)


def test_combine_chains():
    x = np.arange(24).reshape(2, 3, 4)
    c = combine_chains_f(x)
    assert c.shape == (2, 12) and list(c[0, :3]) == [0, 4, 8] and list(c[0, 3:6]) == [1, 5, 9]


@pytest.mark.parametrize("binary", [False, True])
def test_generics_host_logic(emul_lib, binary):
    fit, data = _fit(emul_lib, "emu_", binary=binary)
    _check_surface(fit, data)


def _check_stored(fit, data, lib, prefix):
    """exportBARTState -> createStoredBARTSampler (reference src/init.cpp:409-446): predictions from the rebuilt samplers
    are identical to those of the fitting samplers, which may be gone by then."""
    d, xb, X, groups, rows, groups_t = data
    before = {t: fit.predict(x_bart=xb[:23], X=X[:23], groups=[type(g)(np.asarray(g.levels)[:23], None if g.slopes is None else
                             np.asarray(g.slopes)[:23], g.name) for g in groups], type=t, combine_chains=False) for t in ("ev", "indiv.bart")}
    states = fit.export_bart_states()
    assert all(isinstance(b, bytes) and len(b) > 64 for b in states)
    fit.attach_stored_samplers(states, lib=lib, prefix=prefix)          # closes the live samplers
    for t, ref in before.items():
        got = fit.predict(x_bart=xb[:23], X=X[:23], groups=[type(g)(np.asarray(g.levels)[:23], None if g.slopes is None else
                          np.asarray(g.slopes)[:23], g.name) for g in groups], type=t, combine_chains=False)
        np.testing.assert_array_equal(got, ref)
    assert fit.export_bart_states() == states                           # a stored sampler re-exports the same bytes
    assert len(fit.extract("trees")["tree"]) > 0                        # kept trees travel with the state
    from stan4bart_amd.abi import StoredSampler
    with pytest.raises(RuntimeError, match="exported|truncated"):
        StoredSampler(lib, prefix, states[0][: len(states[0]) // 2])
    with pytest.raises(RuntimeError, match="exported"):
        StoredSampler(lib, prefix, b"\x00" * 64)
    # crafted / corrupted states must be rejected at load time (they would make the prediction kernel read out of bounds or loop):
    # the node records are the tail of the product's byte string, 24 bytes each: {int16 var, uint16 cut, int16 left, int16 right,
    # double mu, int32 n, pad} (the oracle has its own layout)
    if prefix != "orc_":
        import struct
        raw = bytearray(states[0])
        P, T = struct.unpack_from("<II", raw, 8)
        S, num_nodes = struct.unpack_from("<QQ", raw, 20)
        node0 = len(raw) - 24 * num_nodes
        k = next(j for j in range(num_nodes) if struct.unpack_from("<h", raw, node0 + 24 * j)[0] >= 0)    # an internal node
        left = struct.unpack_from("<h", raw, node0 + 24 * k + 4)[0]
        for offset, fmt, value in ((4, "<h", 30000), (6, "<h", left), (0, "<h", P + 3), (2, "<H", 65000)):
            bad = bytearray(raw)                # child link out of range / both links to one child / predictor / cut out of range
            struct.pack_into(fmt, bad, node0 + 24 * k + offset, value)
            with pytest.raises(RuntimeError, match="exported BART state"):
                StoredSampler(lib, prefix, bytes(bad))
    fit.close()


@pytest.mark.parametrize("binary", [False, True])
def test_stored_sampler_host_logic(emul_lib, binary):
    fit, data = _fit(emul_lib, "emu_", binary=binary)
    _check_stored(fit, data, emul_lib, "emu_")


def test_stored_sampler_oracle(oracle_lib):
    fit, data = _fit(oracle_lib, "orc_")
    _check_stored(fit, data, oracle_lib, "orc_")


@pytest.mark.gpu
def test_stored_sampler_on_hip(hip_lib):
    fit, data = _fit(hip_lib, "s4b_")
    _check_stored(fit, data, hip_lib, "s4b_")


def test_predict_needs_kept_trees(emul_lib):
    fit, data = _fit(emul_lib, "emu_", keep_trees=False, chains=1)
    with pytest.raises(ValueError, match="keepTrees"):
        fit.predict(x_bart=data[1][:3], X=data[2][:3], groups=data[5], type="ev")
    with pytest.raises(ValueError, match="keepTrees"):
        fit.extract("trees")


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [False, True])
def test_generics_on_hip(hip_lib, binary):
    fit, data = _fit(hip_lib, "s4b_", binary=binary)
    _check_surface(fit, data)


def test_callback_and_treatment(emul_lib):
    """reference tests/testthat/test-11-callback.R (the callback sees each draw: its pieces equal the extracted test
    components) and test-10-treatment.R (treatment = ... builds the counterfactual test sample)."""
    d, xb, X, groups, rows, groups_t = _data()
    Xm = X.mean(axis=0)
    lev1, lev2 = np.asarray(d["g1"]), np.asarray(d["g2"])

    def cb(yhat_train, yhat_test, stan_pars, names):
        beta = np.array([stan_pars[names.index(f"beta.{k + 1}")] for k in range(2)])
        b = np.array([v for v, nm in zip(stan_pars, names) if nm.startswith("b.")])
        fix = (X_cf - Xm) @ beta
        # Z column order: g.2 (8 levels, intercept) then g.1 (5 levels, intercept + slope)
        ran = b[lev2 - 1] + b[8 + (lev1 - 1) * 2] + b[8 + (lev1 - 1) * 2 + 1] * X_cf[:, 0]
        return np.concatenate([yhat_test, fix, ran])

    X_cf = X.copy()
    X_cf[:, 1] = 1.0 - X[:, 1]
    fit = stan4bart(d["y"], xb, X=X, groups=groups, treatment=("X", 1), callback=cb, chains=2, seed=5, iter=13, warmup=7,
                    bart_args={"n.trees": 7}, make_sampler=lambda a, st: Sampler(emul_lib, "emu_", a, st))
    n = len(d["y"])
    cbk = fit.extract("callback", combine_chains=False)
    assert cbk.shape == (3 * n, 6, 2)
    np.testing.assert_allclose(cbk[:n], fit.extract("indiv.bart", sample="test", combine_chains=False), rtol=1e-12)
    np.testing.assert_allclose(cbk[n:2 * n], fit.extract("indiv.fixef", sample="test", combine_chains=False), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cbk[2 * n:], fit.extract("indiv.ranef", sample="test", combine_chains=False), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cbk[:n] + cbk[n:2 * n] + cbk[2 * n:], fit.extract("ev", sample="test", combine_chains=False), rtol=1e-9)
    # treatment effect samples: observed minus counterfactual differ exactly by the flipped column's coefficient
    diff = fit.extract("ev", combine_chains=False) - fit.extract("ev", sample="test", combine_chains=False)
    beta_z = fit.extract("fixef", combine_chains=False)[1]
    np.testing.assert_allclose(diff, (X[:, 1] - X_cf[:, 1])[:, None, None] * beta_z[None], rtol=1e-8, atol=1e-8)
    assert fit.fitted(sample="test").shape == (n,)
    assert fit.extract("callback", include_warmup=True, combine_chains=False).shape == (3 * n, 13, 2)


def test_kept_trees_match_oracle(oracle_lib, emul_lib):
    a, _ = _fit(oracle_lib, "orc_", chains=1)
    b, _ = _fit(emul_lib, "emu_", chains=1)
    ta, tb = a.extract("trees"), b.extract("trees")
    for k in ("chain", "sample", "tree", "n", "var", "split"):
        np.testing.assert_array_equal(ta[k], tb[k])
    np.testing.assert_allclose(ta["value"], tb["value"], rtol=1e-6, atol=1e-9)
    a.close(); b.close()


@pytest.mark.gpu
def test_kept_trees_match_oracle_on_hip(oracle_lib, hip_lib):
    a, _ = _fit(oracle_lib, "orc_", chains=1)
    b, _ = _fit(hip_lib, "s4b_", chains=1)
    ta, tb = a.extract("trees"), b.extract("trees")
    for k in ("chain", "sample", "tree", "n", "var", "split"):
        np.testing.assert_array_equal(ta[k], tb[k])
    np.testing.assert_allclose(ta["value"], tb["value"], rtol=1e-6, atol=1e-9)
    a.close(); b.close()


def test_concurrent_chains_match_parallel_seeding_rule(emul_lib):
    """cores > 1: chains run concurrently in threads, each seeded like the reference's parallel workers
    (R/stan4bart_fit.R:515-533); every chain equals the same chain fitted alone with that seed."""
    from stan4bart_amd import RRng, fit_worker, make_sampler_args
    from stan4bart_amd.fit import chain_seeds
    d, xb, X, groups, rows, groups_t = _data()
    mk = lambda a, st: Sampler(emul_lib, "emu_", a, st)
    fit = stan4bart(d["y"], xb, X=X, groups=groups, chains=3, cores=3, seed=42, iter=12, warmup=6, bart_args={"n.trees": 5}, make_sampler=mk)
    assert fit.bart_train.shape == (len(d["y"]), 6, 3)
    for c in range(3):
        args = make_sampler_args(d["y"], xb, X=X, groups=groups, iter=12, warmup=6, bart_args={"n.trees": 5})
        alone = fit_worker(mk, args, RRng(int(chain_seeds(42, 3)[c])))
        np.testing.assert_array_equal(alone["sample"]["stan"], fit.stan[:, :, c])
        np.testing.assert_array_equal(alone["sample"]["bart"]["train"], fit.bart_train[:, :, c])


@pytest.mark.gpu
def test_concurrent_chains_on_hip(hip_lib):
    """three samplers driven from three host threads on one device (own stream and graph each), then used from the main
    thread: equal to the same chains fitted one after the other."""
    from stan4bart_amd import RRng, fit_worker, make_sampler_args
    from stan4bart_amd.fit import chain_seeds
    d, xb, X, groups, rows, groups_t = _data(n=3000)
    mk = lambda a, st: Sampler(hip_lib, "s4b_", a, st)
    fit = stan4bart(d["y"], xb, X=X, groups=groups, chains=3, cores=3, seed=42, iter=12, warmup=6,
                    bart_args={"n.trees": 25, "keepTrees": True}, make_sampler=mk)
    def mk_shared(a, st):   # what stan4bart(cores = 3) tells every sampler: three chains share the device
        s = Sampler(hip_lib, "s4b_", a, st)
        s.set_device_sharing(3)
        return s
    for c in range(3):
        args = make_sampler_args(d["y"], xb, X=X, groups=groups, iter=12, warmup=6, bart_args={"n.trees": 25, "keepTrees": True})
        alone = fit_worker(mk_shared, args, RRng(int(chain_seeds(42, 3)[c])))
        np.testing.assert_array_equal(alone["sample"]["stan"], fit.stan[:, :, c])
        np.testing.assert_array_equal(alone["sample"]["bart"]["train"], fit.bart_train[:, :, c])
        if c == 0:   # the same chain with the device to itself (fused tree update): another summation order, the same chain
            solo = fit_worker(mk, args, RRng(int(chain_seeds(42, 3)[c])))
            np.testing.assert_allclose(solo["sample"]["stan"], fit.stan[:, :, c], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(solo["sample"]["bart"]["train"], fit.bart_train[:, :, c], rtol=1e-9, atol=1e-12)
            np.testing.assert_array_equal(solo["sample"]["bart"]["varcount"], fit.bart_varcount[:, :, c])
    p = fit.predict(x_bart=xb[:9], X=X[:9], groups=[type(g)(np.asarray(g.levels)[:9], None if g.slopes is None else np.asarray(g.slopes)[:9], g.name)
                                                    for g in groups], type="ev", combine_chains=False)
    np.testing.assert_allclose(p, fit.extract("ev", combine_chains=False)[:9], rtol=1e-9, atol=1e-9)
    fit.close()


def test_extract_k_of_a_modeled_end_node_sensitivity(emul_lib):
    """bart_args = list(k = chi(1.25, Inf)): the fit carries the k draws (reference R/stan4bart.R:389-399) and extract(fit, "k") returns
    them, chain 1 first; without a hyperprior extract("k") is the reference's error (R/generics.R:223-224)."""
    d, xb, X, groups, rows, groups_t = _data()
    kw = dict(X=X, groups=groups, chains=2, seed=4, iter=12, warmup=5, make_sampler=lambda a, st: Sampler(emul_lib, "emu_", a, st))
    fit = stan4bart(d["y"], xb, bart_args={"n.trees": 6, "k": "chi(1.25, Inf)"}, **kw)
    k = fit.extract("k", combine_chains=False)
    assert k.shape == (7, 2) and np.all(k > 0) and np.std(k) > 0
    np.testing.assert_array_equal(fit.extract("k"), np.concatenate([k[:, 0], k[:, 1]]))
    assert fit.extract("k", include_warmup=True, combine_chains=False).shape == (12, 2) and fit.extract("k", include_warmup="only", combine_chains=False).shape == (5, 2)
    plain = stan4bart(d["y"], xb, bart_args={"n.trees": 6}, **kw)
    with pytest.raises(ValueError, match="end-node sensitivity"):
        plain.extract("k")
