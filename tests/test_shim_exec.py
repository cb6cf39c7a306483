"""The R `.Call` shim (shim/init_shim.cpp) EXECUTED: linked against an in-memory mock of the R API (tests/r_api_mock, test
infrastructure) and driven through the twelve registered routines with the argument lists the reference's R code passes
(R/stan4bart_fit.R:42-57: create, run(warmup), disengageAdaptation, run(sampling); :572-580 exportBARTState /
createStoredBARTSampler; R/generics.R:190 getTrees, :667 predictBART), then compared bit for bit with the ctypes path on the same
library.  CPU: the shim over the emulation of the device layer; `-m gpu`: the shim over libs4b.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, friedman_case

MOCK_DIR = os.path.join(ROOT, "tests", "r_api_mock")
NILSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, RAWSXP, EXTPTRSXP = 0, 10, 13, 14, 16, 19, 24, 22
CLOSURE = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


class MockR:
    """ctypes handle on a shim + mock library: builds SEXPs, calls routines by name, reads results back."""

    def __init__(self, path):
        self.lib = L = C.CDLL(path)
        vp = C.c_void_p
        for name, res, args in [
            ("mock_nil", vp, []), ("mock_real", vp, [C.POINTER(C.c_double), C.c_int64]), ("mock_int", vp, [C.POINTER(C.c_int), C.c_int64]),
            ("mock_lgl", vp, [C.c_int]), ("mock_str", vp, [C.c_char_p]), ("mock_raw", vp, [C.c_char_p, C.c_int64]), ("mock_list", vp, [C.c_int64]),
            ("mock_list_set", None, [vp, C.c_int64, C.c_char_p, vp]), ("mock_set_dim", None, [vp, C.c_int, C.c_int]), ("mock_set_attr", None, [vp, C.c_char_p, vp]),
            ("mock_s4", vp, []), ("mock_set_slot", None, [vp, C.c_char_p, vp]), ("mock_env", vp, []), ("mock_closure", vp, [CLOSURE, vp]),
            ("mock_set_seed", None, [C.POINTER(C.c_int)]), ("mock_get_seed", C.c_int, [C.POINTER(C.c_int)]),
            ("mock_type", C.c_int, [vp]), ("mock_length", C.c_int64, [vp]), ("mock_real_ptr", C.POINTER(C.c_double), [vp]),
            ("mock_int_ptr", C.POINTER(C.c_int), [vp]), ("mock_raw_ptr", C.POINTER(C.c_ubyte), [vp]), ("mock_elt", vp, [vp, C.c_int64]),
            ("mock_chars", C.c_char_p, [vp]), ("mock_get_attr", vp, [vp, C.c_char_p]), ("mock_finalize", None, [vp]),
            ("mock_call", C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(vp)]), ("mock_last_error", C.c_char_p, []),
            ("mock_printed", C.c_char_p, []), ("mock_clear_printed", None, []), ("mock_num_warnings", C.c_int, []), ("mock_protect_depth", C.c_int, []),
            ("mock_protect_underflows", C.c_int, []), ("mock_set_interrupt_pending", None, [C.c_int]), ("mock_init", None, []),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.mock_init()
        self._keep = []

    # ---- building
    def nil(self):
        return self.lib.mock_nil()

    def real(self, x, dim=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel(order="F"))
        s = self.lib.mock_real(a.ctypes.data_as(C.POINTER(C.c_double)), a.size)
        if dim is not None:
            self.lib.mock_set_dim(s, int(dim[0]), int(dim[1]))
        return s

    def integer(self, x):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.int32).ravel())
        return self.lib.mock_int(a.ctypes.data_as(C.POINTER(C.c_int)), a.size)

    def lgl(self, v):
        return self.lib.mock_lgl(int(bool(v)))

    def string(self, s):
        return self.lib.mock_str(s.encode())

    def raw(self, b):
        return self.lib.mock_raw(bytes(b), len(b))

    def named_list(self, items):
        lst = self.lib.mock_list(len(items))
        for i, (k, v) in enumerate(items):
            self.lib.mock_list_set(lst, i, k.encode() if k is not None else None, v)
        return lst

    def s4(self, slots):
        o = self.lib.mock_s4()
        for k, v in slots.items():
            self.lib.mock_set_slot(o, k.encode(), v)
        return o

    def closure(self, pyfn):
        def tramp(a, b, c, user):
            out = pyfn(self.to_numpy(a), None if self.type(b) == NILSXP else self.to_numpy(b), self.to_numpy(c), self.names(c))
            return self.nil() if out is None else self.real(out)
        cb = CLOSURE(tramp)
        self._keep.append(cb)
        return self.lib.mock_closure(cb, None)

    def set_seed(self, state625):
        seed = np.concatenate([[10403], np.asarray(state625, dtype=np.uint32).astype(np.int64)]).astype(np.uint32).view(np.int32)
        self.lib.mock_set_seed(np.ascontiguousarray(seed).ctypes.data_as(C.POINTER(C.c_int)))

    def get_seed(self):
        out = np.zeros(626, dtype=np.int32)
        assert self.lib.mock_get_seed(out.ctypes.data_as(C.POINTER(C.c_int))) == 0
        return out[1:].view(np.uint32).copy()

    # ---- calling / reading
    def call(self, name, *args, expect=0):
        arr = (C.c_void_p * max(1, len(args)))(*args)
        out = C.c_void_p()
        rc = self.lib.mock_call(name.encode(), len(args), arr, C.byref(out))
        assert self.lib.mock_protect_depth() == 0, f"{name}: PROTECT stack not balanced ({self.lib.mock_protect_depth()})"
        assert self.lib.mock_protect_underflows() == 0, f"{name}: UNPROTECT below zero"
        if expect is not None:
            assert rc == expect, (name, rc, self.lib.mock_last_error().decode())
        return out.value if rc == 0 else None

    def error(self):
        return self.lib.mock_last_error().decode()

    def type(self, s):
        return self.lib.mock_type(s)

    def to_numpy(self, s):
        t, n = self.type(s), self.lib.mock_length(s)
        if t == REALSXP:
            a = np.ctypeslib.as_array(self.lib.mock_real_ptr(s), shape=(n,)).copy() if n else np.zeros(0)
        elif t in (INTSXP, LGLSXP):
            a = np.ctypeslib.as_array(self.lib.mock_int_ptr(s), shape=(n,)).copy() if n else np.zeros(0, dtype=np.int32)
        elif t == RAWSXP:
            a = np.ctypeslib.as_array(self.lib.mock_raw_ptr(s), shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint8)
        else:
            raise TypeError(t)
        d = self.lib.mock_get_attr(s, b"dim")
        if self.type(d) != NILSXP:
            dims = np.ctypeslib.as_array(self.lib.mock_int_ptr(d), shape=(self.lib.mock_length(d),))
            a = a.reshape(tuple(int(v) for v in dims), order="F")
        return a

    def strings(self, s):
        return [self.lib.mock_chars(self.lib.mock_elt(s, i)).decode() for i in range(self.lib.mock_length(s))]

    def names(self, s):
        nm = self.lib.mock_get_attr(s, b"names")
        return None if self.type(nm) == NILSXP else self.strings(nm)

    def as_dict(self, lst):
        return {k: self.lib.mock_elt(lst, i) for i, k in enumerate(self.names(lst))}


def r_arguments(R, a, callback=None, drop=None):
    """The six arguments of .Call(C_stan4bart_create, ...) as the reference's R code builds them (R/stan4bart_fit.R:259-365, 436-479)
    from one SamplerArgs: dbarts control / data / model S4 objects, the 44-name stanData list, stanControl, commonControl."""
    xb = np.asarray(a.x_bart, dtype=np.float64)
    n, p = xb.shape
    ns = a.node_scale if a.node_scale is not None else (3.0 if a.is_binary else 0.5)
    control = R.s4({"n.trees": R.integer([a.n_trees]), "n.thin": R.integer([a.n_thin]), "keepTrees": R.lgl(a.keep_trees)})
    data = R.s4({"x": R.real(xb, dim=(n, p)), "n.cuts": R.integer(np.broadcast_to(np.asarray(a.n_cuts), (p,))),
                 "x.test": R.real(a.x_test, dim=np.asarray(a.x_test).shape) if a.x_test is not None and len(a.x_test) else R.nil()})
    pp = a.proposal_probs
    # normal(k): node.prior carries k and node.hyperprior is dbarts' fixed hyperprior; normal(k = chi(df, scale)): node.hyperprior is a
    # dbartsChiHyperprior with the slots degreesOfFreedom and scale (R/stan4bart_fit.R:460-479)
    k_hyper = getattr(a, "k_hyper", None)
    hyper = R.s4({"k": R.real([a.k])}) if k_hyper is None else R.s4({"degreesOfFreedom": R.real([k_hyper[0]]), "scale": R.real([k_hyper[1]])})
    model = R.s4({"tree.prior": R.s4({"power": R.real([a.power]), "base": R.real([a.base])}), "node.prior": R.s4({"k": R.real([a.k])}) if k_hyper is None else R.s4({}),
                  "node.hyperprior": hyper, "node.scale": R.real([ns]),
                  "p.birth_death": R.real([pp[0]]), "p.swap": R.real([pp[1]]), "p.change": R.real([pp[2]]), "p.birth": R.real([pp[3]])})
    X = np.asarray(a.X if a.X is not None else np.zeros((n, 0)), dtype=np.float64).reshape(n, -1)
    K = X.shape[1]
    t = len(a.p)
    q = int(sum(int(pi) * int(li) for pi, li in zip(a.p, a.l)))
    w = np.asarray(a.w if a.w is not None else np.zeros(0), dtype=np.float64)
    sd = [("N", R.integer([n])), ("K", R.integer([K])), ("X", R.real(X, dim=(n, K))), ("len_y", R.integer([n])), ("lb_y", R.real([-np.inf])), ("ub_y", R.real([np.inf])),
          ("y", R.real(a.y)), ("has_intercept", R.integer([0])), ("is_binary", R.integer([int(a.is_binary)])), ("prior_dist", R.integer([a.prior_dist])),
          ("prior_dist_for_intercept", R.integer([0])), ("prior_dist_for_aux", R.integer([a.prior_dist_for_aux])),
          ("has_weights", R.integer([int(a.weights is not None and len(a.weights) > 0)])), ("weights", R.real(a.weights if a.weights is not None else [])),
          ("offset_", R.real(np.zeros(n))), ("prior_scale", R.real(a.prior_scale if a.prior_scale is not None else np.ones(K))),
          ("prior_scale_for_intercept", R.real([0.0])), ("prior_scale_for_aux", R.real([a.prior_scale_for_aux])),
          ("prior_mean", R.real(a.prior_mean if a.prior_mean is not None else np.zeros(K))), ("prior_mean_for_intercept", R.real([0.0])),
          ("prior_mean_for_aux", R.real([a.prior_mean_for_aux])), ("prior_df", R.real(a.prior_df if a.prior_df is not None else np.ones(K))),
          ("prior_df_for_intercept", R.real([1.0])), ("prior_df_for_aux", R.real([a.prior_df_for_aux])),
          ("global_prior_df", R.real([a.global_prior_df])), ("global_prior_scale", R.real([a.global_prior_scale])), ("slab_df", R.real([a.slab_df])),
          ("slab_scale", R.real([a.slab_scale])), ("num_normals", R.integer(a.num_normals if a.num_normals is not None else [])),
          ("t", R.integer([t])), ("p", R.integer(a.p)), ("l", R.integer(a.l)), ("q", R.integer([q])),
          ("len_theta_L", R.integer([int(sum(pi * (pi - 1) // 2 + pi for pi in a.p))])), ("shape", R.real(a.shape)), ("scale", R.real(a.scale)),
          ("len_concentration", R.integer([len(a.concentration)])), ("concentration", R.real(a.concentration)),
          ("len_regularization", R.integer([len(a.regularization)])), ("regularization", R.real(a.regularization)),
          ("num_non_zero", R.integer([len(w)])), ("w", R.real(w)), ("v", R.integer(a.v if a.v is not None else [])),
          ("u", R.integer(a.u if a.u is not None else np.zeros(n + 1)))]
    if drop:
        sd = [kv for kv in sd if kv[0] != drop]
    stan_data = R.named_list(sd)
    stan_control = R.named_list([("seed", R.integer([a.seed])), ("init_r", R.real([a.init_r])), ("skip", R.integer([a.skip])),
                                 ("adapt_gamma", R.real([a.adapt_gamma])), ("adapt_delta", R.real([a.adapt_delta])), ("adapt_kappa", R.real([a.adapt_kappa])),
                                 ("adapt_init_buffer", R.integer([a.adapt_init_buffer])), ("adapt_term_buffer", R.integer([a.adapt_term_buffer])),
                                 ("adapt_window", R.integer([a.adapt_window])), ("adapt_t0", R.real([a.adapt_t0])), ("stepsize", R.real([a.stepsize])),
                                 ("stepsize_jitter", R.real([a.stepsize_jitter])), ("max_treedepth", R.integer([a.max_treedepth])), ("hmc_mode", R.integer([a.hmc_mode]))])
    common = R.named_list([("warmup", R.integer([a.warmup])), ("iter", R.integer([a.iter])), ("verbose", R.integer([a.verbose])), ("refresh", R.integer([a.refresh or 200])),
                           ("is_binary", R.lgl(a.is_binary)), ("offset", R.nil() if a.offset is None else R.real(a.offset)), ("offset_type", R.integer([a.offset_type])),
                           ("bart_offset_init", R.nil() if a.bart_offset_init is None else R.real(a.bart_offset_init)), ("sigma_init", R.real([a.sigma_init])), ("keep_fits", R.lgl(a.keep_fits)),
                           ("callback", R.nil() if callback is None else R.closure(callback)), ("callbackEnv", R.nil() if callback is None else R.lib.mock_env())])
    return control, data, model, stan_data, stan_control, common


def _build():
    r = subprocess.run(["make", "-C", MOCK_DIR], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


@pytest.fixture(scope="module")
def shim_emul(emul_lib):
    _build()
    return MockR(os.path.join(MOCK_DIR, "_build", "libshim_emul.so"))


def drive(R, clib, prefix, n=150, n_test=9):
    """create -> run(warm-up) -> disengageAdaptation -> run(sampling) -> getParametricMean / getBARTDataRange -> exportBARTState ->
    createStoredBARTSampler -> predictBART -> getTrees -> printTrees -> finalize through the shim, and the same chain through ctypes."""
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    args, _ = friedman_case(n=n, T=7, warmup=5, iter=9, n_test=n_test, bart_args={"keepTrees": True})
    rng = RRng(4242)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    state0 = rng.state.copy()
    # ---- ctypes path
    seen_py = []
    import copy
    a2 = copy.copy(args)
    a2.callback = lambda tr, te, sp: seen_py.append(float(tr[0] + sp[0])) or np.array([tr[0] + sp[0], te[0]])
    s = Sampler(clib, prefix, a2, state0)
    w = s.run(args.warmup, True, 0)
    s.disengage_adaptation()
    r = s.run(args.iter - args.warmup, False, 0)
    pm, rg, st_bytes, rng_after = s.get_parametric_mean(), s.get_bart_data_range(), s.export_bart_state(), s.get_r_rng_state()
    from stan4bart_amd.abi import StoredSampler
    stored = StoredSampler(clib, prefix, st_bytes)
    pred = stored.predict_bart(np.asarray(args.x_test))
    trees = stored.get_kept_trees()
    par_names = s.stan_par_names()
    s.free()
    # ---- the shim
    R.set_seed(state0)
    cb_seen = []

    def r_callback(train, test, pars, names):
        assert names == par_names            # the parameter vector is named (reference src/init.cpp:880-895)
        cb_seen.append(float(train[0] + pars[0]))
        return np.array([train[0] + pars[0], test[0]])
    six = r_arguments(R, args, callback=r_callback)
    ptr = R.call("stan4bart_create", *six)
    assert R.type(ptr) == EXTPTRSXP
    R.call("stan4bart_printInitialSummary", ptr)
    rw = R.as_dict(R.call("stan4bart_run", ptr, R.integer([args.warmup]), R.lgl(True), R.string("both")))
    R.call("stan4bart_disengageAdaptation", ptr)
    rs_list = R.call("stan4bart_run", ptr, R.integer([args.iter - args.warmup]), R.lgl(False), R.string("both"))
    rs = R.as_dict(rs_list)
    assert list(rs) == ["stan", "bart", "callback"]
    # result shapes and names of the reference (src/stan_sampler.cpp:577-596, src/bart_util.cpp:13-81)
    stan_s = R.to_numpy(rs["stan"])
    assert R.strings(R.lib.mock_elt(R.lib.mock_get_attr(rs["stan"], b"dimnames"), 0)) == par_names
    bart_s = R.as_dict(rs["bart"])
    import json
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_call_interface.json")))
    assert list(bart_s) == ref["bartResultNames"] == ["sigma", "train", "test", "varcount"]
    assert np.array_equal(R.to_numpy(rw["stan"]), w["stan"]) and np.array_equal(stan_s, r["stan"])
    assert np.array_equal(R.to_numpy(bart_s["train"]), r["bart"]["train"]) and np.array_equal(R.to_numpy(bart_s["test"]), r["bart"]["test"])
    assert np.array_equal(R.to_numpy(bart_s["sigma"]), r["bart"]["sigma"]) and np.array_equal(R.to_numpy(bart_s["varcount"]), r["bart"]["varcount"])
    assert R.to_numpy(bart_s["varcount"]).dtype == np.int32
    cbm = R.to_numpy(rs["callback"])                       # [length of the callback's value x iterations]
    assert cbm.shape == (2, args.iter - args.warmup) and np.array_equal(cbm[0], np.array(seen_py[args.warmup:])) and cb_seen == seen_py
    assert np.array_equal(R.get_seed(), rng_after)         # .Random.seed advanced exactly as the device stream did
    assert np.array_equal(R.to_numpy(R.call("stan4bart_getParametricMean", ptr)), pm)
    assert np.array_equal(R.to_numpy(R.call("stan4bart_getBARTDataRange", ptr)), rg)
    raw = R.call("stan4bart_exportBARTState", ptr)
    assert bytes(R.to_numpy(raw).tobytes()) == bytes(st_bytes)
    stp = R.call("stan4bart_createStoredBARTSampler", six[0], six[1], six[2], R.named_list([(None, raw)]))
    xt = np.asarray(args.x_test, dtype=np.float64)
    assert np.array_equal(R.to_numpy(R.call("stan4bart_predictBART", stp, R.real(xt, dim=xt.shape), R.nil())), pred)
    off = np.linspace(-1, 1, n_test)
    assert np.array_equal(R.to_numpy(R.call("stan4bart_predictBART", stp, R.real(xt, dim=xt.shape), R.real(off))), stored.predict_bart(xt, off))
    df = R.call("stan4bart_getTrees", stp, R.nil(), R.nil(), R.nil(), R.lgl(False))
    cols = R.as_dict(df)
    assert list(cols) == ["sample", "tree", "n", "var", "value"] and R.strings(R.lib.mock_get_attr(df, b"class")) == ["data.frame"]
    assert np.array_equal(R.to_numpy(cols["sample"]), trees["sample"] + 1) and np.array_equal(R.to_numpy(cols["tree"]), trees["tree"] + 1)
    assert np.array_equal(R.to_numpy(cols["n"]), trees["n"]) and np.array_equal(R.to_numpy(cols["value"]), trees["value"])
    assert np.array_equal(R.to_numpy(cols["var"]), np.where(trees["var"] >= 0, trees["var"] + 1, -1))
    sub = R.as_dict(R.call("stan4bart_getTrees", stp, R.nil(), R.integer([2, 4]), R.integer([3]), R.lgl(False)))
    keep = np.isin(trees["sample"], [1, 3]) & (trees["tree"] == 2)
    assert np.array_equal(R.to_numpy(sub["value"]), trees["value"][keep])
    R.lib.mock_clear_printed()
    R.call("stan4bart_printTrees", stp, R.nil(), R.integer([1]), R.integer([1, 2]))
    # reference messages for bad indices (src/init.cpp:467-478) arrive as R errors
    R.call("stan4bart_getTrees", stp, R.nil(), R.integer(list(range(1, 40))), R.nil(), R.lgl(False), expect=1)
    assert "samples specified but only" in R.error()
    # ---- user interrupt during a run: the run stops, the sampler survives and continues
    R.lib.mock_set_interrupt_pending(1)
    R.call("stan4bart_run", ptr, R.integer([3]), R.lgl(False), R.string("both"), expect=2)
    more = R.as_dict(R.call("stan4bart_run", ptr, R.integer([2]), R.lgl(False), R.integer([1])))      # resultsType 1: BART only
    assert list(more) == ["bart", "callback"]
    # ---- finalizers (what R's garbage collector / .onUnload would run)
    R.lib.mock_finalize(stp)
    R.call("stan4bart_finalize")
    R.call("stan4bart_getParametricMean", ptr, expect=1)
    assert "NULL external pointer" in R.error()
    return True


def test_shim_executes_over_the_emulation(shim_emul):
    assert drive(shim_emul, shim_emul.lib, "s4b_")


def test_shim_argument_errors_are_r_errors(shim_emul):
    """validation of the reference (src/stan_sampler.cpp:115-139, src/init.cpp:1015-1051) reaches R as an error with its message"""
    from stan4bart_amd import RRng
    R = shim_emul
    args, _ = friedman_case(n=60, T=3, warmup=2, iter=4)
    rng = RRng(1)
    args.seed = 7
    R.set_seed(rng.state)
    six = list(r_arguments(R, args, drop="prior_df"))
    R.call("stan4bart_create", *six, expect=1)
    assert R.error() == "stanData requires 'prior_df' to be specified"
    six = list(r_arguments(R, args))
    six[5] = R.named_list([("warmup", R.integer([2])), ("iter", R.integer([4])), ("keep_fits", R.lgl(True))])
    R.call("stan4bart_create", *six, expect=1)
    assert "is_binary" in R.error()
    six = list(r_arguments(R, args))
    six[4] = R.named_list([("init_r", R.real([2.0]))])
    R.call("stan4bart_create", *six, expect=1)
    assert R.error() == "stanControl requires 'seed' to be specified"
    R.call("stan4bart_run", R.lib.mock_nil(), R.integer([1]), R.lgl(True), R.string("both"), expect=None)   # NULL pointer object: no crash required of R; skip
    # keep_fits = FALSE: run returns list(callback = NULL) (reference src/init.cpp:725-733, 935-950)
    import copy
    a3 = copy.copy(args)
    a3.keep_fits = False
    R.set_seed(rng.state)
    ptr = R.call("stan4bart_create", *r_arguments(R, a3))
    out = R.call("stan4bart_run", ptr, R.integer([3]), R.lgl(True), R.string("both"))
    assert R.names(out) == ["callback"] and R.type(R.lib.mock_elt(out, 0)) == NILSXP
    R.call("stan4bart_finalize")


def _k_through_the_shim(R, clib, prefix, n=140):
    """bart_args = list(k = chi(1.25, Inf)) through the .Call layer: model@node.hyperprior is unpacked, the bart result list has the
    reference's fifth element "k" (src/bart_util.cpp:17-26,60-64,75-76) and equals the ctypes path's draws bit for bit."""
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    args, _ = friedman_case(n=n, T=6, warmup=4, iter=9, bart_args={"k": "chi(1.25, Inf)"})
    rng = RRng(99)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    state0 = rng.state.copy()
    s = Sampler(clib, prefix, args, state0)
    w = s.run(args.warmup, True, 0)
    s.disengage_adaptation()
    r = s.run(args.iter - args.warmup, False, 0)
    s.free()
    R.set_seed(state0)
    ptr = R.call("stan4bart_create", *r_arguments(R, args))
    rw = R.as_dict(R.as_dict(R.call("stan4bart_run", ptr, R.integer([args.warmup]), R.lgl(True), R.string("both")))["bart"])
    R.call("stan4bart_disengageAdaptation", ptr)
    rs = R.as_dict(R.as_dict(R.call("stan4bart_run", ptr, R.integer([args.iter - args.warmup]), R.lgl(False), R.string("both")))["bart"])
    import json
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_call_interface.json")))
    assert list(rs) == ref["bartResultNamesWithModeledK"] == ["sigma", "train", "test", "varcount", "k"]
    assert np.array_equal(R.to_numpy(rw["k"]), w["bart"]["k"]) and np.array_equal(R.to_numpy(rs["k"]), r["bart"]["k"])
    assert np.array_equal(R.to_numpy(rs["train"]), r["bart"]["train"]) and np.std(r["bart"]["k"]) > 0
    R.lib.mock_finalize(ptr)      # (what R's garbage collector would run; stan4bart_finalize is the package's unload hook and ends the session's bookkeeping)
    return True


def test_shim_returns_the_k_draws_of_a_modeled_k(shim_emul):
    assert _k_through_the_shim(shim_emul, shim_emul.lib, "s4b_")


@pytest.mark.gpu
def test_shim_returns_the_k_draws_over_the_hip_library(hip_lib):
    _build()
    R = MockR(os.path.join(MOCK_DIR, "_build", "libshim_hip.so"))
    assert _k_through_the_shim(R, hip_lib, "s4b_", n=1500)


@pytest.mark.gpu
def test_shim_executes_over_the_hip_library(hip_lib):
    _build()
    R = MockR(os.path.join(MOCK_DIR, "_build", "libshim_hip.so"))
    assert drive(R, hip_lib, "s4b_", n=2000, n_test=17)
