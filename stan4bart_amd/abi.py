"""ctypes view of the C-ABI declared in ``include/stan4bart_amd.h``.

This is the Python stand-in for the R ``.Call`` shim (there is no R in this image): it marshals
numpy arrays into the plain structs of the C boundary, exactly as the R shim in
``INTEGRATION.md`` unpacks SEXPs.  The class is generic over the symbol prefix so that the test
suite can drive the CPU oracle (``orc_*``, test infrastructure only) through the very same code.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

INTERFACE_VERSION = 6     # S4B_INTERFACE_VERSION of include/stan4bart_amd.h (tests/test_host_logic.py compares the two)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)


class BartControl(C.Structure):
    _fields_ = [("n_trees", C.c_int32), ("n_thin", C.c_int32), ("keep_trees", C.c_int32),
                ("node_capacity", C.c_int32),
                ("base", C.c_double), ("power", C.c_double), ("k", C.c_double), ("node_scale", C.c_double),
                ("birth_or_death_prob", C.c_double), ("swap_prob", C.c_double),
                ("change_prob", C.c_double), ("birth_prob", C.c_double),
                ("split_probs", c_double_p), ("use_quantiles", C.c_int32), ("interface_version", C.c_int32),
                ("k_hyper_df", C.c_double), ("k_hyper_scale", C.c_double)]


class BartData(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_int32), ("reserved", C.c_int32),
                ("x", c_double_p), ("n_cuts", c_int32_p),
                ("n_test", C.c_int64), ("x_test", c_double_p)]


class StanData(C.Structure):
    _fields_ = [("N", C.c_int64), ("K", C.c_int32),
                ("is_binary", C.c_int32), ("has_intercept", C.c_int32), ("has_weights", C.c_int32),
                ("prior_dist", C.c_int32), ("prior_dist_for_aux", C.c_int32),
                ("X", c_double_p), ("y", c_double_p), ("weights", c_double_p),
                ("prior_scale", c_double_p), ("prior_mean", c_double_p), ("prior_df", c_double_p),
                ("prior_scale_for_aux", C.c_double), ("prior_mean_for_aux", C.c_double),
                ("prior_df_for_aux", C.c_double),
                ("t", C.c_int32), ("q", C.c_int32),
                ("len_theta_L", C.c_int32), ("len_concentration", C.c_int32),
                ("len_regularization", C.c_int32), ("reserved", C.c_int32),
                ("p", c_int32_p), ("l", c_int32_p),
                ("shape", c_double_p), ("scale", c_double_p),
                ("concentration", c_double_p), ("regularization", c_double_p),
                ("num_non_zero", C.c_int64),
                ("w", c_double_p), ("v", c_int32_p), ("u", c_int32_p),
                ("global_prior_df", C.c_double), ("global_prior_scale", C.c_double),
                ("slab_df", C.c_double), ("slab_scale", C.c_double), ("num_normals", c_int32_p)]


class StanControl(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("skip", C.c_int32), ("init_r", C.c_double),
                ("adapt_gamma", C.c_double), ("adapt_delta", C.c_double),
                ("adapt_kappa", C.c_double), ("adapt_t0", C.c_double),
                ("adapt_init_buffer", C.c_uint32), ("adapt_term_buffer", C.c_uint32),
                ("adapt_window", C.c_uint32), ("reserved", C.c_uint32),
                ("stepsize", C.c_double), ("stepsize_jitter", C.c_double),
                ("max_treedepth", C.c_int32), ("hmc_mode", C.c_int32)]


CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int32)
PROGRESS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32)


class CommonControl(C.Structure):
    _fields_ = [("warmup", C.c_int32), ("iter", C.c_int32), ("verbose", C.c_int32), ("refresh", C.c_int32),
                ("is_binary", C.c_int32), ("offset_type", C.c_int32), ("keep_fits", C.c_int32),
                ("device", C.c_int32),
                ("offset", c_double_p), ("bart_offset_init", c_double_p),
                ("sigma_init", C.c_double),
                ("callback", CALLBACK), ("callback_user", C.c_void_p)]


class Results(C.Structure):
    _fields_ = [("stan", c_double_p), ("bart_sigma", c_double_p), ("bart_train", c_double_p),
                ("bart_test", c_double_p), ("bart_varcount", c_int32_p), ("bart_k", c_double_p)]


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def _f64(a, order="F") -> np.ndarray:
    return np.require(np.asarray(a, dtype=np.float64), requirements=["A", "O"] + (["F"] if order == "F" else ["C"]))


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


@dataclass
class SamplerArgs:
    """The six argument objects of ``stan4bart_create`` (reference R/stan4bart_fit.R:42) as numpy/python."""
    # bart data / control / model
    x_bart: np.ndarray                      # n x p
    n_trees: int = 75
    n_cuts: int | Sequence[int] = 100
    n_thin: int = 1
    x_test: Optional[np.ndarray] = None
    base: float = 0.95
    power: float = 2.0
    k: float = 2.0
    node_scale: Optional[float] = None      # default 0.5 continuous / 3.0 binary
    keep_trees: bool = False
    node_capacity: int = 0
    proposal_probs: Sequence[float] = (0.5, 0.1, 0.4, 0.5)
    split_probs: Optional[Sequence[float]] = None     # cgm(split.probs): one positive weight per BART predictor, or None (uniform)
    use_quantiles: bool = False                       # dbartsControl(useQuantiles): cut points from the distinct values
    k_hyper: Optional[tuple] = None                   # normal(k = chi(degreesOfFreedom, scale)): (df, scale) -> k is sampled; `k` is where it starts
    # stan data
    X: Optional[np.ndarray] = None          # n x K (already centred)
    y: Optional[np.ndarray] = None
    weights: Optional[np.ndarray] = None
    is_binary: bool = False
    prior_dist: int = 1
    prior_dist_for_aux: int = 3
    prior_scale: Optional[np.ndarray] = None
    prior_mean: Optional[np.ndarray] = None
    prior_df: Optional[np.ndarray] = None
    prior_scale_for_aux: float = 1.0
    prior_mean_for_aux: float = 0.0
    prior_df_for_aux: float = 1.0
    global_prior_df: float = 1.0            # hs / hs_plus
    global_prior_scale: float = 0.01
    slab_df: float = 4.0
    slab_scale: float = 2.5
    num_normals: Optional[Sequence[int]] = None   # product_normal
    p: Sequence[int] = ()
    l: Sequence[int] = ()
    shape: Sequence[float] = ()
    scale: Sequence[float] = ()
    concentration: Sequence[float] = ()
    regularization: Sequence[float] = ()
    w: Optional[np.ndarray] = None
    v: Optional[np.ndarray] = None
    u: Optional[np.ndarray] = None
    # stan control
    seed: int = 0
    skip: int = 1
    init_r: float = 2.0
    adapt_gamma: float = 0.05
    adapt_delta: float = 0.8
    adapt_kappa: float = 0.75
    adapt_t0: float = 10.0
    adapt_init_buffer: int = 75
    adapt_term_buffer: int = 50
    adapt_window: int = 25
    stepsize: float = 1.0
    stepsize_jitter: float = 0.0
    max_treedepth: int = 10
    hmc_mode: int = 0
    # common control
    warmup: int = 1000
    iter: int = 2000
    verbose: int = 0
    refresh: int = 0
    offset: Optional[np.ndarray] = None
    offset_type: int = 0
    bart_offset_init: Optional[np.ndarray] = None
    sigma_init: float = 1.0
    keep_fits: bool = True
    callback: Optional[Callable] = None
    device: int = 0
    extras: dict = field(default_factory=dict)


class Sampler:
    """One chain: wraps ``<prefix>create / run / disengage_adaptation / ... / free``."""

    def __init__(self, lib: C.CDLL, prefix: str, args: SamplerArgs, r_rng_state: np.ndarray):
        self._lib, self._pfx = lib, prefix
        self._bind()
        a = args
        self._keep = []  # keep numpy buffers alive for the duration of create()

        def keep(x):
            self._keep.append(x)
            return x

        xb = keep(_f64(a.x_bart))
        n, p = xb.shape
        ncuts = keep(_i32(np.broadcast_to(np.asarray(a.n_cuts), (p,))))
        xt = keep(_f64(a.x_test)) if a.x_test is not None and len(a.x_test) else None
        bd = BartData(n=n, p=p, reserved=0, x=_dp(xb), n_cuts=_ip(ncuts),
                      n_test=0 if xt is None else xt.shape[0], x_test=_dp(xt))
        ns = a.node_scale if a.node_scale is not None else (3.0 if a.is_binary else 0.5)
        pp = a.proposal_probs
        bc = BartControl(n_trees=a.n_trees, n_thin=a.n_thin, keep_trees=int(a.keep_trees),
                         node_capacity=a.node_capacity, base=a.base, power=a.power, k=a.k, node_scale=ns,
                         birth_or_death_prob=pp[0], swap_prob=pp[1], change_prob=pp[2], birth_prob=pp[3],
                         use_quantiles=int(bool(a.use_quantiles)), interface_version=INTERFACE_VERSION,
                         k_hyper_df=0.0 if a.k_hyper is None else float(a.k_hyper[0]),
                         k_hyper_scale=float("inf") if a.k_hyper is None else float(a.k_hyper[1]))
        self.k_modeled = a.k_hyper is not None
        if a.split_probs is not None:
            sp = keep(_f64(np.asarray(a.split_probs, dtype=np.float64)))
            if sp.shape != (xb.shape[1],):
                raise ValueError("split_probs needs one weight per BART predictor")
            bc.split_probs = _dp(sp)

        X = keep(_f64(a.X if a.X is not None else np.zeros((n, 0))))
        if X.ndim == 1:
            X = keep(_f64(X.reshape(n, 1)))
        K = X.shape[1]
        y = keep(_f64(a.y))
        t = len(a.p)
        q = int(sum(int(pi) * int(li) for pi, li in zip(a.p, a.l)))
        len_theta_L = int(sum(pi * (pi - 1) // 2 + pi for pi in a.p))
        ps = keep(_f64(a.prior_scale if a.prior_scale is not None else np.ones(K)))
        pm = keep(_f64(a.prior_mean if a.prior_mean is not None else np.zeros(K)))
        pdf = keep(_f64(a.prior_df if a.prior_df is not None else np.ones(K)))
        wts = keep(_f64(a.weights)) if a.weights is not None and len(a.weights) else None
        p_arr, l_arr = keep(_i32(a.p)), keep(_i32(a.l))
        shape, scale = keep(_f64(a.shape)), keep(_f64(a.scale))
        conc, reg = keep(_f64(a.concentration)), keep(_f64(a.regularization))
        w = keep(_f64(a.w if a.w is not None else np.zeros(0)))
        v = keep(_i32(a.v if a.v is not None else np.zeros(0)))
        u = keep(_i32(a.u if a.u is not None else np.zeros(n + 1)))
        sd = StanData(N=n, K=K, is_binary=int(a.is_binary), has_intercept=0, has_weights=int(wts is not None),
                      prior_dist=a.prior_dist, prior_dist_for_aux=a.prior_dist_for_aux,
                      X=_dp(X), y=_dp(y), weights=_dp(wts), prior_scale=_dp(ps), prior_mean=_dp(pm), prior_df=_dp(pdf),
                      prior_scale_for_aux=a.prior_scale_for_aux, prior_mean_for_aux=a.prior_mean_for_aux,
                      prior_df_for_aux=a.prior_df_for_aux, t=t, q=q, len_theta_L=len_theta_L,
                      len_concentration=len(conc), len_regularization=len(reg), reserved=0,
                      p=_ip(p_arr), l=_ip(l_arr), shape=_dp(shape), scale=_dp(scale),
                      concentration=_dp(conc), regularization=_dp(reg), num_non_zero=len(w),
                      w=_dp(w), v=_ip(v), u=_ip(u),
                      global_prior_df=a.global_prior_df, global_prior_scale=a.global_prior_scale, slab_df=a.slab_df,
                      slab_scale=a.slab_scale,
                      num_normals=_ip(keep(_i32(a.num_normals))) if a.num_normals is not None else None)
        sc = StanControl(seed=a.seed & 0xFFFFFFFF, skip=a.skip, init_r=a.init_r, adapt_gamma=a.adapt_gamma,
                         adapt_delta=a.adapt_delta, adapt_kappa=a.adapt_kappa, adapt_t0=a.adapt_t0,
                         adapt_init_buffer=a.adapt_init_buffer, adapt_term_buffer=a.adapt_term_buffer,
                         adapt_window=a.adapt_window, reserved=0, stepsize=a.stepsize,
                         stepsize_jitter=a.stepsize_jitter, max_treedepth=a.max_treedepth, hmc_mode=a.hmc_mode)
        off = keep(_f64(a.offset)) if a.offset is not None and len(a.offset) else None
        boi = keep(_f64(a.bart_offset_init)) if a.bart_offset_init is not None else None
        self._py_callback = a.callback
        self._cb = CALLBACK(self._trampoline) if a.callback is not None else CALLBACK()
        cc = CommonControl(warmup=a.warmup, iter=a.iter, verbose=a.verbose, refresh=a.refresh,
                           is_binary=int(a.is_binary), offset_type=a.offset_type, keep_fits=int(a.keep_fits),
                           device=a.device, offset=_dp(off), bart_offset_init=_dp(boi), sigma_init=a.sigma_init,
                           callback=self._cb, callback_user=None)
        st = np.ascontiguousarray(r_rng_state, dtype=np.uint32)
        assert st.shape == (625,)
        self._h = C.c_void_p()
        rc = self._f("create")(C.byref(bc), C.byref(bd), C.byref(sd), C.byref(sc), C.byref(cc),
                               st.ctypes.data_as(c_uint32_p), C.byref(self._h))
        self._check(rc)
        self._keep.clear()
        dims = (C.c_int64 * 5)()
        self._check(self._f("get_dims")(self._h, dims))
        self.num_pars, self.n, self.n_test, self.p, self.n_trees = (int(d) for d in dims)
        self.keep_fits = bool(a.keep_fits)
        self.callback_results: list = []
        self._pending_exc = None

    # ------------------------------------------------------------------ plumbing
    def _bind(self):
        """restype + argtypes of every entry point of include/stan4bart_amd.h (64-bit counts travel as c_int64)."""
        vp, i32, i64, dp, ip, up = C.c_void_p, C.c_int32, C.c_int64, c_double_p, c_int32_p, c_uint32_p
        sig = {
            "create": [C.POINTER(BartControl), C.POINTER(BartData), C.POINTER(StanData), C.POINTER(StanControl),
                       C.POINTER(CommonControl), up, C.POINTER(vp)],
            "run": [vp, i32, i32, i32, C.POINTER(Results)],
            "disengage_adaptation": [vp], "print_initial_summary": [vp],
            "get_parametric_mean": [vp, dp], "get_bart_data_range": [vp, dp],
            "get_r_rng_state": [vp, up], "set_r_rng_state": [vp, up],
            "get_dims": [vp, C.POINTER(i64)], "get_stan_par_names": [vp, C.c_char_p, C.c_size_t],
            "get_trees": [vp, i64, ip, ip, ip, ip, dp, C.POINTER(i64)],
            "get_kept_trees": [vp, i64, i64, ip, ip, ip, ip, ip, dp, C.POINTER(i64)],
            "get_kept_trees_indexed": [vp, ip, i64, ip, i64, i64, ip, ip, ip, ip, ip, dp, C.POINTER(i64)],
            "export_bart_state": [vp, vp, i64, C.POINTER(i64)],
            "create_stored_bart_sampler": [vp, i64, i32, C.POINTER(vp)],
            "predict_bart": [vp, dp, i64, dp, C.POINTER(i64)],
            "predict_bart_offset": [vp, dp, i64, dp, dp, C.POINTER(i64)], "print_trees": [vp, ip, i64, ip, i64],
            "get_state": [vp, vp, i64, C.POINTER(i64)], "set_state": [vp, vp, i64],
            "set_trace": [vp, i32], "get_trace": [vp, i64, ip, C.POINTER(i64)],
            "get_leaf_assignment": [vp, i32, ip], "get_counters": [vp, C.POINTER(i64)], "get_nuts_stats": [vp, dp],
            "profile_sweep": [vp, i32, dp], "profile_leapfrog": [vp, i32, dp],
            "set_progress": [vp, PROGRESS, vp], "set_device_sharing": [vp, i32],
            "set_tree_path": [vp, i32], "get_tree_path": [vp, ip], "get_fused_stats": [vp, C.POINTER(i64)], "get_sweep_stats": [vp, C.POINTER(i64)], "get_sweep_busy": [vp, C.POINTER(i64)], "get_sweep_spec": [vp, C.POINTER(i64)], "set_test_hook": [vp, i32, i64], "set_hmc_mode": [vp, i32], "get_hmc_mode": [vp, ip],
        }
        for name, argtypes in sig.items():
            fn = getattr(self._lib, self._pfx + name, None)
            if fn is None:          # optional entry points (the oracle exports the subset the tests drive)
                continue
            fn.restype = C.c_int
            fn.argtypes = argtypes
        getattr(self._lib, self._pfx + "last_error").restype = C.c_char_p
        getattr(self._lib, self._pfx + "last_error").argtypes = []
        getattr(self._lib, self._pfx + "free").restype = None
        getattr(self._lib, self._pfx + "free").argtypes = [vp]

    def _f(self, name):
        return getattr(self._lib, self._pfx + name)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._f("last_error")().decode())

    def _trampoline(self, user, yhat_train, yhat_test, stan_pars, num_pars):
        tr = np.ctypeslib.as_array(yhat_train, shape=(self.n,)).copy()
        te = np.ctypeslib.as_array(yhat_test, shape=(self.n_test,)).copy() if self.n_test else None
        sp = np.ctypeslib.as_array(stan_pars, shape=(num_pars,)).copy()
        try:
            self.callback_results.append(self._py_callback(tr, te, sp))
        except BaseException as e:       # ctypes would swallow it: remember it, stop the run (non-zero return), re-raise after
            if self._pending_exc is None:
                self._pending_exc = e
            return 1
        return 0

    # ------------------------------------------------------------------ the .Call surface
    def run(self, num_iter: int, is_warmup: bool, results_type: int = 0) -> dict:
        """``stan4bart_run``: returns dict(stan=[num_pars x S], bart=dict(sigma, train, test, varcount))."""
        S = num_iter if self.keep_fits else 1
        stan = np.zeros((self.num_pars, S), order="F")
        sigma = np.zeros(S)
        train = np.zeros((self.n, S), order="F")
        test = np.zeros((self.n_test, S), order="F")
        varcount = np.zeros((self.p, S), dtype=np.int32, order="F")
        kdraws = np.zeros(S)
        res = Results(stan=_dp(stan), bart_sigma=_dp(sigma), bart_train=_dp(train),
                      bart_test=_dp(test) if self.n_test else None, bart_varcount=_ip(varcount),
                      bart_k=_dp(kdraws) if getattr(self, "k_modeled", False) else None)
        self.callback_results = []
        self._pending_exc = None
        rc = self._f("run")(self._h, num_iter, int(is_warmup), results_type, C.byref(res))
        if self._pending_exc is not None:           # an exception inside the per-iteration callback or the progress hook
            e, self._pending_exc = self._pending_exc, None
            raise e
        self._check(rc)
        out = {}
        if results_type in (0, 2):
            out["stan"] = stan
        if results_type in (0, 1):
            out["bart"] = dict(sigma=sigma, train=train, test=test, varcount=varcount)
            if getattr(self, "k_modeled", False):        # the reference's fifth result element (src/bart_util.cpp:17-26,75-76): only when k is modeled
                out["bart"]["k"] = kdraws
        if self._py_callback is not None:
            out["callback"] = list(self.callback_results)
        return out

    def disengage_adaptation(self):
        self._check(self._f("disengage_adaptation")(self._h))

    def print_initial_summary(self):
        self._check(self._f("print_initial_summary")(self._h))

    def get_parametric_mean(self) -> np.ndarray:
        out = np.zeros(self.n)
        self._check(self._f("get_parametric_mean")(self._h, _dp(out)))
        return out

    def get_bart_data_range(self) -> np.ndarray:
        out = np.zeros(2)
        self._check(self._f("get_bart_data_range")(self._h, _dp(out)))
        return out

    def get_r_rng_state(self) -> np.ndarray:
        st = np.zeros(625, dtype=np.uint32)
        self._check(self._f("get_r_rng_state")(self._h, st.ctypes.data_as(c_uint32_p)))
        return st

    def set_r_rng_state(self, st: np.ndarray):
        st = np.ascontiguousarray(st, dtype=np.uint32)
        self._check(self._f("set_r_rng_state")(self._h, st.ctypes.data_as(c_uint32_p)))

    def stan_par_names(self) -> list[str]:
        buf = C.create_string_buffer(64 * (self.num_pars + 8))
        self._check(self._f("get_stan_par_names")(self._h, buf, len(buf)))
        return buf.value.decode().split("\n")

    def get_trees(self) -> dict:
        """Flattened live trees (``stan4bart_getTrees(current = TRUE)`` layout)."""
        nn = C.c_int64()
        self._check(self._f("get_trees")(self._h, 0, None, None, None, None, None, C.byref(nn)))
        m = nn.value
        tree, nobs, var, split = (np.zeros(m, dtype=np.int32) for _ in range(4))
        value = np.zeros(m)
        self._check(self._f("get_trees")(self._h, m, _ip(tree), _ip(nobs), _ip(var), _ip(split), _dp(value), C.byref(nn)))
        return dict(tree=tree, n=nobs, var=var, split=split, value=value)

    def get_kept_trees(self, sample: int = -1) -> dict:
        """Flattened trees of the kept draws (``stan4bart_getTrees(current = FALSE)``; ``sample`` 0-based, -1 = all)."""
        nn = C.c_int64()
        self._check(self._f("get_kept_trees")(self._h, C.c_int64(sample), 0, None, None, None, None, None, None, C.byref(nn)))
        m = nn.value
        smp, tree, nobs, var, split = (np.zeros(m, dtype=np.int32) for _ in range(5))
        value = np.zeros(m)
        self._check(self._f("get_kept_trees")(self._h, C.c_int64(sample), m, _ip(smp), _ip(tree), _ip(nobs), _ip(var), _ip(split), _dp(value),
                                              C.byref(nn)))
        return dict(sample=smp, tree=tree, n=nobs, var=var, split=split, value=value)

    def get_kept_trees_indexed(self, sample_idx=None, tree_idx=None) -> dict:
        """``stan4bart_getTrees(sampleIndices, treeIndices)``: 0-based index vectors (None = all, in order)."""
        if not hasattr(self._lib, self._pfx + "get_kept_trees_indexed"):      # (the oracle exports the scalar form only)
            parts = [self.get_kept_trees(-1)] if sample_idx is None else [self.get_kept_trees(int(k)) for k in sample_idx]
            out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
            if tree_idx is not None:
                order = np.concatenate([np.flatnonzero((out["sample"] == sm) & (out["tree"] == t)) for sm in (np.unique(out["sample"]) if sample_idx is None else sample_idx) for t in tree_idx] or [np.zeros(0, dtype=np.int64)])
                out = {k: v[order.astype(np.int64)] for k, v in out.items()}
            return out
        si = None if sample_idx is None else _i32(sample_idx)
        ti = None if tree_idx is None else _i32(tree_idx)
        args = (_ip(si), 0 if si is None else len(si), _ip(ti), 0 if ti is None else len(ti))
        nn = C.c_int64()
        self._check(self._f("get_kept_trees_indexed")(self._h, *args, 0, None, None, None, None, None, None, C.byref(nn)))
        m = nn.value
        smp, tree, nobs, var, split = (np.zeros(m, dtype=np.int32) for _ in range(5))
        value = np.zeros(m)
        self._check(self._f("get_kept_trees_indexed")(self._h, *args, m, _ip(smp), _ip(tree), _ip(nobs), _ip(var), _ip(split), _dp(value), C.byref(nn)))
        return dict(sample=smp, tree=tree, n=nobs, var=var, split=split, value=value)

    def print_trees(self, sample_idx=None, tree_idx=None):
        si = None if sample_idx is None else _i32(sample_idx)
        ti = None if tree_idx is None else _i32(tree_idx)
        self._check(self._f("print_trees")(self._h, _ip(si), 0 if si is None else len(si), _ip(ti), 0 if ti is None else len(ti)))

    def set_progress(self, fn):
        """``fn(iter, num_iter, is_warmup) -> bool`` (True cancels the run), or None."""
        self._progress_py = fn

        def hook(user, it, n, w):
            try:
                return int(bool(fn(it, n, bool(w))))
            except BaseException as e:   # cancel the run (non-zero) and re-raise once s4b_run has returned
                if self._pending_exc is None:
                    self._pending_exc = e
                return 1
        self._progress_c = PROGRESS(hook) if fn is not None else PROGRESS()
        self._check(self._f("set_progress")(self._h, self._progress_c, None))

    def set_device_sharing(self, chains: int):
        """Hint: ``chains`` samplers share this sampler's GPU (three or more: the tree update that leaves room for the others)."""
        fn = getattr(self._lib, self._pfx + "set_device_sharing", None)
        if fn is not None:          # (the CPU oracle has no such notion)
            self._check(fn(self._h, int(chains)))

    TREE_PATHS = {"auto": 0, "two-kernel": 1, "fused": 2, "persistent": 4, "stream": 5}

    def set_tree_path(self, path):
        """Device code of a tree update: "auto", "two-kernel" (k_tree + k_control), "fused" (k_step), "persistent" (k_sweep: one
        launch per sweep, the residual in registers) or "stream" (k_sweep_stream: the same launch with a streaming pass; on request
        only — it is slower than the other paths at every size)."""
        fn = getattr(self._lib, self._pfx + "set_tree_path", None)
        if fn is not None:          # (the CPU oracle has one path)
            self._check(fn(self._h, int(self.TREE_PATHS.get(path, path))))

    def get_tree_path(self):
        """(requested, in effect) as names."""
        fn = getattr(self._lib, self._pfx + "get_tree_path", None)
        if fn is None:
            return ("auto", "oracle")
        out = np.zeros(2, dtype=np.int32)
        self._check(fn(self._h, _ip(out)))
        names = {v: k for k, v in self.TREE_PATHS.items()}
        return (names[int(out[0])], names[int(out[1])])

    def set_hmc_mode(self, mode: int):
        self._check(self._f("set_hmc_mode")(self._h, int(mode)))

    def get_hmc_mode(self) -> int:
        fn = getattr(self._lib, self._pfx + "get_hmc_mode", None)
        if fn is None:
            return 1          # (the oracle evaluates the likelihood per leapfrog, like the reference)
        out = np.zeros(1, dtype=np.int32)
        self._check(fn(self._h, _ip(out)))
        return int(out[0])

    def get_fused_stats(self):
        """(evaluations of the fused O(N) Stan sums, evaluations repeated in plain doubles after a failed range check)"""
        out = (C.c_int64 * 2)()
        fn = getattr(self._lib, self._pfx + "get_fused_stats", None)
        if fn is not None:
            self._check(fn(self._h, out))
        return int(out[0]), int(out[1])

    def get_sweep_stats(self):
        """(sweeps run by the persistent kernel, of which handed over to k_step part-way) since creation."""
        fn = getattr(self._lib, self._pfx + "get_sweep_stats", None)
        if fn is None:
            return (0, 0)
        out = (C.c_int64 * 2)()
        self._check(fn(self._h, out))
        return (int(out[0]), int(out[1]))

    def get_sweep_spec(self):
        """(persistent launches that ran, tree updates decided inside them, steps published before their verdict, steps borne out) since creation."""
        fn = getattr(self._lib, self._pfx + "get_sweep_spec", None)
        if fn is None:
            return (0, 0, 0, 0)
        out = (C.c_int64 * 4)()
        self._check(fn(self._h, out))
        return tuple(int(v) for v in out)

    def set_test_hook(self, hook: int, value: int):
        """TEST HOOK (include/stan4bart_amd.h): hook 1, value k — every k-th persistent launch reports a busy device (0 = off)."""
        self._check(self._f("set_test_hook")(self._h, int(hook), int(value)))

    def get_sweep_busy(self) -> int:
        """Persistent launches that found the device shared (roll call failed; their sweeps ran as k_step launches) since creation."""
        fn = getattr(self._lib, self._pfx + "get_sweep_busy", None)
        if fn is None:
            return 0
        out = C.c_int64(0)
        self._check(fn(self._h, C.byref(out)))
        return int(out.value)

    def set_trace(self, enable: bool):
        self._check(self._f("set_trace")(self._h, int(enable)))

    def get_trace(self, cap: int = 1 << 20) -> np.ndarray:
        nn = C.c_int64()
        out = np.zeros((cap, 5), dtype=np.int32)
        self._check(self._f("get_trace")(self._h, cap, _ip(out), C.byref(nn)))
        return out[: nn.value].copy()

    def get_leaf_assignment(self, tree: int) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.int32)
        self._check(self._f("get_leaf_assignment")(self._h, tree, _ip(out)))
        return out

    def get_counters(self) -> np.ndarray:
        out = (C.c_int64 * 3)()
        self._check(self._f("get_counters")(self._h, out))
        return np.array(list(out), dtype=np.int64)

    def get_nuts_stats(self) -> dict:
        """Totals over all NUTS transitions since creation."""
        out = (C.c_double * 4)()
        self._check(self._f("get_nuts_stats")(self._h, out))
        return dict(transitions=int(out[0]), sum_treedepth=int(out[1]), sum_n_leapfrog=int(out[2]), divergent=int(out[3]))

    def predict_bart(self, x_test: np.ndarray, offset_test: Optional[np.ndarray] = None) -> np.ndarray:
        """``stan4bart_predictBART(x_test, offset_test)``: BART fit of every kept draw (keep_trees) at new rows, [n_test x samples]."""
        xt = _f64(x_test)
        ns = C.c_int64()
        self._check(self._f("predict_bart")(self._h, _dp(xt), xt.shape[0], None, C.byref(ns)))
        out = np.zeros((xt.shape[0], ns.value), order="F")
        if ns.value:
            if offset_test is not None:
                off = _f64(offset_test)
                if off.shape != (xt.shape[0],):
                    raise ValueError("length of offset_test must equal number of rows in x_test")
                self._check(self._f("predict_bart_offset")(self._h, _dp(xt), xt.shape[0], _dp(off), _dp(out), C.byref(ns)))
            else:
                self._check(self._f("predict_bart")(self._h, _dp(xt), xt.shape[0], _dp(out), C.byref(ns)))
        return out

    def profile_leapfrog(self, n_evals: int = 10) -> dict:
        """Per-leapfrog O(N) sums of the hmc_mode 1 path timed with HIP events (measurement hook of the HIP library)."""
        out = (C.c_double * 8)()
        self._check(self._f("profile_leapfrog")(self._h, n_evals, out))
        return dict(kernels_us=out[0], with_fetch_us=out[1], launches=out[2], n=int(out[3]), algorithmic_bytes=out[4])

    def get_state(self) -> bytes:
        """``s4b_get_state``: the whole sampler state (NUTS + adaptation, trees, fits, offsets, both generators)."""
        size = C.c_int64()
        self._check(self._f("get_state")(self._h, None, 0, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        self._check(self._f("get_state")(self._h, buf, size.value, C.byref(size)))
        return buf.raw[: size.value]

    def set_state(self, state: bytes):
        """``s4b_set_state``: resume from / be teacher-forced to a state produced by ``get_state`` of either implementation."""
        buf = C.create_string_buffer(state, len(state))
        self._check(self._f("set_state")(self._h, buf, len(state)))

    def export_bart_state(self) -> bytes:
        """``stan4bart_exportBARTState``: the kept trees + cut points + scales as one byte string."""
        size = C.c_int64()
        self._check(self._f("export_bart_state")(self._h, None, 0, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        self._check(self._f("export_bart_state")(self._h, buf, size.value, C.byref(size)))
        return buf.raw[: size.value]

    def profile_sweep(self, n_sweeps: int = 1) -> dict:
        """Extra BART sweeps timed with HIP events on the sampler's stream (measurement hook of the HIP library)."""
        out = (C.c_double * 8)()
        self._check(self._f("profile_sweep")(self._h, n_sweeps, out))
        return dict(stats_us=out[0], control_us=out[1], apply_us=out[2], launches=[out[3], out[4], out[5]],
                    sweep_wall_us=out[6], n=int(out[7]))

    def free(self):
        if getattr(self, "_h", None) and self._h.value:
            self._f("free")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class StoredSampler:
    """``stan4bart_createStoredBARTSampler`` (reference src/init.cpp:418-446, R/stan4bart_fit.R:572-580): a sampler rebuilt
    from an exported BART state — in another process, after the fitting sampler is gone — that predicts from the kept trees."""

    def __init__(self, lib: C.CDLL, prefix: str, state: bytes, device: int = 0):
        self._lib, self._pfx = lib, prefix
        self._h = C.c_void_p()
        Sampler._bind(self)
        buf = C.create_string_buffer(state, len(state))
        self._check(self._f("create_stored_bart_sampler")(buf, C.c_int64(len(state)), C.c_int32(device), C.byref(self._h)))

    _f = Sampler._f
    _check = Sampler._check
    predict_bart = Sampler.predict_bart
    export_bart_state = Sampler.export_bart_state
    get_kept_trees = Sampler.get_kept_trees
    get_kept_trees_indexed = Sampler.get_kept_trees_indexed
    print_trees = Sampler.print_trees
    free = Sampler.free

    def get_trees(self):
        return self.get_kept_trees()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
