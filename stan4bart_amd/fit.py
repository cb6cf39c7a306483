"""Host-side mirror of the reference's R fit driver for the hot path.

Follows reference R/stan4bart_fit.R: the six argument objects handed to ``stan4bart_create``
(`:259-365` data.stan, `:482-488` control.stan, `:490-493` control.common, `:436-479` dbarts
control/model), the per-chain worker (`:33-60`) and the chain fan-out (`:495-558`).  There is no R
in this image, so the model-frame plumbing of R/stan4bart.R / R/lme4_functions.R is replaced by
explicit arrays: ``y``, the BART predictor matrix, the fixed-effect design and a list of grouping
terms.  Everything numeric that reaches the C boundary is computed as the R code computes it.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np

from .abi import Sampler, SamplerArgs
from .rcompat import RRng

INT_MAX = 2147483647


@dataclass
class GroupTerm:
    """One ``(1 + slopes | g)`` term: ``levels`` are 1-based integer codes, ``slopes`` n x (p-1)."""
    levels: np.ndarray
    slopes: Optional[np.ndarray] = None
    name: str = "g"

    @property
    def p(self) -> int:
        return 1 + (0 if self.slopes is None else np.atleast_2d(self.slopes.T).shape[0])

    @property
    def l(self) -> int:
        return int(np.max(self.levels))


def center_x(X: np.ndarray):
    """reference R/rstanarm_functions.R:420-446 (no intercept column: BART supplies it)."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    xbar = X.mean(axis=0) if X.shape[1] else np.zeros(0)
    return X - xbar, xbar


def make_z_csr(groups: Sequence[GroupTerm], n: int):
    """CSR parts (w, v, u) of Z = t(Zt) (reference R/stan4bart_fit.R:290,313-317; column order =
    lme4 Zt row order: terms by decreasing number of levels, level-major within a term)."""
    order = sorted(range(len(groups)), key=lambda i: -groups[i].l)
    terms = [groups[i] for i in order]
    p = [g.p for g in terms]
    l = [g.l for g in terms]
    z = int(sum(p))
    w = np.zeros((n, z))
    v = np.zeros((n, z), dtype=np.int32)
    base, col = 0, 0
    for g, pi, li in zip(terms, p, l):
        lev = np.asarray(g.levels, dtype=np.int64) - 1
        vals = np.ones((n, pi))
        if pi > 1:
            vals[:, 1:] = np.asarray(g.slopes, dtype=np.float64).reshape(n, pi - 1)
        for c in range(pi):
            w[:, col + c] = vals[:, c]
            v[:, col + c] = base + lev * pi + c
        base += pi * li
        col += pi
    u = (np.arange(n + 1, dtype=np.int64) * z).astype(np.int32)
    return terms, p, l, w.reshape(-1), v.reshape(-1), u, base


def init_fit(y, Xc, groups, n, is_binary):
    """bart_offset_init / sigma_init from the lm / glm fallback of reference R/stan4bart.R:156-178 (``subbars``: the grouping
    factors enter as fixed dummies; lme4 is not available here).  The design [1 | Xc | level dummies] is never formed
    densely: it has 1 + K dense columns and one non-zero per grouping factor and row, so the normal equations are
    assembled from a sparse matrix (m x m with m = 1 + K + sum(levels - 1)) and the rank comes from the small Gram matrix.
    Gaussian: least squares, sigma = sqrt(RSS / (n - rank)) (``sigma(lm)``).  Binomial: probit IRLS (``glm(family =
    binomial("probit"))``); the reference takes ``fitted(init_fit, type = "link")`` (R/stan4bart.R:172), and ``fitted()``
    ignores ``type`` for a glm, so what it hands to BART is the fitted mean Phi(eta) — reproduced here."""
    y = np.asarray(y, dtype=np.float64)
    K = Xc.shape[1]
    D = np.column_stack([np.ones(n), Xc]) if K else np.ones((n, 1))          # dense part: intercept + fixed effects
    facs = []                                                                # per grouping factor: 0-based dummy column of every row, -1 for the reference level
    for g in groups:
        if g.l < 2:
            continue
        lev = np.asarray(g.levels, dtype=np.int64)
        facs.append((np.where(lev >= 2, lev - 2, -1), g.l - 1))
    m = D.shape[1] + sum(w for _, w in facs)

    def apply(coef):                                                         # A @ coef without forming A
        out = D @ coef[:D.shape[1]]
        o = D.shape[1]
        for col, w in facs:
            c = np.concatenate([coef[o:o + w], [0.0]])                       # (index -1: the reference level adds nothing)
            out = out + c[col]
            o += w
        return out

    def wls(wt, z):
        # normal equations of [D | dummies] from per-level sums (numpy only): the Gram matrix is m x m, never n x m
        Dw = D if wt is None else D * wt[:, None]
        G = np.zeros((m, m)); rhs = np.zeros(m)
        d = D.shape[1]
        G[:d, :d] = Dw.T @ D
        rhs[:d] = Dw.T @ z
        ones = np.ones(n) if wt is None else wt
        offs, o = [], d
        for col, w in facs:
            offs.append(o); o += w
        for (col, w), o in zip(facs, offs):
            in_ = col >= 0
            for j in range(d):
                G[j, o:o + w] = np.bincount(col[in_], weights=Dw[in_, j], minlength=w)
            G[o:o + w, :d] = G[:d, o:o + w].T
            rhs[o:o + w] = np.bincount(col[in_], weights=(ones * z)[in_], minlength=w)
            for (col2, w2), o2 in zip(facs, offs):
                both = in_ & (col2 >= 0)
                blk = np.zeros((w, w2))
                np.add.at(blk, (col[both], col2[both]), ones[both])
                G[o:o + w, o2:o2 + w2] = blk
        coef, _, rank, _ = np.linalg.lstsq(G, rhs, rcond=1e-11)
        return coef, int(rank)

    class _A:                                                                # (`A @ coef` below)
        def __matmul__(self, coef):
            return apply(coef)
    A = _A()

    if not is_binary:
        coef, rank = wls(None, y)
        fitted = A @ coef
        sigma = float(np.sqrt(np.sum((y - fitted) ** 2) / max(1, n - rank)))
        return fitted, sigma
    # probit IRLS, glm.fit's scheme: mustart = (y + 0.5) / 2, eta = qnorm(mu), deviance convergence (epsilon = 1e-8, maxit 25)
    try:
        from scipy.special import ndtr                                       # (only the probit path needs scipy: the normal cdf)
    except ImportError as e:                                                 # pragma: no cover
        raise ImportError("family = 'binomial' needs scipy (scipy.special.ndtr) for the probit starting values") from e
    from .rcompat import qnorm
    mu = (y + 0.5) / 2.0
    eta = qnorm(mu)
    dev_old = np.inf
    for _ in range(25):
        dmu = np.exp(-0.5 * eta * eta) / np.sqrt(2.0 * np.pi)
        var = mu * (1.0 - mu)
        # glm.fit: `good <- weights > 0 & mu.eta.val != 0` — observations whose fitted probability has run to 0 or 1 (separation)
        # carry no weight in this step instead of dividing by zero
        good = (dmu > 0.0) & (var > 0.0)
        safe = np.where(good, dmu, 1.0)
        wt = np.where(good, safe * safe / np.where(good, var, 1.0), 0.0)
        z = np.where(good, eta + (y - mu) / safe, eta)
        coef, _ = wls(wt, z)
        eta = A @ coef
        mu = np.clip(ndtr(eta), 1e-15, 1.0 - 1e-15)
        dev = -2.0 * float(np.sum(np.where(y > 0, np.log(mu), np.log1p(-mu))))
        if abs(dev - dev_old) / (abs(dev) + 0.1) < 1e-8:
            break
        dev_old = dev
    return mu, 1.0


def _k_prior(k):
    """bart_args$k (reference R/stan4bart_fit.R:460-465 puts it into dbarts' ``normal(k = ...)``): a number — the fixed k — or a
    hyperprior the way dbarts writes it, ``chi(degreesOfFreedom = 1.25, scale = Inf)`` (reference R/stan4bart.R:202,
    tests/testthat/test-09-bartArgs.R:32): the string of the R call, ``{"chi": (df, scale)}`` / ``{"chi": {...}}``, or ``("chi", df, scale)``.
    Returns (k where the chain starts, None | (df, scale)); a modeled k starts from dbarts' default 2."""
    if isinstance(k, (int, float, np.integer, np.floating)):
        return float(k), None
    df, scale = 1.25, float("inf")          # dbarts' defaults of chi()

    def num(x):
        x = str(x).strip()
        return float("inf") if x in ("Inf", "inf", "+Inf") else float(x)
    if isinstance(k, str):
        txt = k.replace(" ", "")
        if not (txt.startswith("chi(") and txt.endswith(")")) and txt != "chi":
            raise ValueError(f"bart_args['k']: {k!r} is neither a number nor a chi(degreesOfFreedom, scale) hyperprior")
        pos = 0
        for a in [x for x in txt[4:-1].split(",") if x] if txt != "chi" else []:
            if "=" in a:
                nm, v = a.split("=", 1)
                if nm in ("degreesOfFreedom", "df"):
                    df = num(v)
                elif nm == "scale":
                    scale = num(v)
                else:
                    raise ValueError(f"bart_args['k']: unknown argument {nm!r} of chi()")
            else:
                if pos == 0:
                    df = num(a)
                else:
                    scale = num(a)
                pos += 1
    elif isinstance(k, dict) and set(k) == {"chi"}:
        v = k["chi"]
        if isinstance(v, dict):
            df, scale = float(v.get("degreesOfFreedom", v.get("df", df))), float(v.get("scale", scale))
        else:
            v = tuple(v)
            df, scale = (float(v[0]) if len(v) > 0 else df), (float(v[1]) if len(v) > 1 else scale)
    elif isinstance(k, (tuple, list)) and len(k) >= 1 and k[0] == "chi":
        df, scale = (float(k[1]) if len(k) > 1 else df), (float(k[2]) if len(k) > 2 else scale)
    else:
        raise ValueError(f"bart_args['k']: {k!r} is neither a number nor a chi(degreesOfFreedom, scale) hyperprior")
    if not (df > 0 and np.isfinite(df)) or not scale > 0:
        raise ValueError("chi(degreesOfFreedom, scale): both must be positive (scale may be Inf)")
    return 2.0, (df, scale)


def _split_probs(sp, p: int, names=None):
    """cgm(split.probs = ) the way dbarts takes it (reference tests/testthat/test-09-bartArgs.R:20: c(X3 = 2, .default = 1)): a
    sequence of p weights, or a mapping predictor (0-based column index, or its name when bart_args['predictor.names'] is given)
    -> weight with an optional '.default' for the rest (1 otherwise).  Returns p positive weights summing to one, or None."""
    if sp is None:
        return None
    if isinstance(sp, dict):
        out = np.full(p, float(sp.get(".default", 1.0)))
        for key, w in sp.items():
            if key == ".default":
                continue
            j = list(names).index(key) if (names is not None and not isinstance(key, (int, np.integer))) else int(key)
            if not 0 <= j < p:
                raise ValueError(f"split.probs: no predictor {key!r}")
            out[j] = float(w)
    else:
        out = np.asarray(sp, dtype=np.float64)
        if out.shape != (p,):
            raise ValueError("split.probs needs one weight per BART predictor (or a mapping with '.default')")
    if not np.all(np.isfinite(out)) or np.any(out <= 0):
        raise ValueError("split.probs must be positive and finite")
    return out / out.sum()          # (dbarts hands its sampler the normalised weights)


def make_sampler_args(y, x_bart, X=None, groups: Sequence[GroupTerm] = (), x_test=None, family: str = "gaussian",
                      iter: int = 2000, warmup: int = 1000, skip=1, keep_fits: bool = True, callback=None,
                      offset=None, offset_type: str = "default", weights=None,
                      stan_args: Optional[dict] = None, bart_args: Optional[dict] = None, verbose: int = 0,
                      refresh: Optional[int] = None, device: int = 0) -> SamplerArgs:
    stan_args = dict(stan_args or {})
    bart_args = dict(bart_args or {})
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    is_binary = family == "binomial"
    Xc, xbar = center_x(X if X is not None else np.zeros((n, 0)))
    K = Xc.shape[1]

    # priors (reference R/stan4bart_fit.R:100-232 + rstanarm handle_glm_prior): default normal(0, 2.5, autoscale) for
    # the coefficients, exponential(1, autoscale) for sigma.  stan_args["prior"] = dict(dist=..., ...) selects another
    # family with rstanarm's defaults: student_t(df=1) / hs(df=1, global_df=1, global_scale=0.01, slab_df=4,
    # slab_scale=2.5) / hs_plus(df1=1, df2=1, ...) / laplace / lasso(df=1) / product_normal(df=2, scale=1)
    prior = dict(stan_args.get("prior") or {})
    dist = prior.get("dist", "normal")
    dists = {"none": 0, "normal": 1, "student_t": 2, "hs": 3, "hs_plus": 4, "laplace": 5, "lasso": 6, "product_normal": 7}
    if dist not in dists:
        raise ValueError(f"prior dist must be one of {sorted(dists)}")
    prior_dist = dists[dist]
    autoscale = bool(prior.get("autoscale", dist in ("normal", "student_t", "laplace", "lasso")))
    default_scale = 1.0 if dist == "product_normal" else float(stan_args.get("prior_scale", 2.5))
    prior_scale = np.full(K, float(prior.get("scale", default_scale)))
    prior_mean = np.full(K, float(prior.get("location", 0.0)))
    prior_df = np.full(K, float(prior.get("df", prior.get("df1", 1.0))))
    num_normals = None
    hs_args = dict(global_prior_df=float(prior.get("global_df", 1.0)), global_prior_scale=float(prior.get("global_scale", 0.01)),
                   slab_df=float(prior.get("slab_df", 4.0)), slab_scale=float(prior.get("slab_scale", 2.5)))
    if dist == "hs_plus":
        prior_scale = np.full(K, float(prior.get("df2", 1.0)))      # "unorthodox usage of prior_scale as another df" (continuous.stan:396)
    if dist == "product_normal":
        num_normals = np.full(K, int(prior.get("df", 2)), dtype=np.int32)
        if np.any(num_normals < 2):
            raise ValueError("product_normal needs df >= 2")
    ss = float(np.std(y, ddof=1)) if not is_binary else 1.0
    use_qr = bool(stan_args.get("QR", False))
    if prior_dist > 0 and autoscale:
        if not is_binary:
            prior_scale = ss * prior_scale
        for k in range(K if not use_qr else 0):          # (reference R/stan4bart_fit.R:218: not with QR)
            xs = 1.0 if len(np.unique(Xc[:, k])) == 1 else float(np.std(Xc[:, k], ddof=1))
            prior_scale[k] = max(1e-12, prior_scale[k] / xs)
    # stan_args = list(QR = TRUE) (reference R/stan4bart_fit.R:239-258): the sampler sees Q * scale_factor of the thin QR decomposition of
    # the centred design; the coefficient rows of its draws are mapped back with R_inv in fit_worker (R/stan4bart_fit.R:560-570)
    R_inv = None
    xbar_model = xbar
    if use_qr and K > 0:
        if K <= 1:
            raise ValueError("'QR' can only be specified when there are multiple predictors.")
        Q, Rm = np.linalg.qr(Xc)
        scale_factor = float(np.sqrt(n - 1.0)) if autoscale else float(Rm[K - 1, K - 1])
        R_inv = np.linalg.solve(Rm, np.eye(K)) * scale_factor
        Xc = Q * scale_factor
        xbar_model = xbar @ R_inv
    prior_scale_for_aux = 0.0 if is_binary else ss * 1.0

    terms, p, l, w, v, u, q = make_z_csr(groups, n) if len(groups) else ([], [], [], np.zeros(0), np.zeros(0, np.int32),
                                                                        np.zeros(n + 1, np.int32), 0)
    t = len(p)
    decov = stan_args.get("prior_covariance", dict(regularization=1.0, concentration=1.0, shape=1.0, scale=1.0))
    n_conc = int(sum(pi for pi in p if pi > 1))
    n_reg = int(sum(1 for pi in p if pi > 1))

    boi, sigma_init = init_fit(y, Xc, list(groups), n, is_binary)
    if isinstance(skip, (tuple, list)):
        skip_bart, skip_stan = int(skip[0]), int(skip[1] if len(skip) > 1 else skip[0])
    else:
        skip_bart = skip_stan = int(skip)
    if refresh is None:
        refresh = max(iter // 10, 1)
    off_types = ["default", "fixef", "ranef", "bart", "parametric"]
    return SamplerArgs(
        x_bart=np.asarray(x_bart, dtype=np.float64), x_test=x_test,
        n_trees=int(bart_args.get("n.trees", 75)), n_cuts=bart_args.get("n.cuts", 100), n_thin=skip_bart,
        base=float(bart_args.get("base", 0.95)), power=float(bart_args.get("power", 2.0)),
        k=_k_prior(bart_args.get("k", 2.0))[0], k_hyper=_k_prior(bart_args.get("k", 2.0))[1], keep_trees=bool(bart_args.get("keepTrees", False)),
        split_probs=_split_probs(bart_args.get("split.probs"), np.asarray(x_bart).shape[1], bart_args.get("predictor.names")),
        use_quantiles=bool(bart_args.get("useQuantiles", False)),
        node_scale=3.0 if is_binary else 0.5,
        X=Xc, y=y, weights=weights, is_binary=is_binary, prior_dist=prior_dist, prior_dist_for_aux=0 if is_binary else 3,
        prior_scale=prior_scale, prior_mean=prior_mean, prior_df=prior_df, num_normals=num_normals, **hs_args,
        prior_scale_for_aux=prior_scale_for_aux, prior_mean_for_aux=0.0, prior_df_for_aux=1.0,
        p=p, l=l, shape=[float(decov["shape"])] * t, scale=[float(decov["scale"])] * t,
        concentration=[float(decov["concentration"])] * n_conc, regularization=[float(decov["regularization"])] * n_reg,
        w=w, v=v, u=u,
        skip=skip_stan, init_r=float(stan_args.get("init_r", 2.0)),
        adapt_gamma=float(stan_args.get("adapt_gamma", 0.05)), adapt_delta=float(stan_args.get("adapt_delta", 0.8)),
        adapt_kappa=float(stan_args.get("adapt_kappa", 0.75)), hmc_mode=int(stan_args.get("hmc_mode", 0)),
        warmup=warmup, iter=iter, verbose=verbose, refresh=refresh,
        offset=offset, offset_type=off_types.index(offset_type),
        bart_offset_init=boi, sigma_init=sigma_init, keep_fits=keep_fits, callback=callback, device=device,
        extras=dict(xbar=xbar, xbar_model=xbar_model, R_inv=R_inv, group_terms=terms),
    )


def qr_back_transform(args: SamplerArgs, results: dict) -> None:
    """QR = TRUE: the ``beta.*`` rows of a chain's draws back to the scale of the design, in place (reference
    R/stan4bart_fit.R:560-570 does this for every chain).  Every driver of a chain calls it: ``fit_worker`` and ``generics.stan4bart``."""
    R_inv = (args.extras or {}).get("R_inv")
    if R_inv is None:
        return
    rows = [i for i, nm in enumerate(results["par_names"]) if nm.startswith("beta.")]
    for ph in ("warmup", "sample"):
        if ph in results and rows:
            results[ph]["stan"][rows, :] = R_inv @ results[ph]["stan"][rows, :]


def fit_worker(make_sampler: Callable[[SamplerArgs, np.ndarray], Sampler], args: SamplerArgs, rng: RRng) -> dict:
    """reference stan4bart_fit_worker (R/stan4bart_fit.R:33-60): seed Stan from R's stream, create,
    warmup, disengage adaptation, sample.  ``rng`` is advanced in place (PutRNGstate semantics)."""
    args.seed = int(rng.sample_int(INT_MAX, 1)[0])
    sampler = make_sampler(args, rng.state)
    results: dict = {}
    try:
        if args.verbose > 0:
            sampler.print_initial_summary()
        if args.warmup > 0:
            results["warmup"] = sampler.run(args.warmup, True, 0)
        sampler.disengage_adaptation()
        results["sample"] = sampler.run(args.iter - args.warmup, False, 0)
        results["par_names"] = sampler.stan_par_names()
        qr_back_transform(args, results)
        results["range.bart"] = sampler.get_bart_data_range()
        if args.keep_trees:
            results["trees"] = sampler.get_trees()
            results["bart_state"] = sampler.export_bart_state()   # exportBARTState: survives the sampler (R/stan4bart_fit.R:572-580)
        rng.state = sampler.get_r_rng_state()
    finally:
        sampler.free()
    return results


def hip_sampler_factory():
    from ._lib import load_library
    lib = load_library()
    return lambda a, st: Sampler(lib, "s4b_", a, st)


def stan4bart_fit(y, x_bart, X=None, groups: Sequence[GroupTerm] = (), chains: int = 4, seed: Optional[int] = None,
                  make_sampler: Optional[Callable] = None, **kw) -> list:
    """Serial multi-chain fit with the reference's single-threaded seeding rule
    (R/stan4bart_fit.R:545-554: ``set.seed(seed)`` once, chains continue one stream)."""
    make_sampler = make_sampler or hip_sampler_factory()
    rng = RRng(seed if seed is not None else int.from_bytes(os.urandom(4), "little"))
    out = []
    for c in range(chains):
        args = make_sampler_args(y, x_bart, X=X, groups=groups, **kw)
        out.append(fit_worker(make_sampler, args, rng))
    return out


def chain_seeds(seed: int, chains: int) -> np.ndarray:
    """Per-worker seeds of the parallel path (R/stan4bart_fit.R:515-516):
    ``set.seed(seed); sample.int(.Machine$integer.max, chains)``."""
    return RRng(seed).sample_int(INT_MAX, chains)


def fit_chain_parallel(chain: int, seed: int, chains: int, make_sampler: Callable, y, x_bart, **kw) -> dict:
    """One chain of the parallel path: worker ``chain`` does ``set.seed(randomSeeds[chain])`` first
    (R/stan4bart_fit.R:35-36,527-533).  Called by one process per GPU."""
    rng = RRng(int(chain_seeds(seed, chains)[chain]))
    args = make_sampler_args(y, x_bart, **kw)
    return fit_worker(make_sampler, args, rng)
