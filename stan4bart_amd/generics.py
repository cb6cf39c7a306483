"""Host-side mirror of the reference's user surface around the hot path: ``stan4bart()`` (fit + package the draws,
reference R/stan4bart.R:299-455 ``package_samples``) and the generics ``extract`` / ``fitted`` / ``predict``
(reference R/generics.R:168-484, 486-508, 608-724).

There is no R in this image, so the formula / model-frame front end is replaced by explicit arrays (see ``fit.py``);
everything after that point — array layouts ``[.., iterations, chains]``, chain combination, how the components are put
together, the user-offset rules, the probit link, prediction from the kept trees — follows the R code.  Posterior
predictive noise (``type = "ppd"``) and random effects of unseen grouping levels are drawn with numpy's generator,
not R's stream: they are post-hoc draws outside the sampler.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from .fit import GroupTerm, chain_seeds, hip_sampler_factory, make_sampler_args, make_z_csr, qr_back_transform
from .rcompat import RRng

INT_MAX = 2147483647
EXTRACT_TYPES = ("ev", "ppd", "fixef", "indiv.fixef", "ranef", "indiv.ranef", "indiv.bart", "sigma", "Sigma", "k", "varcount",
                 "stan", "trees", "callback")


def combine_chains_f(x: np.ndarray) -> np.ndarray:
    """reference R/generics.R:1-16: the last two dims (iterations, chains) become one, chain 1's draws first."""
    x = np.asarray(x)
    if x.ndim > 2:
        return np.swapaxes(x, -1, -2).reshape(x.shape[:-2] + (x.shape[-2] * x.shape[-1],))
    if x.ndim == 2:
        return x.T.reshape(-1)
    return x


def _pnorm(x):
    return 0.5 * (1.0 + np.vectorize(math.erf)(np.asarray(x) / math.sqrt(2.0)))


def _theta_to_sigma(theta: np.ndarray, p: int) -> np.ndarray:
    """lme4 mkVarCorr(sc = 1, ...) for one term (reference R/generics.R:248): theta holds the lower triangle of the
    covariance factor column by column, Sigma = L L'."""
    L = np.zeros((p, p))
    k = 0
    for c in range(p):
        for r in range(c, p):
            L[r, c] = theta[k]
            k += 1
    return L @ L.T


@dataclass
class Stan4bartFit:
    family: str
    par_names: list
    stan: np.ndarray            # [num_pars, iterations, chains]
    bart_train: np.ndarray      # [n, iterations, chains]
    bart_test: Optional[np.ndarray]
    bart_varcount: np.ndarray   # [P, iterations, chains]
    warmup: Optional[dict]      # same four arrays for the warmup phase
    X: np.ndarray               # fixed-effect design as given (not centred), [n, K]
    X_means: np.ndarray
    X_test: Optional[np.ndarray]
    terms: list                 # grouping terms in Z column order
    terms_test: Optional[list]
    offset: Optional[np.ndarray]
    offset_test: Optional[np.ndarray]
    offset_type: str
    range_bart: np.ndarray      # [2, chains] (min, max) of the BART response scale
    samplers: list = field(default_factory=list)   # live samplers holding the kept trees (bart_args keepTrees)
    trees: Optional[list] = None
    callback: Optional[np.ndarray] = None   # [len(result), iterations, chains]
    weights: Optional[np.ndarray] = None    # observation weights of the training sample
    k: Optional[np.ndarray] = None          # [iterations, chains] draws of a modeled end-node sensitivity k (bart_args k = chi(...)), else None

    # ------------------------------------------------------------------ helpers
    def _get(self, name: str, include_warmup, only_warmup):
        """reference get_samples (R/generics.R:134-166)."""
        cur = getattr(self, name)
        if not include_warmup:
            return cur
        if self.warmup is None or self.warmup.get(name) is None:
            raise ValueError("model was fit without warmup draws kept")
        w = self.warmup[name]
        return w if only_warmup else np.concatenate([w, cur], axis=-2)

    def _rows(self, prefix: str):
        return [i for i, nm in enumerate(self.par_names) if nm.startswith(prefix)]

    @property
    def n_chains(self) -> int:
        return self.bart_train.shape[2]

    def _ranef_arrays(self, include_warmup, only_warmup):
        stan = self._get("stan", include_warmup, only_warmup)
        rows = self._rows("b.")
        b = stan[rows]
        out, base = {}, 0
        for g in self.terms:
            blk = b[base: base + g.p * g.l]                    # level-major, the p coefficients adjacent
            out[g.name] = blk.reshape(g.l, g.p, *blk.shape[1:]).transpose(1, 0, 2, 3)   # [predictor, group, iter, chain]
            base += g.p * g.l
        return out

    def _sigma_arrays(self, include_warmup, only_warmup):
        stan = self._get("stan", include_warmup, only_warmup)
        th = stan[self._rows("theta_L.")]
        out, base = {}, 0
        for g in self.terms:
            k = g.p * (g.p + 1) // 2
            blk = th[base: base + k]
            S = np.zeros((g.p, g.p) + blk.shape[1:])
            for i in range(blk.shape[1]):
                for c in range(blk.shape[2]):
                    S[:, :, i, c] = _theta_to_sigma(blk[:, i, c], g.p)
            out[g.name] = S
            base += k
        return out

    def _fitted_fixed(self, x, include_warmup, only_warmup):
        """reference fitted_fixed (R/generics.R:510-551): x beta - sum(beta * X_means)."""
        fixef = self._get("stan", include_warmup, only_warmup)[self._rows("beta.")]      # [K, iter, chain]
        xc = np.asarray(x, dtype=np.float64).reshape(-1, fixef.shape[0]) - self.X_means
        return np.einsum("nk,ksc->nsc", xc, fixef)

    def _fitted_random(self, terms_new, include_warmup, only_warmup, sample_new_levels, rng):
        """reference fitted_random (R/generics.R:553-606) + levelfun (R/lme4_functions.R:1337-1378)."""
        re = self._ranef_arrays(include_warmup, only_warmup)
        names = {g.name: g for g in self.terms}
        n = len(terms_new[0].levels)
        any_arr = next(iter(re.values()))
        out = np.zeros((n,) + any_arr.shape[2:])
        Sig = None
        for g in terms_new:
            if g.name not in names:
                raise ValueError("grouping factors specified that were not present in original model")
            old = names[g.name]
            if g.p != old.p:
                raise ValueError("random effects specified that were not present in original model")
            b = re[g.name]                                       # [p, l, iter, chain]
            lev = np.asarray(g.levels, dtype=np.int64)
            n_new = int(max(0, lev.max() - old.l))
            if n_new:
                ext = np.zeros((b.shape[0], n_new) + b.shape[2:])
                if sample_new_levels:
                    Sig = Sig or self._sigma_arrays(include_warmup, only_warmup)
                    S = Sig[g.name]
                    for i in range(b.shape[2]):
                        for c in range(b.shape[3]):
                            L = np.linalg.cholesky(S[:, :, i, c] + 1e-300 * np.eye(b.shape[0]))
                            ext[:, :, i, c] = L @ rng.standard_normal((b.shape[0], n_new))
                b = np.concatenate([b, ext], axis=1)
            vals = np.ones((n, g.p))
            if g.p > 1:
                vals[:, 1:] = np.asarray(g.slopes, dtype=np.float64).reshape(n, g.p - 1)
            out += np.einsum("np,pnsc->nsc", vals, b[:, lev - 1])
        return out

    # ------------------------------------------------------------------ extract
    def extract(self, type: str = "ev", sample: str = "train", combine_chains: bool = True, sample_new_levels: bool = True,
                include_warmup=False, seed: Optional[int] = None, **kw):
        if type not in EXTRACT_TYPES:
            raise ValueError(f"'type' must be one of {EXTRACT_TYPES}")
        if type == "trees":
            # reference R/generics.R:186-193: the kept draws' trees (bart_args keepTrees), optionally one draw / some trees;
            # columns chain, sample, tree, n, var (-1 = leaf), value (cut point | leaf mean on the BART scale)
            if not self.samplers:
                raise ValueError("extracting trees requires stan4bart to be called with `bart_args = {'keepTrees': True}`")
            sample_nums, tree_nums, chain_nums = kw.get("sampleNums"), kw.get("treeNums"), kw.get("chainNums")
            cols = {k: [] for k in ("chain", "sample", "tree", "n", "var", "split", "value")}
            for c, smp in enumerate(self.samplers):
                if chain_nums is not None and c not in np.atleast_1d(chain_nums):
                    continue
                # the index vectors go to the C boundary as they are (stan4bart_getTrees takes them: reference src/init.cpp:514-671)
                tr = smp.get_kept_trees_indexed(None if sample_nums is None else np.atleast_1d(sample_nums),
                                                None if tree_nums is None else np.atleast_1d(tree_nums))
                cols["chain"].append(np.full(len(tr["tree"]), c, dtype=np.int32))
                for k in ("sample", "tree", "n", "var", "split", "value"):
                    cols[k].append(tr[k])
            return {k: np.concatenate(v) if v else np.zeros(0) for k, v in cols.items()}
        if sample not in ("train", "test"):
            raise ValueError("'sample' must be 'train' or 'test'")
        if isinstance(include_warmup, str):
            if include_warmup != "only":
                raise ValueError("'include_warmup' must be logical or \"only\"")
            include_warmup, only_warmup = True, True
        else:
            include_warmup, only_warmup = bool(include_warmup), False
        done = (lambda r: ({k: combine_chains_f(v) for k, v in r.items()} if isinstance(r, dict) else combine_chains_f(r))
                if combine_chains else r)
        if type == "callback":
            if self.callback is None:
                raise ValueError("cannot extract callback samples for model fit without callback function")
            return done(self._get("callback", include_warmup, only_warmup))
        is_bernoulli = self.family == "binomial"
        if type == "sigma" and is_bernoulli:
            raise ValueError("cannot extract 'sigma': binary outcome model does not have a residual standard error parameter")
        n_fixef, n_terms = len(self._rows("beta.")), len(self.terms)
        if type == "fixef":
            if not n_fixef:
                raise ValueError("cannot extract fixef for model with no unmodeled parameters")
            return done(self._get("stan", include_warmup, only_warmup)[self._rows("beta.")])
        if type == "ranef":
            if not n_terms:
                raise ValueError("cannot extract ranef for model with no modeled parameters")
            return done(self._ranef_arrays(include_warmup, only_warmup))
        if type == "Sigma":
            if not n_terms:
                raise ValueError("cannot extract Sigma for model with no modeled parameters")
            return done(self._sigma_arrays(include_warmup, only_warmup))
        if type == "sigma":
            return done(self._get("stan", include_warmup, only_warmup)[self.par_names.index("aux.1")])
        if type == "k":      # reference R/generics.R:223-224, 280-284
            if self.k is None:
                raise ValueError("cannot extract 'k': model was not fit with end-node sensitivity as a modeled parameter")
            return done(self._get("k", include_warmup, only_warmup))
        if type == "varcount":
            return done(self._get("bart_varcount", include_warmup, only_warmup))
        if type == "stan":
            return done(self._get("stan", include_warmup, only_warmup))

        rng = np.random.default_rng(seed)
        if sample == "train":
            X, terms, offset = self.X, self.terms, self.offset
        else:
            if self.bart_test is None:
                raise ValueError("model was fit without test data")
            X, terms, offset = self.X_test, self.terms_test, self.offset_test
        ot = self.offset_type
        fix = ran = bart = 0.0
        if type in ("ev", "ppd", "indiv.fixef") and n_fixef:
            if offset is not None and ot in ("fixef", "parametric") and type != "indiv.fixef":
                fix = np.asarray(offset)[:, None, None]
            else:
                fix = self._fitted_fixed(X, include_warmup, only_warmup)
        if type in ("ev", "ppd", "indiv.ranef") and n_terms:
            if offset is not None and ot in ("ranef", "parametric") and type != "indiv.ranef":
                ran = 0.0 if ot == "parametric" else np.asarray(offset)[:, None, None]
            else:
                ran = self._fitted_random(terms, include_warmup, only_warmup, sample_new_levels, rng)
        if type in ("ev", "ppd", "indiv.bart"):
            if offset is not None and ot == "bart" and type != "indiv.bart":
                bart = np.asarray(offset)[:, None, None]
            else:
                bart = self._get("bart_train" if sample == "train" else "bart_test", include_warmup, only_warmup)
        result = {"ev": lambda: bart + fix + ran, "ppd": lambda: bart + fix + ran, "indiv.fixef": lambda: fix,
                  "indiv.ranef": lambda: ran, "indiv.bart": lambda: bart}[type]()
        if np.isscalar(result):
            raise ValueError(f"model has no component for type '{type}'")
        if type in ("ev", "ppd") and offset is not None and ot == "default":
            result = result + np.asarray(offset)[:, None, None]
        if type in ("ev", "ppd") and is_bernoulli:
            result = _pnorm(result)
        if type == "ppd":
            if is_bernoulli:
                result = (rng.random(result.shape) < result).astype(np.float64)
            else:
                sig = self._get("stan", include_warmup, only_warmup)[self.par_names.index("aux.1")]   # [iter, chain]
                sd = sig[None]
                if sample == "train" and self.weights is not None:     # reference R/generics.R:452-459: sd = sigma sqrt(1 / w)
                    sd = sd * np.sqrt(1.0 / self.weights)[:, None, None]
                result = result + rng.standard_normal(result.shape) * sd
        return done(result)

    # ------------------------------------------------------------------ fitted
    def fitted(self, type: str = "ev", sample: str = "train", sample_new_levels: bool = True, seed: Optional[int] = None):
        """reference fitted.stan4bartFit (R/generics.R:486-508): posterior mean over draws and chains."""
        s = self.extract(type, sample, combine_chains=True, sample_new_levels=sample_new_levels, seed=seed)
        avg = lambda x: np.mean(x, axis=-1)
        return {k: avg(v) for k, v in s.items()} if isinstance(s, dict) else avg(s)

    # ------------------------------------------------------------------ predict
    def predict(self, x_bart=None, X=None, groups: Optional[Sequence[GroupTerm]] = None, offset=None, type: str = "ev",
                combine_chains: bool = True, sample_new_levels: bool = True, seed: Optional[int] = None):
        """reference predict.stan4bartFit (R/generics.R:608-724); BART part from the kept trees
        (``stan4bart_predictBART``, reference src/init.cpp:354-403) — needs bart_args keepTrees."""
        if type not in ("ev", "ppd", "indiv.fixef", "indiv.ranef", "indiv.bart"):
            raise ValueError("'type' must be one of ev, ppd, indiv.fixef, indiv.ranef, indiv.bart")
        if x_bart is None and X is None and groups is None:
            return self.extract(type, combine_chains=combine_chains)
        rng = np.random.default_rng(seed)
        is_bernoulli = self.family == "binomial"
        n_fixef, n_terms = len(self._rows("beta.")), len(self.terms)
        fix = ran = bart = 0.0
        if type in ("ev", "ppd", "indiv.fixef"):
            if X is not None and n_fixef:
                fix = self._fitted_fixed(X, False, False)
            elif type == "indiv.fixef":
                raise ValueError("predict called with type 'indiv.fixef', but model does not include fixed effect terms")
        if type in ("ev", "ppd", "indiv.bart"):
            if not self.samplers:
                raise ValueError("predict for bart components requires 'bart_args' to contain 'keepTrees' as True")
            bart = np.stack([s.predict_bart(np.asarray(x_bart, dtype=np.float64)) for s in self.samplers], axis=2)
        if type in ("ev", "ppd", "indiv.ranef"):
            if groups is not None and len(groups) and n_terms:
                order = {g.name: g for g in groups}
                ran = self._fitted_random([order[g.name] for g in self.terms if g.name in order], False, False, sample_new_levels, rng)
            elif type == "indiv.ranef":
                raise ValueError("predict called with type 'indiv.ranef', but model does not include random effect terms")
        off = 0.0 if offset is None else np.asarray(offset, dtype=np.float64)[:, None, None]
        result = {"ev": lambda: bart + fix + ran + off, "ppd": lambda: bart + fix + ran + off, "indiv.fixef": lambda: fix,
                  "indiv.ranef": lambda: ran, "indiv.bart": lambda: bart}[type]()
        if type in ("ev", "ppd") and is_bernoulli:
            result = _pnorm(result)
        if type == "ppd":
            if is_bernoulli:
                result = (rng.random(result.shape) < result).astype(np.float64)
            else:
                sig = self.stan[self.par_names.index("aux.1")]
                result = result + rng.standard_normal(result.shape) * sig[None]
        return combine_chains_f(result) if combine_chains else result

    def export_bart_states(self) -> list:
        """``stan4bart_exportBARTState`` per chain (reference R/stan4bart_fit.R:572-580): byte strings that
        ``attach_stored_samplers`` turns back into predict-capable samplers, in this or another process."""
        if not self.samplers:
            raise ValueError("exporting the BART state requires bart_args keepTrees")
        return [s.export_bart_state() for s in self.samplers]

    def attach_stored_samplers(self, states: Sequence[bytes], lib=None, prefix: str = "s4b_", device: int = 0):
        """``stan4bart_createStoredBARTSampler`` for every chain: afterwards ``predict`` works without the fitting samplers."""
        from .abi import StoredSampler
        if lib is None:
            from ._lib import load_library
            lib = load_library()
        self.close()
        self.samplers = [StoredSampler(lib, prefix, st, device) for st in states]

    def close(self):
        for s in self.samplers:
            s.free()
        self.samplers = []


def _stack(chain_results, phase, key, sub=None):
    arrs = []
    for r in chain_results:
        a = r[phase][key] if sub is None else r[phase][key][sub]
        arrs.append(np.asarray(a))
    return np.stack(arrs, axis=-1)


def stan4bart(y, x_bart, X=None, groups: Sequence[GroupTerm] = (), x_bart_test=None, X_test=None,
              groups_test: Optional[Sequence[GroupTerm]] = None, offset=None, offset_test=None, offset_type: str = "default",
              family: str = "gaussian", chains: int = 4, seed: Optional[int] = None, iter: int = 2000, warmup: int = 1000,
              keep_warmup: bool = True, make_sampler: Optional[Callable] = None, treatment=None, callback: Optional[Callable] = None,
              cores: int = 1, **kw) -> Stan4bartFit:
    """The reference's ``stan4bart()`` after its formula front end (R/stan4bart.R:1-297 -> arrays): runs ``chains``
    chains with the single-threaded seeding rule (R/stan4bart_fit.R:545-554) and packages the draws
    (``package_samples``, R/stan4bart.R:299-455).  With ``bart_args = {"keepTrees": True}`` the samplers stay alive
    inside the fit (``object$sampler.bart``) so that ``predict`` can use the kept trees; call ``close()`` when done.

    ``treatment = ("X" | "x_bart", column)`` names a binary treatment column: the test sample becomes the training rows
    with the treatment flipped, i.e. the counterfactual (reference R/stan4bart.R:93-120, tests/testthat/test-10-treatment.R).
    ``callback(yhat_train, yhat_test, stan_pars, par_names)`` is evaluated after every iteration inside the sampler
    (reference src/init.cpp:849-911, tests/testthat/test-11-callback.R); its results come back as ``extract("callback")``.

    ``cores > 1`` runs the chains concurrently with the reference's parallel seeding rule (R/stan4bart_fit.R:515-533: worker c
    does ``set.seed(sample.int(.Machine$integer.max, chains)[c])``), here as host threads sharing the device: one chain leaves
    most of an MI355X idle at n = 1e6, four interleaved chains finish about twice as fast as four in a row."""
    make_sampler = make_sampler or hip_sampler_factory()
    if treatment is not None:
        if x_bart_test is not None or X_test is not None:
            raise ValueError("'treatment' builds the test sample itself: do not pass test data as well")
        where, col = treatment
        if where not in ("X", "x_bart"):
            raise ValueError("treatment must be ('X', column) or ('x_bart', column)")
        src = np.asarray(X if where == "X" else x_bart, dtype=np.float64)
        vals = np.unique(src[:, col])
        if len(vals) != 2:
            raise ValueError("treatment must be binary")
        flipped = src.copy()
        flipped[:, col] = np.where(src[:, col] == vals[0], vals[1], vals[0])
        x_bart_test = flipped if where == "x_bart" else np.asarray(x_bart, dtype=np.float64)
        X_test = flipped if where == "X" else (None if X is None else np.asarray(X, dtype=np.float64))
        groups_test = list(groups) if len(groups) else None
        offset_test = offset
    names_box: list = []
    cb = None
    if callback is not None:
        cb = lambda tr, te, sp: np.asarray(callback(tr, te, sp, names_box[0]), dtype=np.float64)
    import os
    rng = RRng(seed if seed is not None else int.from_bytes(os.urandom(4), "little") % INT_MAX)
    keep_trees = bool((kw.get("bart_args") or {}).get("keepTrees", False))
    results, samplers = [None] * chains, [None] * chains
    args_box = [None]

    def one_chain(c, chain_rng, sharing=1):
        args = make_sampler_args(y, x_bart, X=X, groups=groups, x_test=x_bart_test, family=family, iter=iter, warmup=warmup,
                                 offset=offset, offset_type=offset_type, keep_fits=True, callback=cb, **kw)
        args_box[0] = args
        args.seed = int(chain_rng.sample_int(INT_MAX, 1)[0])
        s = make_sampler(args, chain_rng.state)
        r = {}
        try:
            if sharing > 1 and hasattr(s, "set_device_sharing"):
                s.set_device_sharing(sharing)                 # host threads that share the GPU while this chain runs (its batch)
            names_box[:] = [s.stan_par_names()]
            if warmup > 0:
                r["warmup"] = s.run(warmup, True, 0)
            s.disengage_adaptation()
            r["sample"] = s.run(iter - warmup, False, 0)
            r["par_names"] = s.stan_par_names()
            qr_back_transform(args, r)       # stan_args = {"QR": True}: beta rows back to the design's scale (R/stan4bart_fit.R:560-570)
            r["range.bart"] = s.get_bart_data_range()
            chain_rng.state = s.get_r_rng_state()
        except Exception:
            s.free()
            raise
        if keep_trees:
            samplers[c] = s
        else:
            s.free()
        results[c] = r

    if cores <= 1 or chains == 1:
        try:
            for c in range(chains):        # one stream continued from chain to chain (R/stan4bart_fit.R:545-554)
                one_chain(c, rng)
        except Exception:                  # a later chain failed: the kept samplers of the earlier ones hold device memory
            for sm in samplers:
                if sm is not None:
                    sm.free()
            raise
    else:
        import threading
        seeds = chain_seeds(int(rng.sample_int(INT_MAX, 1)[0]) if seed is None else seed, chains)
        errors = []

        def guarded(c, sharing):
            try:
                one_chain(c, RRng(int(seeds[c])), sharing)
            except Exception as e:   # surfaced after the join
                errors.append(e)
        pending = list(range(chains))
        while pending:
            batch, pending = pending[:cores], pending[cores:]
            th = [threading.Thread(target=guarded, args=(c, len(batch))) for c in batch]
            [t.start() for t in th]
            [t.join() for t in th]
        if errors:
            for sm in samplers:
                if sm is not None:
                    sm.free()
            raise errors[0]
    samplers = [sm for sm in samplers if sm is not None]
    args = args_box[0]

    def pack(phase):
        return dict(stan=_stack(results, phase, "stan"), bart_train=_stack(results, phase, "bart", "train"),
                    bart_test=_stack(results, phase, "bart", "test") if x_bart_test is not None else None,
                    bart_varcount=_stack(results, phase, "bart", "varcount"),
                    k=_stack(results, phase, "bart", "k") if "k" in results[0][phase]["bart"] else None,     # (R/stan4bart.R:389-399)
                    callback=(np.stack([np.stack(r[phase]["callback"], axis=1) for r in results], axis=2)
                              if callback is not None else None))
    smp = pack("sample")
    n = len(y)
    Xa = np.zeros((n, 0)) if X is None else np.asarray(X, dtype=np.float64).reshape(n, -1)
    terms = args.extras["group_terms"]
    terms_test = None
    if groups_test is not None and len(groups_test):
        by = {g.name: g for g in groups_test}
        terms_test = [by[g.name] for g in terms]
    return Stan4bartFit(
        family=family, par_names=results[0]["par_names"], stan=smp["stan"], bart_train=smp["bart_train"], bart_test=smp["bart_test"],
        bart_varcount=smp["bart_varcount"], warmup=pack("warmup") if (warmup > 0 and keep_warmup) else None,
        X=Xa, X_means=np.asarray(args.extras["xbar"]), X_test=None if X_test is None else np.asarray(X_test, dtype=np.float64),
        terms=terms, terms_test=terms_test, offset=None if offset is None else np.asarray(offset, dtype=np.float64),
        offset_test=None if offset_test is None else np.asarray(offset_test, dtype=np.float64), offset_type=offset_type,
        range_bart=np.stack([r["range.bart"] for r in results], axis=1), samplers=samplers, callback=smp["callback"],
        weights=None if kw.get("weights") is None else np.asarray(kw["weights"], dtype=np.float64), k=smp["k"])
