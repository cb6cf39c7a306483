"""Seeded random configurations at the sizes where k_sweep fills the device (n = 5e4 ... 1.04e6: up to 255 pass workgroups exchange their
bin partials), BART block only.  Used by tools/fuzz_large.py (campaigns of hundreds of seeds outside the suite) and by
tests/test_gpu_large.py (eight of them in the driver-run suite)."""
import numpy as np


def large_case(seed, n=None, iters=None, deep=None, k_chi=None, trees=None, split_probs=False, weights=False):
    """`n`, `iters` (warm-up, total), `deep` (True: trees of tens of leaves from the prior, i.e. hand-overs to k_step), `k_chi` ((df, scale) of a
    modeled k) and `trees` override what the seed drew; everything else stays the seed's.  `split_probs`: cgm(split.probs = ) with weights drawn from a
    generator of their own (the seed's other draws stay what they are) — the persistent sweep then runs as k_sweep_sp / k_sweep_few_sp.  `weights`: observation weights from a generator of their own, a
    scale between 1e-3 and 1 and a few weights a thousand times smaller than the rest (k_sweep_w); not for binary responses, not together with split_probs."""
    from stan4bart_amd import GroupTerm, make_sampler_args
    g = np.random.default_rng(500000 + seed)
    n0 = int(g.choice([g.integers(50000, 200000), g.integers(200000, 700000), g.integers(700000, 1044000)]))
    n = n0 if n is None else int(n)
    p = int(g.integers(1, 9))
    cols = [(g.random(n) < 0.3).astype(np.float64) if g.random() < 0.25 else (g.normal(size=n) if g.random() < 0.5 else g.random(n)) for _ in range(p)]
    xb = np.column_stack(cols)
    x4 = g.random(n)
    f = 3.0 * np.sin(2.0 * xb[:, 0]) + (xb[:, -1] > np.median(xb[:, -1])) * 2.0 + 1.5 * x4
    binary = bool(g.random() < 0.15) and n < 150000              # (probit latents are serial: keep those cases small)
    yc = f + g.normal(size=n) * g.choice([0.1, 1.0, 3.0])
    y = (yc > np.median(yc)).astype(np.float64) if binary else yc * g.choice([1.0, 1e-3, 250.0])
    bart_args = {"n.trees": int(g.integers(1, 13)), "n.cuts": int(g.choice([1, 5, 100])), "k": float(g.choice([0.5, 2.0, 4.0]))}
    r = g.random()
    if deep is not None:
        r = 0.4 if deep else 0.9
    if r < 0.3:
        bart_args.update(base=0.99, power=0.5)
    elif r < 0.45:
        bart_args.update(base=0.99, power=0.3, k=0.3)             # trees of tens of leaves from the prior: hand-overs
    if g.random() < 0.2:
        bart_args["useQuantiles"] = True
    warmup = int(g.integers(1, 5)); it = warmup + int(g.integers(2, 8))
    if iters is not None:
        warmup, it = iters
    if trees is not None:
        bart_args["n.trees"] = int(trees)
    if k_chi is not None:
        bart_args["k"] = ("chi", float(k_chi[0]), float(k_chi[1]))
    if split_probs:
        bart_args["split.probs"] = [float(w) for w in np.random.default_rng(900000 + seed).choice([0.05, 0.5, 1.0, 1.0, 3.0, 8.0], size=p)]
    groups = [GroupTerm(g.integers(1, 6, size=n), None, "g.1")] if g.random() < 0.4 else []
    w = None
    if weights and not binary and not split_probs:
        gw = np.random.default_rng(700000 + seed)
        # (weights of O(1) and below.  A weight multiplies an observation's precision; the reference — dbarts, restated in oracle/bart_ref.hpp — gives a branch with an
        # EMPTY leaf the log-likelihood -1e7 and compares it with the FULL integrated likelihood of the other branch, within-leaf sum of squares included, while the
        # device code carries the terms that do not cancel between two non-empty branches only: once sum of w d^2 / sigma^2 of a branch exceeds 2e7 — weights of 40 at
        # n = 4e5 — the reference ACCEPTS a rule that leaves a leaf empty and the device code rejects it: seed 79 of the first version of this generator)
        w = gw.uniform(0.25, 4.0, n) * float(gw.choice([1e-3, 0.03, 0.125, 1.0, 1.0]))
        w[gw.integers(0, n, 100)] *= 1e-3
    args = make_sampler_args(y, xb, X=x4[:, None], groups=groups, family="binomial" if binary else "gaussian", iter=it, warmup=warmup, bart_args=bart_args, weights=w,
                             x_test=xb[:50].copy() if g.random() < 0.3 else None)
    if bart_args.get("power") == 0.3:
        args.node_capacity = 1024
    return args, dict(n=n, p=p, binary=binary, **{k: v for k, v in bart_args.items() if k != "split.probs"}, split_probs=bool(split_probs), weights=w is not None)
