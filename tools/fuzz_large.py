#!/usr/bin/env python3
"""Seeded random configurations at the sizes where k_sweep fills the device (n = 5e4 ... 1.04e6: up to 255 pass workgroups exchange their
bin partials), BART block only, against the oracle:  python tools/fuzz_large.py 0 60 [persistent|fused|two-kernel]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import assert_chain_parity, run_chain
from stan4bart_amd import GroupTerm, make_sampler_args
from stan4bart_amd._lib import load_library


def large_case(seed):
    g = np.random.default_rng(500000 + seed)
    n = int(g.choice([g.integers(50000, 200000), g.integers(200000, 700000), g.integers(700000, 1044000)]))
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
    if r < 0.3:
        bart_args.update(base=0.99, power=0.5)
    elif r < 0.45:
        bart_args.update(base=0.99, power=0.3, k=0.3)             # trees of tens of leaves from the prior: hand-overs
    if g.random() < 0.2:
        bart_args["useQuantiles"] = True
    warmup = int(g.integers(1, 5)); it = warmup + int(g.integers(2, 8))
    groups = [GroupTerm(g.integers(1, 6, size=n), None, "g.1")] if g.random() < 0.4 else []
    args = make_sampler_args(y, xb, X=x4[:, None], groups=groups, family="binomial" if binary else "gaussian", iter=it, warmup=warmup, bart_args=bart_args,
                             x_test=xb[:50].copy() if g.random() < 0.3 else None)
    if bart_args.get("power") == 0.3:
        args.node_capacity = 1024
    return args, dict(n=n, p=p, binary=binary, **bart_args)


if __name__ == "__main__":
    lo, hi, path = int(sys.argv[1]), int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "persistent")
    olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so")); hlib = load_library()
    bad, refused, handed, t0 = [], 0, 0, time.time()
    for seed in range(lo, hi):
        args, what = large_case(seed)
        print("seed", seed, what, flush=True)
        a = run_chain(olib, "orc_", args, results_type=1)
        try:
            b = run_chain(hlib, "s4b_", args, results_type=1, tree_path=path)
        except RuntimeError as e:
            if "node capacity exceeded" in str(e) or "outgrew node_capacity" in str(e):
                refused += 1; continue
            raise
        handed += b["sweep_stats"][1]
        try:
            assert_chain_parity(a, b, stan=False)
        except AssertionError as e:
            bad.append(seed); print("  FAILED", str(e)[:300], flush=True)
    print(f"large seeds {lo}..{hi - 1} on the {path} path: {hi - lo - len(bad) - refused} ok, {refused} refused for their node capacity, failed {bad}; {handed} sweeps handed over; {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
