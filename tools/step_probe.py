#!/usr/bin/env python3
"""Tuning probe for the BART sweep on the GPU box: Friedman-like data from numpy's generator (fast to make; the sweep's cost
does not depend on which generator made x), a few Gibbs iterations, then s4b_profile_sweep.  Environment switches of the
library apply (S4B_GRIDF, S4B_GRID, S4B_LIB_PATH=.../libs4b_timing.so for the in-kernel phase timers); --path picks the tree update.
    python tools/step_probe.py --n 1000000 --p 50 --trees 200 --sweeps 3"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--p", type=int, default=50)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--path", default="auto", choices=["auto", "two-kernel", "fused", "persistent", "stream"])
    ap.add_argument("--split-probs", action="store_true", help="cgm(split.probs = ): weights 3 on the first five predictors, 0.2 on the last five, 1 elsewhere")
    ap.add_argument("--weights", action="store_true", help="observation weights, uniform on (0.3, 3)")
    a = ap.parse_args()
    from stan4bart_amd import RRng, make_sampler_args
    from stan4bart_amd._lib import load_library
    from stan4bart_amd.abi import Sampler
    g = np.random.default_rng(99)
    xb = np.empty((a.n, a.p - 1), order="F")
    for j in range(a.p - 1):
        xb[:, j] = g.random(a.n)
    x4 = g.random(a.n)
    z = (g.random(a.n) < 0.2).astype(np.float64)
    y = 10 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 20 * (xb[:, 2] - 0.5) ** 2 + 5 * xb[:, 3] + 10 * x4 + 5 * z + g.standard_normal(a.n)
    args = make_sampler_args(y, xb, X=np.column_stack([x4, z]), groups=[], iter=2 * a.iters, warmup=a.iters, keep_fits=False, weights=g.uniform(0.3, 3.0, a.n) if a.weights else None,
                             bart_args=dict({"n.trees": a.trees}, **({"split.probs": [3.0 if j < 5 else (0.2 if j >= a.p - 6 else 1.0) for j in range(a.p - 1)]} if a.split_probs else {})))
    rng = RRng(4321)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(load_library(), "s4b_", args, rng.state)
    s.set_tree_path(a.path)
    s.run(a.iters, True, 0)
    spec0 = s.get_sweep_spec()
    prof = s.profile_sweep(a.sweeps)
    spec1 = s.get_sweep_spec()
    prof["sweep_spec_profiled"] = dict(zip(("launches", "tree_updates", "published_before_verdict", "borne_out"), (b - c for b, c in zip(spec1, spec0))))
    s.free()
    prof["per_tree_wall_us"] = prof["sweep_wall_us"] / a.trees
    prof["GBs_per_tree_wall"] = 22.0 * a.n / (prof["per_tree_wall_us"] * 1e-6) / 1e9
    print(json.dumps(prof))


if __name__ == "__main__":
    main()
