#!/usr/bin/env python3
"""Cost of the probit latent draws per observation on the GPU box: a binary-response chain with few trees, timed per Gibbs
iteration with the table kernel (default) and with S4B_LATENTS=1 (serial-bookkeeping kernel) — run it once per setting.
    python tools/latents_probe.py --n 1000000"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--trees", type=int, default=10)
    ap.add_argument("--iters", type=int, default=6)
    a = ap.parse_args()
    from stan4bart_amd import RRng, make_sampler_args
    from stan4bart_amd._lib import load_library
    from stan4bart_amd.abi import Sampler
    g = np.random.default_rng(5)
    xb = np.asfortranarray(g.random((a.n, 5)))
    x4 = g.random(a.n)
    eta = 2.0 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 2.0 * (xb[:, 2] - 0.5) + 1.5 * (x4 - 0.5)
    y = (eta + g.standard_normal(a.n) > 1.0).astype(np.float64)
    args = make_sampler_args(y, xb, X=x4.reshape(-1, 1), groups=[], family="binomial", iter=2 * a.iters, warmup=a.iters, keep_fits=False,
                             bart_args={"n.trees": a.trees})
    rng = RRng(77)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(load_library(), "s4b_", args, rng.state)
    s.run(2, True, 1)
    t0 = time.perf_counter()
    s.run(a.iters, True, 1)           # BART block only: sweep + latents
    dt = (time.perf_counter() - t0) / a.iters
    st = s.get_r_rng_state()
    s.free()
    print(f"n={a.n} trees={a.trees}: {dt * 1e3:.3f} ms per BART iteration = {dt / a.n * 1e6:.4f} us per observation (sweep included); mean(y)={y.mean():.3f}; rng checksum {int(st.astype(np.uint64).sum())}")


if __name__ == "__main__":
    main()
