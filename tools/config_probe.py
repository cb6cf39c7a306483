#!/usr/bin/env python3
"""One of bench.py's extra configurations alone, in its stationary regime, for a profiler or the host-side timers (S4B_HOST_TIMING=1 with the tuning library):
    python tools/config_probe.py c2|c4 [--burn 1000] [--iters 400]
    rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/config_probe.py c4"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c2", "c4"])
    ap.add_argument("--burn", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=400)
    a = ap.parse_args()
    import bench
    from stan4bart_amd import RRng, generate_friedman_data
    from stan4bart_amd._lib import load_library
    from stan4bart_amd.abi import Sampler
    if a.config == "c2":
        d = generate_friedman_data(100_000, ranef=False, causal=True, p=10)
        args = bench.case_from_design(d, 10, 200, 0, a.burn, a.burn + a.iters, ranef=False)
    else:
        from stan4bart_amd.cases import ihdp_case
        args = ihdp_case(warmup=a.burn, iter=a.burn + a.iters, T=75)
    rng = RRng(12345)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(load_library(), "s4b_", args, rng.state)
    s.run(a.burn, True, 0)
    s.disengage_adaptation()
    n0 = s.get_nuts_stats()
    t0 = time.perf_counter()
    s.run(a.iters, False, 0)
    dt = time.perf_counter() - t0
    n1 = s.get_nuts_stats()
    out = {"config": a.config, "iters_per_sec": a.iters / dt, "ms_per_iteration": 1e3 * dt / a.iters, "tree_path": s.get_tree_path()[1],
           "n_leapfrog_per_iteration": (n1["sum_n_leapfrog"] - n0["sum_n_leapfrog"]) / a.iters}
    s.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
