#!/usr/bin/env python3
"""Benchmark of the stan4bart Gibbs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Gibbs iterations / second at n = 1e6, p = 50, ntree = 200 — BASELINE config 3:
Friedman data, formula y ~ bart(. - X4 - z - g.1 - g.2) + X4 + z + (1 + X4 | g.1) + (1 | g.2), one chain per GPU.
A "step" is one Gibbs iteration = one NUTS transition of the Stan block + one BART sweep (200 tree updates).
`value` is the whole-job aggregate over all N chains (chains are independent, so scaling is weak);
`per_chain` is the BASELINE per-chain figure.  Inputs are resident in HBM before the timed region.

Extra objects on the same JSON line:
  roofline      dominant kernel of the sweep, HIP-event timing on the sampler's own stream, against 8 TB/s HBM
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this box's host, rank 0 at N = 1 only, on the same workload
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def build_case(n, p, trees, device, warmup, steps):
    from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
    d = generate_friedman_data(n, ranef=True, causal=True, p=p)
    x = d["x"]
    xb = np.asfortranarray(x[:, [j for j in range(p) if j != 3]])
    X = np.column_stack([x[:, 3], d["z"]])
    groups = [GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")]
    args = make_sampler_args(d["y"], xb, X=X, groups=groups, iter=warmup + steps, warmup=warmup, keep_fits=False,
                             bart_args={"n.trees": trees}, device=device)
    return args


def target_roofline_leg(lib, n, p, trees, device, sweeps):
    """north_star's roofline target is quoted at n = 1e7, p = 50, ntree = 200 (larger than the metric's workload): measure
    the same dominant kernel there too, live, on a BART-sized case (Friedman surface, fixed effects X4 + z, numpy's
    generator for the 5e8 uniforms — the R-compatible stream would take minutes on the host and the sweep's cost does
    not depend on which generator made x)."""
    from stan4bart_amd import RRng, make_sampler_args
    from stan4bart_amd.abi import Sampler
    g = np.random.default_rng(99)
    x = np.empty((n, p), order="F")
    for j in range(p):
        x[:, j] = g.random(n)
    z = (g.random(n) < 0.2).astype(np.float64)
    y = (10.0 * np.sin(np.pi * x[:, 0] * x[:, 1]) + 20.0 * (x[:, 2] - 0.5) ** 2 + 5.0 * x[:, 4] + 10.0 * x[:, 3] + 5.0 * z
         + g.standard_normal(n))
    X = np.column_stack([x[:, 3], z])
    xb = np.asfortranarray(np.delete(x, 3, axis=1))
    del x
    args = make_sampler_args(y, xb, X=X, groups=[], iter=8, warmup=4, keep_fits=False, bart_args={"n.trees": trees},
                             device=device)
    rng = RRng(4321)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "s4b_", args, rng.state)
    s.run(2, True, 0)
    prof = s.profile_sweep(sweeps)
    s.free()
    achieved = 22.0 * n / (prof["stats_us"] * 1e-6) / 1e9
    return {"workload": f"Friedman n={n}, p={p}, ntree={trees} (north_star roofline target config)", "kernel": "k_tree",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "avg_launch_us": prof["stats_us"], "k_control_us": prof["control_us"], "algorithmic_bytes_per_launch": 22.0 * n,
            "sweep_wall_us": prof["sweep_wall_us"]}


def cpu_baseline(n, p, trees, iters):
    """Time the CPU oracle (single thread, the reference's execution model: R/stan4bart_fit.R:437-439) on the
    same workload for a bounded number of Gibbs iterations."""
    import subprocess
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    lib = ctypes.CDLL(so)
    args = build_case(n, p, trees, 0, 0, iters)
    rng = RRng(12345)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "orc_", args, rng.state)
    s.run(1, True, 0)                      # one untimed iteration (page-in)
    t0 = time.perf_counter()
    s.run(iters, True, 0)
    dt = time.perf_counter() - t0
    s.free()
    return iters / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--p", type=int, default=50)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-sweeps", type=int, default=2)
    ap.add_argument("--target-n", type=int, default=10_000_000,
                    help="also measure the sweep kernel at north_star's roofline-target size (0 = skip; N = 1 only)")
    a = ap.parse_args()

    import torch
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd._lib import load_library
    from stan4bart_amd.fit import chain_seeds
    from stan4bart_amd.parallel import all_gather_array, init_process_group

    # S4B_BENCH_BACKEND=gloo + S4B_BENCH_ONE_DEVICE=1: rehearsal of the N-rank path on a single-GPU box (all ranks share
    # device 0, the collectives run over gloo); the driver's real runs use the default: one GPU per rank, RCCL
    rank, local_rank, world = init_process_group(os.environ.get("S4B_BENCH_BACKEND"))
    if os.environ.get("S4B_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    lib = load_library()

    args = build_case(a.n, a.p, a.trees, local_rank, a.warmup, a.steps)
    rng = RRng(int(chain_seeds(20260101, max(1, world))[rank]))
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    sampler = Sampler(lib, "s4b_", args, rng.state)     # uploads everything: inputs are HBM-resident from here on
    if a.warmup > 0:
        sampler.run(a.warmup, True, 0)                  # W untimed warm-up Gibbs iterations (adaptation engaged)
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    barrier()
    c0 = sampler.get_counters()
    t0 = time.perf_counter()
    out = sampler.run(a.steps, True, 0)                 # exactly K timed Gibbs iterations (run() returns synchronised)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = sampler.get_counters()
    barrier()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt_max = float(t.item())
    else:
        dt_max = dt
    sig = all_gather_array(np.array([float(out["bart"]["sigma"][-1])]))   # the only collective: chain summaries

    prof = sampler.profile_sweep(a.profile_sweeps) if rank == 0 else None
    lf = sampler.profile_leapfrog(20) if rank == 0 else None
    probe = None
    if rank == 0:   # measured streaming ceiling of this device (read-only and in-place update over 1 GB)
        po = (ctypes.c_double * 4)()
        lib.s4b_stream_probe.restype = ctypes.c_int
        if lib.s4b_stream_probe(ctypes.c_int32(local_rank), ctypes.c_int64(1 << 27), ctypes.c_int32(5), po) == 0:
            probe = {"read_GBs": po[0], "update_in_place_GBs": po[1]}
    sampler.free()
    del args
    target = None
    if world == 1 and a.target_n > 0 and a.target_n != a.n:
        target = target_roofline_leg(lib, a.target_n, a.p, a.trees, local_rank, a.profile_sweeps)

    if rank == 0:
        n = a.n
        per_chain = a.steps / dt_max
        # dominant kernel of the sweep and its algorithmic bytes per launch (DESIGN.md "Roofline accounting")
        # k_tree<true> = the whole O(N) part of one tree update (finish tree t-1 + statistics of tree t):
        # R read 8 + R write 8 + leaf(t-1) read 2 + leaf(t) read 2 + binned predictor 2 = 22 B per observation
        dom, dom_us, dom_bytes = "k_tree", prof["stats_us"], 22.0 * n
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        tree_update_us = prof["stats_us"] + prof["control_us"]
        # HBM traffic of the same kernel from the PMC passes committed under profiles/ (rocprofv3 cannot wrap this process
        # from inside; the counters were collected on the same command line, see profiles/pmc_traffic.json)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(str(n), {}).get("bytes_per_launch")
        except (OSError, ValueError):
            pass
        rec = {
            "metric": "gibbs_iters_per_sec", "value": per_chain * world, "unit": "Gibbs iterations/s (all chains)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_max / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_chain": per_chain,
            "config": {"workload": f"Friedman n={n}, p={a.p}, ntree={a.trees}, (1+X4|g.1)+(1|g.2), one chain per GPU"
                                   + (" (BASELINE config 3)" if (n, a.p, a.trees) == (1_000_000, 50, 200) else ""),
                       "chains": world, "hmc_mode": "sufficient-statistics", "n_leapfrog_timed": int(c1[0] - c0[0]),
                       "tree_updates_timed": int(c1[1] - c0[1]), "sigma_last": [float(s[0]) for s in sig]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_us": dom_us, "algorithmic_bytes_per_launch": dom_bytes,
                         "tree_update": {"k_tree_us": prof["stats_us"], "k_control_us": prof["control_us"],
                                         "final_k_apply_us": prof["apply_us"], "algorithmic_bytes": 22.0 * n,
                                         "achieved_GBs_incl_control": 22.0 * n / (tree_update_us * 1e-6) / 1e9},
                         "sweep_wall_us": prof["sweep_wall_us"],
                         "measured_stream": probe,
                         "frac_of_measured_update_stream": (achieved / probe["update_in_place_GBs"]) if probe else None},
        }
        # second kernel group of the path: the O(N) sums one leapfrog costs when the gradient is evaluated on the device
        # (hmc_mode 1, the reference's cost model).  The timed region above uses hmc_mode 0, where a leapfrog is O((K+q)^2)
        # on the host from sufficient statistics gathered once per Gibbs iteration by these same kernels.
        rec["roofline_hmc"] = {"bound": "hbm", "kernel": "k_stan_inputs + k_zt_chunks + k_stan_finalize (direct)",
                               "achieved": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "avg_eval_us": lf["kernels_us"], "avg_eval_us_with_result_fetch": lf["with_fetch_us"],
                               "launches_per_eval": lf["launches"], "algorithmic_bytes_per_eval": lf["algorithmic_bytes"],
                               "note": "N (8K + 12z + 20) bytes per leapfrog (SURVEY 8d B_lf); not in the timed region (hmc_mode 0)"}
        if target is not None:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    target["traffic"] = json.load(f).get(str(a.target_n), {}).get("bytes_per_launch")
            except (OSError, ValueError):
                target["traffic"] = None
            if probe:
                target["frac_of_measured_update_stream"] = target["achieved"] / probe["update_in_place_GBs"]
            rec["roofline_target_config"] = target
        if world == 1 and not a.no_cpu_baseline:
            v, secs = cpu_baseline(a.n, a.p, a.trees, a.cpu_iters)
            rec["cpu_baseline"] = {"value": v, "unit": "Gibbs iterations/s/chain", "cores": 1, "kind": "port",
                                   "sample": f"same workload (n={n}, p={a.p}, ntree={a.trees}), {a.cpu_iters} Gibbs iterations, "
                                             f"{secs:.1f} s of single-thread CPU time, host has {os.cpu_count()} cores"}
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
