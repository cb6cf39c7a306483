#!/usr/bin/env python3
"""Benchmark of the stan4bart Gibbs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Gibbs iterations / second at n = 1e6, p = 50, ntree = 200 — BASELINE config 3:
Friedman data, formula y ~ bart(. - X4 - z - g.1 - g.2) + X4 + z + (1 + X4 | g.1) + (1 | g.2), one chain per GPU.
A "step" is one Gibbs iteration = one NUTS transition of the Stan block + one BART sweep (200 tree updates).
`value` is the whole-job aggregate over all N chains (chains are independent, so scaling is weak);
`per_chain` is the BASELINE per-chain figure.  Inputs are resident in HBM before the timed region.

What is timed is a RUNNING chain, in the reference's phase order (R/stan4bart_fit.R:49-51: warm-up, disengage adaptation,
sample): `--burn-in` untimed warm-up iterations with adaptation engaged (step size, metric windows) bring the chain to its
stationary regime, adaptation is disengaged, then W untimed and exactly K timed iterations of the SAMPLING phase follow.
Leapfrogs per transition and mean tree depth of the timed iterations are reported in `config` (on this workload the adapted
chain needs ~10 leapfrogs per transition, tree depth 3); the warm-up phase's own rate is `warmup_phase_iters_per_sec`.

Extra objects on the same JSON line:
  roofline      dominant kernel of the sweep, HIP-event timing on the sampler's own stream, against 8 TB/s HBM
  cpu_baseline  the CPU oracle (oracle/, "port") timed on this box's host, rank 0 at N = 1 only, on the same workload
  extra_configs BASELINE configs 1 and 2 (CPU port for both, the HIP path for config 2), small and quick
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def friedman_design(n, p, rank, world, barrier):
    """The synthetic design of the metric's workload.  R-compatible generator (reference inst/common/friedmanData.R with
    set.seed(99)), ~13 s of host time at n = 1e6: made ONCE per node by rank 0 and shared through /dev/shm, not once per rank."""
    from stan4bart_amd import generate_friedman_data
    keys = ("x", "y", "z", "g1", "g2")
    if world == 1:
        return generate_friedman_data(n, ranef=True, causal=True, p=p)
    path = f"/dev/shm/s4b_bench_design_{os.environ.get('MASTER_PORT', '0')}_{n}_{p}.npz"
    if rank == 0:
        d = generate_friedman_data(n, ranef=True, causal=True, p=p)
        np.savez(path + ".tmp.npz", **{k: d[k] for k in keys})
        os.replace(path + ".tmp.npz", path)
    barrier()
    if rank != 0:
        with np.load(path) as f:
            d = {k: f[k] for k in keys}
    barrier()
    if rank == 0:
        os.remove(path)
    return d


def case_from_design(d, p, trees, device, warmup, iters, ranef=True, keep_fits=False):
    from stan4bart_amd import GroupTerm, make_sampler_args
    x = d["x"]
    xb = np.asfortranarray(x[:, [j for j in range(p) if j != 3]])
    X = np.column_stack([x[:, 3], d["z"]])
    groups = [GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")] if ranef else []
    return make_sampler_args(d["y"], xb, X=X, groups=groups, iter=iters, warmup=warmup, keep_fits=keep_fits,
                             bart_args={"n.trees": trees}, device=device)


def target_roofline_leg(lib, n, p, trees, device, sweeps):
    """north_star's roofline target is quoted at n = 1e7, p = 50, ntree = 200 (larger than the metric's workload): measure
    the same dominant kernel there too, live, on a BART-sized case (Friedman surface, fixed effects X4 + z, numpy's
    generator for the 5e8 uniforms — the R-compatible stream would take minutes on the host and the sweep's cost does
    not depend on which generator made x)."""
    from stan4bart_amd import RRng, make_sampler_args
    from stan4bart_amd.abi import Sampler
    g = np.random.default_rng(99)
    xb = np.empty((n, p - 1), order="F")
    for j in range(p - 1):
        xb[:, j] = g.random(n)
    x4 = g.random(n)
    z = (g.random(n) < 0.2).astype(np.float64)
    from stan4bart_amd import GroupTerm
    g1, g2 = g.integers(1, 6, size=n), g.integers(1, 9, size=n)
    b1 = g.standard_normal((5, 2)) @ np.linalg.cholesky(np.array([[2.25, 0.2], [0.2, 1.0]])).T
    y = (10.0 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 20.0 * (xb[:, 2] - 0.5) ** 2 + 5.0 * xb[:, 3] + 10.0 * x4 + 5.0 * z
         + b1[g1 - 1, 0] + b1[g1 - 1, 1] * x4 + 1.1 * g.standard_normal(8)[g2 - 1] + g.standard_normal(n))
    args = make_sampler_args(y, xb, X=np.column_stack([x4, z]), groups=[GroupTerm(g1, x4, "g.1"), GroupTerm(g2, None, "g.2")],
                             iter=8, warmup=4, keep_fits=False, bart_args={"n.trees": trees}, device=device)
    del xb
    rng = RRng(4321)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "s4b_", args, rng.state)
    s.run(2, True, 0)
    prof = s.profile_sweep(sweeps)
    lf = s.profile_leapfrog(10)
    s.free()
    fused = prof["control_us"] == 0.0
    achieved = 22.0 * n / (prof["stats_us"] * 1e-6) / 1e9
    return {"workload": f"Friedman n={n}, p={p}, ntree={trees} (north_star roofline target config)",
            "kernel": "k_step" if fused else "k_tree", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "avg_launch_us": prof["stats_us"], "k_control_us": prof["control_us"],
            "algorithmic_bytes_per_launch": 22.0 * n, "sweep_wall_us": prof["sweep_wall_us"],
            "achieved_GBs_whole_sweep": 22.0 * n * trees / (prof["sweep_wall_us"] * 1e-6) / 1e9,
            "hmc": {"kernel": "k_stan_fused (direct), K=2, z=3", "bound": "hbm", "achieved": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "avg_eval_us": lf["kernels_us"], "avg_eval_us_with_result_fetch": lf["with_fetch_us"], "launches_per_eval": lf["launches"],
                    "algorithmic_bytes_per_eval": lf["algorithmic_bytes"]}}


def oracle_lib():
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return ctypes.CDLL(so)


def time_chain(lib, prefix, args, seed, warm, iters):
    """`warm` untimed + `iters` timed warm-up-phase Gibbs iterations of one chain; returns (iterations / s, seconds)."""
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    rng = RRng(seed)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, prefix, args, rng.state)
    if warm:
        s.run(warm, True, 0)
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
    ap.add_argument("--burn-in", type=int, default=150, help="untimed warm-up-phase iterations before the W + K sampling-phase iterations")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--p", type=int, default=50)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--profile-sweeps", type=int, default=2)
    ap.add_argument("--target-n", type=int, default=10_000_000,
                    help="also measure the sweep kernel at north_star's roofline-target size (0 = skip; N = 1 only)")
    a = ap.parse_args()
    t_start = time.perf_counter()

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

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    d = friedman_design(a.n, a.p, rank, world, barrier)
    t_design = time.perf_counter()
    # the NUTS adaptation windows are laid out over the warm-up phase (reference interruptable_sampler.hpp:171): warmup = burn-in
    total = a.burn_in + a.warmup + a.steps
    args = case_from_design(d, a.p, a.trees, local_rank, a.burn_in, total)
    rng = RRng(int(chain_seeds(20260101, max(1, world))[rank]))
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    sampler = Sampler(lib, "s4b_", args, rng.state)     # uploads everything: inputs are HBM-resident from here on
    t_created = time.perf_counter()
    # ---- phase 1 (untimed for the metric): warm-up with adaptation engaged, to the stationary regime
    warm_rate = None
    if a.burn_in > 0:
        t0 = time.perf_counter()
        sampler.run(a.burn_in, True, 0)
        warm_rate = a.burn_in / (time.perf_counter() - t0)
    sampler.disengage_adaptation()
    # ---- phase 2: sampling.  W untimed, then exactly K timed Gibbs iterations
    if a.warmup > 0:
        sampler.run(a.warmup, False, 0)
    barrier()
    c0, s0 = sampler.get_counters(), sampler.get_nuts_stats()
    t0 = time.perf_counter()
    out = sampler.run(a.steps, False, 0)                # (run() returns synchronised)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1, s1 = sampler.get_counters(), sampler.get_nuts_stats()
    barrier()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt_max = float(t.item())
    else:
        dt_max = dt
    # the only collective: chain summaries (last sigma, per-rank set-up seconds)
    summ = all_gather_array(np.array([float(out["bart"]["sigma"][-1]), t_design - t_start, t_created - t_design]))

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
        trans = max(1, s1["transitions"] - s0["transitions"])
        fused = prof["control_us"] == 0.0
        # dominant kernel of the sweep and its algorithmic bytes per launch (DESIGN.md "Roofline accounting"): the O(N) part of
        # one tree update (finish tree t-1 + statistics of tree t):
        # R read 8 + R write 8 + leaf(t-1) read 2 + leaf(t) read 2 + binned predictor 2 = 22 B per observation.
        # fused path: k_step is the WHOLE tree update (control code included); two-kernel path: k_tree, with k_control beside it
        dom = "k_step" if fused else "k_tree"
        dom_us, dom_bytes = prof["stats_us"], 22.0 * n
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        # HBM traffic of the tree kernel: REPLAYED from the PMC passes committed under profiles/ (rocprofv3 cannot wrap this
        # process from inside; collected on this command line by tools/profile_round.sh, see profiles/pmc_traffic.json)
        traffic, traffic_note = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            traffic = pj.get(str(n), {}).get("bytes_per_launch")
            traffic_note = "replayed from profiles/pmc_traffic.json (" + str(pj.get(str(n), {}).get("kernel", "tree kernel")) + "), not measured in this run"
        except (OSError, ValueError):
            pass
        rec = {
            "metric": "gibbs_iters_per_sec", "value": per_chain * world, "unit": "Gibbs iterations/s (all chains)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_max / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_chain": per_chain,
            "phase_timed": f"sampling (after {a.burn_in} untimed warm-up iterations with adaptation, disengage, {a.warmup} untimed sampling iterations)",
            "warmup_phase_iters_per_sec": warm_rate,
            "config": {"workload": f"Friedman n={n}, p={a.p}, ntree={a.trees}, (1+X4|g.1)+(1|g.2), one chain per GPU"
                                   + (" (BASELINE config 3)" if (n, a.p, a.trees) == (1_000_000, 50, 200) else ""),
                       "chains": world, "hmc_mode": "sufficient-statistics", "burn_in": a.burn_in,
                       "n_leapfrog_timed": int(s1["sum_n_leapfrog"] - s0["sum_n_leapfrog"]),
                       "n_leapfrog_per_step": (s1["sum_n_leapfrog"] - s0["sum_n_leapfrog"]) / a.steps,
                       "mean_treedepth_timed": (s1["sum_treedepth"] - s0["sum_treedepth"]) / trans,
                       "divergent_timed": int(s1["divergent"] - s0["divergent"]),
                       "gradient_evals_timed": int(c1[0] - c0[0]), "tree_updates_timed": int(c1[1] - c0[1]),
                       "sigma_last": [float(s[0]) for s in summ],
                       "setup_seconds_per_rank": {"design (rank 0 generates, others load)": [float(s[1]) for s in summ],
                                                  "create (binning, upload, init)": [float(s[2]) for s in summ]}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "avg_launch_us": dom_us, "algorithmic_bytes_per_launch": dom_bytes,
                         "timing": "HIP events around every launch on the sampler's stream (adds ~2 us per launch; profiles/ has rocprofv3)",
                         "separate_control_kernel_us": None if fused else prof["control_us"],
                         "last_launch_of_sweep_us": prof["apply_us"],
                         "sweep_wall_us": prof["sweep_wall_us"], "tree_update_wall_us": prof["sweep_wall_us"] / a.trees,
                         "achieved_GBs_whole_sweep": dom_bytes * a.trees / (prof["sweep_wall_us"] * 1e-6) / 1e9,
                         "measured_stream": probe,
                         "frac_of_measured_update_stream": (achieved / probe["update_in_place_GBs"]) if probe else None},
        }
        # second kernel group of the path: the O(N) sums one leapfrog costs when the gradient is evaluated on the device
        # (hmc_mode 1, the reference's cost model).  The timed region above uses hmc_mode 0, where a leapfrog is O((K+q)^2)
        # on the host from sufficient statistics gathered once per Gibbs iteration by the same kernel.
        rec["roofline_hmc"] = {"bound": "hbm", "kernel": "k_stan_fused (direct)",
                               "achieved": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                               "avg_eval_us": lf["kernels_us"], "avg_eval_us_with_result_fetch": lf["with_fetch_us"],
                               "launches_per_eval": lf["launches"], "algorithmic_bytes_per_eval": lf["algorithmic_bytes"],
                               "note": "N (8K + 12z + 20) bytes per leapfrog (SURVEY 8d B_lf); not in the timed region (hmc_mode 0)"}
        if target is not None:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    target["traffic"] = json.load(f).get(str(a.target_n), {}).get("bytes_per_launch")
                target["traffic_note"] = "replayed from profiles/pmc_traffic.json, not measured in this run"
            except (OSError, ValueError):
                target["traffic"] = None
            if probe:
                target["frac_of_measured_update_stream"] = target["achieved"] / probe["update_in_place_GBs"]
            rec["roofline_target_config"] = target
        if world == 1 and not a.no_cpu_baseline:
            olib = oracle_lib()
            cargs = case_from_design(d, a.p, a.trees, 0, 0, a.cpu_iters + 1)
            v, secs = time_chain(olib, "orc_", cargs, 12345, 1, a.cpu_iters)
            rec["cpu_baseline"] = {"value": v, "unit": "Gibbs iterations/s/chain", "cores": 1, "kind": "port",
                                   "sample": f"same workload (n={n}, p={a.p}, ntree={a.trees}), {a.cpu_iters} Gibbs iterations from a cold chain "
                                             f"(warm-up phase, 3-4 leapfrogs each), {secs:.1f} s of single-thread CPU time, host has {os.cpu_count()} cores",
                                   "reference_R_package": "R is absent on this box (" + ("Rscript not found" if shutil.which("Rscript") is None else "Rscript present but the reference is not installed")
                                                          + "): the CPU figure is the repo's C++ restatement (oracle/), not vdorie/stan4bart itself"}
        if world == 1 and not a.no_extra_configs:
            # BASELINE configs 1 and 2: small and quick, reported beside the headline
            from stan4bart_amd import generate_friedman_data
            olib = oracle_lib()
            extra = {}
            d1 = generate_friedman_data(100, ranef=True, causal=True, p=10)
            from stan4bart_amd import GroupTerm, make_sampler_args
            x1 = d1["x"]
            a1 = make_sampler_args(d1["y"], x1[:, [j for j in range(10) if j != 3]], X=np.column_stack([x1[:, 3], d1["z"]]),
                                   groups=[GroupTerm(d1["g1"], None, "g.1"), GroupTerm(d1["g2"], None, "g.2")], iter=400, warmup=200,
                                   keep_fits=False, bart_args={"n.trees": 50})
            v1, s1c = time_chain(olib, "orc_", a1, 12345, 100, 300)
            extra["config1"] = {"workload": "Friedman n=100, ntree=50, (1|g.1)+(1|g.2) (BASELINE config 1, CPU plumbing case)",
                                "cpu_port_iters_per_sec": v1, "cpu_seconds": s1c}
            d2 = generate_friedman_data(100_000, ranef=False, causal=True, p=10)
            a2c = case_from_design(d2, 10, 200, 0, 0, 40, ranef=False)
            v2c, s2c = time_chain(olib, "orc_", a2c, 12345, 2, 30)
            a2g = case_from_design(d2, 10, 200, local_rank, 0, 400, ranef=False)
            v2g, s2g = time_chain(lib, "s4b_", a2g, 12345, 100, 200)
            extra["config2"] = {"workload": "Friedman n=1e5, p=10, ntree=200, fixed effects only (BASELINE config 2)",
                                "gpu_iters_per_sec": v2g, "gpu_seconds": s2g, "cpu_port_iters_per_sec": v2c, "cpu_seconds": s2c, "cpu_cores": 1}
            rec["extra_configs"] = extra
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
