#!/usr/bin/env python3
"""Benchmark of the stan4bart Gibbs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Gibbs iterations / second at n = 1e6, p = 50, ntree = 200 — BASELINE config 3:
Friedman data, formula y ~ bart(. - X4 - z - g.1 - g.2) + X4 + z + (1 + X4 | g.1) + (1 | g.2), one chain per GPU.
A "step" is one Gibbs iteration = one NUTS transition of the Stan block + one BART sweep (200 tree updates).
`value` is the whole-job aggregate over all N chains (chains are independent, so scaling is weak);
`per_chain` is the BASELINE per-chain figure.  Inputs are resident in HBM before the timed region.

What is timed is a RUNNING chain in its STATIONARY regime, in the reference's phase order (R/stan4bart_fit.R:49-51: warm-up,
disengage adaptation, sample): `--burn-in` untimed warm-up iterations with adaptation engaged (default 1000 — the reference's default
fit is 1000 warm-up + 1000 sampling iterations), adaptation is disengaged, then W untimed and exactly K timed iterations of the SAMPLING
phase follow.  At this size the chain needs several hundred iterations before NUTS reaches the tree depth it then keeps (~1000 leapfrogs
per transition: DESIGN.md 8); a short burn-in times a transient that is 30 times cheaper in leapfrogs.  `config.stationarity` therefore
compares the leapfrogs per iteration of the timed window with those of the next `--mode-iters` iterations of the same chain and says
whether they agree within 20 %; `value` is only a stationary figure when it says so.

Extra objects on the same JSON line:
  roofline            dominant kernel of the sweep, HIP-event timing on the sampler's own stream, against 8 TB/s HBM
  per_chain_hmc_mode1 the same chain continued for K more timed iterations with one O(N) device evaluation per leapfrog
                      (the reference's cost model); the headline uses sufficient statistics gathered once per iteration
  cpu_baseline        the CPU oracle (oracle/, "port") timed on this box's host, rank 0 at N = 1 only, on the same workload,
                      started from the GPU chain's state after the burn-in (same regime: same leapfrogs per transition)
  extra_configs       BASELINE configs 1 (CPU port), 2, 4 and 5 (HIP path; each timed in its stationary regime like the headline: warm-up with
                      adaptation, disengage, sampling iterations, a `stationarity` record and the roofline of the chain that was timed)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os

# ROCr / RCCL read their environment when the runtime initialises: set it before anything can touch HIP (the host driver of this
# pool only supports dmabuf IPC; without this RCCL fails with `hipIpcGetMemHandle: invalid argument`)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import shutil
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)
DOMINANT = {"fused": "k_step", "two-kernel": "k_tree", "persistent": "k_sweep", "stream": "k_sweep_stream"}


def source_sha16():
    """Fingerprint of what produced a line: bench.py + the library's sources.  A committed N = 1 reference line is only compared with a
    line of the same sources (profiles/bench_n1_reference.json)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "stan4bart_amd", "csrc")
    files = [os.path.abspath(__file__)] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc", ".hpp")) or f == "Makefile")
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pin_rank_to_cores(local_rank: int, ranks_on_node: int):
    """One chain = one host thread (NUTS control flow, launches).  N ranks on one socket otherwise migrate across each other's
    cores: give every rank a disjoint, contiguous core set.  Returns the cores this process may run on."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        if ranks_on_node <= 1 or len(cores) < ranks_on_node:
            return cores
        per = len(cores) // ranks_on_node
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return []


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


def case_from_design(d, p, trees, device, warmup, iters, ranef=True, keep_fits=False, hmc_mode=0):
    from stan4bart_amd import GroupTerm, make_sampler_args
    x = d["x"]
    xb = np.asfortranarray(x[:, [j for j in range(p) if j != 3]])
    X = np.column_stack([x[:, 3], d["z"]])
    groups = [GroupTerm(d["g1"], x[:, 3], "g.1"), GroupTerm(d["g2"], None, "g.2")] if ranef else []
    return make_sampler_args(d["y"], xb, X=X, groups=groups, iter=iters, warmup=warmup, keep_fits=keep_fits,
                             bart_args={"n.trees": trees}, device=device, stan_args={"hmc_mode": hmc_mode})


def sweep_roofline(prof, path, n, trees):
    """Roofline record of the dominant kernel of one sweep.  Algorithmic bytes of a tree update: R read 8 + R write 8 + leaf id of
    the finished tree 2 + leaf id of the tree whose statistics are gathered 2 + binned predictor 2 = 22 B per observation."""
    # (the persistent path runs a whole sweep — `trees` tree updates — in ONE launch of k_sweep: its launch does `trees` times the work)
    per_launch = 22.0 * n * (trees if path in ("persistent", "stream") else 1)
    rec = {"bound": "hbm", "kernel": DOMINANT[path], "tree_path": path, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "avg_launch_us": prof["stats_us"], "algorithmic_bytes_per_launch": per_launch,
           "timing": "HIP events around every launch on the sampler's stream (adds ~2 us per launch; profiles/ has rocprofv3)",
           "sweep_wall_us": prof["sweep_wall_us"], "tree_update_wall_us": prof["sweep_wall_us"] / trees,
           "achieved_GBs_whole_sweep": 22.0 * n * trees / (prof["sweep_wall_us"] * 1e-6) / 1e9}
    rec["achieved"] = per_launch / (prof["stats_us"] * 1e-6) / 1e9
    rec["frac"] = rec["achieved"] / HBM_PEAK_GBS
    if path in ("persistent", "stream"):
        rec["tree_updates_per_launch"] = trees
        rec["avg_tree_update_us"] = prof["stats_us"] / trees
        rec["sweeps_handed_over_to_k_step"] = prof["control_us"]     # a tree outgrew the 64 node slots of the wave-register control path
        rec["persistent_sweeps"] = prof["launches"][1]
        rec["note"] = ("algorithmic bytes (SURVEY 8d: 22 B per observation and tree update) / launch duration; the launch itself moves far less: "
                       "the residual stays in registers for the whole sweep (per tree update 2 B leaf id + 2-6 B predictor columns read, 2 B written under an accepted move)"
                       if path == "persistent" else
                       "one launch per sweep; the pass waves stream residual (read + write), both leaf planes and the predictor column of the pending rules per tree: "
                       "the 22 algorithmic bytes per observation and tree update really move")
        return rec
    if path == "two-kernel":
        rec["separate_control_kernel_us"] = prof["control_us"]
        rec["last_launch_of_sweep_us"] = prof["apply_us"]
    elif path == "fused":
        rec["last_launch_of_sweep_us"] = prof["apply_us"]
    return rec


def target_roofline_leg(lib, n, p, trees, device, sweeps, burn_in):
    """north_star's roofline target is quoted at n = 1e7, p = 50, ntree = 200 (larger than the metric's workload): measure
    the same dominant kernel there too, live, on a BART-sized case (Friedman surface, fixed effects X4 + z, numpy's
    generator for the 5e8 uniforms — the R-compatible stream would take minutes on the host and the sweep's cost does
    not depend on which generator made x)."""
    from stan4bart_amd import GroupTerm, RRng, make_sampler_args
    from stan4bart_amd.abi import Sampler
    g = np.random.default_rng(99)
    xb = np.empty((n, p - 1), order="F")
    for j in range(p - 1):
        xb[:, j] = g.random(n)
    x4 = g.random(n)
    z = (g.random(n) < 0.2).astype(np.float64)
    g1, g2 = g.integers(1, 6, size=n), g.integers(1, 9, size=n)
    b1 = g.standard_normal((5, 2)) @ np.linalg.cholesky(np.array([[2.25, 0.2], [0.2, 1.0]])).T
    y = (10.0 * np.sin(np.pi * xb[:, 0] * xb[:, 1]) + 20.0 * (xb[:, 2] - 0.5) ** 2 + 5.0 * xb[:, 3] + 10.0 * x4 + 5.0 * z
         + b1[g1 - 1, 0] + b1[g1 - 1, 1] * x4 + 1.1 * g.standard_normal(8)[g2 - 1] + g.standard_normal(n))
    args = make_sampler_args(y, xb, X=np.column_stack([x4, z]), groups=[GroupTerm(g1, x4, "g.1"), GroupTerm(g2, None, "g.2")],
                             iter=burn_in + 8, warmup=burn_in, keep_fits=False, bart_args={"n.trees": trees}, device=device)
    del xb
    rng = RRng(4321)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "s4b_", args, rng.state)
    # (the sweep is measured on a chain past its burn-in, like the headline: the trees of a young chain are smaller and accept five times as often)
    t0 = time.perf_counter()
    s.run(burn_in, True, 0)
    t_burn = time.perf_counter() - t0
    s.disengage_adaptation()
    path = s.get_tree_path()[1]
    prof = s.profile_sweep(sweeps)
    lf = s.profile_leapfrog(10)
    rec = sweep_roofline(prof, path, n, trees)
    s.free()
    rec["workload"] = f"Friedman n={n}, p={p}, ntree={trees} (north_star roofline target config)"
    rec["burn_in"] = burn_in
    rec["burn_in_seconds"] = t_burn
    rec["hmc"] = {"kernel": "k_stan_fused (direct), K=2, z=3", "bound": "hbm", "achieved": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                  "avg_eval_us": lf["kernels_us"], "avg_eval_us_with_result_fetch": lf["with_fetch_us"], "launches_per_eval": lf["launches"],
                  "algorithmic_bytes_per_eval": lf["algorithmic_bytes"]}
    return rec


def oracle_lib():
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


def cpu_baseline_from_state(olib, args, state, iters):
    """The CPU oracle on the same workload, in the same regime: its chain state is REPLACED by the GPU chain's state after the
    burn-in (adapted step size and metric, trees, residual), adaptation disengaged, then `iters` sampling iterations are timed.
    Returns the record's value fields and the leapfrogs it needed."""
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    rng = RRng(12345)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(olib, "orc_", args, rng.state)
    try:
        s.disengage_adaptation()
        s.set_state(state)
        s.set_trace(True)
        n0 = s.get_nuts_stats()
        t0 = time.perf_counter()
        out = s.run(iters, False, 0)
        dt = time.perf_counter() - t0
        n1 = s.get_nuts_stats()
        chain = {"trace": s.get_trace(), "row": out["stan"][:, -1].copy(), "rng": s.get_r_rng_state()}
    finally:
        s.free()
    return iters / dt, dt, (n1["sum_n_leapfrog"] - n0["sum_n_leapfrog"]) / iters, chain


def chain_check(gpu, cpu, trees):
    """The CPU leg doubles as a parity check of the regime the headline is measured in: both chains ran the same
    iterations from the same state.  The first iteration's tree moves must be identical (anything else is a fault of the HIP path or of the
    oracle: the run stops); over the following iterations NUTS may amplify rounding differences, so the rest is reported."""
    tg, tc = gpu["trace"], cpu["trace"]
    first = bool(len(tg) >= trees and len(tc) >= trees and np.array_equal(tg[:trees], tc[:trees]))
    same = bool(tg.shape == tc.shape and np.array_equal(tg, tc))
    rec = {"first_iteration_tree_moves_identical": first, "all_tree_moves_identical": same, "tree_moves_compared": int(min(len(tg), len(tc))),
           "r_generator_state_identical": bool(np.array_equal(gpu["rng"], cpu["rng"])),
           "nuts_depth_leapfrogs_divergent_identical_last_iteration": bool(np.array_equal(gpu["row"][3:6], cpu["row"][3:6])),
           "max_rel_diff_stan_row_last_iteration": float(np.max(np.abs(gpu["row"] - cpu["row"]) / (1e-9 + np.abs(cpu["row"]))))}
    if not first:
        raise SystemExit("bench.py: from the state after the burn-in the CPU oracle's first iteration does not reproduce the GPU chain's tree moves: " + json.dumps(rec))
    return rec


def stationary_leg(lib, args, seed, burn, timed, nxt, profile_sweeps=0):
    """One chain in the reference's phase order (R/stan4bart_fit.R:49-51): `burn` warm-up iterations with adaptation, disengage, `timed` sampling
    iterations timed, `nxt` more to say whether the window was stationary (same leapfrogs per iteration within 20 %: the headline's rule)."""
    import torch
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    rng = RRng(seed)
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    s = Sampler(lib, "s4b_", args, rng.state)
    try:
        t0 = time.perf_counter()
        s.run(burn, True, 0)
        t_burn = time.perf_counter() - t0
        s.disengage_adaptation()
        n0 = s.get_nuts_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(timed, False, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n1 = s.get_nuts_stats()
        s.run(nxt, False, 0)
        n2 = s.get_nuts_stats()
        lf_win = (n1["sum_n_leapfrog"] - n0["sum_n_leapfrog"]) / timed
        lf_next = (n2["sum_n_leapfrog"] - n1["sum_n_leapfrog"]) / nxt
        rec = {"gpu_iters_per_sec": timed / dt, "ms_per_step": 1e3 * dt / timed, "gpu_seconds": dt, "n_leapfrog_per_step": lf_win, "hmc_mode": s.get_hmc_mode(),
               "tree_path": s.get_tree_path()[1],
               "stationarity": {"burn_in": burn, "burn_in_seconds": t_burn, "timed_iterations": timed, "n_leapfrog_per_step_timed_window": lf_win,
                                "n_leapfrog_per_step_next_iterations": lf_next, "next_iterations": nxt, "ratio": lf_win / lf_next if lf_next else None,
                                "stationary": bool(lf_next and 0.8 <= lf_win / lf_next <= 1.25)}}
        prof = s.profile_sweep(profile_sweeps) if profile_sweeps else None
        lf = s.profile_leapfrog(5) if profile_sweeps else None
    finally:
        s.free()
    return rec, prof, lf


def reference_r_leg(n, p, trees, iters):
    """bench/reference_cpu.R: the installed CRAN package on this box's host cores, when R and the package exist."""
    rscript = shutil.which("Rscript")
    if rscript is None:
        return {"status": "Rscript not found on this box: the CPU figure is the repo's C++ restatement (oracle/), not vdorie/stan4bart itself",
                "script": "bench/reference_cpu.R"}
    try:
        out = subprocess.run([rscript, os.path.join(ROOT, "bench", "reference_cpu.R"), str(n), str(p), str(trees), str(iters)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3600)
    except subprocess.TimeoutExpired:
        return {"status": "bench/reference_cpu.R timed out after 3600 s", "script": "bench/reference_cpu.R"}
    for line in out.stdout.splitlines()[::-1]:
        if line.startswith("{"):
            try:
                rec = json.loads(line)
                rec["script"] = "bench/reference_cpu.R"
                return rec
            except ValueError:
                pass
    return {"status": "bench/reference_cpu.R did not report (is the stan4bart package installed?): " + out.stderr.strip()[-300:], "script": "bench/reference_cpu.R"}


def self_launch(a, argv):
    """`python bench.py --gpus N` (N > 1) without a launcher: the chain fan-out of the reference (R/stan4bart_fit.R:498-533 starts one
    worker process per chain) needs one process per GPU, so this process — BEFORE any GPU call — starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child (never exec: a process that has
    touched the GPU must not be replaced) and relays its JSON line and exit code."""
    if not a.emul:
        import torch
        have = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
        if have < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but only {have} GPU(s) are visible: one chain per GPU needs one GPU per rank")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [("--num-obs" if x == "--n" else x) for x in argv]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if child.returncode != 0:
        raise SystemExit(child.returncode)
    if len(lines) != 1:
        raise SystemExit(f"the {a.gpus}-rank run printed {len(lines)} JSON lines, expected one")
    rec = json.loads(lines[0])
    if rec.get("n_gpus") != a.gpus:
        raise SystemExit(f"the run reports n_gpus = {rec.get('n_gpus')}, asked for --gpus {a.gpus}")
    print(lines[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--burn-in", type=int, default=1000, help="untimed warm-up-phase iterations before the W + K sampling-phase iterations (the reference's default warm-up)")
    ap.add_argument("--n", "--num-obs", dest="n", type=int, default=1_000_000)   # (--num-obs: torchrun's own parser trips over a bare --n)
    ap.add_argument("--p", type=int, default=50)
    ap.add_argument("--trees", type=int, default=200)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--no-hmc-mode1", action="store_true")
    ap.add_argument("--mode-iters", type=int, default=100, help="iterations of the like-for-like leg that times both gradient modes from one saved state")
    ap.add_argument("--profile-sweeps", type=int, default=2)
    ap.add_argument("--tree-path", default="auto", choices=["auto", "two-kernel", "fused", "persistent", "stream"])
    ap.add_argument("--c5-n", type=int, default=10_000_000, help="observations of the BASELINE config 5 leg of extra_configs (0 = skip; N = 1 only)")
    ap.add_argument("--target-burn-in", type=int, default=300, help="warm-up iterations of the n = --target-n leg before its sweeps are profiled")
    ap.add_argument("--c5-burn-in", type=int, default=1000, help="warm-up iterations of the config 5 leg before its 20 timed sampling iterations")
    ap.add_argument("--target-n", type=int, default=10_000_000,
                    help="also measure the sweep kernel at north_star's roofline-target size (0 = skip; N = 1 only)")
    ap.add_argument("--emul", action="store_true",
                    help="TEST ONLY (pytest -m 'not gpu'): run the N-rank plumbing of this script over the CPU emulation of the device layer "
                         "(tests/emul) and gloo; prints a line whose numbers mean nothing")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a, sys.argv[1:])
    t_start = time.perf_counter()

    import torch
    from stan4bart_amd import RRng
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd.fit import chain_seeds
    from stan4bart_amd.parallel import all_gather_array, dist_env, init_process_group

    # S4B_BENCH_BACKEND=gloo + S4B_BENCH_ONE_DEVICE=1: rehearsal of the N-rank path on a single-GPU box (all ranks share
    # device 0, the collectives run over gloo); the driver's real runs use the default: one GPU per rank, RCCL
    rank0, local0, world0 = dist_env()
    if world0 != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world0}")
    one_device = bool(os.environ.get("S4B_BENCH_ONE_DEVICE"))
    if not a.emul:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if not one_device and torch.cuda.device_count() < world0:
            raise SystemExit(f"WORLD_SIZE={world0} ranks but only {torch.cuda.device_count()} GPU(s) are visible: one chain per GPU needs one GPU per rank")
    cores = pin_rank_to_cores(local0, world0)       # before the runtimes start their helper threads
    rank, local_rank, world = init_process_group("gloo" if a.emul else os.environ.get("S4B_BENCH_BACKEND"))
    if one_device or a.emul:
        local_rank = 0
    if a.emul:
        lib, prefix = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "_build", "libs4b_emul.so")), "emu_"
    else:
        from stan4bart_amd._lib import load_library
        torch.cuda.set_device(local_rank)
        lib, prefix = load_library(), "s4b_"

    def barrier():
        if not a.emul:
            torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        if not a.emul:
            torch.cuda.synchronize()

    d = friedman_design(a.n, a.p, rank, world, barrier)
    t_design = time.perf_counter()
    # the NUTS adaptation windows are laid out over the warm-up phase (reference interruptable_sampler.hpp:171): warmup = burn-in
    total = a.burn_in + a.warmup + a.steps
    args = case_from_design(d, a.p, a.trees, local_rank, a.burn_in, total)
    rng = RRng(int(chain_seeds(20260101, max(1, world))[rank]))
    args.seed = int(rng.sample_int(2147483647, 1)[0])
    sampler = Sampler(lib, prefix, args, rng.state)     # uploads everything: inputs are HBM-resident from here on
    if one_device and world > 1 and not a.emul:
        sampler.set_device_sharing(world)     # rehearsal mode: all ranks share device 0 (persistent launches of different processes are sorted out by their roll call)
    sampler.set_tree_path(a.tree_path)
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
    state_after_burn_in = sampler.get_state() if (rank == 0 and world == 1 and not a.emul and not (a.no_cpu_baseline and a.no_hmc_mode1)) else None
    barrier()
    c0, s0 = sampler.get_counters(), sampler.get_nuts_stats()
    t0 = time.perf_counter()
    out = sampler.run(a.steps, False, 0)                # (run() returns synchronised)
    if not a.emul:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1, s1 = sampler.get_counters(), sampler.get_nuts_stats()
    barrier()
    hmc_mode = sampler.get_hmc_mode()
    tree_path = sampler.get_tree_path()[1] if not a.emul else "emulation"
    # ---- both gradient modes over the SAME iterations: the chain is saved (s4b_get_state), `--mode-iters` sampling iterations are
    # timed with sufficient statistics gathered once per iteration (mode 0, the headline's mode), the state is put back and the same
    # iterations are timed again with one O(N) device evaluation per leapfrog (mode 1, the reference's cost model).  The two legs
    # start from the same state and generator positions, so they do the same leapfrogs up to rounding.
    dt1 = None
    modes = None
    # ---- the headline's own K iterations once more with one O(N) device evaluation per leapfrog (hmc_mode 1): the chain goes back to the
    # state it had when the timed region began (same trees, same generator positions: the same leapfrogs up to rounding)
    window1 = None
    if state_after_burn_in is not None and not a.no_hmc_mode1:
        sampler.set_state(state_after_burn_in)
        sampler.set_hmc_mode(1)
        m0 = sampler.get_nuts_stats()
        t0w = time.perf_counter()
        sampler.run(a.steps, False, 0)
        torch.cuda.synchronize()
        dtw = time.perf_counter() - t0w
        window1 = {"value": a.steps / dtw, "unit": "Gibbs iterations/s/chain", "ms_per_step": 1e3 * dtw / a.steps,
                   "n_leapfrog_per_step": (sampler.get_nuts_stats()["sum_n_leapfrog"] - m0["sum_n_leapfrog"]) / a.steps,
                   "note": "hmc_mode 1 over the SAME iterations as the headline (chain put back to the state at the start of the timed region)"}
        sampler.set_hmc_mode(hmc_mode)
    if not a.no_hmc_mode1 and not a.emul:
        state0 = sampler.get_state()
        legs = {}
        for mode in (0, 1):
            sampler.set_state(state0)
            sampler.set_hmc_mode(mode)
            barrier()
            m0 = sampler.get_nuts_stats()
            t0 = time.perf_counter()
            sampler.run(a.mode_iters, False, 0)
            torch.cuda.synchronize()
            legs[mode] = (time.perf_counter() - t0, (sampler.get_nuts_stats()["sum_n_leapfrog"] - m0["sum_n_leapfrog"]) / a.mode_iters)
            barrier()
        sampler.set_hmc_mode(hmc_mode)
        dt1 = legs[1][0]
        modes = legs

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())
    dt_max = max_over_ranks(dt)
    dt1_max = max_over_ranks(dt1) if dt1 is not None else None
    # the only collective: chain summaries (last sigma, per-rank set-up seconds, per-rank timed seconds)
    summ = all_gather_array(np.array([float(out["bart"]["sigma"][-1]), t_design - t_start, t_created - t_design, dt, float(len(cores))]))

    prof = lf = probe = None
    if rank == 0 and not a.emul:
        prof = sampler.profile_sweep(a.profile_sweeps)
        lf = sampler.profile_leapfrog(20)
        po = (ctypes.c_double * 4)()   # measured streaming ceiling of this device (read-only and in-place update over 1 GB)
        lib.s4b_stream_probe.restype = ctypes.c_int
        if lib.s4b_stream_probe(ctypes.c_int32(local_rank), ctypes.c_int64(1 << 27), ctypes.c_int32(5), po) == 0:
            probe = {"read_GBs": po[0], "update_in_place_GBs": po[1]}
    fused_stats = sampler.get_fused_stats()
    sweep_spec, sweep_busy = (sampler.get_sweep_spec(), sampler.get_sweep_busy()) if not a.emul else ((0, 0, 0, 0), 0)
    gpu_chain = None
    if state_after_burn_in is not None and not a.no_cpu_baseline:
        # the iterations the CPU leg below runs from the same state, once more on the GPU with the tree-move trace on (untimed)
        sampler.set_state(state_after_burn_in)
        sampler.set_hmc_mode(hmc_mode)
        sampler.set_trace(True)
        og = sampler.run(a.cpu_iters, False, 0)
        gpu_chain = {"trace": sampler.get_trace(), "row": og["stan"][:, -1].copy(), "rng": sampler.get_r_rng_state()}
        sampler.set_trace(False)
    sampler.free()
    target = None
    if world == 1 and a.target_n > 0 and a.target_n != a.n and not a.emul:
        target = target_roofline_leg(lib, a.target_n, a.p, a.trees, local_rank, a.profile_sweeps, a.target_burn_in)

    if rank == 0:
        n = a.n
        per_chain = a.steps / dt_max
        trans = max(1, s1["transitions"] - s0["transitions"])
        per_rank = [a.steps / float(s[3]) for s in summ]
        rec = {
            "metric": "gibbs_iters_per_sec", "value": per_chain * world, "unit": "Gibbs iterations/s (all chains)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_max / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_chain": per_chain, "per_chain_by_rank": per_rank,
            "phase_timed": f"sampling (after {a.burn_in} untimed warm-up iterations with adaptation, disengage, {a.warmup} untimed sampling iterations)",
            "warmup_phase_iters_per_sec": warm_rate,
            "config": {"workload": f"Friedman n={n}, p={a.p}, ntree={a.trees}, (1+X4|g.1)+(1|g.2), one chain per GPU"
                                   + (" (BASELINE config 3)" if (n, a.p, a.trees) == (1_000_000, 50, 200) else ""),
                       "chains": world, "hmc_mode": hmc_mode,
                       "hmc_mode_meaning": "0: O(N) sums folded into sufficient statistics once per Gibbs iteration (leapfrogs cost O(nnz Gram) on the host); 1: one O(N) device evaluation per leapfrog",
                       "tree_path": tree_path, "burn_in": a.burn_in, "source_sha16": source_sha16(),
                       "n_leapfrog_timed": int(s1["sum_n_leapfrog"] - s0["sum_n_leapfrog"]),
                       "n_leapfrog_per_step": (s1["sum_n_leapfrog"] - s0["sum_n_leapfrog"]) / a.steps,
                       "mean_treedepth_timed": (s1["sum_treedepth"] - s0["sum_treedepth"]) / trans,
                       "divergent_timed": int(s1["divergent"] - s0["divergent"]),
                       "gradient_evals_timed": int(c1[0] - c0[0]), "tree_updates_timed": int(c1[1] - c0[1]),
                       "fused_sum_evaluations": fused_stats[0], "fused_sum_fallbacks_to_doubles": fused_stats[1],
                       "persistent_launches": {"ran": sweep_spec[0], "tree_updates_inside": sweep_spec[1], "published_before_the_verdict": sweep_spec[2],
                                               "borne_out": sweep_spec[3], "found_the_device_busy": sweep_busy},
                       "sigma_last": [float(s[0]) for s in summ],
                       "host_cores_per_rank": [int(s[4]) for s in summ],
                       "setup_seconds_per_rank": {"design (rank 0 generates, others load)": [float(s[1]) for s in summ],
                                                  "create (binning, upload, init)": [float(s[2]) for s in summ]}},
        }
        if world > 1:
            # scaling efficiency against the committed single-GPU line of this round (the driver computes its own from its N = 1 run)
            try:
                with open(os.path.join(ROOT, "profiles", "bench_n1_reference.json")) as f:
                    ref1 = json.load(f)
                rc = ref1.get("config", {})
                same = (rc.get("workload") == rec["config"]["workload"] and rc.get("tree_path") == tree_path and rc.get("burn_in") == a.burn_in
                        and rc.get("source_sha16") == rec["config"]["source_sha16"])
                if same:
                    rec["scaling_efficiency_vs_committed_n1"] = {"value": min(per_rank) / ref1["per_chain"], "n1_per_chain": ref1["per_chain"],
                                                                 "definition": "slowest rank's per-chain rate / per-chain rate of profiles/bench_n1_reference.json"}
                else:
                    rec["scaling_efficiency_vs_committed_n1"] = {"value": None, "status": "profiles/bench_n1_reference.json was made by other sources, another tree path or "
                                                                 "another burn-in: refused (the driver computes the efficiency from its own N = 1 run)"}
            except (OSError, ValueError, KeyError):
                pass
        if dt1_max is not None:
            lf_win, lf_next = rec["config"]["n_leapfrog_per_step"], modes[0][1]
            rec["config"]["stationarity"] = {
                "n_leapfrog_per_step_timed_window": lf_win, "n_leapfrog_per_step_next_iterations": lf_next, "next_iterations": a.mode_iters,
                "ratio": lf_win / lf_next if lf_next else None, "stationary": bool(lf_next and 0.8 <= lf_win / lf_next <= 1.25),
                "rate_next_iterations_same_mode": a.mode_iters / modes[0][0],
                "note": "the timed window and the chain's next iterations (hmc_mode 0, from the state the window left) must cost the same leapfrogs per "
                        "iteration (within 20 %) for `value` to be a figure of the stationary regime"}
            rec["gradient_modes_same_iterations"] = {
                "iterations": a.mode_iters, "from": "one saved state (s4b_get_state / s4b_set_state), sampling phase, rank 0's chain",
                "hmc_mode0": {"iters_per_sec": a.mode_iters / modes[0][0], "ms_per_step": 1e3 * modes[0][0] / a.mode_iters, "n_leapfrog_per_step": modes[0][1]},
                "hmc_mode1": {"iters_per_sec": a.mode_iters / modes[1][0], "ms_per_step": 1e3 * modes[1][0] / a.mode_iters, "n_leapfrog_per_step": modes[1][1]}}
            rec["per_chain_hmc_mode1"] = {"value": a.mode_iters / dt1_max, "unit": "Gibbs iterations/s/chain", "ms_per_step": 1e3 * dt1_max / a.mode_iters,
                                          "n_leapfrog_per_step": modes[1][1],
                                          "note": "hmc_mode 1 (every leapfrog launches k_stan_fused over all N observations) over the iterations of gradient_modes_same_iterations"}
        if window1 is not None:
            rec["per_chain_hmc_mode1_headline_window"] = window1
        if prof is not None:
            # HBM traffic of the tree kernel: REPLAYED from the PMC passes committed under profiles/ (rocprofv3 cannot wrap this
            # process from inside; collected on this command line by tools/profile_round.sh, see profiles/pmc_traffic.json)
            traffic, traffic_note = None, None
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pj = json.load(f)
                ent = pj.get(str(n), {})
                if ent.get("kernel", "").startswith(DOMINANT[tree_path]):
                    traffic = ent.get("bytes_per_launch")
                    traffic_note = "replayed from profiles/pmc_traffic.json (" + str(ent.get("kernel")) + "), not measured in this run"
            except (OSError, ValueError):
                pass
            rf = sweep_roofline(prof, tree_path, n, a.trees)
            rf["traffic"], rf["traffic_note"] = traffic, traffic_note
            rf["measured_stream"] = probe
            rf["frac_of_measured_update_stream"] = (rf["achieved"] / probe["update_in_place_GBs"]) if probe else None
            rec["roofline"] = rf
            # second kernel group of the path: the O(N) sums one leapfrog costs when the gradient is evaluated on the device
            rec["roofline_hmc"] = {"bound": "hbm", "kernel": "k_stan_fused (direct)",
                                   "achieved": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": lf["algorithmic_bytes"] / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "avg_eval_us": lf["kernels_us"], "avg_eval_us_with_result_fetch": lf["with_fetch_us"],
                                   "launches_per_eval": lf["launches"], "algorithmic_bytes_per_eval": lf["algorithmic_bytes"],
                                   "bytes_the_kernel_reads_per_eval": float(n) * (8 * 2 + 12 * 3 + 8),
                                   "frac_against_bytes_read": float(n) * (8 * 2 + 12 * 3 + 8) / (lf["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "note": "frac: N (8K + 12z + 20) bytes per leapfrog (SURVEY 8d B_lf) / time; the DIRECT kernel reads e0 = y - offset instead of y and offset and "
                                           "no row pointers (fixed row length): N (8K + 12z + 8) bytes really move, frac_against_bytes_read is what HBM sustains; "
                                           "in the timed region of per_chain_hmc_mode1, not of the headline"}
        if target is not None:
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    ent = json.load(f).get(str(a.target_n), {})
                target["traffic"] = ent.get("bytes_per_launch") if ent.get("kernel", "").startswith(target["kernel"]) else None
                target["traffic_note"] = "replayed from profiles/pmc_traffic.json, not measured in this run"
            except (OSError, ValueError):
                target["traffic"] = None
            if probe:
                target["frac_of_measured_update_stream"] = target["achieved"] / probe["update_in_place_GBs"]
            rec["roofline_target_config"] = target
        if state_after_burn_in is not None and not a.no_cpu_baseline:
            olib = oracle_lib()
            cargs = case_from_design(d, a.p, a.trees, 0, a.burn_in, total)
            v, secs, lfc, cpu_chain = cpu_baseline_from_state(olib, cargs, state_after_burn_in, a.cpu_iters)
            rec["cpu_baseline"] = {"value": v, "unit": "Gibbs iterations/s/chain", "cores": 1, "kind": "port",
                                   "same_chain_as_the_gpu": chain_check(gpu_chain, cpu_chain, a.trees),
                                   "sample": f"same workload (n={n}, p={a.p}, ntree={a.trees}), {a.cpu_iters} sampling-phase Gibbs iterations started from the GPU chain's state "
                                             f"after the burn-in (adapted step size and metric; {lfc:.1f} leapfrogs per iteration, each O(N) as in the reference), "
                                             f"{secs:.1f} s of single-thread CPU time, host has {os.cpu_count()} cores",
                                   "reference_R_package": reference_r_leg(n, a.p, a.trees, a.cpu_iters)}
        if world == 1 and not a.no_extra_configs and not a.emul:
            # BASELINE configs 1 and 2: small and quick, reported beside the headline
            from stan4bart_amd import GroupTerm, generate_friedman_data, make_sampler_args
            olib = oracle_lib()
            extra = {}
            d1 = generate_friedman_data(100, ranef=True, causal=True, p=10)
            x1 = d1["x"]
            a1 = make_sampler_args(d1["y"], x1[:, [j for j in range(10) if j != 3]], X=np.column_stack([x1[:, 3], d1["z"]]),
                                   groups=[GroupTerm(d1["g1"], None, "g.1"), GroupTerm(d1["g2"], None, "g.2")], iter=400, warmup=200,
                                   keep_fits=False, bart_args={"n.trees": 50})
            v1, s1c = time_chain(olib, "orc_", a1, 12345, 100, 300)
            extra["config1"] = {"workload": "Friedman n=100, ntree=50, (1|g.1)+(1|g.2) (BASELINE config 1, CPU plumbing case)",
                                "cpu_port_iters_per_sec": v1, "cpu_seconds": s1c}
            # configs 2, 4, 5 on the GPU: each in the reference's phase order (warm-up with adaptation, disengage, sample) and timed in its
            # STATIONARY regime like the headline, with its own `stationarity` record and the roofline of the chain that was timed
            d2 = generate_friedman_data(100_000, ranef=False, causal=True, p=10)
            a2c = case_from_design(d2, 10, 200, 0, 0, 40, ranef=False)
            v2c, s2c = time_chain(olib, "orc_", a2c, 12345, 2, 30)
            a2g = case_from_design(d2, 10, 200, local_rank, 1000, 1400, ranef=False)
            r2, p2, _ = stationary_leg(lib, a2g, 12345, 1000, 200, 100, profile_sweeps=a.profile_sweeps)
            r2.update({"workload": "Friedman n=1e5, p=10, ntree=200, fixed effects only (BASELINE config 2)",
                       "cpu_port_iters_per_sec": v2c, "cpu_seconds": s2c, "cpu_cores": 1, "cpu_port_note": "30 warm-up-phase iterations from the start of a chain",
                       "roofline_tree_kernel": sweep_roofline(p2, r2["tree_path"], 100_000, 200)})
            extra["config2"] = r2
            # BASELINE config 4 at its shape: the 747 x 25 IHDP covariates, 26-level group with a random slope on the treatment,
            # binary outcome / probit link (the latents are drawn on the device from R's stream), counterfactual test rows
            from stan4bart_amd.cases import c5_case, ihdp_case
            a4 = ihdp_case(warmup=1000, iter=1600, T=75)
            a4.device = local_rank
            r4, p4, _ = stationary_leg(lib, a4, 12345, 1000, 400, 100, profile_sweeps=a.profile_sweeps)
            r4.update({"workload": "IHDP shape: n=747, 25 covariates + treatment, ntree=75, (1+z|g1) with 26 levels, binary outcome / probit link, "
                                   "747 counterfactual test rows (BASELINE config 4; the outcome is this repository's surrogate, DESIGN.md 7)",
                       "roofline_tree_kernel": sweep_roofline(p4, r4["tree_path"], 747, 75)})
            extra["config4"] = r4
            if a.c5_n > 0:
                # BASELINE config 5 ("HBM-roofline run"): n = 1e7, P = 100, ntree = 400, 200 groups with random slopes (q = 400, z = 2)
                t5 = time.perf_counter()
                a5, xb5 = c5_case(a.c5_n, warmup=a.c5_burn_in, iter=a.c5_burn_in + 60, keep_fits=False, device=local_rank)
                del xb5
                t5c = time.perf_counter()
                r5, p5, l5 = stationary_leg(lib, a5, 777, a.c5_burn_in, 20, 20, profile_sweeps=1)
                rf5 = sweep_roofline(p5, r5["tree_path"], a.c5_n, 400)
                K5, z5 = 2, 2
                read5 = float(a.c5_n) * (8 * K5 + 12 * z5 + 8)
                r5.update({"workload": f"n={a.c5_n}, P=100, ntree=400, (1+X4|g.1) with 200 groups: q=400, z=2 (BASELINE config 5)",
                           "setup_seconds": {"data + binning on the host": t5c - t5},
                           "roofline_tree_kernel": rf5,
                           "roofline_hmc": {"kernel": "k_stan_fused (direct), K=2, z=2, 400-column Z'e histogram in LDS", "avg_eval_us": l5["kernels_us"],
                                            "avg_eval_us_with_result_fetch": l5["with_fetch_us"],
                                            "algorithmic_bytes_per_eval": l5["algorithmic_bytes"],
                                            "frac": l5["algorithmic_bytes"] / (l5["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                            "bytes_the_kernel_reads_per_eval": read5,
                                            "frac_against_bytes_read": read5 / (l5["kernels_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}})
                extra["config5"] = r5
            rec["extra_configs"] = extra
        if rec["n_gpus"] != a.gpus:
            raise SystemExit(f"internal: n_gpus {rec['n_gpus']} != --gpus {a.gpus}")
        print(json.dumps(rec))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
