"""Chain fan-out over torch.distributed (gloo, world size 2, CPU): every rank fits its own chain with the
reference's parallel seeding rule (R/stan4bart_fit.R:515-516), then the draws are all-gathered.  The device
layer is the CPU emulation (tests/emul) because there is no GPU here; on the GPU box the same code runs with the
HIP library and backend "nccl" (RCCL) — see bench.py."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, friedman_case


def _worker(rank, world, port, tmpdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from stan4bart_amd import generate_friedman_data, GroupTerm
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd.parallel import init_process_group, run_chains_distributed
    init_process_group("gloo")
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "_build", "libs4b_emul.so"))
    d = generate_friedman_data(100, ranef=True, causal=True)
    x = d["x"]
    res = run_chains_distributed(lambda a, st: Sampler(lib, "emu_", a, st), 777, d["y"], x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]],
                                 X=np.column_stack([x[:, 3], d["z"]]), groups=[GroupTerm(d["g1"]), GroupTerm(d["g2"])],
                                 iter=13, warmup=7, bart_args={"n.trees": 11, "keepTrees": True})
    # every rank can predict from every chain's kept trees (exported, gathered, rebuilt as stored samplers)
    from stan4bart_amd.abi import StoredSampler
    preds = [StoredSampler(lib, "emu_", st).predict_bart(x[:5, [0, 1, 2, 4, 5, 6, 7, 8, 9]]) for st in res["bart_states"]]
    np.save(os.path.join(tmpdir, f"pred_{rank}.npy"), np.stack(preds))
    np.save(os.path.join(tmpdir, f"train_{rank}.npy"), res["local"]["sample"]["bart"]["train"][:5])
    np.save(os.path.join(tmpdir, f"stan_{rank}.npy"), res["stan"])
    np.save(os.path.join(tmpdir, f"local_{rank}.npy"), res["local"]["sample"]["stan"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_chains_two_ranks_gloo(emul_lib, tmp_path):
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "stan_0.npy"), np.load(tmp_path / "stan_1.npy")
    assert g0.shape[0] == 2 and np.array_equal(g0, g1)                       # every rank holds both chains
    for r in range(2):
        assert np.array_equal(g0[r], np.load(tmp_path / f"local_{r}.npy"))   # chain r came from rank r
    assert not np.array_equal(g0[0], g0[1])                                  # different seeds => different chains
    p0, p1 = np.load(tmp_path / "pred_0.npy"), np.load(tmp_path / "pred_1.npy")
    assert p0.shape == (2, 5, 6) and np.array_equal(p0, p1)                  # [chain, row, kept draw] on both ranks
    for r in range(2):                                                       # predict(training rows) == training fit of chain r
        np.testing.assert_allclose(p0[r], np.load(tmp_path / f"train_{r}.npy"), rtol=1e-9, atol=1e-9)
    # the same chains fitted serially with the same per-chain seeds are identical
    from stan4bart_amd import GroupTerm, RRng, generate_friedman_data, make_sampler_args, fit_worker
    from stan4bart_amd.abi import Sampler
    from stan4bart_amd.fit import chain_seeds
    d = generate_friedman_data(100, ranef=True, causal=True)
    x = d["x"]
    for r in range(2):
        rng = RRng(int(chain_seeds(777, 2)[r]))
        args = make_sampler_args(d["y"], x[:, [0, 1, 2, 4, 5, 6, 7, 8, 9]], X=np.column_stack([x[:, 3], d["z"]]),
                                 groups=[GroupTerm(d["g1"]), GroupTerm(d["g2"])], iter=13, warmup=7, bart_args={"n.trees": 11, "keepTrees": True}, device=r)
        res = fit_worker(lambda a, st: Sampler(emul_lib, "emu_", a, st), args, rng)
        assert np.array_equal(res["sample"]["stan"], g0[r])


def test_split_rhat():
    from stan4bart_amd.parallel import split_rhat
    rng = np.random.default_rng(0)
    same = rng.normal(size=(4, 3, 200))
    assert np.all(np.abs(split_rhat(same) - 1.0) < 0.05)
    shifted = same + np.arange(4)[:, None, None] * 3.0
    assert np.all(split_rhat(shifted) > 1.5)


def test_bench_two_ranks_over_gloo(emul_lib):
    """The torchrun line the driver uses for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`), here
    with world size 2 over gloo and the CPU emulation of the device layer (`--emul`, test only): rendezvous on 127.0.0.1, the design
    generated once by rank 0 and shared through /dev/shm, per-rank core pinning, barrier + max-over-ranks timing, the end-of-run
    all-gather, ONE JSON line from rank 0 with the contract's fields."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--burn-in", "4", "--num-obs", "300", "--p", "10", "--trees", "5",
           "--emul", "--no-extra-configs", "--no-cpu-baseline", "--target-n", "0"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["config"]["chains"] == 2
    assert len(rec["per_chain_by_rank"]) == 2 and abs(rec["value"] - 2 * rec["per_chain"]) < 1e-9
    assert len(rec["config"]["sigma_last"]) == 2 and rec["config"]["sigma_last"][0] != rec["config"]["sigma_last"][1]   # two different chains
    cores = rec["config"]["host_cores_per_rank"]
    assert len(cores) == 2 and (os.cpu_count() < 2 or cores[0] <= os.cpu_count() // 2)      # disjoint core sets


def test_bench_refuses_more_ranks_than_gpus():
    """one chain per GPU: WORLD_SIZE above the visible device count must fail loudly, not oversubscribe a device"""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and ("MI355X" in out.stderr or "GPU" in out.stderr)


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_launches_its_own_ranks(emul_lib):
    """`python bench.py --gpus 2` with NO launcher on the command line (the form the driver uses for N = 1): the parent starts the
    two ranks itself (torch.distributed.run as a child process, before anything touches the GPU), relays the one JSON line and
    checks n_gpus == --gpus.  Reference: the chain fan-out of R/stan4bart_fit.R:498-533."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--burn-in", "4", "--n", "300", "--p", "10",
           "--trees", "5", "--emul", "--no-extra-configs", "--no-cpu-baseline", "--target-n", "0"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=dict(_clean_env(), OMP_NUM_THREADS="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["chains"] == 2 and len(rec["per_chain_by_rank"]) == 2


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N with fewer than N visible GPUs exits non-zero — it must never run fewer chains than asked for and print n_gpus: 1"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert out.returncode != 0 and "GPU" in out.stderr and not any(ln.startswith("{") for ln in out.stdout.splitlines())


def test_bench_refuses_world_size_mismatch():
    """under a launcher, WORLD_SIZE must equal --gpus (also WORLD_SIZE = 1 with --gpus 2)"""
    import subprocess
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--emul"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_rocr_environment_is_set_before_any_gpu_call():
    """HSA_ENABLE_IPC_MODE_LEGACY is read when ROCr initialises: bench.py and parallel.py must set it at import time, not after
    torch.cuda.* has run"""
    import subprocess
    code = ("import os; os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); import stan4bart_amd.parallel; "
            "assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'")
    assert subprocess.run([sys.executable, "-c", code], cwd=ROOT).returncode == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY"') < src.index("import numpy")
