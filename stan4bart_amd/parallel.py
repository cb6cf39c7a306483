"""Chain fan-out: one process per GPU, one chain per process (reference R/stan4bart_fit.R:495-533 runs
chains as independent PSOCK worker processes; nothing is exchanged while sampling).

`torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU) is used only for the
rendezvous, the timing barrier and one end-of-run all-gather of per-chain draws/summaries — there is no
data-path collective because the chains share nothing.
"""
from __future__ import annotations

import os

# read by ROCr when the runtime initialises, i.e. it must be in place before anything touches HIP (the host driver only
# supports dmabuf IPC; RCCL fails with `hipIpcGetMemHandle: invalid argument` without it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from typing import Callable, Optional

import numpy as np

from .fit import chain_seeds, fit_worker, make_sampler_args
from .rcompat import RRng


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def all_gather_array(x: np.ndarray) -> list:
    """All-gather one equally-shaped float64 array per rank (RCCL when the group is nccl)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [np.asarray(x)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def run_chains_distributed(make_sampler: Callable, seed: int, y, x_bart, **kw) -> dict:
    """Each rank fits chain `rank` with the reference's parallel seeding rule (R/stan4bart_fit.R:515-516),
    then the draws of the Stan block are all-gathered; every rank returns the same stacked array
    [chains, num_pars, samples] plus split R-hat per parameter."""
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    seeds = chain_seeds(seed, world)
    rng = RRng(int(seeds[rank]))
    args = make_sampler_args(y, x_bart, device=local_rank, **kw)
    res = fit_worker(make_sampler, args, rng)
    draws = np.stack(all_gather_array(res["sample"]["stan"]))
    sigma_mean = np.stack(all_gather_array(np.array([res["sample"]["bart"]["sigma"].mean()])))
    out = dict(local=res, stan=draws, sigma_mean=sigma_mean[:, 0], rhat=split_rhat(draws), par_names=res["par_names"])
    if "bart_state" in res:   # the kept trees of every chain, as exported byte strings (any rank can rebuild predict-capable samplers)
        states = [None] * world
        if world > 1:
            dist.all_gather_object(states, res["bart_state"])
        else:
            states = [res["bart_state"]]
        out["bart_states"] = states
    return out


def split_rhat(draws: np.ndarray) -> np.ndarray:
    """Split-R-hat over chains for every parameter row (draws: [chains, pars, samples])."""
    c, p, s = draws.shape
    h = s // 2
    if h < 2:
        return np.full(p, np.nan)
    parts = np.concatenate([draws[:, :, :h], draws[:, :, h:2 * h]], axis=0)   # [2c, p, h]
    w = parts.var(axis=2, ddof=1).mean(axis=0)
    b = h * parts.mean(axis=2).var(axis=0, ddof=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.sqrt(((h - 1) / h * w + b / h) / w)
