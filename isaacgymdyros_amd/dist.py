"""Multi-GPU layout: one process per GPU, each with its own block of environments (the reference's Horovod
layout, utils/rlgames_utils.py:71-81).  Environments are physically independent, so `step` needs no exchange;
the only collective is a logging one, issued once per rollout horizon: an all-gather of finished-episode
statistics over RCCL/xGMI (backend "nccl" on ROCm) -- or gloo on CPU tensors in the tests.  Optional, also once per horizon:
`sync_perturbation_gate`, which lets the ranks' push-perturbation gates latch on the GLOBAL population means, as one simulation of
all the envs would."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import abi


def init_from_env(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total_envs: int, rank: int, world: int):
    """Contiguous block of env ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_episode_stats(env_state: torch.Tensor) -> torch.Tensor:
    """[4] float64 on env_state's device: (sum of last finished-episode returns over envs that finished at least
    one episode, sum of their lengths, number of such envs, total finished episodes)."""
    ret = abi.es_view(env_state, "last_episode_return").double()
    length = abi.es_view(env_state, "epi_len_log").double()
    eps = abi.es_view(env_state, "episodes_finished")
    done = eps > 0
    return torch.stack([(ret * done).sum(), (length * done).sum(), done.sum().double(), eps.sum().double()])


def gather_episode_stats(env_state: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of the per-rank statistics: returns [world, 4] on every rank (logging only; 32 B per rank)."""
    local = local_episode_stats(env_state)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local.unsqueeze(0)
    out = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, local, group=group)
    return torch.stack(out)


def summarize(stats: torch.Tensor) -> dict:
    s = stats.sum(0).tolist()
    n = max(s[2], 1.0)
    return {"mean_episode_return": s[0] / n, "mean_episode_length": s[1] / n, "envs_with_episode": int(s[2]),
            "episodes": int(s[3])}


def sync_perturbation_gate(gate_acc: torch.Tensor, steps_done: int, num_envs: int, max_episode_length: float = 8000.0,
                           pert_period: float = 2000.0, group=None) -> torch.Tensor:
    """The one cross-env coupling of the step is the perturbation gate: pushes start once mean(epi_len_log) > max_episode_length - period
    (6000) AND mean(contact_reward_mean) > 0.165 over ALL envs (tasks/dyros_dynamic_walk.py:489), and stay on (the latch).  Sharded, every rank
    takes those means over its own envs -- the reference's Horovod layout.  This is the optional emulation of ONE simulation of all the envs
    (SURVEY.md section 8e): an all-reduce of the newest step's two sums and the env count (3 doubles), the condition evaluated on the global
    means, the latch word of every rank's gate_acc set if it holds.  Call it once per rollout horizon, after `steps_done` steps (the sums of
    step s are in slot s % 3 of gate_acc: include/dyros_walk.h DW_GATE_*).  No host sync: everything stays on gate_acc's device.
    Returns the 0-d bool tensor `gate is open globally`."""
    nb, latch = abi.K["DW_GATE_BUCKETS"], abi.K["DW_GATE_LATCH"]
    if gate_acc.dtype != torch.int64 or gate_acc.numel() <= latch:
        raise ValueError("sync_perturbation_gate: gate_acc must be the int64 [DW_GATE_WORDS] buffer of the env")
    if steps_done < 1:
        raise ValueError("sync_perturbation_gate: no step has filled a slot yet")
    slot = (int(steps_done) - 1) % 3
    sums = gate_acc[slot * nb * 2:(slot + 1) * nb * 2].view(nb, 2).sum(0).double()          # (sum of epi_len_log, sum of contact_reward_mean * 2^32)
    vec = torch.cat([sums, torch.tensor([float(num_envs)], dtype=torch.float64, device=gate_acc.device)])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, group=group)
    is_open = (vec[0] / vec[2] > float(max_episode_length) - float(pert_period)) & (vec[1] / 4294967296.0 / vec[2] > 0.165)
    gate_acc[latch] = torch.maximum(gate_acc[latch], is_open.to(torch.int64))
    return is_open
