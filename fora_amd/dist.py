"""Multi-GPU plumbing: one process per GPU, sources sharded i mod G, no data-path
collective for `query`; one all-gather of fixed-size top-k lists for `topk`
(SURVEY.md 8e).  torch.distributed is plumbing only (backend "nccl" is RCCL on ROCm;
"gloo" in the CPU tests)."""
import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_sources(sources, rank, world):
    """source i of the global query list runs on rank i mod world (query.h:1471-1476 has no
    cross-query state, so any partition is valid)."""
    sources = np.asarray(sources)
    return sources[rank::world].copy()


def unshard_index(total, world):
    """position in the rank-major gathered layout of each global query index"""
    per = (total + world - 1) // world
    g = np.arange(total)
    return (g % world) * per + (g // world)


def gather_topk(ids, scores, total, rank, world, device=None):
    """All-gather per-rank [nq_local, k] top-k lists (padded to ceil(total/world) rows) and
    return them in global query order on every rank."""
    import torch
    import torch.distributed as dist

    k = ids.shape[1]
    per = (total + world - 1) // world
    pad_ids = np.zeros((per, k), dtype=np.int32)
    pad_sc = np.zeros((per, k), dtype=np.float64)
    pad_ids[: ids.shape[0]] = ids
    pad_sc[: scores.shape[0]] = scores
    dev = device if device is not None else "cpu"
    t_ids = torch.from_numpy(pad_ids).to(dev)
    t_sc = torch.from_numpy(pad_sc).to(dev)
    if world > 1:
        out_ids = torch.empty((world * per, k), dtype=t_ids.dtype, device=dev)
        out_sc = torch.empty((world * per, k), dtype=t_sc.dtype, device=dev)
        dist.all_gather_into_tensor(out_ids, t_ids) if dev != "cpu" else dist.all_gather(
            list(out_ids.chunk(world)), t_ids)
        dist.all_gather_into_tensor(out_sc, t_sc) if dev != "cpu" else dist.all_gather(
            list(out_sc.chunk(world)), t_sc)
    else:
        out_ids, out_sc = t_ids, t_sc
    pos = unshard_index(total, world)
    return out_ids.cpu().numpy()[pos], out_sc.cpu().numpy()[pos]


def max_over_ranks(value, world, device=None):
    import torch.distributed as _d
    if world == 1 and not (_d.is_available() and _d.is_initialized()):
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, world, device=None):
    import torch.distributed as _d
    if world == 1 and not (_d.is_available() and _d.is_initialized()):
        return [float(v) for v in values]
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def min_over_ranks(value, world, device=None):
    import torch.distributed as _d
    if world == 1 and not (_d.is_available() and _d.is_initialized()):
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())
