"""Process-group plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl").

Replaces the slice of python/sglang/multimodal_gen/runtime/distributed/parallel_state.py:310-472
the DiT path needs.  Flux scales by DATA parallelism over the CFG x request batch: requests are
independent, so there is no collective inside a denoise step, only a scatter of the work list
and a gather of the final latents.  Ulysses all-to-all (layers/usp.py:49-158) is provided for the
sequence-parallel attention wrapper.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun contract); returns (rank, world, local)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_requests(num_requests: int, world_size: int, rank: int) -> range:
    """Static contiguous split of the request list; the first `num_requests % world_size`
    ranks take one extra request.  A request's CFG branches stay on one rank, so the CFG
    combine (pipelines_core/stages/denoising.py:1486-1488) needs no communication."""
    base, rem = divmod(num_requests, world_size)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_latents(local: torch.Tensor, counts: Sequence[int], group=None) -> Optional[torch.Tensor]:
    """Gather per-rank final latents [n_r, ...] to rank 0 (once per batch).  Ranks may hold
    different request counts: pad to the maximum, all_gather, trim."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    nmax = max(counts)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    if dist.get_rank(group) != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def ulysses_all_to_all(x: torch.Tensor, group, scatter_dim: int, gather_dim: int) -> torch.Tensor:
    """[B, S_local, H, D] <-> [B, S, H_local, D] exchange (layers/usp.py:49-158): split
    `scatter_dim` into P parts, send part p to rank p, concatenate what arrives along
    `gather_dim`.  One all_to_all_single per tensor; over xGMI every rank talks to all 7
    peers at once, which is the pattern the point-to-point fabric is built for."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    if x.shape[scatter_dim] % P:
        raise ValueError(f"dim {scatter_dim} of size {x.shape[scatter_dim]} not divisible by {P}")
    parts = torch.stack(x.chunk(P, dim=scatter_dim), dim=0).contiguous()
    recv = torch.empty_like(parts)
    dist.all_to_all_single(recv, parts, group=group)
    return torch.cat(list(recv.unbind(0)), dim=gather_dim)
