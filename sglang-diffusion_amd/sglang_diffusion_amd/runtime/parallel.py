"""Process-group plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl").

Replaces the slice of python/sglang/multimodal_gen/runtime/distributed/parallel_state.py:310-472
the DiT path needs.  Flux scales by DATA parallelism over the CFG x request batch: requests are
independent, so there is no collective inside a denoise step, only a scatter of the work list
and a gather of the final latents.  Ulysses all-to-all (layers/usp.py:49-158) is provided for the
sequence-parallel attention wrapper, and ring attention (layers/usp.py:161-252) for sequences
that are sharded over the keys as well.
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


def _hip_attention(q, k, v, scale):
    from .. import _C

    B, S, H, D = q.shape
    out = torch.empty_like(q)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    _C.attention(q, k, v, out, lse, scale)
    return out, lse


def _hip_merge(out, lse, a, a_lse, b, b_lse):
    from .. import _C

    _C.merge_attn_states(out, lse, a, a_lse, b, b_lse, lse_layout="bhs")


def ring_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, group=None,
                   softmax_scale: Optional[float] = None, is_causal: bool = False,
                   attn_fn=None, merge_fn=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense attention of the local queries over the keys of EVERY rank of `group`
    (layers/usp.py:161-252, which drives torch's _templated_ring_attention).

    q, k, v: this rank's shards [B, S_local, H, D] (bf16).  In step s the rank attends to the
    K/V shard that started on rank (r - s) mod P while that shard is already on its way to rank
    r + 1 (point-to-point isend/irecv: one xGMI link per direction, overlapped with the
    attention kernel), and folds the partial result into the running one with its
    log-sum-exp (mi355_merge_attn_states).  The layout stays [B, S, H, D] throughout: the
    kernels take strides, so the reference's [B, H, S, D] permutes have no counterpart.
    Returns (out [B, S_local, H, D], lse [B, H, S_local]).

    attn_fn(q, k, v, scale) -> (out, lse [B, H, S]) and merge_fn(out, lse, a, a_lse, b, b_lse)
    default to the HIP kernels; the multi-process CPU tests pass reference callables."""
    if is_causal:
        raise NotImplementedError("ring attention: causal masking is not built (the DiT path is dense)")
    attn_fn = attn_fn or _hip_attention
    merge_fn = merge_fn or _hip_merge
    scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
    P = dist.get_world_size(group) if dist.is_initialized() else 1
    if P == 1:
        return attn_fn(q, k, v, scale)
    rank = dist.get_rank(group)
    nxt = dist.get_global_rank(group, (rank + 1) % P) if group is not None else (rank + 1) % P
    prv = dist.get_global_rank(group, (rank - 1) % P) if group is not None else (rank - 1) % P
    k_cur, v_cur = k.contiguous(), v.contiguous()
    out = lse = None
    for step in range(P):
        reqs = []
        if step + 1 < P:
            k_nxt, v_nxt = torch.empty_like(k_cur), torch.empty_like(v_cur)
            ops = [dist.P2POp(dist.isend, k_cur, nxt, group), dist.P2POp(dist.isend, v_cur, nxt, group),
                   dist.P2POp(dist.irecv, k_nxt, prv, group), dist.P2POp(dist.irecv, v_nxt, prv, group)]
            reqs = dist.batch_isend_irecv(ops)
        o_s, lse_s = attn_fn(q, k_cur, v_cur, scale)
        if out is None:
            out, lse = o_s, lse_s
        else:
            merge_fn(out, lse, out, lse, o_s, lse_s)
        for r in reqs:
            r.wait()
        if step + 1 < P:
            k_cur, v_cur = k_nxt, v_nxt
    return out, lse


def _hip_scale(x: torch.Tensor, scale: float) -> torch.Tensor:
    """bf16(scale * x) through the guidance-combine kernel with a zero second operand:
    bf16(0 + bf16(scale * bf16(x - 0)))."""
    from .. import _C

    out = torch.empty_like(x)
    _C.cfg_combine(out, x, torch.zeros_like(x), float(scale))
    return out


def cfg_parallel_combine(noise_pred_local: torch.Tensor, guidance_scale: float, group,
                         scale_fn=None) -> torch.Tensor:
    """Classifier-free guidance with the two branches on the two ranks of `group`
    (pipelines_core/stages/denoising.py:1436-1446): rank 0 holds the conditional prediction and
    contributes s * cond, rank 1 holds the unconditional one and contributes (1 - s) * uncond;
    the sum is one all-reduce of B * tokens * channels bf16 values per step (0.5 MB per
    sample at 1024 x 1024: a single xGMI link moves that in microseconds).  Returns the combined
    prediction on both ranks.  Note the reference's arithmetic differs from its serial form
    u + s (c - u); both are kept as they are."""
    if dist.get_world_size(group) != 2:
        raise ValueError("CFG parallelism splits the two guidance branches over exactly 2 ranks")
    scale_fn = scale_fn or _hip_scale
    s = float(guidance_scale)
    partial = scale_fn(noise_pred_local, s if dist.get_rank(group) == 0 else 1.0 - s)
    dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial
