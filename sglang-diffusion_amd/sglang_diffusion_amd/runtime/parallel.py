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


def _wire(t: torch.Tensor, group=None) -> torch.Tensor:
    """The tensor a collective of `group` can carry: gloo moves host memory only, so device
    tensors are staged through the host there; RCCL moves device memory only, so host tensors
    are staged through this rank's GPU."""
    backend = dist.get_backend(group)
    if t.is_cuda and backend == "gloo":
        return t.cpu()
    if not t.is_cuda and backend == "nccl":
        return t.to(torch.device("cuda", torch.cuda.current_device()))
    return t


def all_gather_tensors(x: torch.Tensor, group=None) -> List[torch.Tensor]:
    """dist.all_gather of equally shaped tensors; under gloo, device tensors travel through the
    host and come back on x's device."""
    P = dist.get_world_size(group)
    w = _wire(x.contiguous(), group)
    parts = [torch.empty_like(w) for _ in range(P)]
    dist.all_gather(parts, w, group=group)
    return [t.to(x.device) for t in parts] if w.device != x.device else parts


def _a2a(recv: torch.Tensor, send: torch.Tensor, group) -> None:
    """all_to_all_single into `recv` (staged as `_wire` says when the backend cannot carry it)."""
    w = _wire(send, group)
    if w.device == send.device:
        dist.all_to_all_single(recv, send, group=group)
        return
    r = torch.empty(recv.shape, dtype=recv.dtype, device=w.device)
    dist.all_to_all_single(r, w, group=group)
    recv.copy_(r)


def gather_latents(local: torch.Tensor, counts: Sequence[int], group=None) -> Optional[torch.Tensor]:
    """Gather per-rank final latents [n_r, ...] to rank 0 of `group` (once per batch): a
    `dist.gather`, so only rank 0 receives anything.  Ranks may hold different request counts:
    every rank pads to the maximum, rank 0 trims."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    nmax = max(counts)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    pad = _wire(pad, group)
    root = dist.get_rank(group) == 0
    bufs = [torch.empty_like(pad) for _ in range(world)] if root else None
    dist.gather(pad, bufs, dst=_global_rank(group, 0), group=group)
    if not root:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).to(local.device)


_REQ_TENSORS = ("latents", "prompt_embeds", "pooled_embeds", "negative_prompt_embeds",
                "neg_pooled_embeds")


def scatter_requests(reqs, assign_fn, group=None, src: int = 0, device=None):
    """Send every rank its share of a request list that only rank `src` of `group` holds.

    `reqs` is the list on `src` and None elsewhere; `assign_fn(costs) -> [[positions], ...]`
    maps per-request DiT-sample costs to ranks.  The scalar fields and tensor shapes of the
    whole list travel as ONE broadcast object (the reference broadcasts each pickled request to
    every rank, managers/scheduler.py:291-314); a request's tensors go point to point to the one
    rank that runs it, all sends of a rank batched.  Returns (meta, mine): `meta[i]` describes
    request i on every rank (rank, batch, latent shape/dtype), `mine` maps this rank's
    positions to Req objects on `device` (default: this rank's current GPU when rank `src` keeps
    its requests on a GPU, else the host)."""
    import dataclasses

    from .denoise import Req

    me = dist.get_rank(group)
    box = [None]
    if me == src:
        costs = [r.batch_size * (2 if r.do_classifier_free_guidance else 1) for r in reqs]
        parts = assign_fn(costs)
        owner = {i: rk for rk, p in enumerate(parts) for i in p}
        scalars = [f.name for f in dataclasses.fields(Req) if f.name not in _REQ_TENSORS]
        box[0] = [dict(rank=owner[i], batch=r.batch_size, shape=tuple(r.latents.shape[1:]),
                       dtype=r.latents.dtype,
                       scalars={k: getattr(r, k) for k in scalars},
                       tensors={k: (tuple(getattr(r, k).shape), getattr(r, k).dtype)
                                for k in _REQ_TENSORS if getattr(r, k) is not None},
                       device=str(r.latents.device))
                  for i, r in enumerate(reqs)]
    dist.broadcast_object_list(box, src=_global_rank(group, src), group=group)
    meta = box[0]
    if device is None:
        # the KIND of device rank `src` keeps its requests on, but this rank's own GPU: with one
        # process per GPU "cuda:0" in rank 0's metadata is somebody else's device everywhere else
        device = torch.device(meta[0]["device"]) if meta else torch.device("cpu")
        if device.type == "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
    on_host = dist.get_backend(group) == "gloo"
    # RCCL moves device memory only: host-resident requests are staged through this rank's GPU
    wire = torch.device("cpu") if on_host else torch.device("cuda", torch.cuda.current_device())
    ops, mine, landing = [], {}, []
    for i, m in enumerate(meta):
        if me == src and m["rank"] == src:
            mine[i] = reqs[i]
        elif me == src:
            for k in m["tensors"]:
                t = getattr(reqs[i], k).contiguous()
                ops.append(dist.P2POp(dist.isend, t.to(wire), _global_rank(group, m["rank"]), group))
        elif me == m["rank"]:
            got = {}
            for k, (shape, dtype) in m["tensors"].items():
                got[k] = torch.empty(shape, dtype=dtype, device=wire)
                ops.append(dist.P2POp(dist.irecv, got[k], _global_rank(group, src), group))
            landing.append((i, got))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for i, got in landing:
        mine[i] = Req(**{k: v.to(device) for k, v in got.items()}, **meta[i]["scalars"])
    slim = [dict(rank=m["rank"], batch=m["batch"], shape=m["shape"], dtype=m["dtype"]) for m in meta]
    return slim, mine


def _global_rank(group, group_rank: int) -> int:
    return dist.get_global_rank(group, group_rank) if group is not None else group_rank


def ulysses_all_to_all(x: torch.Tensor, group, scatter_dim: int, gather_dim: int) -> torch.Tensor:
    """[B, S_local, H, D] <-> [B, S, H_local, D] exchange (layers/usp.py:49-158): split
    `scatter_dim` into P parts, send part p to rank p, concatenate what arrives along
    `gather_dim`.  One all_to_all_single per tensor; over xGMI every rank talks to all 7
    peers at once, which is the pattern the point-to-point fabric is built for.

    Copies: the send buffer is the input itself when its P parts are already contiguous
    blocks (B = 1 and the sequence is scattered), else ONE packing copy; the received blocks
    are the result itself when B = 1 and the sequence is gathered (a view), else ONE
    interleaving copy."""
    P = dist.get_world_size(group)
    if P == 1:
        return x
    if x.dim() != 4 or {scatter_dim, gather_dim} != {1, 2}:
        raise ValueError("ulysses_all_to_all exchanges dims 1 (sequence) and 2 (heads) of [B, S, H, D]")
    if x.shape[scatter_dim] % P:
        raise ValueError(f"dim {scatter_dim} of size {x.shape[scatter_dim]} not divisible by {P}")
    B, S, H, D = x.shape
    if scatter_dim == 2:   # heads out, sequence in: [B, S, H, D] -> [B, P*S, H/P, D]
        send = x.reshape(B, S, P, H // P, D).permute(2, 0, 1, 3, 4).contiguous()  # [P, B, S, H/P, D]
        recv = torch.empty_like(send)
        _a2a(recv, send, group)
        if B == 1:
            return recv.view(1, P * S, H // P, D)
        return recv.permute(1, 0, 2, 3, 4).reshape(B, P * S, H // P, D)
    # sequence out, heads in: [B, S, H, D] -> [B, S/P, P*H, D]
    if B == 1 and x.is_contiguous():
        send = x.view(P, 1, S // P, H, D)                                       # already P blocks
    else:
        send = x.reshape(B, P, S // P, H, D).permute(1, 0, 2, 3, 4).contiguous()
    recv = torch.empty((P, B, S // P, H, D), dtype=x.dtype, device=x.device)
    _a2a(recv, send, group)
    return recv.permute(1, 2, 0, 3, 4).reshape(B, S // P, P * H, D)


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
    staged = k_cur.is_cuda and dist.get_backend(group) == "gloo"  # gloo moves host memory only
    out = lse = None
    for step in range(P):
        reqs = []
        if step + 1 < P:
            k_snd, v_snd = (k_cur.cpu(), v_cur.cpu()) if staged else (k_cur, v_cur)
            k_nxt, v_nxt = torch.empty_like(k_snd), torch.empty_like(v_snd)
            ops = [dist.P2POp(dist.isend, k_snd, nxt, group), dist.P2POp(dist.isend, v_snd, nxt, group),
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
            k_cur, v_cur = (k_nxt.to(k.device), v_nxt.to(v.device)) if staged else (k_nxt, v_nxt)
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
    w = _wire(partial, group)
    dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
    return partial if w is partial else partial.copy_(w)
