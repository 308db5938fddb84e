"""Data-parallel batch scheduler for the denoise loop.

The reference scheduler pops ONE request per iteration and runs it on all GPUs
(python/sglang/multimodal_gen/runtime/managers/scheduler.py:182-190); `dp_size > 1` raises
"DP is not yet supported" (runtime/server_args.py:1201-1202).  This scheduler keeps the same
shape -- a FIFO `waiting_queue` of (identity, Req) items, `get_next_batch_to_run()` handing the
event loop the items to execute next -- but a batch is a MICRO-BATCH of compatible requests
(batch = prompts x outputs as in pipelines_core/schedule_batch.py:231-243, a request with true
CFG costing two DiT samples), and the queue is spread over the ranks of one node by DiT-sample
cost.  Each rank holds a full weight replica and a request's cond / uncond branches stay on
one rank, so nothing is exchanged inside a step: requests travel to their rank once (the
counterpart of the reference's `broadcast_pyobj` of every request to every rank,
scheduler.py:291-314) and final latents return with one gather per latent shape.
"""
from __future__ import annotations

from collections import deque
from typing import Any, Deque, Dict, List, Optional, Sequence, Tuple

import torch

from .denoise import DenoisingStage, Req
from .parallel import gather_latents, scatter_requests

Item = Tuple[Any, Req]  # (identity, request), the element type of the reference's waiting queue


def request_cost(r: Req) -> int:
    """DiT samples per denoise step: prompts x outputs, twice that under true CFG."""
    return r.batch_size * (2 if r.do_classifier_free_guidance else 1)


def _never_merged(r: Req) -> bool:
    # the norm cap compares WHOLE-tensor norms of the batch (denoising.py:1490-1499): merging
    # two requests would change each one's result, so such a request runs alone
    return bool(r.cfg_normalization) and float(r.cfg_normalization) > 0


def _compat_key(r: Req) -> Tuple:
    return (r.height, r.width, r.num_inference_steps, r.do_classifier_free_guidance,
            float(r.guidance_scale), float(r.embedded_cfg_scale), float(r.guidance_rescale),
            float(r.cfg_normalization or 0.0), tuple(r.latents.shape[1:]), r.latents.dtype,
            tuple(r.prompt_embeds.shape[1:]))


def merge_requests(reqs: Sequence[Req]) -> Req:
    """Concatenate compatible requests along the batch dimension; every per-request option of
    the denoise stage is carried over (the compatibility key makes them equal)."""
    r0 = reqs[0]
    if len(reqs) > 1 and any(_never_merged(r) for r in reqs):
        raise ValueError("a request with cfg_normalization > 0 cannot share a micro-batch: the "
                         "cap is a norm over the whole batch")
    if any(_compat_key(r) != _compat_key(r0) for r in reqs):
        raise ValueError("merge_requests: requests differ in shape, step count or guidance options")
    cat = lambda xs: torch.cat(list(xs), dim=0)  # noqa: E731
    out = Req(latents=cat(r.latents for r in reqs),
              prompt_embeds=cat(r.prompt_embeds for r in reqs),
              pooled_embeds=cat(r.pooled_embeds for r in reqs), height=r0.height, width=r0.width,
              num_inference_steps=r0.num_inference_steps, guidance_scale=r0.guidance_scale,
              embedded_cfg_scale=r0.embedded_cfg_scale, cfg_normalization=r0.cfg_normalization,
              guidance_rescale=r0.guidance_rescale,
              request_ids=[i for r in reqs for i in r.request_ids])
    if r0.negative_prompt_embeds is not None:
        out.negative_prompt_embeds = cat(r.negative_prompt_embeds for r in reqs)
        out.neg_pooled_embeds = cat(r.neg_pooled_embeds for r in reqs)
    return out


def _take_microbatch(pending: List[int], reqs: Sequence[Req], max_samples: int) -> List[int]:
    """The next micro-batch out of `pending` (queue positions, FIFO): the earliest request plus
    the later compatible ones, in order, until one of them no longer fits.  Stopping at the
    first misfit keeps a key's requests strictly FIFO."""
    head = pending[0]
    cost = request_cost(reqs[head])
    if cost > max_samples:
        raise ValueError(f"request {head} needs {cost} DiT samples > max_samples={max_samples}")
    take = [head]
    if not _never_merged(reqs[head]):
        key = _compat_key(reqs[head])
        for i in pending[1:]:
            if _compat_key(reqs[i]) != key:
                continue
            c = request_cost(reqs[i])
            if cost + c > max_samples:
                break
            take.append(i)
            cost += c
    return take


def plan_microbatches(reqs: Sequence[Req], max_samples: int) -> List[List[int]]:
    """The micro-batches `get_next_batch_to_run` would hand out for this queue if nothing were
    admitted meanwhile, as lists of queue positions."""
    pending = list(range(len(reqs)))
    plans: List[List[int]] = []
    while pending:
        take = _take_microbatch(pending, reqs, max_samples)
        plans.append(take)
        taken = set(take)
        pending = [i for i in pending if i not in taken]
    return plans


def assign_requests(costs: Sequence[int], world_size: int) -> List[List[int]]:
    """Spread requests over ranks by DiT-sample cost (a CFG request weighs twice a plain one):
    longest-processing-time first -- heaviest request to the least loaded rank, ties to the
    earlier request and the lower rank -- which is within 4/3 of the optimal makespan and exact
    for the equal-cost queues of the common case.  Each rank's list is in queue order."""
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in sorted(range(len(costs)), key=lambda j: (-costs[j], j)):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in out]


class DataParallelBatchScheduler:
    """One instance per rank.  Two ways to feed it:

    * `queue_on_rank0=False` (default): every rank is handed the same requests in the same
      order (as the reference's ranks all receive every request) and keeps its share;
    * `queue_on_rank0=True`: only rank 0 holds the requests; `run()` first sends every other
      rank its share (metadata as one broadcast object, tensors point to point).
    """

    def __init__(self, stage: DenoisingStage, rank: int = 0, world_size: int = 1,
                 max_samples_per_step: int = 8, group=None, queue_on_rank0: bool = False):
        self.stage = stage
        self.rank, self.world_size = rank, world_size
        self.max_samples = max_samples_per_step
        self.group = group
        self.queue_on_rank0 = queue_on_rank0
        self.waiting_queue: Deque[Item] = deque()
        self._seq = 0

    # the reference names its queue `waiting_queue`; `queue` is kept as an alias
    @property
    def queue(self) -> Deque[Item]:
        return self.waiting_queue

    def add_request(self, req: Req, identity: Any = None) -> None:
        """Admission: may be called at any time between micro-batches."""
        if identity is None:
            identity = self._seq
        self._seq += 1
        self.waiting_queue.append((identity, req))

    def get_next_batch_to_run(self) -> Optional[List[Item]]:
        """Pop the next micro-batch from the waiting queue: the earliest request and the
        compatible ones behind it that fit `max_samples_per_step` (scheduler.py:182-190 pops
        exactly one).  None when the queue is empty."""
        if not self.waiting_queue:
            return None
        items = list(self.waiting_queue)
        take = _take_microbatch(list(range(len(items))), [r for _, r in items], self.max_samples)
        taken = set(take)
        self.waiting_queue = deque(it for i, it in enumerate(items) if i not in taken)
        return [items[i] for i in take]

    @torch.no_grad()
    def execute(self, items: List[Item], num_steps: Optional[int] = None) -> List[torch.Tensor]:
        """Run one micro-batch through the denoise stage; final latents per item."""
        merged = merge_requests([r for _, r in items])
        self.stage.forward(merged, num_steps=num_steps)
        outs, o = [], 0
        for _, r in items:
            outs.append(merged.latents[o:o + r.batch_size])
            o += r.batch_size
        return outs

    def local_slice(self, n: int) -> List[int]:
        """Queue positions this rank runs when all `n` queued requests cost the same."""
        return assign_requests([1] * n, self.world_size)[self.rank]

    @torch.no_grad()
    def run(self, num_steps: Optional[int] = None, gather: bool = True,
            timing: Optional[dict] = None):
        """Drain the queue over all ranks.  Returns the final latents of every request in queue
        order (a list, one tensor per request) on rank 0 and None on the other ranks; with
        `gather=False` (or one rank) every rank returns its own requests' latents, in queue
        order.  `timing`, when given, receives this rank's wall-clock seconds per phase
        (`scatter_s`, `compute_s`, `gather_s`; the device is synchronised between phases)."""
        import time

        def mark(name, t0):
            if timing is not None:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                timing[name] = timing.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()

        t = time.perf_counter()
        if self.world_size == 1:
            order = [ident for ident, _ in self.waiting_queue]
            done: Dict[Any, torch.Tensor] = {}
            while (items := self.get_next_batch_to_run()) is not None:
                for (ident, _), o in zip(items, self.execute(items, num_steps)):
                    done[ident] = o
            mark("compute_s", t)
            return [done[i] for i in order]

        # ---- who runs what: metadata of the whole queue is known on every rank ------------
        if self.queue_on_rank0:
            meta, mine = scatter_requests(
                [r for _, r in self.waiting_queue] if self.rank == 0 else None,
                lambda costs: assign_requests(costs, self.world_size), self.group,
                device=self._own_device())
            self.waiting_queue.clear()
        else:
            reqs = [r for _, r in self.waiting_queue]
            self.waiting_queue.clear()
            parts = assign_requests([request_cost(r) for r in reqs], self.world_size)
            meta = [dict(rank=None, batch=r.batch_size, shape=tuple(r.latents.shape[1:]),
                         dtype=r.latents.dtype) for r in reqs]
            for rk, p in enumerate(parts):
                for i in p:
                    meta[i]["rank"] = rk
            mine = {i: reqs[i] for i in parts[self.rank]}
        for pos in sorted(mine):
            self.waiting_queue.append((pos, mine[pos]))
        t = mark("scatter_s", t)
        done = {}
        while (items := self.get_next_batch_to_run()) is not None:
            for (pos, _), o in zip(items, self.execute(items, num_steps)):
                done[pos] = o
        t = mark("compute_s", t)
        if not gather:
            return [done[p] for p in sorted(done)]

        # ---- one gather per latent shape ---------------------------------------------------
        dev = next(iter(done.values())).device if done else self.stage_device()
        results: List[Optional[torch.Tensor]] = [None] * len(meta)
        buckets: Dict[Tuple, List[int]] = {}
        for pos, m in enumerate(meta):
            buckets.setdefault((m["shape"], m["dtype"]), []).append(pos)
        for (shape, dtype), members in buckets.items():
            counts = [sum(meta[p]["batch"] for p in members if meta[p]["rank"] == rk)
                      for rk in range(self.world_size)]
            local = [done[p] for p in members if meta[p]["rank"] == self.rank]
            loc = (torch.cat(local, dim=0) if local
                   else torch.zeros((0,) + tuple(shape), dtype=dtype, device=dev))
            full = gather_latents(loc, counts, self.group)
            if full is None:
                continue
            # `full` is rank-major; walk it back into queue positions
            o = 0
            for rk in range(self.world_size):
                for p in members:
                    if meta[p]["rank"] == rk:
                        results[p] = full[o:o + meta[p]["batch"]]
                        o += meta[p]["batch"]
        mark("gather_s", t)
        return results if self.rank == 0 else None

    def _own_device(self):
        """Where this rank's requests must land: the stage's device if its model has one, else
        whatever scatter_requests derives from the sender."""
        d = self.stage_device()
        return d if d.type != "cpu" or getattr(self.stage, "transformer", None) is not None else None

    def stage_device(self):
        """Device of the stage's model (used to shape empty contributions to a gather)."""
        tr = getattr(self.stage, "transformer", None)
        try:
            return next(tr.parameters()).device
        except (AttributeError, StopIteration, TypeError):
            return torch.device("cpu")
