"""Data-parallel batch scheduler for the denoise loop.

The reference scheduler pops ONE request per iteration and runs it on all GPUs
(python/sglang/multimodal_gen/runtime/managers/scheduler.py:182-190); `dp_size > 1` raises
"DP is not yet supported" (runtime/server_args.py:1201-1202).  Here the queue is drained in
FIFO order into micro-batches of compatible requests (same height/width/steps/CFG mode,
batch = prompts x outputs as in pipelines_core/schedule_batch.py:231-243), the request list is
split by request across the ranks of one node, and each rank runs its micro-batches with a
full weight replica.  A request's cond/uncond branches stay together, so nothing is exchanged
inside a step; results return with one gather per drained queue.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass
from typing import Deque, Dict, List, Optional, Sequence, Tuple

import torch

from .denoise import DenoisingStage, Req
from .parallel import gather_latents, shard_requests


def _compat_key(r: Req) -> Tuple:
    return (r.height, r.width, r.num_inference_steps, r.do_classifier_free_guidance,
            float(r.guidance_scale), float(r.embedded_cfg_scale), r.prompt_embeds.shape[1])


def merge_requests(reqs: Sequence[Req]) -> Req:
    """Concatenate compatible requests along the batch dimension."""
    r0 = reqs[0]
    cat = lambda xs: torch.cat(list(xs), dim=0)  # noqa: E731
    out = Req(latents=cat(r.latents for r in reqs),
              prompt_embeds=cat(r.prompt_embeds for r in reqs),
              pooled_embeds=cat(r.pooled_embeds for r in reqs), height=r0.height, width=r0.width,
              num_inference_steps=r0.num_inference_steps, guidance_scale=r0.guidance_scale,
              embedded_cfg_scale=r0.embedded_cfg_scale)
    if r0.negative_prompt_embeds is not None:
        out.negative_prompt_embeds = cat(r.negative_prompt_embeds for r in reqs)
        out.neg_pooled_embeds = cat(r.neg_pooled_embeds for r in reqs)
    return out


def plan_microbatches(reqs: Sequence[Req], max_samples: int) -> List[List[int]]:
    """FIFO greedy packing: a micro-batch holds consecutive-in-queue compatible requests whose
    DiT sample count (requests x CFG branches) fits `max_samples`.  Returns request indices."""
    plans: List[List[int]] = []
    open_by_key: Dict[Tuple, List[int]] = {}
    load: Dict[int, int] = {}
    for idx, r in enumerate(reqs):
        cost = r.batch_size * (2 if r.do_classifier_free_guidance else 1)
        if cost > max_samples:
            raise ValueError(f"request {idx} needs {cost} DiT samples > max_samples={max_samples}")
        key = _compat_key(r)
        cur = open_by_key.get(key)
        if cur is not None and load[id(cur)] + cost <= max_samples:
            cur.append(idx)
            load[id(cur)] += cost
        else:
            cur = [idx]
            plans.append(cur)
            open_by_key[key] = cur
            load[id(cur)] = cost
    return plans


class DataParallelBatchScheduler:
    def __init__(self, stage: DenoisingStage, rank: int = 0, world_size: int = 1,
                 max_samples_per_step: int = 8, group=None):
        self.stage = stage
        self.rank, self.world_size = rank, world_size
        self.max_samples = max_samples_per_step
        self.group = group
        self.queue: Deque[Req] = deque()

    def add_request(self, req: Req) -> None:
        self.queue.append(req)

    def local_slice(self, n: int) -> range:
        return shard_requests(n, self.world_size, self.rank)

    @torch.no_grad()
    def run(self, num_steps: Optional[int] = None, gather: bool = True):
        """Drain the queue: every rank sees the same queue, keeps its slice, runs it in
        micro-batches, and rank 0 gets every request's final latents in queue order."""
        reqs = list(self.queue)
        self.queue.clear()
        mine = list(self.local_slice(len(reqs)))
        local = [reqs[i] for i in mine]
        outs: List[Optional[torch.Tensor]] = [None] * len(local)
        for plan in plan_microbatches(local, self.max_samples):
            merged = merge_requests([local[i] for i in plan])
            self.stage.forward(merged, num_steps=num_steps)
            o = 0
            for i in plan:
                b = local[i].batch_size
                outs[i] = merged.latents[o:o + b]
                o += b
        if not gather or self.world_size == 1:
            return outs
        same = len({tuple(r.latents.shape[1:]) for r in reqs}) == 1
        if not same:
            raise NotImplementedError("gather of mixed-shape requests: gather per shape bucket")
        counts = [sum(reqs[i].batch_size for i in shard_requests(len(reqs), self.world_size, r))
                  for r in range(self.world_size)]
        dev = reqs[0].latents.device
        loc = (torch.cat(outs, dim=0) if outs else
               torch.zeros((0,) + tuple(reqs[0].latents.shape[1:]), dtype=reqs[0].latents.dtype,
                           device=dev))
        return gather_latents(loc, counts, self.group)
