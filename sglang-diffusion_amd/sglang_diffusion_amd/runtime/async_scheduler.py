"""Data-parallel scheduling UNDER LOAD: requests are admitted while ranks compute, and every
request's result returns as soon as that request is done.

`DataParallelBatchScheduler.run()` (batch_scheduler.py) drains a queue that is complete when
it starts: rank 0 scatters, every rank computes, one gather.  A serving loop is different --
the reference's `Scheduler.event_loop` (python/sglang/multimodal_gen/runtime/managers/
scheduler.py:326-418) receives requests at any time (`recv_reqs`, :291-314), picks the next
batch (`get_next_batch_to_run`, :182-190), executes it and replies per request (:373-383).
This module is that loop over the ranks of one node:

  * rank 0 owns admission.  `submit()` assigns a request to the rank with the least
    outstanding DiT-sample cost (ties to the lower rank) and sends it THERE at once -- header
    and tensors point to point -- whether or not that rank is in the middle of a micro-batch:
    a receiver thread on the rank appends it to the rank's `waiting_queue`, and the rank's
    next `get_next_batch_to_run()` sees it.  Nothing waits for the slowest rank.
  * a rank that finishes a micro-batch sends each request's final latents to rank 0 by
    itself (one point-to-point message per request); rank 0 hands them out through
    `results()` in completion order.  No collective is involved, so a rank with nothing to
    do takes part in nothing.
  * the traffic rides on a gloo group over host memory, like the reference's request path
    (pickled objects over ZMQ and gloo CPU groups): a request is 4.7 MB per sample and a
    result 0.5 MB, against seconds of compute.  The denoise steps themselves never
    communicate (a request's cond / uncond pair stays on one rank).

Threads: per peer one sender and one receiver thread on rank 0, one receiver thread on every
other rank; each (peer, direction) has its own tag, so messages of a pair stay ordered.
"""
from __future__ import annotations

import dataclasses
import io
import pickle
import queue
import threading
from collections import deque
from typing import Any, Deque, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .batch_scheduler import Item, _take_microbatch, merge_requests, request_cost
from .denoise import Req

_REQ_TENSORS = ("latents", "prompt_embeds", "pooled_embeds", "negative_prompt_embeds",
                "neg_pooled_embeds")
_TAG_DOWN, _TAG_UP = 11, 12     # rank 0 -> rank r, rank r -> rank 0


def _send_obj(obj, dst: int, group, tag: int) -> None:
    buf = io.BytesIO()
    pickle.dump(obj, buf)
    data = torch.frombuffer(bytearray(buf.getvalue()), dtype=torch.uint8)
    dist.send(torch.tensor([data.numel()], dtype=torch.int64), dst, group=group, tag=tag)
    dist.send(data, dst, group=group, tag=tag)


def _recv_obj(src: int, group, tag: int):
    n = torch.zeros(1, dtype=torch.int64)
    dist.recv(n, src, group=group, tag=tag)
    data = torch.empty(int(n.item()), dtype=torch.uint8)
    dist.recv(data, src, group=group, tag=tag)
    return pickle.loads(data.numpy().tobytes())


class AsyncDataParallelScheduler:
    """One instance per rank of `group` (a gloo group; default: a new one over all ranks).

    rank 0:   start(); submit(req, identity) at any time; poll results() / wait_all();
              serve_local() runs rank 0's own share; shutdown() ends every rank's loop.
    rank r:   start(); serve() blocks until rank 0 shuts down.
    `device`: where this rank's requests live while they run (its GPU); None keeps them on
    the host (CPU tests)."""

    def __init__(self, stage, rank: int, world_size: int, max_samples_per_step: int = 8,
                 group=None, device=None, num_steps: Optional[int] = None):
        self.stage, self.rank, self.world_size = stage, rank, world_size
        self.max_samples = max_samples_per_step
        self.group = group
        self.device = device
        self.num_steps = num_steps
        self.waiting_queue: Deque[Item] = deque()
        self._cv = threading.Condition()
        self._stop = False
        self._threads: List[threading.Thread] = []
        # rank 0 only
        self._outstanding = [0] * world_size          # DiT-sample cost in flight per rank
        self._owner: Dict[Any, Tuple[int, int]] = {}  # identity -> (rank, cost)
        self._outbox: List[queue.Queue] = [queue.Queue() for _ in range(world_size)]
        self._done: "queue.Queue[Tuple[Any, torch.Tensor]]" = queue.Queue()
        self._submitted = 0
        self._returned = 0
        self.log: List[Tuple[str, Any, float]] = []   # (event, identity, perf_counter) for tests

    # ------------------------------------------------------------------------- plumbing --
    def _note(self, what: str, ident) -> None:
        import time

        self.log.append((what, ident, time.perf_counter()))

    def start(self) -> None:
        if self.world_size == 1:
            return
        if self.group is None:
            self.group = dist.new_group(backend="gloo")
        if self.rank == 0:
            for r in range(1, self.world_size):
                for fn in (self._sender, self._collector):
                    t = threading.Thread(target=fn, args=(r,), daemon=True)
                    t.start()
                    self._threads.append(t)
        else:
            t = threading.Thread(target=self._receiver, daemon=True)
            t.start()
            self._threads.append(t)

    def _peer(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    # rank 0: one thread per peer drains that peer's outbox
    def _sender(self, r: int) -> None:
        while True:
            msg = self._outbox[r].get()
            if msg is None:
                _send_obj({"op": "stop"}, self._peer(r), self.group, _TAG_DOWN)
                return
            ident, req = msg
            scalars = {f.name: getattr(req, f.name) for f in dataclasses.fields(Req)
                       if f.name not in _REQ_TENSORS}
            tensors = {k: getattr(req, k) for k in _REQ_TENSORS if getattr(req, k) is not None}
            head = {"op": "req", "identity": ident, "scalars": scalars,
                    "tensors": {k: (tuple(v.shape), v.dtype) for k, v in tensors.items()}}
            _send_obj(head, self._peer(r), self.group, _TAG_DOWN)
            for v in tensors.values():
                dist.send(v.detach().to("cpu").contiguous(), self._peer(r), group=self.group,
                          tag=_TAG_DOWN)
            self._note("sent", ident)

    # rank 0: one thread per peer receives that peer's finished requests
    def _collector(self, r: int) -> None:
        while True:
            head = _recv_obj(self._peer(r), self.group, _TAG_UP)
            if head["op"] == "bye":
                return
            out = torch.empty(head["shape"], dtype=head["dtype"])
            dist.recv(out, self._peer(r), group=self.group, tag=_TAG_UP)
            self._finish(head["identity"], out)

    # rank r: admission from rank 0 while the main thread computes
    def _receiver(self) -> None:
        src = self._peer(0)
        while True:
            head = _recv_obj(src, self.group, _TAG_DOWN)
            if head["op"] == "stop":
                with self._cv:
                    self._stop = True
                    self._cv.notify_all()
                return
            got = {}
            for k, (shape, dtype) in head["tensors"].items():
                t = torch.empty(shape, dtype=dtype)
                dist.recv(t, src, group=self.group, tag=_TAG_DOWN)
                got[k] = t if self.device is None else t.to(self.device, non_blocking=False)
            req = Req(**got, **head["scalars"])
            with self._cv:
                self.waiting_queue.append((head["identity"], req))
                self._note("admitted", head["identity"])
                self._cv.notify_all()

    def _finish(self, ident, latents: torch.Tensor) -> None:
        with self._cv:
            rk, cost = self._owner.pop(ident)
            self._outstanding[rk] -= cost
            self._returned += 1
            self._note("returned", ident)
            self._cv.notify_all()
        self._done.put((ident, latents))

    # ---------------------------------------------------------------------------- rank 0 --
    def submit(self, req: Req, identity: Any = None) -> Any:
        """Admit a request NOW: it is assigned to the least loaded rank and on its way before
        this call returns, whatever the ranks are doing."""
        if self.rank != 0:
            raise RuntimeError("submit() belongs to rank 0, which owns admission")
        cost = request_cost(req)
        if cost > self.max_samples:
            raise ValueError(f"request needs {cost} DiT samples > max_samples={self.max_samples}")
        with self._cv:
            if identity is None:
                identity = self._submitted
            self._submitted += 1
            rk = min(range(self.world_size), key=lambda k: (self._outstanding[k], k))
            self._outstanding[rk] += cost
            self._owner[identity] = (rk, cost)
            self._note("submitted", identity)
            if rk == 0:
                if self.device is not None:
                    req = dataclasses.replace(req, **{
                        k: getattr(req, k).to(self.device) for k in _REQ_TENSORS
                        if getattr(req, k) is not None})
                self.waiting_queue.append((identity, req))
                self._cv.notify_all()
        if rk != 0:
            self._outbox[rk].put((identity, req))
        return identity

    def results(self, block: bool = False, timeout: Optional[float] = None):
        """Finished requests as (identity, final latents on the host), in completion order."""
        out = []
        try:
            if block:
                out.append(self._done.get(timeout=timeout))
            while True:
                out.append(self._done.get_nowait())
        except queue.Empty:
            pass
        return out

    def pending(self) -> int:
        with self._cv:
            return self._submitted - self._returned

    # ------------------------------------------------------------------------ every rank --
    def get_next_batch_to_run(self) -> Optional[List[Item]]:
        with self._cv:
            if not self.waiting_queue:
                return None
            items = list(self.waiting_queue)
            take = _take_microbatch(list(range(len(items))), [r for _, r in items], self.max_samples)
            taken = set(take)
            self.waiting_queue = deque(it for i, it in enumerate(items) if i not in taken)
            return [items[i] for i in take]

    @torch.no_grad()
    def _execute_and_reply(self, items: List[Item]) -> None:
        for ident, _ in items:
            self._note("started", ident)
        merged = merge_requests([r for _, r in items])
        self.stage.forward(merged, num_steps=self.num_steps)
        o = 0
        for ident, r in items:
            lat = merged.latents[o:o + r.batch_size].detach().to("cpu").contiguous()
            o += r.batch_size
            if self.rank == 0:
                self._finish(ident, lat)
            else:
                _send_obj({"op": "done", "identity": ident, "shape": tuple(lat.shape),
                           "dtype": lat.dtype}, self._peer(0), self.group, _TAG_UP)
                dist.send(lat, self._peer(0), group=self.group, tag=_TAG_UP)

    def serve_local(self, until_idle: bool = True, wait_s: float = 0.05) -> int:
        """Rank 0: run its own share.  Returns the number of micro-batches executed.  With
        `until_idle` it keeps going while ANY request is outstanding anywhere (sleeping on the
        condition variable when its own queue is empty), so that `results()` fills up."""
        n = 0
        while True:
            items = self.get_next_batch_to_run()
            if items is not None:
                self._execute_and_reply(items)
                n += 1
                continue
            with self._cv:
                if not until_idle or self._submitted == self._returned:
                    return n
                self._cv.wait(wait_s)

    def serve(self) -> int:
        """Ranks other than 0: compute whatever arrives until rank 0 says stop."""
        n = 0
        while True:
            items = self.get_next_batch_to_run()
            if items is not None:
                self._execute_and_reply(items)
                n += 1
                continue
            with self._cv:
                if self._stop and not self.waiting_queue:
                    break
                self._cv.wait(0.05)
        _send_obj({"op": "bye"}, self._peer(0), self.group, _TAG_UP)
        return n

    def shutdown(self) -> None:
        """Rank 0: tell every rank to leave its loop (after what it already holds)."""
        if self.rank == 0:
            for r in range(1, self.world_size):
                self._outbox[r].put(None)
        for t in self._threads:
            t.join(30)
