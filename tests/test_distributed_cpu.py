"""world_size-2 `gloo` tests of the multi-process paths: the gather of final latents with
unequal per-rank request counts, the Ulysses all-to-all, and the data-parallel batch
scheduler draining one queue on two ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sglang_diffusion_amd.runtime.batch_scheduler import DataParallelBatchScheduler, merge_requests
from sglang_diffusion_amd.runtime.parallel import (cfg_parallel_combine, gather_latents,
                                                   init_distributed, ring_attention,
                                                   ulysses_all_to_all)
from sglang_diffusion_amd.runtime.synthetic import make_synthetic_request

WORLD = 2


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, fn, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, WORLD)
    try:
        ret[rank] = fn(rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(fn):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, fn, ret)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return dict(ret)


def _gather_case(rank):
    counts = [3, 1]
    local = torch.full((counts[rank], 4, 2), float(rank + 1)) + torch.arange(counts[rank]).view(-1, 1, 1)
    out = gather_latents(local.bfloat16(), counts)
    return None if out is None else out.float()


def test_gather_latents_uneven_counts():
    ret = _run(_gather_case)
    assert ret[1] is None
    out = ret[0]
    assert out.shape == (4, 4, 2)
    assert out[:, 0, 0].tolist() == [1.0, 2.0, 3.0, 2.0]


def _ulysses_case(rank):
    g = dist.group.WORLD
    ok_fwd = ok_back = True
    # B = 1 takes the copy-free views, B = 2 the packed form; a strided input (a slice of a
    # fused q|k|v buffer) must give the same result as a dense one
    for B, strided in ((2, False), (1, False), (1, True), (2, True)):
        S, H, D = 6, 4, 8
        full = torch.arange(B * S * H * D, dtype=torch.float32).view(B, S, H, D)
        local = full[:, rank * S // WORLD:(rank + 1) * S // WORLD]             # sequence shard
        if strided:
            buf = torch.zeros(B, S // WORLD, 3, H, D)
            buf[:, :, 1] = local
            local = buf[:, :, 1]
        else:
            local = local.contiguous()
        heads = ulysses_all_to_all(local, g, scatter_dim=2, gather_dim=1)      # -> all S, H/P heads
        ok_fwd &= torch.equal(heads, full[:, :, rank * H // WORLD:(rank + 1) * H // WORLD])
        back = ulysses_all_to_all(heads, g, scatter_dim=1, gather_dim=2)       # inverse exchange
        ok_back &= torch.equal(back, local)
    try:
        ulysses_all_to_all(torch.zeros(1, 3, 3, 2), g, scatter_dim=2, gather_dim=1)
        raised = False
    except ValueError:
        raised = True
    return ok_fwd, ok_back, raised


def test_ulysses_all_to_all_round_trip():
    ret = _run(_ulysses_case)
    for rank in range(WORLD):
        assert ret[rank] == (True, True, True)


class _FakeStage:
    def forward(self, req, num_steps=None):
        req.latents = req.latents * 2 + req.prompt_embeds.float().mean(dim=(1, 2)).view(-1, 1, 1)
        return req


def _queue():
    return [make_synthetic_request(1 + (s % 2), height=64, width=64, num_inference_steps=2,
                                   device="cpu", seed=s, text_len=4, joint_dim=8, pooled_dim=4)
            for s in range(5)]


def _dp_case(rank):
    sch = DataParallelBatchScheduler(_FakeStage(), rank=rank, world_size=WORLD, max_samples_per_step=3)
    for r in _queue():
        sch.add_request(r)
    out = sch.run()
    return None if out is None else [o.float() for o in out]


def test_dp_scheduler_two_ranks_equals_serial():
    ret = _run(_dp_case)
    assert ret[1] is None
    expect = [_FakeStage().forward(merge_requests([r])).latents for r in _queue()]
    assert len(ret[0]) == len(expect)
    for o, e in zip(ret[0], expect):
        assert torch.equal(o, e.float())


def _mixed_queue(with_cfg=True):
    """Two latent shapes, plain and CFG requests interleaved: costs 1,2,1,4,1,2,2."""
    out = []
    for s, (b, hw, cfg) in enumerate([(1, 64, False), (1, 64, True), (1, 128, False), (2, 64, True),
                                      (1, 128, False), (2, 64, False), (1, 128, True)]):
        out.append(make_synthetic_request(b, height=hw, width=hw, num_inference_steps=2, device="cpu",
                                          seed=s, text_len=4, joint_dim=8, pooled_dim=4,
                                          true_cfg=cfg and with_cfg))
        out[-1].request_ids = [f"r{s}"]
    return out


def _dp_mixed_case(rank, on_rank0):
    sch = DataParallelBatchScheduler(_FakeStage(), rank=rank, world_size=WORLD, max_samples_per_step=4,
                                     queue_on_rank0=on_rank0)
    if not on_rank0 or rank == 0:
        for r in _mixed_queue():
            sch.add_request(r)
    out = sch.run()
    return None if out is None else [o.float() for o in out]


def _dp_mixed_replicated(rank):
    return _dp_mixed_case(rank, False)


def _dp_mixed_scattered(rank):
    return _dp_mixed_case(rank, True)


@pytest.mark.parametrize("case", [_dp_mixed_replicated, _dp_mixed_scattered])
def test_dp_scheduler_mixed_shapes_and_cfg_costs(case):
    """Requests of two latent shapes and uneven CFG costs: one gather per shape returns every
    request's latents to rank 0 in queue order, whether every rank was handed the queue or rank
    0 alone held it and sent the other rank its share."""
    ret = _run(case)
    assert ret[1] is None
    expect = [_FakeStage().forward(merge_requests([r])).latents for r in _mixed_queue()]
    assert [tuple(o.shape) for o in ret[0]] == [tuple(e.shape) for e in expect]
    for o, e in zip(ret[0], expect):
        assert torch.equal(o, e.float())


def _ring_case(rank):
    """Two ranks, queries and keys both sharded along the sequence: the ring result for the
    local queries equals dense attention over the concatenated keys.  The per-step attention
    and the merge are the CPU oracle's (the HIP kernels are the default on a GPU)."""
    from oracle import ops as O

    B, S, H, D = 2, 24, 3, 16
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, S, H, D, generator=g).bfloat16()
    k = torch.randn(B, S, H, D, generator=g).bfloat16()
    v = torch.randn(B, S, H, D, generator=g).bfloat16()
    sl = slice(rank * S // WORLD, (rank + 1) * S // WORLD)

    def attn(q_, k_, v_, scale):
        return O.attention(q_, k_, v_, scale, return_lse=True)

    def merge(out, lse, a, a_lse, b, b_lse):
        # oracle layout: [tokens, heads, D] and [tokens, heads]
        Bq, Sq = a.shape[:2]
        vm, sm = O.merge_attn_states(a.reshape(Bq * Sq, H, D), a_lse.transpose(1, 2).reshape(Bq * Sq, H),
                                     b.reshape(Bq * Sq, H, D), b_lse.transpose(1, 2).reshape(Bq * Sq, H))
        out.copy_(vm.view(Bq, Sq, H, D))
        lse.copy_(sm.view(Bq, Sq, H).transpose(1, 2))

    out, lse = ring_attention(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(),
                              dist.group.WORLD, attn_fn=attn, merge_fn=merge)
    ref, ref_lse = O.attention(q[:, sl], k, v, D ** -0.5, return_lse=True)
    err = (out.float() - ref.float()).norm() / ref.float().norm()
    lse_err = (lse - ref_lse).abs().max()
    try:
        ring_attention(q, k, v, dist.group.WORLD, is_causal=True, attn_fn=attn, merge_fn=merge)
        raised = False
    except NotImplementedError:
        raised = True
    return float(err), float(lse_err), raised


def test_ring_attention_two_ranks_equals_dense():
    ret = _run(_ring_case)
    for rank in range(WORLD):
        err, lse_err, raised = ret[rank]
        # one extra bf16 rounding of the partial results before the merge
        assert err <= 6e-3 and lse_err <= 1e-5 and raised, ret[rank]


class _OracleImpl:
    """Attention implementation stand-in for the CPU: the oracle's dense attention."""

    def forward(self, q, k, v, attn_metadata=None, return_softmax_lse=False):
        from oracle import ops as O

        return O.attention(q, k, v, return_lse=return_softmax_lse)

    @staticmethod
    def merge_states(out, lse, a, a_lse, b, b_lse):
        from oracle import ops as O

        B, S, H, D = a.shape
        vm, sm = O.merge_attn_states(a.reshape(B * S, H, D), a_lse.transpose(1, 2).reshape(B * S, H),
                                     b.reshape(B * S, H, D), b_lse.transpose(1, 2).reshape(B * S, H))
        out.copy_(vm.view(B, S, H, D))
        lse.copy_(sm.view(B, S, H).transpose(1, 2))


def _ulysses_replicated_case(rank):
    """UlyssesAttention with a replicated text part (layers/attention/layer.py:84-152): the
    video tokens are sharded over the ranks, the text tokens are on both; the result equals one
    dense attention over [video | text] on a single rank."""
    from oracle import ops as O
    from sglang_diffusion_amd.layers.attention import UlyssesAttention
    from sglang_diffusion_amd.layers.forward_context import set_forward_context

    B, S, T, H, D = 1, 16, 6, 4, 16
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(3, B, S, H, D, generator=g).bfloat16()
    tqkv = torch.randn(3, B, T, H, D, generator=g).bfloat16()
    sl = slice(rank * S // WORLD, (rank + 1) * S // WORLD)
    layer = UlyssesAttention(H, D, group=dist.group.WORLD, attn_impl=_OracleImpl())
    with set_forward_context(current_timestep=0, attn_metadata=None):
        out, rep = layer(qkv[0][:, sl].contiguous(), qkv[1][:, sl].contiguous(),
                         qkv[2][:, sl].contiguous(), tqkv[0], tqkv[1], tqkv[2])
    ref = O.attention(torch.cat([qkv[0], tqkv[0]], 1), torch.cat([qkv[1], tqkv[1]], 1),
                      torch.cat([qkv[2], tqkv[2]], 1))
    return torch.equal(out, ref[:, sl]), torch.equal(rep, ref[:, S:])


def test_ulysses_attention_with_replicated_text():
    ret = _run(_ulysses_replicated_case)
    for rank in range(WORLD):
        assert ret[rank] == (True, True), ret[rank]


def _usp_case(rank):
    """USPAttention on two ranks: (a) Flux joint attention with the text prefix replicated
    (layer.py:414-470) equals dense attention over [text | image] bit for bit; (b) the ring
    variant over a sequence sharded in q AND k/v equals dense attention within the extra
    rounding of the partial results."""
    from oracle import ops as O
    from sglang_diffusion_amd.layers.attention import USPAttention
    from sglang_diffusion_amd.layers.forward_context import set_forward_context

    B, T, S, H, D = 1, 5, 16, 4, 16
    g = torch.Generator().manual_seed(21)
    txt = torch.randn(3, B, T, H, D, generator=g).bfloat16()
    img = torch.randn(3, B, S, H, D, generator=g).bfloat16()
    sl = slice(rank * S // WORLD, (rank + 1) * S // WORLD)
    local = [torch.cat([txt[i], img[i][:, sl]], dim=1) for i in range(3)]
    ref = O.attention(*[torch.cat([txt[i], img[i]], dim=1) for i in range(3)])
    want = torch.cat([ref[:, :T], ref[:, T:][:, sl]], dim=1)
    with set_forward_context(current_timestep=0, attn_metadata=None):
        layer = USPAttention(H, D, ulysses_group=dist.group.WORLD, attn_impl=_OracleImpl())
        out = layer(local[0], local[1], local[2], num_replicated_prefix=T)
        ring = USPAttention(H, D, ring_group=dist.group.WORLD, attn_impl=_OracleImpl())
        r_out = ring(img[0][:, sl].contiguous(), img[1][:, sl].contiguous(), img[2][:, sl].contiguous())
    r_ref = O.attention(img[0][:, sl], img[1], img[2])
    r_err = float((r_out.float() - r_ref.float()).norm() / r_ref.float().norm())
    return torch.equal(out, want), r_err


def test_usp_attention_replicated_prefix_and_ring():
    ret = _run(_usp_case)
    for rank in range(WORLD):
        ok, r_err = ret[rank]
        assert ok and r_err <= 6e-3, ret[rank]


def _cfg_parallel_case(rank):
    """The two guidance branches on two ranks (denoising.py:1436-1446): the all-reduced sum is
    s * cond + (1 - s) * uncond with eager bf16 rounding of each product and of the sum."""
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(2, 16, 8, generator=g).bfloat16()
    uncond = torch.randn(2, 16, 8, generator=g).bfloat16()
    s = 3.5
    out = cfg_parallel_combine(cond if rank == 0 else uncond, s, dist.group.WORLD,
                               scale_fn=lambda x, a: a * x)
    want = s * cond + (1 - s) * uncond
    return torch.equal(out, want)


def test_cfg_parallel_combine_two_ranks():
    ret = _run(_cfg_parallel_case)
    assert ret[0] and ret[1]


def _cfg_parallel_stage_case(rank):
    """DenoisingStage.predict_noise_with_cfg on a 2-rank CFG group with the norm cap and the
    std rescale requested: rank 0's conditional prediction is broadcast so both ranks apply
    the two steps to identical operands.  The device kernels are replaced by the oracle's
    statements of them (no GPU here); the stage's control flow and collectives are what runs."""
    from oracle import ops
    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.runtime import denoise as D
    from sglang_diffusion_amd.runtime import parallel as P

    _C.cfg_normalize = lambda n, c, f, scratch=None: n.copy_(ops.cfg_normalize(n, c, f))
    _C.guidance_rescale = lambda n, c, r, scratch=None: n.copy_(ops.rescale_noise_cfg(n, c, r))
    P._hip_scale = lambda x, a: a * x
    g = torch.Generator().manual_seed(11)
    cond = torch.randn(2, 32, 8, generator=g).bfloat16()
    uncond = (torch.randn(2, 32, 8, generator=g) * 0.9).bfloat16()
    pos, neg = torch.zeros(2, 4, 8), torch.ones(2, 4, 8)

    class FakeDiT:
        class config:
            guidance_embeds = False

        def forward_fused(self, hidden_states, encoder_hidden_states, **kw):
            return (cond if encoder_hidden_states.sum() == 0 else uncond).clone()

    req = D.Req(latents=torch.zeros(2, 32, 8).bfloat16(), prompt_embeds=pos,
                pooled_embeds=torch.zeros(2, 4), negative_prompt_embeds=neg,
                neg_pooled_embeds=torch.zeros(2, 4), guidance_scale=4.0, cfg_normalization=1.0,
                guidance_rescale=0.7)
    stage = D.DenoisingStage(FakeDiT(), fused=True, cfg_group=dist.group.WORLD)
    out = stage.predict_noise_with_cfg(req, req.latents, torch.zeros(2), None, None, 0)
    want = 4.0 * cond + (1 - 4.0) * uncond
    want = ops.rescale_noise_cfg(ops.cfg_normalize(want, cond, 1.0), cond, 0.7)
    plain = D.DenoisingStage(FakeDiT(), fused=True, cfg_group=dist.group.WORLD)
    req.cfg_normalization, req.guidance_rescale = 0.0, 0.0
    out_plain = plain.predict_noise_with_cfg(req, req.latents, torch.zeros(2), None, None, 0)
    return torch.equal(out, want), torch.equal(out_plain, 4.0 * cond + (1 - 4.0) * uncond)


def test_cfg_parallel_stage_with_post_combine_steps():
    ret = _run(_cfg_parallel_stage_case)
    assert ret[0] == (True, True) and ret[1] == (True, True)


# ------------------------------------------------------------ scheduling under load --------
class _SleepStage:
    """Stand-in denoiser: 0.25 s per DiT sample, final latents = 2 * latents + 1."""

    transformer = None

    def forward(self, batch, num_steps=None):
        import time

        from sglang_diffusion_amd.runtime.batch_scheduler import request_cost

        time.sleep(0.25 * request_cost(batch))
        batch.latents = batch.latents * 2 + 1
        return batch


def _tiny_request(batch, fill):
    from sglang_diffusion_amd.runtime.denoise import Req

    return Req(latents=torch.full((batch, 4, 8), float(fill)), prompt_embeds=torch.zeros(batch, 2, 8),
               pooled_embeds=torch.zeros(batch, 8), height=64, width=64, num_inference_steps=2)


def _async_case(rank):
    """Staggered arrivals on two ranks.  Rank 0 gets a long request (4 samples = 1 s) and
    computes it; meanwhile a submitter thread keeps admitting short ones, which all go to rank
    1 (least outstanding cost) WHILE it computes, are micro-batched there, and come back to
    rank 0 one by one before rank 0's own long request is done."""
    import threading
    import time

    from sglang_diffusion_amd.runtime.async_scheduler import AsyncDataParallelScheduler

    sch = AsyncDataParallelScheduler(_SleepStage(), rank, WORLD, max_samples_per_step=4)
    sch.start()
    if rank != 0:
        n = sch.serve()
        return {"microbatches": n, "log": [(w, i) for w, i, _ in sch.log],
                "times": {(w, i): t for w, i, t in sch.log}}
    t0 = time.perf_counter()

    def arrivals():
        sch.submit(_tiny_request(4, 10), "long")          # -> rank 0 (both idle, lower rank)
        sch.submit(_tiny_request(1, 1), "a")               # -> rank 1
        time.sleep(0.1)
        sch.submit(_tiny_request(1, 2), "b")               # rank 1 is computing "a": admitted anyway
        time.sleep(0.05)
        sch.submit(_tiny_request(1, 3), "c")
        time.sleep(0.5)
        sch.submit(_tiny_request(1, 4), "d")               # after "a" returned: still rank 1

    th = threading.Thread(target=arrivals)
    th.start()
    time.sleep(0.02)                                       # let the first submissions land
    sch.serve_local()
    th.join()
    while sch.pending():
        sch.serve_local()
    got = dict(sch.results())
    sch.shutdown()
    times = {(w, i): t - t0 for w, i, t in sch.log}
    return {"values": {k: v.flatten()[0].item() for k, v in got.items()},
            "shapes": {k: tuple(v.shape) for k, v in got.items()}, "times": times,
            "order": [i for w, i, _ in sch.log if w == "returned"]}


def test_async_scheduler_admits_while_computing_and_returns_per_request():
    ret = _run(_async_case)
    r0, r1 = ret[0], ret[1]
    # every request came back with the right latents
    assert r0["values"] == {"long": 21.0, "a": 3.0, "b": 5.0, "c": 7.0, "d": 9.0}
    assert r0["shapes"]["long"] == (4, 4, 8) and r0["shapes"]["a"] == (1, 4, 8)
    # per-request return: rank 1's short requests are back before rank 0's long one is done
    order = r0["order"]
    assert order.index("a") < order.index("long") and order.index("b") < order.index("long")
    assert r0["times"][("returned", "a")] < r0["times"][("returned", "long")] - 0.3
    # admission while computing: "b" and "c" reached rank 1 while "a" was still running there
    # (0.25 s), and were then run together as ONE micro-batch
    log1 = r1["log"]
    assert log1.index(("admitted", "b")) < log1.index(("started", "b"))
    t1 = r1["times"]
    assert t1[("admitted", "b")] < t1[("started", "a")] + 0.25
    assert abs(t1[("started", "b")] - t1[("started", "c")]) < 0.05
    assert r1["microbatches"] == 3                          # a | b + c | d
