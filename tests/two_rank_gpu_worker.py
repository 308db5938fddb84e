"""Rank process of tests/test_zy_two_rank_exchange_gpu.py: two of these share cuda:0 and talk over
gloo (device tensors are staged through the host by runtime/parallel.py).  Every exchange of the
sequence-parallel attention wrappers runs here with the HIP attention / merge kernels inside it
and is compared with dense HIP attention on one rank and with the CPU oracle.  Rank 0 prints one
JSON line with the findings."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import ops as O  # noqa: E402
from sglang_diffusion_amd import _C  # noqa: E402
from sglang_diffusion_amd.layers.attention import UlyssesAttention, USPAttention  # noqa: E402
from sglang_diffusion_amd.layers.forward_context import set_forward_context  # noqa: E402
from sglang_diffusion_amd.runtime.parallel import (cfg_parallel_combine, init_distributed,  # noqa: E402
                                                   ring_attention, ulysses_all_to_all)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def dense(q, k, v):
    out = torch.empty_like(q)
    _C.attention(q, k, v, out, None, q.shape[-1] ** -0.5)
    return out


def half_v_guard_with_cfg_parallel(rank, dev, group):
    """The denoise stage with the two CFG branches on the two ranks (an all-reduce and, with
    guidance rescale, a broadcast in EVERY step) when only ONE rank's half-format V leaves
    half's range: both ranks must re-run the batch with bf16 V together -- a rank that went on
    alone would sit in a collective its peer never joins.  The overflow is a real one, counted
    by the device counters: rank 1 converts a tensor holding 1e6 to half right behind its
    first pass over the batch."""
    import warnings

    from sglang_diffusion_amd.models.flux import FluxArchConfig, FluxTransformer2DModel
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights, make_synthetic_request
    from tests.flux_small import SMALL

    cfg = FluxArchConfig(**SMALL, guidance_embeds=True)
    model = FluxTransformer2DModel(cfg, quant_config=None)
    init_synthetic_weights(model, seed=0, host_rng=True)
    model = model.to(dev)
    model.process_weights_after_loading()

    def request():
        return make_synthetic_request(2, height=96, width=128, num_inference_steps=2, device=dev,
                                      seed=3, text_len=24, joint_dim=cfg.joint_attention_dim,
                                      pooled_dim=cfg.pooled_projection_dim, true_cfg=True,
                                      guidance_scale=3.5)

    # what the pair computes with bf16 V from the start
    want = DenoisingStage(model, fused=True, cfg_group=group, value_half=False).forward(request()).latents
    stage = DenoisingStage(model, fused=True, cfg_group=group)
    passes = []
    inner = stage._loop

    def loop(batch, num_steps=None):
        out = inner(batch, num_steps)
        passes.append(_C.value_half())
        if rank == 1 and len(passes) == 1:
            big = torch.full((8, 128), 1e6, dtype=torch.bfloat16, device=dev)
            _C.bf16_to_f16(big, torch.empty_like(big))
        return out

    stage._loop = loop
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = stage.forward(request()).latents
    torch.cuda.synchronize()
    warned = any("half's range" in str(w.message) for w in caught)
    ok = passes == [True, False] and warned and bool(torch.equal(got, want))
    flag = torch.tensor([int(ok)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return {"half_v_guard_ranks_decide_together_exact": bool(flag.item())}


def main():
    rank, world, _ = init_distributed("gloo")
    assert world == 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = dist.group.WORLD
    res = {}
    B, S, T, H, D = 1, 512, 96, 8, 128          # S video / image tokens sharded, T text tokens
    gen = torch.Generator().manual_seed(11)
    qkv = torch.randn(3, B, S, H, D, generator=gen).bfloat16().to(dev)
    tqkv = torch.randn(3, B, T, H, D, generator=gen).bfloat16().to(dev)
    sl = slice(rank * S // world, (rank + 1) * S // world)

    with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=None):
        # ---- the exchange itself on device tensors (strided input: a slice of a q|k|v buffer)
        full = torch.arange(B * S * H * D, dtype=torch.float32, device=dev).view(B, S, H, D).bfloat16()
        heads = ulysses_all_to_all(full[:, sl], g, scatter_dim=2, gather_dim=1)
        res["a2a_forward_exact"] = bool(torch.equal(heads, full[:, :, rank * H // 2:(rank + 1) * H // 2]))
        back = ulysses_all_to_all(heads, g, scatter_dim=1, gather_dim=2)
        res["a2a_round_trip_exact"] = bool(torch.equal(back, full[:, sl]))

        # ---- UlyssesAttention with replicated text (HunyuanVideo), HIP kernel inside
        layer = UlyssesAttention(H, D, group=g)
        out, rep = layer(qkv[0][:, sl].contiguous(), qkv[1][:, sl].contiguous(),
                         qkv[2][:, sl].contiguous(), tqkv[0], tqkv[1], tqkv[2])
        joint = [torch.cat([qkv[i], tqkv[i]], dim=1) for i in range(3)]
        one = dense(*joint)                                     # the same kernel on one rank
        res["ulysses_vs_dense_hip_exact"] = bool(torch.equal(out, one[:, :S][:, sl]) and
                                                 torch.equal(rep, one[:, S:]))
        ref = O.attention(*[t.cpu() for t in joint])
        res["ulysses_vs_oracle"] = max(rel(out, ref[:, :S][:, sl]), rel(rep, ref[:, S:]))

        # ---- USPAttention: Flux joint attention with the text prefix replicated
        local = [torch.cat([tqkv[i], qkv[i][:, sl]], dim=1) for i in range(3)]
        usp = USPAttention(H, D, ulysses_group=g)
        o2 = usp(local[0], local[1], local[2], num_replicated_prefix=T)
        joint2 = [torch.cat([tqkv[i], qkv[i]], dim=1) for i in range(3)]
        one2 = dense(*joint2)
        want2 = torch.cat([one2[:, :T], one2[:, T:][:, sl]], dim=1)
        res["usp_prefix_vs_dense_hip_exact"] = bool(torch.equal(o2, want2))
        ref2 = O.attention(*[t.cpu() for t in joint2])
        res["usp_prefix_vs_oracle"] = rel(o2, torch.cat([ref2[:, :T], ref2[:, T:][:, sl]], dim=1))

        # ---- ring attention: K/V shards travel, partial results merged by the HIP merge kernel
        r_out, r_lse = ring_attention(qkv[0][:, sl].contiguous(), qkv[1][:, sl].contiguous(),
                                      qkv[2][:, sl].contiguous(), g)
        one3 = torch.empty_like(qkv[0][:, sl])
        lse3 = torch.empty(B, H, S // world, dtype=torch.float32, device=dev)
        _C.attention(qkv[0][:, sl].contiguous(), qkv[1], qkv[2], one3, lse3, D ** -0.5)
        res["ring_vs_dense_hip"] = rel(r_out, one3)
        res["ring_lse_max_abs"] = float((r_lse - lse3).abs().max())
        ref3 = O.attention(qkv[0][:, sl].cpu(), qkv[1].cpu(), qkv[2].cpu())
        res["ring_vs_oracle"] = rel(r_out, ref3)
        ring_layer = USPAttention(H, D, ring_group=g)
        r2 = ring_layer(qkv[0][:, sl].contiguous(), qkv[1][:, sl].contiguous(), qkv[2][:, sl].contiguous())
        res["usp_ring_equals_ring_exact"] = bool(torch.equal(r2, r_out))

        # ---- CFG parallel: s * cond on rank 0 + (1 - s) * uncond on rank 1, one all-reduce
        cond = torch.randn(2, 256, 64, generator=gen).bfloat16()
        uncond = torch.randn(2, 256, 64, generator=gen).bfloat16()
        mine = (cond if rank == 0 else uncond).to(dev)
        comb = cfg_parallel_combine(mine, 3.5, g)
        res["cfg_parallel_exact"] = bool(torch.equal(comb.cpu(), 3.5 * cond + (1 - 3.5) * uncond))
    res.update(half_v_guard_with_cfg_parallel(rank, dev, g))
    torch.cuda.synchronize()
    dist.barrier()
    flags = torch.tensor([int(all(v for k, v in res.items() if k.endswith("exact")))])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    res["exact_on_every_rank"] = bool(flags.item())
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
