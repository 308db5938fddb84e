"""The half-format V operand at the edges of IEEE half's range.

The hot path writes the V third of the fused q|k|v projection in half (`f16_cols` of
mi355_linear_ex) and multiplies half probabilities with it (mi355_flash_attn_fwd_ex).  Half
holds bf16 values exactly only for 2^-14 <= |v| <= 65504; these tests send V with magnitudes
from 1e-7 to 1e5 and mixed per-head scales through exactly that pair of kernels and check,
against the CPU oracle, that (a) inside the bound the result holds the 1e-3 bar head by head,
(b) outside it the device-side range counters fire, (c) the bf16-V kernel the denoise stage
falls back to is right there, and (d) the stage does fall back on its own.
"""
import warnings

import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu

B, S, H, DH, K = 2, 320, 4, 128, 256


def _setup(device, head_scales, seed=0):
    """x [B*S, K], a V projection whose output rows of head h are scaled by head_scales[h];
    returns q, k (bf16 [B,S,H,DH] on the device), the projection operands, and the bf16 V the
    plain GEMM gives (the oracle's V)."""
    from sglang_diffusion_amd import _C

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * S, K, generator=g).bfloat16()
    w = torch.randn(H * DH, K, generator=g) * K ** -0.5
    w = (w.view(H, DH, K) * torch.tensor(head_scales).view(H, 1, 1)).view(H * DH, K).bfloat16()
    q = torch.randn(B, S, H, DH, generator=g).bfloat16()
    k = torch.randn(B, S, H, DH, generator=g).bfloat16()
    xd, wd = x.to(device), w.to(device)
    v_bf16 = torch.empty(B * S, H * DH, dtype=torch.bfloat16, device=device)
    _C.linear_ex(xd, wd, v_bf16, M=B * S, N=H * DH, K=K, lda=K, ldw=K, ldo=H * DH)
    return q.to(device), k.to(device), xd, wd, v_bf16


def _half_path(device, q, k, xd, wd):
    """The hot path's pair: GEMM with half output columns, attention with half V."""
    from sglang_diffusion_amd import _C

    v_half = torch.empty(B * S, H * DH, dtype=torch.bfloat16, device=device)   # holds half bits
    _C.linear_ex(xd, wd, v_half, M=B * S, N=H * DH, K=K, lda=K, ldw=K, ldo=H * DH,
                 f16_cols=(0, H * DH))
    out = torch.empty_like(q)
    _C.flash_attn_fwd(q, k, v_half.view(B, S, H, DH), out, None, DH ** -0.5, False, True)
    return out


def _per_head_err(out, ref):
    return [rel_l2(out[:, :, h], ref[:, :, h]) for h in range(H)]


def test_half_v_inside_the_bound_mixed_head_scales(device):
    from oracle import ops as O
    from sglang_diffusion_amd import _C

    # per-head V magnitudes ~ 4e-3, 1, 60 and 6e3 (largest values ~ 4 sigma: 2.5e4 < 65504)
    q, k, xd, wd, v = _setup(device, [4e-3, 1.0, 60.0, 6e3])
    assert v.float().abs().max() < 65504 and v.float().abs().max() > 1e4
    _C.f16_range_reset()
    out = _half_path(device, q, k, xd, wd)
    assert _C.f16_range_status() == (0, 0)
    ref = O.attention(q.cpu(), k.cpu(), v.cpu().view(B, S, H, DH))
    errs = _per_head_err(out, ref)
    print("half V, per-head scales 4e-3 .. 6e3: per-head rel", ["%.2e" % e for e in errs])
    assert max(errs) <= 1e-3
    # single tiny values among normal ones are harmless and are NOT flagged
    v2 = v.clone()
    v2.view(-1)[::97] = 1e-7
    half = torch.empty_like(v2)
    _C.bf16_to_f16(v2, half)
    assert _C.f16_range_status() == (0, 0)
    out2 = torch.empty_like(q)
    _C.flash_attn_fwd(q, k, half.view(B, S, H, DH), out2, None, DH ** -0.5, False, True)
    ref2 = O.attention(q.cpu(), k.cpu(), v2.cpu().view(B, S, H, DH))
    assert max(_per_head_err(out2, ref2)) <= 1e-3


@pytest.mark.parametrize("scales,which", [([1.0, 3e5, 1.0, 1.0], "overflow"),
                                          ([1.0, 1.0, 1e-7, 1.0], "tiny")])
def test_half_v_outside_the_bound_is_counted_and_bf16_v_is_right(device, scales, which):
    from oracle import ops as O
    from sglang_diffusion_amd import _C

    q, k, xd, wd, v = _setup(device, scales, seed=1)
    amax = v.float().abs().view(B * S, H, DH).amax(dim=(0, 2))
    print(f"{which}: per-head max |v| =", ["%.1e" % a for a in amax.tolist()])
    _C.f16_range_reset()
    out_half = _half_path(device, q, k, xd, wd)
    over, tiny = _C.f16_range_status()
    print(f"{which}: range counters after the GEMM's half columns: overflow {over}, tiny {tiny}")
    assert (over > 0) if which == "overflow" else (tiny > 0 and over == 0)
    assert _C.f16_range_status() == (0, 0)                       # reading resets
    # the stand-alone conversion kernel counts the same way
    half = torch.empty_like(v)
    _C.bf16_to_f16(v, half)
    o2, t2 = _C.f16_range_status()
    assert (o2 > 0) if which == "overflow" else (t2 > 0 and o2 == 0)
    ref = O.attention(q.cpu(), k.cpu(), v.cpu().view(B, S, H, DH))
    bad_head = 1 if which == "overflow" else 2
    e_half = _per_head_err(out_half, ref)
    print(f"{which}: half-V per-head rel", ["%.2e" % e for e in e_half])
    assert e_half[bad_head] > 1e-3          # this is what the counters protect against
    # the fallback: bf16 V, probabilities rounded to bf16 (the flash-kernel form; its distance
    # from an fp32-softmax result is 2-3e-3, tests/test_kernels_gpu.py)
    out = torch.empty_like(q)
    _C.flash_attn_fwd(q, k, v.view(B, S, H, DH), out, None, DH ** -0.5, False, False)
    e_bf16 = _per_head_err(out, ref)
    print(f"{which}: bf16-V per-head rel", ["%.2e" % e for e in e_bf16])
    assert max(e_bf16) <= 4e-3


def test_denoise_stage_falls_back_to_bf16_v_on_its_own(device):
    """A model whose value projection overflows half (to_v scaled by 1e5, to_out by 1e-5 so the
    block stays sane): the stage must notice, warn, redo the batch with bf16 V and return the
    oracle's latents -- not saturated ones."""
    from oracle import flux_oracle
    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.models.flux import FluxTransformer2DModel
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights, make_synthetic_request
    from tests.flux_small import SMALL, build_oracle
    from sglang_diffusion_amd.models.flux import FluxArchConfig

    cfg = FluxArchConfig(**SMALL, guidance_embeds=True)
    model = FluxTransformer2DModel(cfg, quant_config=None)
    init_synthetic_weights(model, seed=0, host_rng=True)
    with torch.no_grad():
        a = model.transformer_blocks[0].attn
        a.to_v.weight.mul_(1e5)
        a.to_v.bias.mul_(1e5)
        a.to_out[0].weight.mul_(1e-5)
    state = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}
    model = model.to(device)
    model.process_weights_after_loading()
    oracle = build_oracle(state, cfg, False)
    req = make_synthetic_request(2, height=96, width=128, num_inference_steps=2, device="cpu", seed=3,
                                 text_len=24, joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    ref = flux_oracle.denoise(oracle, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=96, width=128, num_steps=2)
    for name in ("latents", "prompt_embeds", "pooled_embeds"):
        setattr(req, name, getattr(req, name).to(device))
    assert _C.VALUE_HALF
    stage = DenoisingStage(model, fused=True)
    first = {k_: getattr(req, k_).clone() for k_ in ("latents",)}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = stage.forward(req).latents
    assert any("half's range" in str(w.message) for w in caught)
    over, tiny = stage.range_counts()
    assert over > 0
    # the fallback belongs to THIS batch of THIS stage: the process default is untouched, and
    # the stage itself tries half again on its next batch
    assert _C.VALUE_HALF is True and _C.value_half() is True and stage.value_half is None
    e = rel_l2(out, ref)
    print(f"denoise with an overflowing value projection, after the fallback: rel {e:.2e}")
    assert e <= 8e-3
    # a second failing batch in a row: the stage (not the process) gives half up for itself
    req.latents = first["latents"]
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out2 = stage.forward(req).latents
    assert any("keeps bf16 V" in str(w.message) for w in caught)
    assert stage.value_half is False and _C.VALUE_HALF is True
    assert torch.equal(out2, out)
    req.latents = first["latents"]
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out3 = stage.forward(req).latents              # bf16 V from the start: no warning, same bits
    assert not any("half's range" in str(w.message) for w in caught)
    assert torch.equal(out3, out)


def test_range_counts_after_a_graph_replay(device):
    """A caller that replays a captured request batch cannot have the stage read counters in
    the middle: the zeroing, the loop and the copy of the counters to pinned host memory are
    captured with the batch, and `range_counts()` after a replay tells whether that replay's V
    left half's range.  One model that does (to_v scaled by 1e5), one that does not."""
    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.models.flux import FluxArchConfig, FluxTransformer2DModel
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights, make_synthetic_request
    from tests.flux_small import SMALL

    cfg = FluxArchConfig(**SMALL, guidance_embeds=True)
    for scale_v, expect_over in ((1.0, False), (1e5, True)):
        model = FluxTransformer2DModel(cfg, quant_config=None)
        init_synthetic_weights(model, seed=0, host_rng=True)
        with torch.no_grad():
            a = model.transformer_blocks[0].attn
            a.to_v.weight.mul_(scale_v)
            a.to_v.bias.mul_(scale_v)
            a.to_out[0].weight.mul_(1.0 / scale_v)
        model = model.to(device)
        model.process_weights_after_loading()
        stage = DenoisingStage(model, fused=True)
        req = make_synthetic_request(2, height=96, width=128, num_inference_steps=2, device=device,
                                     seed=3, text_len=24, joint_dim=cfg.joint_attention_dim,
                                     pooled_dim=cfg.pooled_projection_dim)
        static_lat = req.latents.clone()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eager = stage.forward(req, num_steps=1).latents.clone()   # warm-up: workspaces exist
        stage.value_half, stage._half_failures = None, 0
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        req.latents = static_lat
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                captured = stage.forward(req, num_steps=1).latents
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(2):
            _C.f16_range_reset(device)
            graph.replay()
            torch.cuda.synchronize()
            over, tiny = stage.range_counts(wait=False)
            print(f"to_v x {scale_v:g}: counters after a replay: overflow {over}, tiny {tiny}")
            assert (over > 0) == expect_over
        if not expect_over:
            assert torch.equal(captured, eager)          # the replay IS the eager batch
