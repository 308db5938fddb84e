"""The benchmark configurations other than the headline one, at sizes a test can afford.

  C1  SD3-Medium-like MM-DiT, 256x256, one denoise step, batch 1, on the CPU reference path:
      the reference has no native SD3 model, so this is the same joint-attention block code
      at SD3-like proportions (24 heads of 64), run by the CPU oracle only (plumbing).
  C2  Flux.1-schnell: no guidance embedder, bf16 everywhere, 4 steps, batch 1.
"""
import pytest
import torch

from oracle import flux_oracle
from sglang_diffusion_amd.models.flux import FluxArchConfig, FluxTransformer2DModel
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights, make_synthetic_request
from tests.flux_small import build_oracle, build_small
from tests.helpers import rel_l2

SD3_LIKE = dict(num_layers=2, num_single_layers=0, attention_head_dim=64, num_attention_heads=24,
                joint_attention_dim=256, pooled_projection_dim=128, in_channels=64,
                axes_dims_rope=(16, 24, 24))


def test_c1_sd3_like_one_step_on_the_cpu_reference_path():
    cfg = FluxArchConfig(**SD3_LIKE, guidance_embeds=False)
    model = FluxTransformer2DModel(cfg, quant_config=None)
    init_synthetic_weights(model, seed=3, host_rng=True)
    orc = build_oracle({k: v.detach().clone() for k, v in model.state_dict().items()}, cfg, False)
    assert orc.D == 1536 and not orc.guidance_embeds
    req = make_synthetic_request(1, height=256, width=256, num_inference_steps=1, device="cpu",
                                 seed=5, text_len=154, joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    assert req.latents.shape == (1, 256, 64)            # (256/16)^2 tokens of 2x2x16 channels
    out = flux_oracle.denoise(orc, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=256, width=256, num_steps=1)
    assert out.shape == req.latents.shape and torch.isfinite(out.float()).all()
    # one Euler step from sigma=1 to 0 is x - v: the update must have moved every token
    assert (out.float() - req.latents.float()).abs().mean() > 1e-3
    # the step is deterministic
    again = flux_oracle.denoise(orc, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                                height=256, width=256, num_steps=1)
    assert torch.equal(out, again)


@pytest.mark.gpu
def test_c1_proportions_on_the_hip_path(device):
    """Same SD3-like proportions (head size 64, 24 heads, no single blocks) through the kernels."""
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage

    model, state, cfg = build_small(device, fp8=False, guidance_embeds=False, seed=3, **SD3_LIKE)
    orc = build_oracle(state, cfg, False)
    req = make_synthetic_request(1, height=256, width=256, num_inference_steps=1, device="cpu",
                                 seed=5, text_len=154, joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    ref = flux_oracle.denoise(orc, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=256, width=256, num_steps=1)
    for name in ("latents", "prompt_embeds", "pooled_embeds"):
        setattr(req, name, getattr(req, name).to(device))
    for fused in (False, True):
        r = make_synthetic_request(1, height=256, width=256, num_inference_steps=1, device=device,
                                   seed=5, text_len=154, joint_dim=cfg.joint_attention_dim,
                                   pooled_dim=cfg.pooled_projection_dim)
        r.latents, r.prompt_embeds, r.pooled_embeds = (req.latents.clone(), req.prompt_embeds,
                                                       req.pooled_embeds)
        out = DenoisingStage(model, fused=fused).forward(r).latents
        e = rel_l2(out, ref)
        print(f"SD3-like, fused={fused}: rel {e:.2e}")
        assert e <= 8e-3


@pytest.mark.gpu
def test_c2_schnell_four_steps_bf16_batch1(device):
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage

    model, state, cfg = build_small(device, fp8=False, guidance_embeds=False)
    orc = build_oracle(state, cfg, False)
    req = make_synthetic_request(1, height=128, width=128, num_inference_steps=4, device="cpu",
                                 seed=7, text_len=32, joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    ref = flux_oracle.denoise(orc, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=128, width=128, num_steps=4)
    for name in ("latents", "prompt_embeds", "pooled_embeds"):
        setattr(req, name, getattr(req, name).to(device))
    out = DenoisingStage(model, fused=True).forward(req).latents
    e = rel_l2(out, ref)
    print(f"schnell-like 4 steps: rel {e:.2e}")
    # four steps of a network whose single forward sits at the 5e-3 two-implementations floor
    assert e <= 1.5e-2


@pytest.mark.gpu
def test_block_quantised_fp8_checkpoint_end_to_end(device):
    """An fp8-serialised checkpoint with 128x128 weight blocks (what the reference's
    tools/convert_hf_to_fp8.py writes) through the plugin layers: per-token-group activation
    quantisation + block-scaled GEMM, against the oracle's restatement of the same path."""
    from sglang_diffusion_amd.runtime.denoise import DenoisingStage
    from tests.flux_small import build_small_block_fp8

    model, state, cfg = build_small_block_fp8(device)
    from oracle.flux_oracle import FluxOracle

    orc = FluxOracle(state, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                     num_heads=cfg.num_attention_heads, head_dim=cfg.attention_head_dim,
                     guidance_embeds=True, fp8_block=True, axes_dims_rope=cfg.axes_dims_rope)
    req = make_synthetic_request(2, height=128, width=96, num_inference_steps=2, device="cpu",
                                 seed=11, text_len=24, joint_dim=cfg.joint_attention_dim,
                                 pooled_dim=cfg.pooled_projection_dim)
    ref = flux_oracle.denoise(orc, req.latents.clone(), req.prompt_embeds, req.pooled_embeds,
                              height=128, width=96, num_steps=2)
    for name in ("latents", "prompt_embeds", "pooled_embeds"):
        setattr(req, name, getattr(req, name).to(device))
    import copy

    out_layers = DenoisingStage(model, fused=False).forward(copy.copy(req)).latents
    out = DenoisingStage(model, fused=True).forward(copy.copy(req)).latents
    e = rel_l2(out, ref)
    print(f"block-quantised checkpoint, 2 steps: rel {e:.2e}")
    assert e <= 2.5e-2
    # the fused schedule keeps every rounding point of the layer path
    assert torch.equal(out, out_layers)
