"""A small Flux-shaped DiT shared by the GPU parity tests, smoke() and fixtures."""
import torch

from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.flux import FluxArchConfig, FluxTransformer2DModel
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

SMALL = dict(num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
             joint_attention_dim=128, pooled_projection_dim=64, in_channels=64)


def build_small(device, fp8: bool, guidance_embeds: bool = True, seed: int = 0, **over):
    cfg = FluxArchConfig(**{**SMALL, **over}, guidance_embeds=guidance_embeds)
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    model = FluxTransformer2DModel(cfg, quant_config=qc)
    init_synthetic_weights(model, seed=seed, host_rng=True)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    model.process_weights_after_loading()
    return model, state, cfg


def build_oracle(state, cfg, fp8: bool):
    from oracle.flux_oracle import FluxOracle

    return FluxOracle(state, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                      num_heads=cfg.num_attention_heads, head_dim=cfg.attention_head_dim,
                      guidance_embeds=cfg.guidance_embeds, fp8=fp8, axes_dims_rope=cfg.axes_dims_rope)


def small_inputs(B, Ti, Tt, cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(B, Ti, cfg.in_channels, generator=g).bfloat16()
    enc = (torch.randn(B, Tt, cfg.joint_attention_dim, generator=g) * 0.5).bfloat16()
    pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
    t = torch.full((B,), 873.5)
    guid = torch.full((B,), 3.5).to(torch.bfloat16) * 1000.0
    return hid, enc, pooled, t, guid


def build_small_block_fp8(device, guidance_embeds: bool = True, seed: int = 0, **over):
    """The small DiT as an fp8-SERIALISED checkpoint with 128x128 weight blocks: bf16 weights are
    drawn as in build_small, the quantised layers go through the offline converter's rule on
    the host and are loaded as `weight` (fp8) + `weight_scale_inv`, like the reference's loader
    fills them.  Returns (model, bf16 state dict, cfg)."""
    from oracle import ops as O

    cfg = FluxArchConfig(**{**SMALL, **over}, guidance_embeds=guidance_embeds)
    plain = FluxTransformer2DModel(cfg, quant_config=None)
    init_synthetic_weights(plain, seed=seed, host_rng=True)
    state = {k: v.detach().clone() for k, v in plain.state_dict().items()}
    qc = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="dynamic",
                   weight_block_size=[128, 128])
    model = FluxTransformer2DModel(cfg, quant_config=qc)
    with torch.no_grad():
        params = dict(model.named_parameters())
        for name, p in params.items():
            if name.endswith(".weight_scale_inv"):
                continue
            if p.dtype == torch.float8_e4m3fn:
                q, s = O.quantize_weight_block(state[name])
                p.data.copy_(q)
                params[name[:-len("weight")] + "weight_scale_inv"].data.copy_(s)
            else:
                p.data.copy_(state[name].to(p.dtype))
    model = model.to(device)
    model.process_weights_after_loading()
    return model, state, cfg
