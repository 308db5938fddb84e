"""One rank's compute of a HunyuanVideo step under 8-way Ulysses, timed on ONE GPU.

The sequence-parallel configuration (latents [1, 16, 33, 90, 160], patch (1, 2, 2): 118 800
image tokens + 256 text tokens, 20 double + 40 single blocks, 24 heads of 128) gives each of
8 ranks 14 850 image tokens for the linear layers and 3 of the 24 heads over the whole sequence
for attention (layers/attention/layer.py:38-154, layers/usp.py:49-158 of the reference).  With
one GPU per box the exchange itself cannot run, so this script times the two halves a rank
computes and prints their sum:

  t_model   full-size model, latents [1, 16, 33, 90, 20] (the rank's 14 850 tokens) + 256 text
            tokens, 24 local heads over 15 106 tokens
  t_local   the attention launches inside t_model (24 heads, S = 15 106), timed alone x 60
  t_ulysses attention with 3 heads over S = 119 056, timed alone x 60

  rank step (no exchange) = t_model - t_local + t_ulysses

Usage: python tools/hunyuan_time.py [--fp8] [--layers 20 40]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "sglang-diffusion_amd"))

from sglang_diffusion_amd import _C  # noqa: E402
from sglang_diffusion_amd.layers.attention import Mi355AttentionMetadataBuilder  # noqa: E402
from sglang_diffusion_amd.layers.forward_context import set_forward_context  # noqa: E402
from sglang_diffusion_amd.layers.quantization import Fp8Config  # noqa: E402
from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig,  # noqa: E402
                                                      HunyuanVideoTransformer3DModel,
                                                      process_weights_after_loading)
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def attention_time(S, H, iters, dev):
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn(1, S, H, 128, generator=g, device=dev).bfloat16() for _ in range(3))
    vh = torch.empty_like(v, dtype=torch.float16)
    out = torch.empty_like(q)

    def run():
        _C.bf16_to_f16(v.view(S, H * 128), vh.view(S, H * 128))
        _C.flash_attn_fwd(q, k, vh, out, None, 128 ** -0.5, v_is_f16=True)
    return timed(run, iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--layers", type=int, nargs=2, default=[20, 40])
    ap.add_argument("--iters", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nd, ns = a.layers
    cfg = HunyuanVideoArchConfig(num_layers=nd, num_single_layers=ns, guidance_embeds=True)
    qc = (Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic")
          if a.fp8 else None)
    with torch.device(dev):
        model = HunyuanVideoTransformer3DModel(cfg, quant_config=qc)
    init_synthetic_weights(model, seed=0, host_rng=False)
    process_weights_after_loading(model)
    for p in model.parameters():
        p.requires_grad_(False)

    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, 16, 33, 90, 20, generator=g, device=dev).bfloat16()
    enc = (torch.randn(1, 1 + 256, cfg.text_embed_dim, generator=g, device=dev) * 0.1).bfloat16()
    # row 0 of the text input carries the pooled vector in its first pooled_projection_dim columns
    t = torch.tensor([900.0], device=dev)
    guid = torch.tensor([6016.0], device=dev).bfloat16()
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)

    def fwd():
        with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=md):
            return model(x, enc, t, guid)

    out = fwd()
    assert out.shape == x.shape and bool(torch.isfinite(out.float()).all())
    t_model = timed(fwd, a.iters)
    S_loc, S_all = 14850 + 256, 118800 + 256
    t_local = attention_time(S_loc, 24, 4, dev)
    t_uly = attention_time(S_all, 3, 2, dev)
    blocks = nd + ns
    step = t_model - blocks * t_local + blocks * t_uly
    print(json.dumps({
        "workload": "hunyuanvideo 720p 129 frames, one rank of ulysses-8 (no exchange)",
        "dtype": "fp8 W8A8 + bf16" if a.fp8 else "bf16", "blocks": [nd, ns],
        "model_S15106_H24_ms": round(t_model * 1e3, 2),
        "attention_S15106_H24_us": round(t_local * 1e6, 1),
        "attention_S119056_H3_ms": round(t_uly * 1e3, 3),
        "attention_S119056_H3_tflops": round(4.0 * S_all * S_all * 128 * 3 / t_uly / 1e12, 1),
        "rank_step_ms": round(step * 1e3, 1),
    }))


if __name__ == "__main__":
    main()
