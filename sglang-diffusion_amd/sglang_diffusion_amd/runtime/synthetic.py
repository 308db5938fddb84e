"""Synthetic weights and inputs of the real shapes (no checkpoints or datasets offline)."""
from __future__ import annotations

import zlib
from typing import Optional

import torch

from . import pipeline_config as pc
from .denoise import Req


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF


@torch.no_grad()
def init_synthetic_weights(model, seed: int = 0, host_rng: bool = True) -> None:
    """weights ~ N(0,1) * fan_in^-0.5, biases ~ N(0,1) * 0.02, RMSNorm weights 1.

    host_rng=True draws every tensor from a CPU generator seeded by the parameter NAME, so a
    model on any device (and the CPU oracle) gets bit-identical bf16 weights; host_rng=False
    draws on the parameter's device (full-size benchmarking: 12 B parameters)."""
    for name, p in model.named_parameters():
        if p.dtype == torch.float32 and p.dim() == 1:  # RMSNorm weight
            p.data.fill_(1.0)
            continue
        dev = torch.device("cpu") if host_rng else p.device
        g = torch.Generator(device=dev)
        g.manual_seed(_seed_for(name, seed))
        if p.dim() == 2:
            w = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) \
                * (p.shape[1] ** -0.5)
        else:
            w = torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * 0.02
        p.data.copy_(w.to(p.dtype))


def hashed_tensor(name: str, shape, seed: int, std: float, device="cpu",
                  dtype=torch.bfloat16) -> torch.Tensor:
    """A pseudo-random tensor that is a pure function of (name, seed, element index): integer
    hashing and ONE fp32 multiply, so a CPU and a GPU produce the very same bits at any size
    (torch.randn streams differ between devices, and 12 B host-side normal draws take minutes).
    Each element is the sum of four 16-bit fields of two 32-bit hashes of its flat index (an
    Irwin-Hall variate: bell-shaped, bounded at +-3.46 std), scaled to standard deviation `std`."""
    n = 1
    for d in shape:
        n *= int(d)
    M = 0xFFFFFFFF
    out = torch.empty(n, dtype=dtype, device=device)
    c = torch.tensor((3.0 ** 0.5) / 65536.0 * float(std), dtype=torch.float32, device=device)
    key = _seed_for(name, seed)
    step = 1 << 24                                   # bounded scratch: 16 M elements at a time

    def mix(x):                                      # every product stays below 2^63
        x = (((x >> 16) ^ x) * 0x45D9F3B) & M
        x = (((x >> 16) ^ x) * 0x45D9F3B) & M
        return (x >> 16) ^ x

    for lo in range(0, n, step):
        hi = min(n, lo + step)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h1 = mix((idx * 2 + key) & M)
        h2 = mix((idx * 2 + 1 + key * 3) & M)
        s = (h1 & 0xFFFF) + (h1 >> 16) + (h2 & 0xFFFF) + (h2 >> 16) - 2 * 65535
        out[lo:hi] = (s.to(torch.float32) * c).to(dtype)
    return out.view(*shape)


def hashed_std(shape) -> float:
    """The scale rule of init_synthetic_weights: fan_in^-0.5 for matrices, 0.02 for vectors."""
    return float(shape[1]) ** -0.5 if len(shape) == 2 else 0.02


@torch.no_grad()
def init_hashed_weights(model, seed: int = 0) -> None:
    """init_synthetic_weights with `hashed_tensor` draws: every device (and a CPU oracle that
    calls hashed_tensor with the same names) gets bit-identical bf16 weights, generated where
    the parameter lives.  Used where the FULL-SIZE network must exist on both sides."""
    for name, p in model.named_parameters():
        if p.dtype == torch.float32 and p.dim() == 1:  # RMSNorm weight
            p.data.fill_(1.0)
            continue
        p.data.copy_(hashed_tensor(name, tuple(p.shape), seed, hashed_std(p.shape), p.device,
                                   torch.bfloat16).to(p.dtype))


@torch.no_grad()
def init_synthetic_block_fp8(model, seed: int = 0, block=(128, 128)) -> None:
    """Fill a model built for an fp8-SERIALISED block-quantised checkpoint: every fp8 weight is
    drawn like init_synthetic_weights draws its bf16 original on the device and then goes
    through the offline converter's block rule (tools/convert_hf_to_fp8.py); everything else is
    initialised as usual."""
    from ..tools.convert_hf_to_fp8 import block_fp8

    params = dict(model.named_parameters())
    for name, p in params.items():
        if name.endswith(".weight_scale_inv"):
            continue
        if p.dtype == torch.float8_e4m3fn:
            g = torch.Generator(device=p.device)
            g.manual_seed(_seed_for(name, seed))
            w = (torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
                 * (p.shape[1] ** -0.5)).to(torch.bfloat16)
            q, s_ = block_fp8(w, block)
            p.data.copy_(q)
            params[name[:-len("weight")] + "weight_scale_inv"].data.copy_(s_)
        elif p.dtype == torch.float32 and p.dim() == 1:
            p.data.fill_(1.0)
        else:
            g = torch.Generator(device=p.device)
            g.manual_seed(_seed_for(name, seed))
            std = (p.shape[1] ** -0.5) if p.dim() == 2 else 0.02
            p.data.copy_((torch.randn(p.shape, generator=g, device=p.device,
                                      dtype=torch.float32) * std).to(p.dtype))


def make_synthetic_request(batch_size: int, height: int = 1024, width: int = 1024,
                           num_inference_steps: int = 50, device="cuda", seed: int = 0,
                           true_cfg: bool = False, guidance_scale: float = 3.5,
                           text_len: int = pc.TEXT_LEN, joint_dim: int = 4096,
                           pooled_dim: int = 768, in_channels: int = 64) -> Req:
    """Latents randn[B,C,H/8,W/8] packed to [B, tokens, 4C] (latent_preparation.py:84-95),
    text states randn*0.1, pooled randn; generated on the host for reproducibility."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    shape = pc.prepare_latent_shape(batch_size, height, width, in_channels)
    lat = pc.pack_latents(torch.randn(shape, generator=g)).to(torch.bfloat16)
    emb = (torch.randn(batch_size, text_len, joint_dim, generator=g) * 0.1).to(torch.bfloat16)
    pooled = torch.randn(batch_size, pooled_dim, generator=g).to(torch.bfloat16)
    req = Req(latents=lat.to(device), prompt_embeds=emb.to(device), pooled_embeds=pooled.to(device),
              height=height, width=width, num_inference_steps=num_inference_steps)
    if true_cfg:
        nemb = (torch.randn(batch_size, text_len, joint_dim, generator=g) * 0.1).to(torch.bfloat16)
        npool = torch.randn(batch_size, pooled_dim, generator=g).to(torch.bfloat16)
        req.negative_prompt_embeds = nemb.to(device)
        req.neg_pooled_embeds = npool.to(device)
        req.guidance_scale = guidance_scale
    return req
