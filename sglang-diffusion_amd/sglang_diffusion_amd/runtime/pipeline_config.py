"""Flux pipeline constants and latent/rope preparation.

Mirrors python/sglang/multimodal_gen/configs/pipeline_configs/flux.py:42-197 (latent shape,
packing, RoPE ids, embedded_cfg_scale), configs/pipeline_configs/qwen_image.py:60-69
(_pack_latents, shared by Flux), configs/pipeline_configs/base.py:776-788 (unpack) and
configs/sample/flux.py:9-22 (sampling defaults).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch

VAE_SCALE_FACTOR = 8
EMBEDDED_CFG_SCALE = 3.5
TEXT_LEN = 512  # T5 text_len for both dev and schnell (configs/models/encoders/t5.py:47)


def prepare_latent_shape(batch_size: int, height: int, width: int, in_channels: int = 64):
    h = 2 * (height // (VAE_SCALE_FACTOR * 2))
    w = 2 * (width // (VAE_SCALE_FACTOR * 2))
    return (batch_size, in_channels // 4, h, w)


def pack_latents(latents: torch.Tensor) -> torch.Tensor:
    """[B, C, H, W] -> [B, (H/2)(W/2), 4C]."""
    b, c, h, w = latents.shape
    x = latents.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latents(latents: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """[B, tokens, 4C] -> [B, C, H/8, W/8] (post_denoising_loop, flux.py:157-167)."""
    b, _, ch = latents.shape
    h = 2 * (int(height) // (VAE_SCALE_FACTOR * 2))
    w = 2 * (int(width) // (VAE_SCALE_FACTOR * 2))
    x = latents.view(b, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(b, ch // 4, h, w)


def prepare_latent_image_ids(height: int, width: int, device) -> torch.Tensor:
    h = int(height) // (VAE_SCALE_FACTOR * 2)
    w = int(width) // (VAE_SCALE_FACTOR * 2)
    ids = torch.zeros(h, w, 3, device=device)
    ids[..., 1] = ids[..., 1] + torch.arange(h, device=device)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w, device=device)[None, :]
    return ids.reshape(h * w, 3)


def get_freqs_cis(text_len: int, width: int, height: int, device, rotary_emb
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """txt ids are zeros, img ids (0, row, col); cos/sin concatenated text first."""
    txt_ids = torch.zeros(text_len, 3, device=device)
    img_ids = prepare_latent_image_ids(height, width, device)
    img_cos, img_sin = rotary_emb.forward(img_ids)
    txt_cos, txt_sin = rotary_emb.forward(txt_ids)
    return (torch.cat([txt_cos, img_cos], dim=0).to(device=device),
            torch.cat([txt_sin, img_sin], dim=0).to(device=device))
