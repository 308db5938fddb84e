"""Timestep / text conditioning embeddings of the Flux DiT.

Timesteps.forward mirrors python/sglang/multimodal_gen/runtime/layers/visual_embedding.py:89-106
(fp32 sinusoid); TimestepEmbedding, PixArtAlphaTextProjection and the Combined*Embeddings
forward restate diffusers==0.36.0 (instantiated at visual_embedding.py:109-145); FeedForward
restates layers/mlp.py:68-118 (diffusers GELU(approximate="tanh") + Linear).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _C
from .layernorm import _ModLinear, _silu


class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float,
                 scale: int = 1):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift
        self.scale = scale

    def forward(self, timesteps: torch.Tensor, out_dtype: torch.dtype = torch.float32):
        t = timesteps.to(torch.float32).contiguous()
        out = torch.empty((t.shape[0], self.num_channels), dtype=out_dtype, device=t.device)
        _C.timestep_embedding(t, out, self.flip_sin_to_cos, self.downscale_freq_shift, self.scale,
                              10000.0)
        return out


class TimestepEmbedding(nn.Module):
    """linear_2(silu(linear_1(x)))."""

    def __init__(self, in_channels: int, time_embed_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.linear_1 = _ModLinear(in_channels, time_embed_dim, dtype=dtype)
        self.linear_2 = _ModLinear(time_embed_dim, time_embed_dim, dtype=dtype)

    def forward(self, sample: torch.Tensor) -> torch.Tensor:
        return self.linear_2(_silu(self.linear_1(sample)))


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features: int, hidden_size: int, act_fn: str = "silu",
                 dtype=torch.bfloat16):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError("Flux uses act_fn='silu'")
        self.linear_1 = _ModLinear(in_features, hidden_size, dtype=dtype)
        self.linear_2 = _ModLinear(hidden_size, hidden_size, dtype=dtype)

    def forward(self, caption: torch.Tensor) -> torch.Tensor:
        return self.linear_2(_silu(self.linear_1(caption)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    """Flux.1-schnell conditioning (no guidance embedding)."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim, dtype=dtype)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim,
                                                       dtype=dtype)

    def forward(self, timestep: torch.Tensor, pooled_projection: torch.Tensor) -> torch.Tensor:
        t = self.timestep_embedder(self.time_proj(timestep, out_dtype=pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    """Flux.1-dev conditioning: (t_emb + g_emb) + text_emb, bf16 adds."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim, dtype=dtype)
        self.guidance_embedder = TimestepEmbedding(256, embedding_dim, dtype=dtype)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim,
                                                       dtype=dtype)

    def forward(self, timestep, guidance, pooled_projection) -> torch.Tensor:
        dt = pooled_projection.dtype
        t = self.timestep_embedder(self.time_proj(timestep, out_dtype=dt))
        g = self.guidance_embedder(self.time_proj(guidance, out_dtype=dt))
        return (t + g) + self.text_embedder(pooled_projection)


class _GELUProj(nn.Module):
    """diffusers GELU(dim_in, dim_out, approximate='tanh'): gelu(proj(x)); the activation is
    fused in the GEMM epilogue after the bf16 rounding of the linear output."""

    def __init__(self, dim_in: int, dim_out: int, dtype=torch.bfloat16):
        super().__init__()
        self.proj = _ModLinear(dim_in, dim_out, dtype=dtype)

    def forward(self, x):
        return self.proj(x, act=1)


class FeedForward(nn.Module):
    """mlp.py:68-118: net = [GELU(dim, 4*dim, tanh), Dropout(0), Linear(4*dim, dim_out)]
    -> parameter names ff.net.0.proj.*, ff.net.2.*."""

    def __init__(self, dim: int, dim_out: int | None = None, mult: int = 4,
                 activation_fn: str = "gelu-approximate", dtype=torch.bfloat16):
        super().__init__()
        if activation_fn != "gelu-approximate":
            raise NotImplementedError("Flux uses activation_fn='gelu-approximate'")
        inner = int(dim * mult)
        self.net = nn.ModuleList([_GELUProj(dim, inner, dtype), nn.Identity(),
                                  _ModLinear(inner, dim_out or dim, dtype=dtype)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net[2](self.net[0](x))
