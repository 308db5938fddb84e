"""Rotary embeddings of the DiT path.

NDRotaryEmbedding.forward_uncached mirrors python/sglang/multimodal_gen/runtime/layers/
rotary_embedding/mrope.py:96-125,259-294 (float64 frequency table, built once per shape on
the host: it is a [S, 64] table, not per-step work); apply_flashinfer_rope_qk_inplace mirrors
rotary_embedding/utils.py:42-121 and runs the fused HIP kernel.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .. import _C


class NDRotaryEmbedding(torch.nn.Module):
    def __init__(self, rope_dim_list: List[int], rope_theta: float, use_real: bool = False,
                 repeat_interleave_real: bool = False, dtype: torch.dtype = torch.float64):
        super().__init__()
        if use_real or repeat_interleave_real:
            raise NotImplementedError("the Flux path uses use_real=False")
        self.rope_dim_list = rope_dim_list
        self.rope_theta = rope_theta
        self.dtype = dtype

    def forward_uncached(self, pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """pos [num_tokens, ndim] -> cos, sin [num_tokens, sum(dims)/2] fp32.  Per axis:
        freqs = 1/theta^(arange(0,dim,2)/dim) in self.dtype, outer(pos, freqs), cos/sin cast
        to fp32 (mrope.py:96-125)."""
        cos_parts, sin_parts = [], []
        for i, dim in enumerate(self.rope_dim_list):
            pos_i = pos[:, i].to(self.dtype)
            freqs = 1.0 / (self.rope_theta ** (
                torch.arange(0, dim, 2, dtype=self.dtype, device=pos.device)[: dim // 2] / dim))
            f = torch.outer(pos_i, freqs)
            cos_parts.append(f.cos().float())
            sin_parts.append(f.sin().float())
        return torch.cat(cos_parts, dim=1).float(), torch.cat(sin_parts, dim=1).float()

    def forward(self, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.forward_uncached(positions)


def apply_flashinfer_rope_qk_inplace(q: torch.Tensor, k: torch.Tensor,
                                     cos_sin_cache: torch.Tensor, *,
                                     head_size: Optional[int] = None, is_neox: bool = False,
                                     positions: Optional[torch.Tensor] = None,
                                     round_cos_sin_to_input_dtype: bool = False):
    """utils.py:42-121.  q,k [bsz, seqlen, nheads, head_size]; cos_sin_cache [seqlen, D] =
    cos | sin (fp32).  `round_cos_sin_to_input_dtype=True` reproduces the reference's
    non-flashinfer fallback, which casts cos/sin to q.dtype first (utils.py:80-83)."""
    if q.dim() != 4 or k.dim() != 4:
        raise ValueError("Expected q/k to be 4D [bsz, seqlen, nheads, head_size], "
                         f"got q:{tuple(q.shape)} k:{tuple(k.shape)}")
    if q.shape != k.shape:
        raise ValueError(f"q and k must have the same shape, got {q.shape} vs {k.shape}")
    if not (isinstance(cos_sin_cache, torch.Tensor) and cos_sin_cache.dim() == 2):
        raise ValueError("cos_sin_cache must be a 2D torch.Tensor")
    bsz, seqlen, nheads, d = q.shape
    if head_size is None:
        head_size = d
    if head_size != d:
        raise ValueError(f"head_size mismatch: inferred {d}, but head_size={head_size}")
    half = cos_sin_cache.shape[-1] // 2
    if positions is not None:
        if positions.dtype != torch.long or positions.dim() != 1:
            raise ValueError("positions must be a 1D torch.long Tensor")
        if positions.numel() != bsz * seqlen:
            raise ValueError(f"positions length must be bsz*seqlen={bsz * seqlen}, "
                             f"got {positions.numel()}")
        qf, kf = q.reshape(bsz * seqlen, nheads * d), k.reshape(bsz * seqlen, nheads * d)
        cache = cos_sin_cache.float().contiguous()
        _C.rotary_embedding(positions, qf, kf, d, cache, is_neox)
        return qf.view(bsz, seqlen, nheads, d), kf.view(bsz, seqlen, nheads, d)
    cos = cos_sin_cache[:seqlen, :half]
    sin = cos_sin_cache[:seqlen, half:]
    if round_cos_sin_to_input_dtype:
        cos, sin = cos.to(q.dtype), sin.to(q.dtype)
    cos, sin = cos.float().contiguous(), sin.float().contiguous()
    if q.stride() != k.stride() or q.stride(-1) != 1:
        q, k = q.contiguous(), k.contiguous()
    _C.qk_norm_rope(q, k, None, None, None, None, 0, 0.0, cos, sin, not is_neox)
    return q, k
