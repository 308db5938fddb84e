"""Norm and rotary wrappers (sgl-kernel/python/sgl_kernel/elementwise.py:10-83,270-366)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _ops  # noqa: F401


def rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
            out: Optional[torch.Tensor] = None, enable_pdl: Optional[bool] = None) -> torch.Tensor:
    """out[i] = input[i] / RMS(input[i]) * weight; `enable_pdl` is accepted and has no meaning
    on AMD hardware."""
    if out is None:
        out = torch.empty_like(input)
    torch.ops.sgl_kernel.rmsnorm.default(out, input, weight, eps, bool(enable_pdl))
    return out


def fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                      eps: float = 1e-6, enable_pdl: Optional[bool] = None) -> None:
    """residual += input; input = rmsnorm(residual) * weight (both in place)."""
    torch.ops.sgl_kernel.fused_add_rmsnorm.default(input, residual, weight, eps, bool(enable_pdl))


def _view_3d(x, head_size):
    return x.view(x.shape[0], -1, head_size)


def apply_rope_with_cos_sin_cache_inplace(positions: torch.Tensor, query: torch.Tensor,
                                          key: torch.Tensor, head_size: int,
                                          cos_sin_cache: torch.Tensor, is_neox: bool = True,
                                          fused_set_kv_buffer_arg=None,
                                          enable_pdl: Optional[bool] = None) -> None:
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    if fused_set_kv_buffer_arg is not None:
        raise NotImplementedError("fused_set_kv_buffer_arg is an LLM-serving feature")
    torch.ops.sgl_kernel.apply_rope_pos_ids_cos_sin_cache.default(
        _view_3d(query, head_size), _view_3d(key, head_size), _view_3d(query, head_size),
        _view_3d(key, head_size), cos_sin_cache, positions.long(), (not is_neox), bool(enable_pdl),
        None, None, None, None)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: Optional[torch.Tensor],
                     head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool = True):
    torch.ops.sgl_kernel.rotary_embedding.default(positions, query, key, head_size, cos_sin_cache, is_neox)
