"""Norm layers of the DiT path.

RMSNorm / apply_qk_norm mirror python/sglang/multimodal_gen/runtime/layers/layernorm.py:41-215,
531-568; the AdaLayerNorm* modules restate the diffusers==0.36.0 classes the reference's
Flux DiT instantiates (models/dits/flux.py:23-27,438,575-576,823): parameter names are kept
so checkpoints load unchanged.  All forwards run hand-written HIP kernels.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _C
from .quantization import UnquantizedLinearMethod


class CustomOp(nn.Module):
    """layers/custom_op.py:18-114: forward() dispatches to the platform method; on this build
    the platform is always HIP (forward_hip), no CUDA/CPU/NPU variants exist."""

    def forward(self, *args, **kwargs):
        return self.forward_hip(*args, **kwargs)

    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError

    def forward_cuda(self, *args, **kwargs):
        return self.forward_hip(*args, **kwargs)


class RMSNorm(CustomOp):
    """x -> w * x / sqrt(E[x^2] + eps), fp32 math, weight is an fp32 ones(hidden) parameter
    (layernorm.py:49-65)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype = torch.float32,
                 var_hidden_size: Optional[int] = None) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)
        self.variance_epsilon = eps
        self.hidden_size = hidden_size
        if var_hidden_size not in (None, hidden_size):
            raise NotImplementedError("variance_size_override is not used by the DiT path")

    def forward_hip(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None):
        shape = x.shape
        if shape[-1] != self.hidden_size:
            raise ValueError(f"Expected hidden_size to be {self.hidden_size}, but found: {shape[-1]}")
        x2 = x.reshape(-1, shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if residual is not None:
            r2 = residual.reshape(-1, shape[-1]).contiguous()
            x2 = x2.clone()
            _C.fused_add_rmsnorm(x2, r2, self.weight.data, self.variance_epsilon)
            return x2.view(shape), r2.view(residual.shape)
        out = torch.empty_like(x2)
        _C.rmsnorm(out, x2, self.weight.data, self.variance_epsilon)
        return out.view(shape)

    def extra_repr(self) -> str:
        return f"hidden_size={self.weight.data.size(0)}, eps={self.variance_epsilon}"


def apply_qk_norm(q: torch.Tensor, k: torch.Tensor, q_norm: RMSNorm, k_norm: RMSNorm,
                  head_dim: int, allow_inplace: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """layernorm.py:531-568: per-head RMSNorm of q and k; the fused in-place kernel
    (fused_inplace_qknorm, jit_kernel/norm.py:66-90) is the one hand-written HIP launch here."""
    if q_norm.variance_epsilon != k_norm.variance_epsilon or q.shape != k.shape \
            or q.stride() != k.stride() or q.dim() != 4:
        return q_norm(q.reshape(-1, head_dim)).view(q.shape), \
            k_norm(k.reshape(-1, head_dim)).view(k.shape)
    if not allow_inplace:
        q, k = q.clone(), k.clone()
    _C.qk_norm_rope(q, k, q_norm.weight.data, k_norm.weight.data, q_norm.weight.data,
                    k_norm.weight.data, 0, q_norm.variance_epsilon, None, None)
    return q, k


class _ModLinear(nn.Module):
    """nn.Linear stand-in (same parameter names) running the bf16 kernels."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True,
                 dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype),
                                   requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype),
                                 requires_grad=False) if bias else None
        self._method = UnquantizedLinearMethod()

    def forward(self, x: torch.Tensor, act: int = 0) -> torch.Tensor:
        return self._method.apply(self, x, self.bias, act)


def _silu(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    out = torch.empty_like(x2)
    _C.activation(out, x2, 0)
    return out.view(x.shape)


def layernorm_modulate(x: torch.Tensor, shift: Optional[torch.Tensor],
                       scale: Optional[torch.Tensor], eps: float = 1e-6) -> torch.Tensor:
    """LayerNorm(no affine) then *(1+scale[:,None]) + shift[:,None]; x [B,T,D]."""
    B, T, D = x.shape
    x2 = x.reshape(B * T, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    out = torch.empty_like(x2)
    _C.layernorm_modulate(out, x2, shift, scale, B, T, eps)
    return out.view(B, T, D)


class AdaLayerNormZero(nn.Module):
    """diffusers AdaLayerNormZero(dim): emb = linear(silu(emb)) -> 6 chunks;
    x = norm(x)*(1+scale_msa[:,None]) + shift_msa[:,None] (unpacked at flux.py:628-634)."""

    def __init__(self, embedding_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.linear = _ModLinear(embedding_dim, 6 * embedding_dim, dtype=dtype)

    def forward(self, x: torch.Tensor, emb: torch.Tensor):
        e = self.linear(_silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = e.chunk(6, dim=1)
        return layernorm_modulate(x, shift_msa, scale_msa), gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    """diffusers AdaLayerNormZeroSingle(dim) (flux.py:438,511)."""

    def __init__(self, embedding_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.linear = _ModLinear(embedding_dim, 3 * embedding_dim, dtype=dtype)

    def forward(self, x: torch.Tensor, emb: torch.Tensor):
        shift, scale, gate = self.linear(_silu(emb)).chunk(3, dim=1)
        return layernorm_modulate(x, shift, scale), gate


class AdaLayerNormContinuous(nn.Module):
    """diffusers AdaLayerNormContinuous(dim, cond_dim, elementwise_affine=False, eps=1e-6):
    scale, shift = linear(silu(cond)).chunk(2) -- scale FIRST (flux.py:823,912)."""

    def __init__(self, embedding_dim: int, conditioning_embedding_dim: int, eps: float = 1e-6,
                 dtype=torch.bfloat16):
        super().__init__()
        self.eps = eps
        self.linear = _ModLinear(conditioning_embedding_dim, 2 * embedding_dim, dtype=dtype)

    def forward(self, x: torch.Tensor, conditioning_embedding: torch.Tensor) -> torch.Tensor:
        scale, shift = self.linear(_silu(conditioning_embedding)).chunk(2, dim=1)
        return layernorm_modulate(x, shift, scale, self.eps)


class LayerNormNoAffine(nn.Module):
    """torch.nn.LayerNorm(dim, eps, elementwise_affine=False) (flux.py:591-592)."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return layernorm_modulate(x, None, None, self.eps)


def _rows_per_sample(t: Optional[torch.Tensor], B: int, D: int, what: str) -> Optional[torch.Tensor]:
    """[B, D] or [B, 1, D] modulation tensors as [B, D] rows (last dim contiguous)."""
    if t is None:
        return None
    if t.dim() == 3 and t.shape[1] == 1:
        t = t[:, 0]
    if t.dim() != 2 or t.shape != (B, D):
        raise NotImplementedError(f"{what}: expected one [{D}] row per sample, got {tuple(t.shape)} "
                                  "(per-frame [B, F, 1, D] modulation is not built)")
    return t if t.stride(-1) == 1 else t.contiguous()


class LayerNormScaleShift(CustomOp):
    """layers/layernorm.py:450-524 (`_NormScaleShift`, norm_type "layer"): FP32LayerNorm without
    affine parameters, then normed * (1 + scale) + shift, in one kernel with eager's bf16
    rounding points (mi355_layernorm_modulate)."""

    norm_type = "layer"

    def __init__(self, hidden_size: int, eps: float = 1e-6, elementwise_affine: bool = False,
                 dtype: torch.dtype = torch.float32, prefix: str = ""):
        super().__init__()
        if elementwise_affine:
            raise NotImplementedError("LayerNormScaleShift with affine LayerNorm weights is not "
                                      "built (the DiT blocks use elementwise_affine=False)")
        self.hidden_size = hidden_size
        self.eps = eps

    def forward_hip(self, x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        return layernorm_modulate(x, _rows_per_sample(shift, B, D, "LayerNormScaleShift"),
                                  _rows_per_sample(scale, B, D, "LayerNormScaleShift"), self.eps)



class ScaleResidualLayerNormScaleShift(CustomOp):
    """layers/layernorm.py:332-443 (`_ScaleResidualNormScaleShift`, norm_type "layer"):
    residual_out = residual + gate * x, normed = LayerNorm(residual_out),
    out = normed * (1 + scale) + shift; returns (out, residual_out).  One pass over the rows
    (mi355_scale_residual_layernorm_modulate); gate is a [B, D] / [B, 1, D] tensor or the int 1."""

    norm_type = "layer"

    def __init__(self, hidden_size: int, eps: float = 1e-6, elementwise_affine: bool = False,
                 dtype: torch.dtype = torch.float32, prefix: str = ""):
        super().__init__()
        if elementwise_affine:
            raise NotImplementedError("ScaleResidualLayerNormScaleShift with affine LayerNorm "
                                      "weights is not built")
        self.hidden_size = hidden_size
        self.eps = eps

    def forward_hip(self, residual: torch.Tensor, x: torch.Tensor, gate, shift: torch.Tensor,
                    scale: torch.Tensor):
        B, T, D = x.shape
        if isinstance(gate, int):
            if gate != 1:
                raise ValueError(f"Only gate value of 1 is supported for int type, but got {gate}")
            g = None
        elif isinstance(gate, torch.Tensor):
            g = _rows_per_sample(gate, B, D, "ScaleResidualLayerNormScaleShift")
        else:
            raise ValueError(f"Gate type {type(gate)} not supported")
        x2, r2 = x.reshape(B * T, D), residual.reshape(B * T, D)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        r2 = r2 if r2.stride(-1) == 1 else r2.contiguous()
        out = torch.empty((B * T, D), dtype=x.dtype, device=x.device)
        res_out = torch.empty((B * T, D), dtype=x.dtype, device=x.device)
        _C.scale_residual_layernorm_modulate(
            out, res_out, r2, x2, g,
            _rows_per_sample(shift, B, D, "ScaleResidualLayerNormScaleShift"),
            _rows_per_sample(scale, B, D, "ScaleResidualLayerNormScaleShift"), B, T, self.eps)
        return out.view(B, T, D), res_out.view(B, T, D)

