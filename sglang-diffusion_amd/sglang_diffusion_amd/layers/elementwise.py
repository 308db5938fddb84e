"""Fused elementwise layers of the DiT blocks (python/sglang/multimodal_gen/runtime/layers/
elementwise.py)."""
from __future__ import annotations

import torch

from .. import _C
from .layernorm import CustomOp, _rows_per_sample


class MulAdd(CustomOp):
    """elementwise.py:7-38: a * (k + b) + c with b one row per sample ([B, D] / [B, 1, D]) and
    a, c [B, T, D]: the gated residual update of the MM-DiT blocks.  k = 0 (the only value the
    DiT blocks use) runs mi355_gate_residual: bf16(c + bf16(a * b))."""

    def __init__(self, prefix: str = ""):
        super().__init__()

    def forward_hip(self, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, k: int = 0) -> torch.Tensor:
        if k != 0:
            raise NotImplementedError("MulAdd with k != 0 is not built")
        B, T, D = a.shape
        gate = _rows_per_sample(b, B, D, "MulAdd")
        a2, c2 = a.reshape(B * T, D), c.reshape(B * T, D)
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        c2 = c2 if c2.is_contiguous() else c2.contiguous()
        out = torch.empty_like(c2)
        _C.gate_residual(out, c2, gate, a2, B, T)
        return out.view(B, T, D)

