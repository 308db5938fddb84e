"""`merge_state_v2` with the reference's signature (sgl-kernel/python/sgl_kernel/attention.py:25-45)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _ops  # noqa: F401


def merge_state_v2(v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor,
                   v_merged: Optional[torch.Tensor] = None,
                   s_merged: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    s_a = s_a.to(torch.float32)
    s_b = s_b.to(torch.float32)
    if v_merged is None:
        v_merged = torch.empty_like(v_a)
    if s_merged is None:
        s_merged = torch.empty_like(s_a)
    torch.ops.sgl_kernel.merge_state_v2.default(v_a, s_a, v_b, s_b, v_merged, s_merged)
    return v_merged, s_merged
