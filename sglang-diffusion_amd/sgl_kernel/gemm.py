"""Python wrappers with the reference's names and argument order
(sgl-kernel/python/sgl_kernel/gemm.py:24-42,99-159)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _ops  # noqa: F401  (registers torch.ops.sgl_kernel.*)


def fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias=None):
    return torch.ops.sgl_kernel.fp8_scaled_mm.default(mat_a, mat_b, scales_a, scales_b, out_dtype, bias)


def fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype):
    return torch.ops.sgl_kernel.fp8_blockwise_scaled_mm.default(mat_a, mat_b, scales_a, scales_b, out_dtype)


def sgl_per_token_group_quant_8bit(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor,
                                   group_size: int, eps: float, fp8_min: float, fp8_max: float,
                                   scale_ue8m0: bool = False, fuse_silu_and_mul: bool = False,
                                   masked_m: Optional[torch.Tensor] = None,
                                   enable_v2: Optional[bool] = None) -> None:
    if enable_v2:
        raise NotImplementedError("sgl_per_token_group_quant_8bit_v2 is not on the diffusion path")
    assert not fuse_silu_and_mul, "only v2 support fuse_silu_and_mul"
    assert masked_m is None, "only v2 support masked_m"
    torch.ops.sgl_kernel.sgl_per_token_group_quant_8bit.default(
        input, output_q, output_s, group_size, eps, fp8_min, fp8_max, scale_ue8m0)


sgl_per_token_group_quant_fp8 = sgl_per_token_group_quant_8bit


def sgl_per_tensor_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor,
                             is_static: bool) -> None:
    torch.ops.sgl_kernel.sgl_per_tensor_quant_fp8.default(input, output_q, output_s, is_static)


def sgl_per_token_quant_fp8(input: torch.Tensor, output_q: torch.Tensor, output_s: torch.Tensor) -> None:
    torch.ops.sgl_kernel.sgl_per_token_quant_fp8.default(input, output_q, output_s)
