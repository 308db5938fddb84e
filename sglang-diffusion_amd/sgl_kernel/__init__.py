"""Drop-in `sgl_kernel` for MI355X: same module names, functions and `torch.ops.sgl_kernel.*`
schemas as the reference's sgl-kernel wheel for the ops the diffusion DiT step uses
(sgl-kernel/python/sgl_kernel/__init__.py), backed by hand-written gfx950 kernels."""
from . import _ops  # noqa: F401
from .attention import merge_state_v2
from .elementwise import (apply_rope_with_cos_sin_cache_inplace, fused_add_rmsnorm, rmsnorm,
                          rotary_embedding)
from .flash_attn import flash_attn_varlen_func
from .gemm import (fp8_blockwise_scaled_mm, fp8_scaled_mm, sgl_per_tensor_quant_fp8,
                   sgl_per_token_group_quant_8bit, sgl_per_token_group_quant_fp8,
                   sgl_per_token_quant_fp8)

__all__ = [
    "apply_rope_with_cos_sin_cache_inplace", "fused_add_rmsnorm", "rmsnorm", "rotary_embedding",
    "flash_attn_varlen_func", "fp8_blockwise_scaled_mm", "fp8_scaled_mm", "sgl_per_tensor_quant_fp8",
    "sgl_per_token_group_quant_8bit", "sgl_per_token_group_quant_fp8", "sgl_per_token_quant_fp8",
    "merge_state_v2",
]
