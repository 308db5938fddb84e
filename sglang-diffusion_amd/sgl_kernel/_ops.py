"""`torch.ops.sgl_kernel.*` for MI355X: the reference's op registry, backed by libmi355dit.so.

The reference registers these ops from C++ (`TORCH_LIBRARY_FRAGMENT(sgl_kernel, m)`,
sgl-kernel/csrc/common_extension.cc:21,66-97,134-158 and flash_extension.cc:21-66) and its
Python layers call them as `torch.ops.sgl_kernel.<name>.default(...)`.  Here the same schemas
are defined from Python and each one is implemented for the `CUDA` dispatch key (= HIP devices
under PyTorch-ROCm) by a call through the C ABI of include/mi355_dit.h.  There is no CPU
implementation: CPU tensors hit the dispatcher's "no kernel for backend CPU" error, and a
missing shared object raises on first use.  `torch.library.register_fake` shims give the
shape/dtype rules for tracing, as the reference does in
python/sglang/srt/layers/quantization/fp8_utils.py:69-81 and fp8_kernel.py:2057-2076.

Conventions kept from the reference: `Tensor!` outputs are caller-allocated and written in
place; functional ops allocate their result on the input's device; kernels are enqueued on the
current stream and never synchronise; unsupported arguments raise RuntimeError instead of being
ignored.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from sglang_diffusion_amd import _C

FP8 = torch.float8_e4m3fn

SCHEMAS = {
    # sgl-kernel/csrc/common_extension.cc:134-158 (gemm section)
    "fp8_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, "
                     "ScalarType out_dtype, Tensor? bias) -> Tensor",
    "fp8_blockwise_scaled_mm": "(Tensor mat_a, Tensor mat_b, Tensor scales_a, Tensor scales_b, "
                               "ScalarType out_dtype) -> Tensor",
    "sgl_per_token_quant_fp8": "(Tensor input, Tensor! output_q, Tensor! output_s) -> ()",
    "sgl_per_token_group_quant_8bit": "(Tensor input, Tensor! output_q, Tensor! output_s, "
                                      "int group_size, float eps, float fp8_min, float fp8_max, "
                                      "bool scale_ue8m0) -> ()",
    "sgl_per_tensor_quant_fp8": "(Tensor input, Tensor! output_q, Tensor! output_s, "
                                "bool is_static) -> ()",
    # sgl-kernel/csrc/common_extension.cc:66-97 (elementwise section)
    "rmsnorm": "(Tensor! output, Tensor input, Tensor weight, float eps, bool enable_pdl) -> ()",
    "fused_add_rmsnorm": "(Tensor! input, Tensor! residual, Tensor weight, float eps, "
                         "bool enable_pdl) -> ()",
    "rotary_embedding": "(Tensor positions, Tensor! query, Tensor!? key, int head_size, "
                        "Tensor cos_sin_cache, bool is_neox) -> ()",
    "apply_rope_pos_ids_cos_sin_cache": "(Tensor q, Tensor k, Tensor! q_rope, Tensor! k_rope, "
                                        "Tensor cos_sin_cache, Tensor pos_ids, bool interleave, "
                                        "bool enable_pdl, Tensor? v, Tensor!? k_buffer, "
                                        "Tensor!? v_buffer, Tensor? kv_cache_loc) -> ()",
    # sgl-kernel/csrc/common_extension.cc:55-56
    "merge_state_v2": "(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged, "
                      "Tensor! s_merged) -> ()",
    # sgl-kernel/csrc/flash_extension.cc:25-63
    "fwd": "(Tensor q, Tensor k, Tensor v, Tensor? k_new, Tensor? v_new, Tensor? q_v, "
           "Tensor? out, Tensor? cu_seqlens_q, Tensor? cu_seqlens_k, Tensor? cu_seqlens_k_new, "
           "Tensor? seqused_q, Tensor? seqused_k, int? max_seqlen_q, int? max_seqlen_k, "
           "Tensor? page_table, Tensor? kv_batch_idx, Tensor? leftpad_k, Tensor? rotary_cos, "
           "Tensor? rotary_sin, Tensor? seqlens_rotary, Tensor? q_descale, Tensor? k_descale, "
           "Tensor? v_descale, float? softmax_scale, bool is_causal, int window_left, "
           "int window_right, int attention_chunk, float softcap, bool is_rotary_interleaved, "
           "Tensor? scheduler_metadata, int num_splits, bool? pack_gqa, int sm_margin, "
           "Tensor? sinks) -> (Tensor, Tensor, Tensor, Tensor)",
}

_lib = torch.library.Library("sgl_kernel", "FRAGMENT")
for _name, _schema in SCHEMAS.items():
    _lib.define(_name + _schema)


def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


# ------------------------------------------------------------------------------------ GEMM --
def _check_mm(mat_a, mat_b, out_dtype, what):
    """Layout preconditions of sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1457-1487."""
    _require(mat_a.dim() == 2 and mat_b.dim() == 2, f"{what}: mat_a and mat_b must be 2-D")
    _require(mat_a.dtype == FP8 and mat_b.dtype == FP8, f"{what}: operands must be float8_e4m3fn")
    _require(mat_a.stride(1) == 1, f"{what}: mat_a must be row major")
    _require(mat_b.stride(0) == 1, f"{what}: mat_b must be column major (a [N,K] weight, transposed)")
    _require(mat_a.shape[1] == mat_b.shape[0], f"{what}: inner dimensions differ")
    _require(out_dtype in (torch.bfloat16, torch.float16),
             f"{what}: out_dtype must be bfloat16 or float16")
    _require(mat_a.shape[1] % 16 == 0, f"{what}: K must be a multiple of 16")
    _require(mat_b.shape[1] % 8 == 0, f"{what}: N must be a multiple of 8")


def _fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias):
    _check_mm(mat_a, mat_b, out_dtype, "fp8_scaled_mm")
    M, K = mat_a.shape
    N = mat_b.shape[1]
    _require(scales_a.numel() == M and scales_b.numel() == N,
             "fp8_scaled_mm: scales_a must have M and scales_b N elements")
    _require(scales_a.dtype == torch.float32 and scales_b.dtype == torch.float32,
             "fp8_scaled_mm: scales must be float32")
    _require(scales_a.is_contiguous() and scales_b.is_contiguous(),
             "fp8_scaled_mm: scales must be contiguous")
    if bias is not None:
        _require(bias.numel() == N and bias.dtype == out_dtype and bias.is_contiguous(),
                 "fp8_scaled_mm: bias must be a contiguous [N] tensor of out_dtype")
    out = torch.empty((M, N), dtype=out_dtype, device=mat_a.device)
    if M:
        _C.fp8_scaled_mm(mat_a, mat_b.t(), scales_a, scales_b, bias, out)
    return out


def _fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype):
    """scales_a [M, K/128] (either stride order), scales_b [K/128, N/128] (either order):
    sgl-kernel/csrc/gemm/fp8_blockwise_gemm_kernel.cu:374-406."""
    _check_mm(mat_a, mat_b, out_dtype, "fp8_blockwise_scaled_mm")
    M, K = mat_a.shape
    N = mat_b.shape[1]
    kb, nb = (K + 127) // 128, (N + 127) // 128
    _require(tuple(scales_a.shape) == (M, kb), "fp8_blockwise_scaled_mm: scales_a must be [M, ceil(K/128)]")
    _require(tuple(scales_b.shape) == (kb, nb),
             "fp8_blockwise_scaled_mm: scales_b must be [ceil(K/128), ceil(N/128)]")
    _require(scales_a.dtype == torch.float32 and scales_b.dtype == torch.float32,
             "fp8_blockwise_scaled_mm: scales must be float32")
    out = torch.empty((M, N), dtype=out_dtype, device=mat_a.device)
    if M:
        _C.fp8_blockwise_scaled_mm(mat_a, mat_b.t(), scales_a, scales_b, out,
                                   (scales_a.stride(0), scales_a.stride(1)),
                                   (scales_b.stride(0), scales_b.stride(1)))
    return out


# ----------------------------------------------------------------------------------- quant --
def _as_rows(x: torch.Tensor, what: str) -> torch.Tensor:
    _require(x.stride(-1) == 1, f"{what}: last dimension must be contiguous")
    if x.dim() == 2:
        return x
    _require(x.is_contiguous(), f"{what}: inputs with more than 2 dimensions must be contiguous")
    return x.view(-1, x.shape[-1])


def _sgl_per_token_quant_fp8(input, output_q, output_s):
    x = _as_rows(input, "sgl_per_token_quant_fp8")
    _require(input.dtype == torch.bfloat16, "sgl_per_token_quant_fp8: input must be bfloat16")
    _require(output_q.dtype == FP8 and output_q.is_contiguous(),
             "sgl_per_token_quant_fp8: output_q must be contiguous float8_e4m3fn")
    _require(output_s.dtype == torch.float32 and output_s.numel() == x.shape[0],
             "sgl_per_token_quant_fp8: output_s must hold one float32 per token")
    _C.per_token_quant_fp8(x, output_q.view(x.shape), output_s)


def _sgl_per_token_group_quant_8bit(input, output_q, output_s, group_size, eps, fp8_min, fp8_max,
                                    scale_ue8m0):
    x = _as_rows(input, "sgl_per_token_group_quant_8bit")
    _require(input.dtype == torch.bfloat16, "sgl_per_token_group_quant_8bit: input must be bfloat16")
    _require(output_q.dtype == FP8, "sgl_per_token_group_quant_8bit: only float8_e4m3fn output "
                                    "is implemented (int8 is not on the diffusion path)")
    _require(output_q.is_contiguous(), "sgl_per_token_group_quant_8bit: output_q must be contiguous")
    M, K = x.shape
    _require(K % group_size == 0, "sgl_per_token_group_quant_8bit: K must divide by group_size")
    s = output_s if output_s.dim() == 2 else output_s.view(M, K // group_size)
    _require(tuple(s.shape) == (M, K // group_size),
             "sgl_per_token_group_quant_8bit: output_s must be [M, K/group_size]")
    _C.per_token_group_quant_fp8(x, output_q.view(M, K), s, group_size, eps, fp8_min, fp8_max,
                                 scale_ue8m0)


def _sgl_per_tensor_quant_fp8(input, output_q, output_s, is_static):
    _require(input.is_contiguous() and output_q.is_contiguous(),
             "sgl_per_tensor_quant_fp8: tensors must be contiguous")
    _require(input.dtype == torch.bfloat16 and output_q.dtype == FP8,
             "sgl_per_tensor_quant_fp8: bfloat16 -> float8_e4m3fn only")
    _require(output_s.dtype == torch.float32 and output_s.numel() == 1,
             "sgl_per_tensor_quant_fp8: output_s must be one float32")
    _C.per_tensor_quant_fp8(input, output_q, output_s, is_static)


# ------------------------------------------------------------------------------ norm / rope --
def _rmsnorm(output, input, weight, eps, enable_pdl):
    x = _as_rows(input, "rmsnorm")
    o = _as_rows(output, "rmsnorm")
    _require(input.dtype == torch.bfloat16 and output.dtype == torch.bfloat16,
             "rmsnorm: bfloat16 activations only")
    _require(weight.numel() == x.shape[1] and weight.is_contiguous(), "rmsnorm: weight must be [hidden]")
    _C.rmsnorm(o, x, weight, eps)


def _fused_add_rmsnorm(input, residual, weight, eps, enable_pdl):
    _require(input.is_contiguous() and residual.is_contiguous(),
             "fused_add_rmsnorm: input and residual must be contiguous")
    _require(input.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16,
             "fused_add_rmsnorm: bfloat16 only")
    _C.fused_add_rmsnorm(input.view(-1, input.shape[-1]), residual.view(-1, input.shape[-1]),
                         weight, eps)


def _rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox):
    _require(positions.dtype == torch.int64 and positions.is_contiguous(),
             "rotary_embedding: positions must be contiguous int64")
    _require(query.dtype == torch.bfloat16, "rotary_embedding: bfloat16 only")
    _require(cos_sin_cache.dtype in (torch.float32, torch.bfloat16) and cos_sin_cache.is_contiguous(),
             "rotary_embedding: cos_sin_cache must be contiguous float32 or bfloat16")
    _C.rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)


def _apply_rope_pos_ids_cos_sin_cache(q, k, q_rope, k_rope, cos_sin_cache, pos_ids, interleave,
                                      enable_pdl, v, k_buffer, v_buffer, kv_cache_loc):
    _require(v is None and k_buffer is None and v_buffer is None and kv_cache_loc is None,
             "apply_rope_pos_ids_cos_sin_cache: the fused KV-cache write is an LLM-serving "
             "feature and is not implemented on the diffusion path")
    _require(q.dim() == 3 and k.dim() == 3, "apply_rope_pos_ids_cos_sin_cache: q, k must be [nnz, H, d]")
    _require(cos_sin_cache.dtype == torch.float32, "cos_sin_cache should be float32")
    if q_rope.data_ptr() != q.data_ptr():
        q_rope.copy_(q)
    if k_rope.data_ptr() != k.data_ptr():
        k_rope.copy_(k)
    _require(q_rope.stride(-1) == 1 and q_rope.stride(1) == q_rope.shape[2]
             and k_rope.stride(-1) == 1 and k_rope.stride(1) == k_rope.shape[2],
             "apply_rope_pos_ids_cos_sin_cache: heads must be packed within a token")
    nnz, _, d = q_rope.shape
    qv = q_rope.as_strided((nnz, q_rope.shape[1] * d), (q_rope.stride(0), 1))
    kv = k_rope.as_strided((nnz, k_rope.shape[1] * d), (k_rope.stride(0), 1))
    _C.rotary_embedding(pos_ids.contiguous(), qv, kv, d, cos_sin_cache, not interleave)


# -------------------------------------------------------------------------------- attention --
def _fwd(q, k, v, k_new, v_new, q_v, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q,
         seqused_k, max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos,
         rotary_sin, seqlens_rotary, q_descale, k_descale, v_descale, softmax_scale, is_causal,
         window_left, window_right, attention_chunk, softcap, is_rotary_interleaved,
         scheduler_metadata, num_splits, pack_gqa, sm_margin, sinks):
    """Dense non-causal attention, the only form the diffusion path issues
    (python/sglang/multimodal_gen/runtime/layers/attention/backends/flash_attn.py:517-563).
    Batched [B,S,H,d] inputs, or packed [T,H,d] inputs whose sequences all have the maximum
    length.  Everything the LLM-serving callers use besides that is rejected."""
    unsupported = dict(k_new=k_new, v_new=v_new, q_v=q_v, cu_seqlens_k_new=cu_seqlens_k_new,
                       seqused_q=seqused_q, seqused_k=seqused_k, page_table=page_table,
                       kv_batch_idx=kv_batch_idx, leftpad_k=leftpad_k, rotary_cos=rotary_cos,
                       rotary_sin=rotary_sin, seqlens_rotary=seqlens_rotary, q_descale=q_descale,
                       k_descale=k_descale, v_descale=v_descale, sinks=sinks)
    bad = [n for n, t in unsupported.items() if t is not None]
    _require(not bad, f"fwd: arguments not supported on the diffusion path: {bad}")
    _require(not is_causal, "fwd: causal attention is not implemented (diffusion attention is dense)")
    _require(window_left < 0 and window_right < 0, "fwd: sliding windows are not implemented")
    _require(attention_chunk == 0 and softcap == 0.0, "fwd: attention_chunk / softcap are not implemented")
    _require(q.dtype == torch.bfloat16 and k.dtype == q.dtype and v.dtype == q.dtype,
             "fwd: bfloat16 q, k, v only")
    _require(q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1,
             "fwd: last dimension of q, k, v must be contiguous")
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** -0.5
    packed = q.dim() == 3
    if packed:
        _require(cu_seqlens_q is not None and cu_seqlens_k is not None and max_seqlen_q and max_seqlen_k,
                 "fwd: packed inputs need cu_seqlens_q/k and max_seqlen_q/k")
        nb = cu_seqlens_q.numel() - 1
        _require(cu_seqlens_k.numel() - 1 == nb, "fwd: cu_seqlens_q and cu_seqlens_k disagree")
        # equal totals <=> every sequence has the maximum length: no device read needed
        _require(q.shape[0] == nb * max_seqlen_q and k.shape[0] == nb * max_seqlen_k,
                 "fwd: ragged packed batches are not implemented; all sequences must have max_seqlen")
        T, H, d = q.shape
        q4 = q.as_strided((nb, max_seqlen_q, H, d), (max_seqlen_q * q.stride(0), q.stride(0), q.stride(1), 1))
        k4 = k.as_strided((nb, max_seqlen_k, k.shape[1], d),
                          (max_seqlen_k * k.stride(0), k.stride(0), k.stride(1), 1))
        v4 = v.as_strided((nb, max_seqlen_k, v.shape[1], d),
                          (max_seqlen_k * v.stride(0), v.stride(0), v.stride(1), 1))
    else:
        _require(q.dim() == 4, "fwd: q must be [B,S,H,d] or [T,H,d]")
        _require(cu_seqlens_q is None and cu_seqlens_k is None,
                 "fwd: cu_seqlens with batched 4-D inputs are not supported")
        q4, k4, v4 = q, k, v
    B, Sq, H, d = q4.shape
    if out is None:
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    else:
        _require(out.shape == q.shape and out.dtype == q.dtype and out.stride(-1) == 1,
                 "fwd: out must match q")
    o4 = out if not packed else out.as_strided(
        (B, Sq, H, d), (Sq * out.stride(0), out.stride(0), out.stride(1), 1))
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    if B and Sq:
        _C.attention(q4, k4, v4, o4, lse, softmax_scale)
    if packed:  # FA3's packed layout of the log-sum-exp is [H, T]
        lse = lse.permute(1, 0, 2).reshape(H, B * Sq)
    empty = torch.empty(0, dtype=torch.float32, device=q.device)
    return out, lse, empty, empty


def _merge_state_v2(v_a, s_a, v_b, s_b, v_merged, s_merged):
    """sgl-kernel/csrc/attention/merge_attn_states.cu:180-204: contiguous [tokens, heads, D] /
    [tokens, heads] operands on one device."""
    for t in (v_a, s_a, v_b, s_b, v_merged, s_merged):
        _require(t.is_cuda and t.is_contiguous(), "merge_state_v2: contiguous device tensors expected")
    _require(v_a.dim() == 3 and s_a.dim() == 2 and v_b.shape == v_a.shape and s_b.shape == s_a.shape
             and v_merged.shape == v_a.shape and s_merged.shape == s_a.shape
             and s_a.shape == v_a.shape[:2], "merge_state_v2: shape mismatch")
    _require(v_merged.dtype == torch.bfloat16 and v_a.dtype == torch.bfloat16
             and v_b.dtype == torch.bfloat16, "merge_state_v2: bfloat16 states only")
    _require(s_a.dtype == torch.float32 and s_b.dtype == torch.float32
             and s_merged.dtype == torch.float32, "merge_state_v2: float32 log-sum-exps only")
    _C.merge_attn_states(v_merged.unsqueeze(0), s_merged.unsqueeze(0), v_a.unsqueeze(0),
                         s_a.unsqueeze(0), v_b.unsqueeze(0), s_b.unsqueeze(0), lse_layout="bsh")


_IMPLS = {
    "merge_state_v2": _merge_state_v2,
    "fp8_scaled_mm": _fp8_scaled_mm,
    "fp8_blockwise_scaled_mm": _fp8_blockwise_scaled_mm,
    "sgl_per_token_quant_fp8": _sgl_per_token_quant_fp8,
    "sgl_per_token_group_quant_8bit": _sgl_per_token_group_quant_8bit,
    "sgl_per_tensor_quant_fp8": _sgl_per_tensor_quant_fp8,
    "rmsnorm": _rmsnorm,
    "fused_add_rmsnorm": _fused_add_rmsnorm,
    "rotary_embedding": _rotary_embedding,
    "apply_rope_pos_ids_cos_sin_cache": _apply_rope_pos_ids_cos_sin_cache,
    "fwd": _fwd,
}
for _name, _fn in _IMPLS.items():
    _lib.impl(_name, _fn, "CUDA")


# --------------------------------------------------------------- fake (meta) implementations --
@torch.library.register_fake("sgl_kernel::fp8_scaled_mm")
def _fake_fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias):
    return mat_a.new_empty((mat_a.shape[0], mat_b.shape[1]), dtype=out_dtype)


@torch.library.register_fake("sgl_kernel::fp8_blockwise_scaled_mm")
def _fake_fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype):
    return mat_a.new_empty((mat_a.shape[0], mat_b.shape[1]), dtype=out_dtype)


def _fake_none(*args, **kwargs):
    return None


for _name in ("sgl_per_token_quant_fp8", "sgl_per_token_group_quant_8bit", "sgl_per_tensor_quant_fp8",
              "rmsnorm", "fused_add_rmsnorm", "rotary_embedding", "apply_rope_pos_ids_cos_sin_cache",
              "merge_state_v2"):
    torch.library.register_fake("sgl_kernel::" + _name)(_fake_none)


@torch.library.register_fake("sgl_kernel::fwd")
def _fake_fwd(q, k, v, k_new, v_new, q_v, out, cu_seqlens_q, cu_seqlens_k, *rest, **kw):
    o = torch.empty_like(q) if out is None else out
    if q.dim() == 4:
        lse = q.new_empty((q.shape[0], q.shape[2], q.shape[1]), dtype=torch.float32)
    else:
        lse = q.new_empty((q.shape[1], q.shape[0]), dtype=torch.float32)
    e = q.new_empty((0,), dtype=torch.float32)
    return o, lse, e, e
