"""CPU restatement of the reference's per-op arithmetic on the DiT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path imports this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
only as the checker or the reported baseline.

Every function restates one reference op with plain torch-on-CPU / numpy
arithmetic and cites the reference lines it follows.  Rounding points (where
eager PyTorch materialises a bf16 tensor) are kept exactly.

Pinning status (see oracle/make_golden.py, tests/test_oracle_pins.py):
  * attention, rmsnorm, fused_add_rmsnorm, rotary embedding, tanh-GELU and the
    block-fp8 weight GEMM are checked against the reference's OWN compiled CPU
    kernels (oracle/_ref, built from sgl-kernel/csrc/cpu), and the fixtures in
    tests/golden/ref_*.npz hold those reference outputs.
  * the quantisation formulas and the fp8 GEMMs follow the torch formulas in
    the reference's kernel tests (cited per function).
  * the diffusers-defined modules (AdaLayerNorm*, FeedForward, embeddings) and
    the scheduler are restated from the call sites; no reference test or
    fixture covers them: parity unpinned for those.
  * the offline checkpoint quantisers (quantize_weight_block, convert_channel_fp8: the
    reference's tools/convert_hf_to_fp8.py) are restated from the source text; the reference
    ships no fixture for them and its Python may not be executed here: parity unpinned.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

FP8_MAX = 448.0
FP8 = torch.float8_e4m3fn


# --------------------------------------------------------------------- quant --
def per_token_quant_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:75-103 (torch formula:
    sgl-kernel/tests/test_per_token_quant_fp8.py:14-27).
    scale = absmax/448 (true division), q = sat(x * (1/scale)), 1/scale := 0 if scale == 0."""
    xf = x.float().numpy()
    amax = np.abs(xf).max(axis=-1)
    scale = (amax / np.float32(FP8_MAX)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(scale == 0, np.float32(0), np.float32(1.0) / scale).astype(np.float32)
    y = np.clip(xf * inv[:, None], -FP8_MAX, FP8_MAX).astype(np.float32)
    q = torch.from_numpy(y).to(FP8)
    return q, torch.from_numpy(scale)


def per_token_group_quant_fp8(
    x: torch.Tensor, group_size: int, eps: float = 1e-10, fp8_min: float = -FP8_MAX,
    fp8_max: float = FP8_MAX,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:45-113 (torch formula:
    python/sglang/test/test_block_fp8.py:41-66).  absmax clamped below by eps,
    y_s = absmax/fp8_max, q = clamp(x / y_s).  Scales returned row-major [M, K/group]."""
    M, K = x.shape
    xf = x.float().numpy().reshape(M, K // group_size, group_size)
    amax = np.maximum(np.abs(xf).max(axis=-1), np.float32(eps)).astype(np.float32)
    y_s = (amax / np.float32(fp8_max)).astype(np.float32)
    y = np.clip(xf / y_s[..., None], fp8_min, fp8_max).astype(np.float32)
    q = torch.from_numpy(y.reshape(M, K)).to(FP8)
    return q, torch.from_numpy(y_s)


def per_tensor_quant_fp8(x: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-94: s = absmax/448 (dynamic),
    q = sat(x * (1/s))."""
    xf = x.float().numpy()
    if scale is None:
        s = np.float32(np.abs(xf).max()) / np.float32(FP8_MAX)
    else:
        s = np.float32(scale.item())
    inv = np.float32(1.0) / np.float32(s)
    y = np.clip(xf * inv, -FP8_MAX, FP8_MAX).astype(np.float32)
    return torch.from_numpy(y).to(FP8), torch.tensor([s], dtype=torch.float32)


def quantize_weight_per_channel(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Online weight quantisation of a bf16 checkpoint on platforms with the
    per-token x per-channel GEMM: per_token_group_quant_fp8(weight, K) i.e. one
    scale per output channel (python/sglang/multimodal_gen/runtime/layers/
    quantization/fp8.py:357-364).  Group quant semantics (eps 1e-10, division)."""
    q, s = per_token_group_quant_fp8(w, w.shape[-1])
    return q, s.reshape(-1)


def quantize_weight_block(w: torch.Tensor, block: int = 128):
    """Offline block quantiser, python/sglang/multimodal_gen/tools/convert_hf_to_fp8.py:48-79:
    zero-pad to whole [block x block] tiles, scale = tile absmax (fp32) / 448,
    q = (w / scale).clamp(+-448).to(fp8); the returned scale is what the checkpoint stores
    as `weight_scale_inv` [ceil(N/block), ceil(K/block)]."""
    N, K = w.shape
    nb, kb = -(-N // block), -(-K // block)
    pad = F.pad(w, (0, kb * block - K, 0, nb * block - N), mode="constant", value=0.0)
    t = pad.reshape(nb, block, kb, block)
    bmax = torch.max(torch.abs(t), dim=1, keepdim=True)[0]
    bmax = torch.max(bmax, dim=3, keepdim=True)[0]
    scale = bmax.to(torch.float32) / FP8_MAX
    q = (t / scale).clamp(min=-FP8_MAX, max=FP8_MAX).reshape(nb * block, kb * block).to(FP8)
    return q[:N, :K].clone(), scale.reshape(nb, kb)


def convert_channel_fp8(w: torch.Tensor):
    """Offline per-channel quantiser, convert_hf_to_fp8.py:82-87."""
    cmax = torch.max(w.abs(), dim=-1, keepdim=True)[0]
    scale = cmax.clamp(min=1e-12).to(torch.float32) / FP8_MAX
    q = (w / scale).clamp(min=-FP8_MAX, max=FP8_MAX).to(FP8)
    return q, scale.reshape(-1)


# ---------------------------------------------------------------------- GEMM --
def fp8_scaled_mm(a_q, w_q, sa, sb, bias=None, out_dtype=torch.bfloat16):
    """sgl-kernel/tests/test_fp8_gemm.py:6-14 (kernel: csrc/gemm/fp8_gemm_kernel.cu:1450-1532):
    o = (a.float() @ b.float()) * scale_a[:,None] * scale_b[None,:] + bias, cast to out_dtype.
    w_q is the weight [N,K] (mat_b = w_q.T, column-major)."""
    o = a_q.float() @ w_q.float().t()
    o = o * sa.view(-1, 1) * sb.view(1, -1)
    if bias is not None:
        o = o + bias.float()
    return o.to(out_dtype)


def fp8_blockwise_scaled_mm(a_q, w_q, sa, sb, out_dtype=torch.bfloat16, block=128):
    """sgl-kernel/tests/test_fp8_blockwise_gemm.py:19-59 / python/sglang/test/test_block_fp8.py:291-343:
    C[m,n] = sum_kb (A[m,kb] . W[n,kb]) * sa[m,kb] * sb[nb,kb].
    sa [M, K/128], sb [N/128, K/128] (the weight_scale_inv layout)."""
    M, K = a_q.shape
    N = w_q.shape[0]
    af, wf = a_q.float(), w_q.float()
    out = torch.zeros(M, N, dtype=torch.float32)
    for kb in range(K // block):
        part = af[:, kb * block:(kb + 1) * block] @ wf[:, kb * block:(kb + 1) * block].t()
        sbn = sb[:, kb].repeat_interleave(block)[:N]
        out += part * sa[:, kb:kb + 1] * sbn[None, :]
    return out.to(out_dtype)


def linear_bf16(x, w, bias=None, act: Optional[str] = None):
    """UnquantizedLinearMethod.apply -> F.linear (python/sglang/multimodal_gen/runtime/
    layers/linear.py:151-159); optional following tanh-GELU as in diffusers' GELU module
    (runtime/layers/mlp.py:94-113): gelu(bf16(linear))."""
    y = F.linear(x, w, bias)
    if act == "gelu_tanh":
        y = F.gelu(y, approximate="tanh")
    elif act == "silu":
        y = F.silu(y)
    return y


def fp8_linear(x, w_q, w_scale, bias=None):
    """apply_fp8_linear on the per-token x per-channel path
    (python/sglang/srt/layers/quantization/fp8_utils.py:1242-1276):
    sgl_per_token_quant_fp8(x) then fp8_scaled_mm(q, W, x_scale, w_scale, bf16, bias)."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    q, s = per_token_quant_fp8(x2)
    o = fp8_scaled_mm(q, w_q, s, w_scale, bias, x.dtype)
    return o.view(*shp[:-1], w_q.shape[0])


def fp8_linear_static(x, w_q, w_scale, input_scale, bias=None):
    """apply_fp8_linear with a calibrated per-tensor activation scale
    (python/sglang/srt/layers/quantization/fp8_utils.py:1234-1240): static_quant_fp8
    (fp8_kernel.py:607-700: q = clamp(x * (1/s))) with the scale repeated per token, then
    fp8_scaled_mm."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    q, s = per_tensor_quant_fp8(x2, input_scale)
    o = fp8_scaled_mm(q, w_q, s.expand(x2.shape[0]).contiguous(), w_scale, bias, x.dtype)
    return o.view(*shp[:-1], w_q.shape[0])


def fp8_block_linear(x, w_q, w_scale_inv, bias=None, block=128):
    """cutlass_w8a8_block_fp8_linear_with_fallback (fp8_utils.py:518-548):
    per_token_group_quant_fp8(x,128) -> fp8_blockwise_scaled_mm -> output += bias (bf16 add)."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    q, s = per_token_group_quant_fp8(x2, block)
    o = fp8_blockwise_scaled_mm(q, w_q, s, w_scale_inv, x.dtype, block)
    if bias is not None:
        o += bias
    return o.view(*shp[:-1], w_q.shape[0])


# ----------------------------------------------------------------- attention --
def attention(q, k, v, scale: Optional[float] = None, return_lse: bool = False):
    """SDPAImpl.forward (python/sglang/multimodal_gen/runtime/layers/attention/backends/
    sdpa.py:55-78) / attention_ref (sgl-kernel/tests/test_flash_attention.py:185-316):
    softmax(q k^T * scale) v in fp32 from bf16 inputs, [B,S,H,D] layout, output bf16."""
    B, Sq, H, D = q.shape
    Hk = k.shape[2]
    if scale is None:
        scale = D ** -0.5
    qf = q.float().transpose(1, 2)
    kf = k.float().transpose(1, 2)
    vf = v.float().transpose(1, 2)
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    s = (qf @ kf.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    o = (p @ vf).transpose(1, 2).to(q.dtype)
    if return_lse:
        return o, torch.logsumexp(s, dim=-1)
    return o


def merge_attn_states(v_a, s_a, v_b, s_b):
    """sgl-kernel/csrc/attention/merge_attn_states.cu:31-109 (the formula its test restates,
    sgl-kernel/tests/test_merge_state_v2.py:11-62): v [tokens, heads, D] in the output dtype,
    s [tokens, heads] fp32 log-sum-exps; +inf counts as -inf.  Returns (v_merged, s_merged)."""
    la = torch.where(torch.isinf(s_a), torch.full_like(s_a, float("-inf")), s_a.float())
    lb = torch.where(torch.isinf(s_b), torch.full_like(s_b, float("-inf")), s_b.float())
    m = torch.maximum(la, lb)
    wa, wb = torch.exp(la - m), torch.exp(lb - m)
    tot = wa + wb
    out = v_a.float() * (wa / tot).unsqueeze(-1) + v_b.float() * (wb / tot).unsqueeze(-1)
    return out.to(v_a.dtype), torch.log(tot) + m


# --------------------------------------------------------------- norm / rope --
def rmsnorm(x, w, eps: float):
    """RMSNorm.forward_native (python/sglang/multimodal_gen/runtime/layers/layernorm.py:106-143)."""
    dt = x.dtype
    xf = x.float()
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return (xf * w.float()).to(dt)


def fused_add_rmsnorm(x, residual, w, eps: float):
    """layernorm.py:114-143 with residual: x = x + residual (fp32); residual = x.to(dtype);
    returns (rmsnorm(x), residual)."""
    dt = x.dtype
    xf = x.float() + residual.float()
    new_res = xf.to(dt)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return (xf * w.float()).to(dt), new_res


def rope_interleaved(x, cos, sin):
    """Triton _rotary_embedding_kernel, interleaved pairs (python/sglang/jit_kernel/diffusion/
    triton/rotary.py:43-64): o1 = fma(-x2, sin, x1*cos); o2 = fma(x1, sin, x2*cos), fp32.
    x [B,S,H,D]; cos,sin [S,D/2] (fp32, or already rounded to bf16 on the ROCm fallback,
    layers/rotary_embedding/utils.py:80-83)."""
    xf = x.float()
    x1, x2 = xf[..., 0::2], xf[..., 1::2]
    c = cos.float()[None, :, None, :]
    s = sin.float()[None, :, None, :]
    o1 = x1 * c - x2 * s
    o2 = x2 * c + x1 * s
    return torch.stack([o1, o2], dim=-1).flatten(-2).to(x.dtype)


def rope_neox(x, cos, sin):
    """_apply_rotary_emb neox branch (layers/rotary_embedding/utils.py:27-37)."""
    xf = x.float()
    x1, x2 = torch.chunk(xf, 2, dim=-1)
    c = cos.float()[None, :, None, :]
    s = sin.float()[None, :, None, :]
    o1 = (x1 * c - x2 * s).to(x.dtype)
    o2 = (x2 * c + x1 * s).to(x.dtype)
    return torch.cat((o1, o2), dim=-1)


def qk_norm_rope(q, k, wq, wk, eps, cos, sin, interleaved=True):
    """apply_qk_norm (layernorm.py:531-568) then apply_flashinfer_rope_qk_inplace
    (rotary_embedding/utils.py:42-121), as FluxAttention.forward chains them
    (models/dits/flux.py:356-395)."""
    if wq is not None:
        q = rmsnorm(q, wq, eps)
        k = rmsnorm(k, wk, eps)
    if cos is not None:
        fn = rope_interleaved if interleaved else rope_neox
        q, k = fn(q, cos, sin), fn(k, cos, sin)
    return q, k


def rotary_embedding_cache(positions, q, k, head_size, cos_sin_cache, is_neox):
    """sgl_kernel rotary_embedding semantics (sgl-kernel/python/sgl_kernel/elementwise.py:270-366;
    CPU twin sgl-kernel/csrc/cpu/rope.cpp): cache [max_pos, rot_dim] = cos | sin."""
    T = positions.numel()
    rot = cos_sin_cache.shape[-1]
    cs = cos_sin_cache[positions].float()
    cos, sin = cs[:, : rot // 2], cs[:, rot // 2:]

    def one(x):
        xv = x.view(T, -1, head_size).float()
        xr = xv[..., :rot]
        if is_neox:
            x1, x2 = xr[..., : rot // 2], xr[..., rot // 2:]
        else:
            x1, x2 = xr[..., 0::2], xr[..., 1::2]
        c, s = cos[:, None, :], sin[:, None, :]
        o1 = x1 * c - x2 * s
        o2 = x2 * c + x1 * s
        if is_neox:
            out = torch.cat([o1, o2], dim=-1)
        else:
            out = torch.stack([o1, o2], dim=-1).flatten(-2)
        res = xv.clone()
        res[..., :rot] = out
        return res.to(x.dtype).view_as(x)

    return one(q), (one(k) if k is not None else None)


# ------------------------------------------------------------ AdaLN / glue ----
def layernorm_modulate(x, shift, scale, eps: float = 1e-6):
    """diffusers AdaLayerNormZero/ZeroSingle/Continuous body and norm2+modulate
    (models/dits/flux.py:628-634,653-661,511,912): x [B,T,D] bf16, shift/scale [B,D] or None.
    norm(x) * (1 + scale[:,None]) + shift[:,None], each op in bf16 as eager does."""
    y = F.layer_norm(x, (x.shape[-1],), eps=eps)
    if scale is not None:
        y = y * (1 + scale[:, None])
    if shift is not None:
        y = y + shift[:, None]
    return y


def gate_residual(residual, gate, y):
    """hidden + gate.unsqueeze(1) * y (flux.py:651-652,664-666,547-550), bf16 ops."""
    return residual + gate.unsqueeze(1) * y


def timestep_embedding(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, scale=1.0,
                       max_period=10000):
    """python/sglang/jit_kernel/tests/test_timestep_embedding.py:17-47 (diffusers get_timestep_embedding)."""
    t = t.to(torch.float32)
    half = dim // 2
    exponent = -torch.log(torch.tensor(max_period, dtype=torch.float32)) * torch.arange(
        0, half, dtype=torch.float32)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = t[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def euler_step(sample, model_output, sigma: float, sigma_next: float):
    """FlowMatchEulerDiscreteScheduler.step (models/schedulers/
    scheduling_flow_match_euler_discrete.py:495-531): sample.float() + dt * model_output, cast
    to model_output.dtype; dt is a 0-dim fp32 tensor so `dt * model_output` is a bf16 product."""
    s = torch.tensor(sigma, dtype=torch.float32)
    sn = torch.tensor(sigma_next, dtype=torch.float32)
    dt = sn - s
    prev = sample.to(torch.float32) + dt * model_output
    return prev.to(model_output.dtype)


def cfg_combine(cond, uncond, scale: float):
    """Serial CFG (pipelines_core/stages/denoising.py:1486-1488)."""
    return uncond + scale * (cond - uncond)


def cfg_normalize(noise_pred, noise_pred_cond, factor: float):
    """Norm cap after the combine (pipelines_core/stages/denoising.py:1490-1500): norms in fp32
    over the whole tensor; the 0-dim fp32 rescale factor multiplies the bf16 tensor.  On the
    device, where the reference runs this, eager casts a 0-dim TENSOR operand to the result
    dtype (bf16) before multiplying, whichever side it is on; CPU eager does that only when
    the 0-dim operand comes first (tools/probes/eager_zero_dim.py prints both) -- hence the
    operand order below, which is the only departure from the reference's text."""
    if not factor or float(factor) <= 0:
        return noise_pred
    ori_norm = torch.linalg.vector_norm(noise_pred_cond.float())
    new_norm = torch.linalg.vector_norm(noise_pred.float())
    max_norm = ori_norm * float(factor)
    if new_norm > max_norm:
        noise_pred = (max_norm / new_norm) * noise_pred
    return noise_pred


def rescale_noise_cfg(noise_cfg, noise_pred_text, guidance_rescale: float = 0.0):
    """DenoisingStage.rescale_noise_cfg (pipelines_core/stages/denoising.py:1164-1191)."""
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    noise_pred_rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * noise_pred_rescaled + (1 - guidance_rescale) * noise_cfg
