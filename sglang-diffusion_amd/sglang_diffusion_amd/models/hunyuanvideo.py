"""HunyuanVideo MM-DiT blocks on the MI355X kernels.

MMDoubleStreamBlock / MMSingleStreamBlock with the reference's module and parameter names
(python/sglang/multimodal_gen/runtime/models/dits/hunyuanvideo.py:49-410), so a checkpoint's
`double_blocks.N.*` / `single_blocks.N.*` tensors land unchanged.  The arithmetic is the
reference's eager ("native") path, every bf16 rounding point kept:

  LayerNormScaleShift                = FP32LayerNorm -> bf16, * (1 + scale) -> bf16, + shift -> bf16
                                       (layers/layernorm.py:450-524; the scale/shift step is
                                       jit_kernel/diffusion/triton/scale_shift.py:64-128)
  ScaleResidualLayerNormScaleShift   = residual + x * gate (two roundings), then the above
                                       (layers/layernorm.py:332-443)
  MulAdd(a, b, c)                    = c + a * b (two roundings)   (layers/elementwise.py:7-38)
  q/k RMSNorm, then interleaved RoPE on the video tokens only
                                       (hunyuanvideo.py:219-228,383-397; rotation formula
                                       jit_kernel/diffusion/triton/rotary.py:43-64)

which are the passes the Flux path already runs (mi355_layernorm_modulate,
mi355_gate_residual, mi355_qk_norm_rope) plus the one-pass
mi355_scale_residual_layernorm_modulate, behind layer classes of the reference's names
(layers/layernorm.py, layers/elementwise.py here), with two differences in layout: the joint sequence is [video | text] (Flux: [text | image]) and the text
tokens are not rotated -- the rotation table carries cos = 1, sin = 0 on their rows, which is
the identity bit for bit.  No fused multi-block schedule exists for this model yet; the blocks
run layer by layer.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _C
from ..layers.attention import UlyssesAttention
from ..layers.elementwise import MulAdd
from ..layers.layernorm import (LayerNormScaleShift, RMSNorm, ScaleResidualLayerNormScaleShift,
                                _silu)
from ..layers.linear import ColumnParallelLinear, ReplicatedLinear
from ..layers.quantization import QuantizationConfig


def _gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out = torch.empty_like(x2)
    _C.activation(out, x2, 1)
    return out.view(x.shape)


class ModulateProjection(nn.Module):
    """layers/visual_embedding.py:223-249: linear(silu(x)), never quantised."""

    def __init__(self, hidden_size: int, factor: int = 2, act_layer: str = "silu",
                 dtype: Optional[torch.dtype] = None, prefix: str = ""):
        super().__init__()
        if act_layer != "silu":
            raise NotImplementedError("ModulateProjection: silu only")
        self.factor, self.hidden_size = factor, hidden_size
        self.linear = ColumnParallelLinear(hidden_size, hidden_size * factor, bias=True,
                                           gather_output=True, params_dtype=dtype or torch.bfloat16)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y, _ = self.linear(_silu(x))
        return y


class MLP(nn.Module):
    """layers/mlp.py:25-65: fc_out(gelu_tanh(fc_in(x))); both linears take the quant config."""

    def __init__(self, input_dim: int, mlp_hidden_dim: int, output_dim: Optional[int] = None,
                 bias: bool = True, act_type: str = "gelu_pytorch_tanh",
                 dtype: Optional[torch.dtype] = None, prefix: str = "",
                 quant_config: Optional[QuantizationConfig] = None):
        super().__init__()
        if act_type != "gelu_pytorch_tanh":
            raise NotImplementedError("MLP: gelu_pytorch_tanh only")
        dt = dtype or torch.bfloat16
        self.fc_in = ColumnParallelLinear(input_dim, mlp_hidden_dim, bias=True, gather_output=False,
                                          params_dtype=dt, quant_config=quant_config)
        self.fc_out = ReplicatedLinear(mlp_hidden_dim, output_dim or input_dim, bias=True,
                                       params_dtype=dt, quant_config=quant_config)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h, _ = self.fc_in(x)
        y, _ = self.fc_out(_gelu_tanh(h))
        return y


def _joint_rope_table(cos: torch.Tensor, sin: torch.Tensor, n_text: int
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rotation table for the joint [video | text] sequence: the text rows rotate by zero."""
    cos = cos.to(torch.float32)
    sin = sin.to(torch.float32)
    one = torch.ones((n_text, cos.shape[1]), dtype=torch.float32, device=cos.device)
    return (torch.cat([cos, one], dim=0).contiguous(),
            torch.cat([sin, torch.zeros_like(one)], dim=0).contiguous())


class MMDoubleStreamBlock(nn.Module):
    """hunyuanvideo.py:49-270."""

    def __init__(self, hidden_size: int, num_attention_heads: int, mlp_ratio: float,
                 dtype: Optional[torch.dtype] = None, supported_attention_backends=None,
                 prefix: str = "", quant_config: Optional[QuantizationConfig] = None,
                 sp_group=None):
        super().__init__()
        dt = dtype or torch.bfloat16
        self.num_attention_heads = num_attention_heads
        head_dim = hidden_size // num_attention_heads
        mlp_hidden_dim = int(hidden_size * mlp_ratio)
        for s in ("img", "txt"):
            pre = f"{prefix}.{s}" if s == "img" else ""   # only the image stream is prefixed
            setattr(self, f"{s}_mod", ModulateProjection(hidden_size, factor=6, dtype=dt))
            setattr(self, f"{s}_attn_norm", LayerNormScaleShift(hidden_size, eps=1e-6, dtype=dt))
            setattr(self, f"{s}_attn_residual_mlp_norm",
                    ScaleResidualLayerNormScaleShift(hidden_size, eps=1e-6, dtype=dt))
            setattr(self, f"{s}_mlp_residual", MulAdd())
            setattr(self, f"{s}_attn_qkv", ReplicatedLinear(
                hidden_size, hidden_size * 3, bias=True, params_dtype=dt,
                prefix=f"{pre}_attn_qkv" if pre else "", quant_config=quant_config))
            setattr(self, f"{s}_attn_q_norm", RMSNorm(head_dim, eps=1e-6, dtype=dt))
            setattr(self, f"{s}_attn_k_norm", RMSNorm(head_dim, eps=1e-6, dtype=dt))
            setattr(self, f"{s}_attn_proj", ReplicatedLinear(
                hidden_size, hidden_size, bias=True, params_dtype=dt,
                prefix=f"{pre}_attn_proj" if pre else "", quant_config=quant_config))
            setattr(self, f"{s}_mlp", MLP(hidden_size, mlp_hidden_dim, bias=True, dtype=dt,
                                          quant_config=quant_config))
        self.attn = UlyssesAttention(num_heads=num_attention_heads, head_size=head_dim, causal=False,
                                     supported_attention_backends=supported_attention_backends,
                                     prefix=f"{prefix}.attn", group=sp_group)

    def _qkv(self, s: str, x_mod: torch.Tensor):
        qkv, _ = getattr(self, f"{s}_attn_qkv")(x_mod)
        B, T = qkv.shape[:2]
        return qkv.view(B, T, 3, self.num_attention_heads, -1)

    def forward(self, img: torch.Tensor, txt: torch.Tensor, vec: torch.Tensor,
                freqs_cis: Tuple[torch.Tensor, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        (i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2) = torch.chunk(self.img_mod(vec), 6, dim=-1)
        (t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2) = torch.chunk(self.txt_mod(vec), 6, dim=-1)
        Ti, Tt = img.shape[1], txt.shape[1]

        # q, k, v of both streams in one joint [video | text] buffer; norm + RoPE in one pass
        qkv = torch.cat([self._qkv("img", self.img_attn_norm(img, i_sh1, i_sc1)),
                         self._qkv("txt", self.txt_attn_norm(txt, t_sh1, t_sc1))], dim=1)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        cos, sin = _joint_rope_table(freqs_cis[0], freqs_cis[1], Tt)
        _C.qk_norm_rope(q, k, self.img_attn_q_norm.weight.data, self.img_attn_k_norm.weight.data,
                        self.txt_attn_q_norm.weight.data, self.txt_attn_k_norm.weight.data, Ti,
                        1e-6, cos, sin, True)
        img_attn, txt_attn = self.attn(q[:, :Ti], k[:, :Ti], v[:, :Ti],
                                       q[:, Ti:], k[:, Ti:], v[:, Ti:])
        B = img.shape[0]
        out = []
        for s, x, attn, g1, sh2, sc2, g2 in (("img", img, img_attn, i_g1, i_sh2, i_sc2, i_g2),
                                             ("txt", txt, txt_attn, t_g1, t_sh2, t_sc2, t_g2)):
            a, _ = getattr(self, f"{s}_attn_proj")(attn.reshape(B, attn.shape[1], -1))
            # x + attn_out * gate, LayerNorm, scale / shift: one pass
            mlp_in, resid = getattr(self, f"{s}_attn_residual_mlp_norm")(x, a, g1, sh2, sc2)
            mlp_out = getattr(self, f"{s}_mlp")(mlp_in)
            out.append(getattr(self, f"{s}_mlp_residual")(mlp_out, g2, resid))
        return out[0], out[1]


class MMSingleStreamBlock(nn.Module):
    """hunyuanvideo.py:273-410; x is the joint [video | text] sequence."""

    def __init__(self, hidden_size: int, num_attention_heads: int, mlp_ratio: float = 4.0,
                 dtype: Optional[torch.dtype] = None, supported_attention_backends=None,
                 prefix: str = "", quant_config: Optional[QuantizationConfig] = None,
                 sp_group=None):
        super().__init__()
        dt = dtype or torch.bfloat16
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        head_dim = hidden_size // num_attention_heads
        self.mlp_hidden_dim = int(hidden_size * mlp_ratio)
        self.linear1 = ReplicatedLinear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim, bias=True,
                                        params_dtype=dt, prefix=f"{prefix}.linear1",
                                        quant_config=quant_config)
        self.linear2 = ReplicatedLinear(hidden_size + self.mlp_hidden_dim, hidden_size, bias=True,
                                        params_dtype=dt, prefix=f"{prefix}.linear2",
                                        quant_config=quant_config)
        self.q_norm = RMSNorm(head_dim, eps=1e-6, dtype=dt)
        self.k_norm = RMSNorm(head_dim, eps=1e-6, dtype=dt)
        self.input_norm_scale_shift = LayerNormScaleShift(hidden_size, eps=1e-6, dtype=dt)
        self.output_residual = MulAdd()
        self.modulation = ModulateProjection(hidden_size, factor=3, dtype=dt)
        self.attn = UlyssesAttention(num_heads=num_attention_heads, head_size=head_dim, causal=False,
                                     supported_attention_backends=supported_attention_backends,
                                     prefix=f"{prefix}.attn", group=sp_group)

    def forward(self, x: torch.Tensor, vec: torch.Tensor, txt_len: int,
                freqs_cis: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
        shift, scale, gate = self.modulation(vec).chunk(3, dim=-1)
        lin1, _ = self.linear1(self.input_norm_scale_shift(x, shift, scale))
        B, S = lin1.shape[:2]
        D = self.hidden_size
        qkv = lin1[:, :, :3 * D].unflatten(-1, (3, self.num_attention_heads, -1))
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        cos, sin = _joint_rope_table(freqs_cis[0], freqs_cis[1], txt_len)
        w_q, w_k = self.q_norm.weight.data, self.k_norm.weight.data
        _C.qk_norm_rope(q, k, w_q, w_k, w_q, w_k, 0, 1e-6, cos, sin, True)
        Ti = S - txt_len
        img_o, txt_o = self.attn(q[:, :Ti], k[:, :Ti], v[:, :Ti], q[:, Ti:], k[:, Ti:], v[:, Ti:])
        attn = torch.cat((img_o, txt_o), dim=1).reshape(B, S, -1)
        combined = torch.cat((attn, _gelu_tanh(lin1[:, :, 3 * D:])), dim=-1)
        out, _ = self.linear2(combined)
        return self.output_residual(out, gate, x)


def process_weights_after_loading(module: nn.Module) -> None:
    """Run every linear's quant-method post-processing (loader/fsdp_load.py:141-146)."""
    for m in module.modules():
        qm = getattr(m, "quant_method", None)
        if qm is not None:
            qm.process_weights_after_loading(m)
