"""Flux.1 MM-DiT on MI355X.

Module structure, parameter names and forward semantics mirror
python/sglang/multimodal_gen/runtime/models/dits/flux.py:182-916 (FluxAttention,
FluxTransformerBlock, FluxSingleTransformerBlock, FluxPosEmbed, FluxTransformer2DModel) and
configs/models/dits/flux.py:11-63 (architecture constants).  Two execution paths share the
same parameters and the same C-ABI kernels:

  * `forward`        - module by module through the plugin interfaces
                       (LinearMethodBase.apply, AttentionImpl.forward, RMSNorm, ...), the
                       structural drop-in;
  * `forward_fused`  - the performance path (models/flux_engine.py): fused QKV projections
                       written straight into the joint text|image buffers, LayerNorm +
                       modulation + fp8 quantisation in one pass, gate/residual folded into
                       GEMM epilogues, all AdaLN modulations computed by one skinny GEMM.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ..layers.attention import USPAttention
from ..layers.embeddings import (CombinedTimestepGuidanceTextProjEmbeddings,
                                 CombinedTimestepTextProjEmbeddings, FeedForward)
from ..layers.layernorm import (AdaLayerNormContinuous, AdaLayerNormZero, AdaLayerNormZeroSingle,
                                LayerNormNoAffine, RMSNorm, apply_qk_norm)
from ..layers.linear import ColumnParallelLinear
from ..layers.quantization import QuantizationConfig
from ..layers.rotary import NDRotaryEmbedding, apply_flashinfer_rope_qk_inplace
from .. import _C


@dataclass
class FluxArchConfig:
    """configs/models/dits/flux.py:11-63 (Flux.1-dev defaults)."""

    patch_size: int = 1
    in_channels: int = 64
    out_channels: Optional[int] = None
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)
    mlp_ratio: float = 4.0

    @property
    def hidden_size(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def flux_dev_config() -> FluxArchConfig:
    return FluxArchConfig()


def flux_schnell_config() -> FluxArchConfig:
    return FluxArchConfig(guidance_embeds=False)


def _gelu(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    out = torch.empty_like(x2)
    _C.activation(out, x2, 1)
    return out.view(x.shape)


def _gate_residual(residual: torch.Tensor, gate: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """residual + gate.unsqueeze(1) * y on [B,T,D] (flux.py:651-652,664-666,547-550)."""
    B, T, D = y.shape
    y2 = y.reshape(B * T, D)
    r2 = residual.reshape(B * T, D)
    if not y2.is_contiguous():
        y2 = y2.contiguous()
    if not r2.is_contiguous():
        r2 = r2.contiguous()
    out = torch.empty_like(r2)
    _C.gate_residual(out, r2, gate, y2, B, T)
    return out.view(B, T, D)


class FluxAttention(nn.Module):
    """flux.py:208-421 (non-Nunchaku structure: separate q/k/v projections)."""

    def __init__(self, query_dim: int, num_heads: int = 8, dim_head: int = 64, bias: bool = False,
                 added_kv_proj_dim: Optional[int] = None, added_proj_bias: bool = True,
                 out_bias: bool = True, eps: float = 1e-5, out_dim: Optional[int] = None,
                 pre_only: bool = False, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", dtype=torch.bfloat16):
        super().__init__()
        self.head_dim = dim_head
        self.inner_dim = out_dim if out_dim is not None else dim_head * num_heads
        self.query_dim = query_dim
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.pre_only = pre_only
        self.heads = out_dim // dim_head if out_dim is not None else num_heads
        self.added_kv_proj_dim = added_kv_proj_dim

        def lin(i, o, b, name):
            # Only the two output projections carry their module path (flux.py:286,332); the
            # q/k/v projections are built with an empty prefix (flux.py:255-276,303-325).  That
            # matters for fp8 checkpoints: the ignore list of a converted checkpoint names the
            # top-level "proj_out", which is matched as a SUBSTRING of a layer's prefix
            # (srt/layers/quantization/utils.py:80), and an empty prefix never matches.
            named = name in ("to_out.0", "to_add_out") and prefix
            return ColumnParallelLinear(i, o, bias=b, gather_output=True, params_dtype=dtype,
                                        quant_config=quant_config,
                                        prefix=f"{prefix}.{name}" if named else "")

        self.norm_q = RMSNorm(dim_head, eps=eps)
        self.norm_k = RMSNorm(dim_head, eps=eps)
        self.to_q = lin(query_dim, self.inner_dim, bias, "to_q")
        self.to_k = lin(query_dim, self.inner_dim, bias, "to_k")
        self.to_v = lin(query_dim, self.inner_dim, bias, "to_v")
        if not pre_only:
            self.to_out = nn.ModuleList([lin(self.inner_dim, self.out_dim, out_bias, "to_out.0")])
        if added_kv_proj_dim is not None:
            self.norm_added_q = RMSNorm(dim_head, eps=eps)
            self.norm_added_k = RMSNorm(dim_head, eps=eps)
            self.add_q_proj = lin(added_kv_proj_dim, self.inner_dim, added_proj_bias, "add_q_proj")
            self.add_k_proj = lin(added_kv_proj_dim, self.inner_dim, added_proj_bias, "add_k_proj")
            self.add_v_proj = lin(added_kv_proj_dim, self.inner_dim, added_proj_bias, "add_v_proj")
            self.to_add_out = lin(self.inner_dim, query_dim, out_bias, "to_add_out")
        self.attn = USPAttention(num_heads=num_heads, head_size=self.head_dim, dropout_rate=0,
                                 softmax_scale=None, causal=False)

    def forward(self, x: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                freqs_cis=None):
        query, _ = self.to_q(x)
        key, _ = self.to_k(x)
        value, _ = self.to_v(x)
        query = query.unflatten(-1, (self.heads, -1))
        key = key.unflatten(-1, (self.heads, -1))
        value = value.unflatten(-1, (self.heads, -1))
        query, key = apply_qk_norm(q=query, k=key, q_norm=self.norm_q, k_norm=self.norm_k,
                                   head_dim=self.head_dim, allow_inplace=True)
        if encoder_hidden_states is not None and self.added_kv_proj_dim is not None:
            eq, _ = self.add_q_proj(encoder_hidden_states)
            ek, _ = self.add_k_proj(encoder_hidden_states)
            ev, _ = self.add_v_proj(encoder_hidden_states)
            eq = eq.unflatten(-1, (self.heads, -1))
            ek = ek.unflatten(-1, (self.heads, -1))
            ev = ev.unflatten(-1, (self.heads, -1))
            eq, ek = apply_qk_norm(q=eq, k=ek, q_norm=self.norm_added_q, k_norm=self.norm_added_k,
                                   head_dim=self.head_dim, allow_inplace=True)
            query = torch.cat([eq, query], dim=1)
            key = torch.cat([ek, key], dim=1)
            value = torch.cat([ev, value], dim=1)
        if freqs_cis is not None:
            cos, sin = freqs_cis
            cos_sin_cache = torch.cat([cos.to(dtype=torch.float32).contiguous(),
                                       sin.to(dtype=torch.float32).contiguous()], dim=-1)
            query, key = apply_flashinfer_rope_qk_inplace(query, key, cos_sin_cache, is_neox=False)
        x = self.attn(query, key, value)
        x = x.flatten(2, 3).to(query.dtype)
        if encoder_hidden_states is not None:
            tl = encoder_hidden_states.shape[1]
            encoder_hidden_states, x = x.split_with_sizes([tl, x.shape[1] - tl], dim=1)
            if not self.pre_only:
                x, _ = self.to_out[0](x)
            encoder_hidden_states, _ = self.to_add_out(encoder_hidden_states)
            return x, encoder_hidden_states
        if not self.pre_only:
            x, _ = self.to_out[0](x)
        return x


class FluxSingleTransformerBlock(nn.Module):
    """flux.py:424-559."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int,
                 mlp_ratio: float = 4.0, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", dtype=torch.bfloat16):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZeroSingle(dim, dtype=dtype)
        # proj_mlp, proj_out and the attention of a single block are built WITHOUT a prefix
        # (flux.py:476-503), so an ignore-list entry "proj_out" (the model's final projection)
        # does not switch off quantisation of every single block's proj_out.
        self.proj_mlp = ColumnParallelLinear(dim, self.mlp_hidden_dim, bias=True,
                                             gather_output=True, params_dtype=dtype,
                                             quant_config=quant_config)
        self.proj_out = ColumnParallelLinear(dim + self.mlp_hidden_dim, dim, bias=True,
                                             gather_output=True, params_dtype=dtype,
                                             quant_config=quant_config)
        self.attn = FluxAttention(query_dim=dim, dim_head=attention_head_dim,
                                  num_heads=num_attention_heads, out_dim=dim, bias=True, eps=1e-6,
                                  pre_only=True, quant_config=quant_config, dtype=dtype)

    def forward(self, hidden_states, encoder_hidden_states, temb, freqs_cis=None,
                joint_attention_kwargs=None):
        text_seq_len = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        residual = hidden_states
        norm_hidden_states, gate = self.norm(hidden_states, emb=temb)
        proj_hidden_states, _ = self.proj_mlp(norm_hidden_states)
        mlp_hidden_states = _gelu(proj_hidden_states)
        attn_output = self.attn(x=norm_hidden_states, freqs_cis=freqs_cis)
        hidden_states = torch.cat([attn_output, mlp_hidden_states], dim=2)
        proj_out, _ = self.proj_out(hidden_states)
        hidden_states = _gate_residual(residual, gate, proj_out)
        return hidden_states[:, :text_seq_len], hidden_states[:, text_seq_len:]


class FluxTransformerBlock(nn.Module):
    """flux.py:562-692."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int,
                 eps: float = 1e-6, quant_config: Optional[QuantizationConfig] = None,
                 prefix: str = "", dtype=torch.bfloat16):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim, dtype=dtype)
        self.norm1_context = AdaLayerNormZero(dim, dtype=dtype)
        self.attn = FluxAttention(query_dim=dim, added_kv_proj_dim=dim,
                                  dim_head=attention_head_dim, num_heads=num_attention_heads,
                                  out_dim=dim, bias=True, eps=eps, quant_config=quant_config,
                                  prefix=f"{prefix}.attn" if prefix else "attn", dtype=dtype)
        self.norm2 = LayerNormNoAffine(dim, eps=1e-6)
        self.norm2_context = LayerNormNoAffine(dim, eps=1e-6)
        self.ff = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", dtype=dtype)
        self.ff_context = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate",
                                      dtype=dtype)

    def forward(self, hidden_states, encoder_hidden_states, temb, freqs_cis=None,
                joint_attention_kwargs=None):
        from ..layers.layernorm import layernorm_modulate

        norm_hidden_states, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(
            hidden_states, emb=temb)
        norm_encoder_hidden_states, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = (
            self.norm1_context(encoder_hidden_states, emb=temb))
        attn_output, context_attn_output = self.attn(
            x=norm_hidden_states, encoder_hidden_states=norm_encoder_hidden_states,
            freqs_cis=freqs_cis)
        hidden_states = _gate_residual(hidden_states, gate_msa, attn_output)
        norm_hidden_states = layernorm_modulate(hidden_states, shift_mlp, scale_mlp, 1e-6)
        ff_output = self.ff(norm_hidden_states)
        hidden_states = _gate_residual(hidden_states, gate_mlp, ff_output)
        encoder_hidden_states = _gate_residual(encoder_hidden_states, c_gate_msa,
                                               context_attn_output)
        norm_encoder_hidden_states = layernorm_modulate(encoder_hidden_states, c_shift_mlp,
                                                        c_scale_mlp, 1e-6)
        context_ff_output = self.ff_context(norm_encoder_hidden_states)
        encoder_hidden_states = _gate_residual(encoder_hidden_states, c_gate_mlp,
                                               context_ff_output)
        return encoder_hidden_states, hidden_states


class FluxPosEmbed(nn.Module):
    """flux.py:695-716: float64 frequency table, cos/sin returned as fp32."""

    def __init__(self, theta: int, axes_dim: List[int]):
        super().__init__()
        self.rope = NDRotaryEmbedding(rope_dim_list=list(axes_dim), rope_theta=theta,
                                      use_real=False, repeat_interleave_real=False,
                                      dtype=torch.float64)

    def forward(self, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        pos = ids.float()
        cos, sin = self.rope.forward_uncached(pos=pos)
        return cos.contiguous().float(), sin.contiguous().float()


class FluxTransformer2DModel(nn.Module):
    """flux.py:719-916."""

    def __init__(self, config: Optional[FluxArchConfig] = None,
                 quant_config: Optional[QuantizationConfig] = None, dtype=torch.bfloat16):
        super().__init__()
        self.config = config or FluxArchConfig()
        c = self.config
        self.dtype = dtype
        self.quant_config = quant_config
        self.out_channels = c.out_channels or c.in_channels
        self.inner_dim = c.num_attention_heads * c.attention_head_dim
        self.rotary_emb = FluxPosEmbed(theta=10000, axes_dim=list(c.axes_dims_rope))
        cls = (CombinedTimestepGuidanceTextProjEmbeddings if c.guidance_embeds
               else CombinedTimestepTextProjEmbeddings)
        self.time_text_embed = cls(embedding_dim=self.inner_dim,
                                   pooled_projection_dim=c.pooled_projection_dim, dtype=dtype)
        # never quantised: built without quant_config (flux.py:788-796,826-831)
        self.context_embedder = ColumnParallelLinear(c.joint_attention_dim, self.inner_dim,
                                                     bias=True, gather_output=True,
                                                     params_dtype=dtype)
        self.x_embedder = ColumnParallelLinear(c.in_channels, self.inner_dim, bias=True,
                                               gather_output=True, params_dtype=dtype)
        self.transformer_blocks = nn.ModuleList([
            FluxTransformerBlock(dim=self.inner_dim, num_attention_heads=c.num_attention_heads,
                                 attention_head_dim=c.attention_head_dim,
                                 quant_config=quant_config, prefix=f"transformer_blocks.{i}",
                                 dtype=dtype) for i in range(c.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([
            FluxSingleTransformerBlock(dim=self.inner_dim,
                                       num_attention_heads=c.num_attention_heads,
                                       attention_head_dim=c.attention_head_dim,
                                       mlp_ratio=c.mlp_ratio, quant_config=quant_config,
                                       prefix=f"single_transformer_blocks.{i}", dtype=dtype)
            for i in range(c.num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(self.inner_dim, self.inner_dim, eps=1e-6,
                                               dtype=dtype)
        self.proj_out = ColumnParallelLinear(self.inner_dim,
                                             c.patch_size * c.patch_size * self.out_channels,
                                             bias=True, gather_output=True, params_dtype=dtype)
        self._engine = None

    # ------------------------------------------------------------------ weights --
    def process_weights_after_loading(self) -> None:
        """Run every linear's quant-method post-processing (the reference does this in its
        loader, loader/component_loaders/transformer_loader.py:164-168)."""
        for m in self.modules():
            qm = getattr(m, "quant_method", None)
            if qm is not None:
                qm.process_weights_after_loading(m)
        self._engine = None

    # ------------------------------------------------------------------ forward --
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None,
                guidance: torch.Tensor = None, freqs_cis=None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None) -> torch.Tensor:
        hidden_states, _ = self.x_embedder(hidden_states)
        if self.config.guidance_embeds and guidance is not None:
            temb = self.time_text_embed(timestep, guidance, pooled_projections)
        else:
            temb = self.time_text_embed(timestep, pooled_projections)
        encoder_hidden_states, _ = self.context_embedder(encoder_hidden_states)
        for block in self.transformer_blocks:
            encoder_hidden_states, hidden_states = block(
                hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                temb=temb, freqs_cis=freqs_cis)
        for block in self.single_transformer_blocks:
            encoder_hidden_states, hidden_states = block(
                hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                temb=temb, freqs_cis=freqs_cis)
        hidden_states = self.norm_out(hidden_states, temb)
        output, _ = self.proj_out(hidden_states)
        return output

    def forward_fused(self, hidden_states, encoder_hidden_states=None, pooled_projections=None,
                      timestep=None, guidance=None, freqs_cis=None,
                      joint_attention_kwargs=None) -> torch.Tensor:
        """Same contract as `forward`, executed by the fused engine."""
        from .flux_engine import FluxEngine

        if self._engine is None:
            self._engine = FluxEngine(self)
        return self._engine.step(hidden_states, encoder_hidden_states, pooled_projections,
                                 timestep, guidance, freqs_cis)


EntryClass = FluxTransformer2DModel
