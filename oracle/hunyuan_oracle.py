"""CPU oracle of the HunyuanVideo MM-DiT (blocks and whole model) -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product never does.  It restates, with plain torch CPU
ops, what MMDoubleStreamBlock.forward and MMSingleStreamBlock.forward compute on the
reference's eager ("native") path (python/sglang/multimodal_gen/runtime/models/dits/
hunyuanvideo.py:177-270,353-410), one function per reference construct, each citing it.

PARITY UNPINNED for this model: the reference ships no fixtures for these blocks, its scale /
shift and rotation steps are Triton kernels that cannot run here (no GPU in this container,
CPU tensors are rejected by their `assert x.is_cuda`), so the per-operation bf16 rounding of
`x * (1 + scale) + shift` is restated from Triton's element-type semantics (bf16 operands give
bf16 results, operation by operation) rather than checked against an output.  The operators
shared with the Flux path (linears, fp8 quantisation, RMSNorm, the rotation formula,
attention) are pinned there.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import ops


class HunyuanBlocksOracle:
    # built without quant_config in the reference (hunyuanvideo.py:470-510,793-841,960-991)
    NEVER_QUANTISED = ("txt_in.", "time_in.", "vector_in.", "guidance_in.", "final_layer.", "img_in.")

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, num_heads: int, fp8: bool = False):
        """state_dict: bf16 weights under the reference's names (`<block>.img_attn_qkv.weight`,
        `<block>.img_mlp.fc_in.weight`, `<block>.linear1.weight` ...).  fp8=True quantises the
        linears the reference builds with quant_config -- attention qkv / proj, both MLP linears,
        linear1 / linear2; never the modulation projections (hunyuanvideo.py:73-165,298-344) --
        per output channel at load time (layers/quantization/fp8.py:357-364)."""
        self.sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        self.H = num_heads
        self.fp8 = fp8
        self.q: Dict[str, tuple] = {}
        if fp8:
            for k, w in self.sd.items():
                if k.endswith(".weight") and w.dim() == 2 and ".linear." not in "." + k \
                        and not k.endswith("_mod.linear.weight") \
                        and not k.endswith("modulation.linear.weight") \
                        and not k.startswith(self.NEVER_QUANTISED):
                    self.q[k[:-7]] = ops.quantize_weight_per_channel(w)

    def lin(self, name: str, x):
        b = self.sd.get(name + ".bias")
        if name in self.q:
            wq, ws = self.q[name]
            return ops.fp8_linear(x, wq, ws, b)
        return F.linear(x, self.sd[name + ".weight"], b)

    # ---- reference constructs ---------------------------------------------------------
    def modulate_projection(self, name: str, vec):
        """ModulateProjection.forward (layers/visual_embedding.py:246-249)."""
        return F.linear(F.silu(vec), self.sd[name + ".linear.weight"], self.sd[name + ".linear.bias"])

    @staticmethod
    def norm_scale_shift(x, shift, scale):
        """LayerNormScaleShift.forward_native (layers/layernorm.py:517-521): FP32LayerNorm
        (:312-325) -> bf16, then fuse_scale_shift_kernel = x * (1 + scale) + shift on bf16
        operands (jit_kernel/diffusion/triton/scale_shift.py:127)."""
        y = F.layer_norm(x.float(), (x.shape[-1],), eps=1e-6).to(x.dtype)
        return (y * (1 + scale[:, None]) + shift[:, None]).to(x.dtype)

    @staticmethod
    def mul_add(a, b, c):
        """MulAdd.forward_native with k = 0 (layers/elementwise.py:30-32): c + a * (0 + b)."""
        return c + a * b[:, None]

    def qk_norm(self, name: str, x):
        """RMSNorm(head_dim, eps=1e-6) over the last dimension, result in x's dtype
        (hunyuanvideo.py:219-220,239-240,381-382)."""
        return ops.rmsnorm(x, self.sd[name + ".weight"], 1e-6)

    def attention(self, img_q, img_k, img_v, txt_q, txt_k, txt_v):
        """UlyssesAttention.forward on one rank (layers/attention/layer.py:112-152): the text
        tokens go BEHIND the video tokens, one dense attention, split again."""
        q = torch.cat([img_q, txt_q], dim=1)
        k = torch.cat([img_k, txt_k], dim=1)
        v = torch.cat([img_v, txt_v], dim=1)
        o = ops.attention(q, k, v)
        n = img_q.shape[1]
        return o[:, :n], o[:, n:]

    # ---- blocks -----------------------------------------------------------------------------
    def double_block(self, pre: str, img, txt, vec, cos, sin):
        """MMDoubleStreamBlock.forward (hunyuanvideo.py:177-270)."""
        H = self.H
        i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = self.modulate_projection(pre + ".img_mod", vec).chunk(6, -1)
        t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = self.modulate_projection(pre + ".txt_mod", vec).chunk(6, -1)
        B, Ti = img.shape[:2]
        Tt = txt.shape[1]
        iqkv = self.lin(pre + ".img_attn_qkv", self.norm_scale_shift(img, i_sh1, i_sc1)).view(B, Ti, 3, H, -1)
        iq = self.qk_norm(pre + ".img_attn_q_norm", iqkv[:, :, 0].contiguous())
        ik = self.qk_norm(pre + ".img_attn_k_norm", iqkv[:, :, 1].contiguous())
        iq, ik = ops.rope_interleaved(iq, cos, sin), ops.rope_interleaved(ik, cos, sin)
        tqkv = self.lin(pre + ".txt_attn_qkv", self.norm_scale_shift(txt, t_sh1, t_sc1)).view(B, Tt, 3, H, -1)
        tq = self.qk_norm(pre + ".txt_attn_q_norm", tqkv[:, :, 0].contiguous())
        tk = self.qk_norm(pre + ".txt_attn_k_norm", tqkv[:, :, 1].contiguous())
        ia, ta = self.attention(iq, ik, iqkv[:, :, 2], tq, tk, tqkv[:, :, 2])
        out = []
        for s, x, a, g1, sh2, sc2, g2 in (("img", img, ia, i_g1, i_sh2, i_sc2, i_g2),
                                          ("txt", txt, ta, t_g1, t_sh2, t_sc2, t_g2)):
            proj = self.lin(f"{pre}.{s}_attn_proj", a.reshape(B, a.shape[1], -1))
            resid = x + proj * g1[:, None]            # ScaleResidual...: layernorm.py:431-433
            m_in = self.norm_scale_shift(resid, sh2, sc2)
            h = F.gelu(self.lin(f"{pre}.{s}_mlp.fc_in", m_in), approximate="tanh")
            out.append(self.mul_add(self.lin(f"{pre}.{s}_mlp.fc_out", h), g2, resid))
        return out[0], out[1]

    def single_block(self, pre: str, x, vec, txt_len: int, cos, sin):
        """MMSingleStreamBlock.forward (hunyuanvideo.py:353-410)."""
        H = self.H
        shift, scale, gate = self.modulate_projection(pre + ".modulation", vec).chunk(3, -1)
        lin1 = self.lin(pre + ".linear1", self.norm_scale_shift(x, shift, scale))
        D = x.shape[-1]
        qkv, mlp = lin1[..., :3 * D], lin1[..., 3 * D:]
        B, S = qkv.shape[:2]
        qkv = qkv.reshape(B, S, 3, H, -1)
        q = self.qk_norm(pre + ".q_norm", qkv[:, :, 0].contiguous())
        k = self.qk_norm(pre + ".k_norm", qkv[:, :, 1].contiguous())
        v = qkv[:, :, 2]
        n = S - txt_len
        iq, ik = ops.rope_interleaved(q[:, :n], cos, sin), ops.rope_interleaved(k[:, :n], cos, sin)
        ia, ta = self.attention(iq, ik, v[:, :n], q[:, n:], k[:, n:], v[:, n:])
        attn = torch.cat((ia, ta), dim=1).reshape(B, S, -1)
        combined = torch.cat((attn, F.gelu(mlp, approximate="tanh")), dim=-1)
        return self.mul_add(self.lin(pre + ".linear2", combined), gate, x)


class HunyuanVideoOracle(HunyuanBlocksOracle):
    """HunyuanVideoTransformer3DModel.forward (hunyuanvideo.py:562-677) on one rank, eager CPU."""

    def __init__(self, state_dict, *, num_heads: int, num_layers: int, num_single_layers: int,
                 num_refiner_layers: int = 2, patch_size=(1, 2, 2), out_channels: int = 16,
                 pooled_dim: int = 768, rope_axes_dim=(16, 56, 56), rope_theta: float = 256.0,
                 guidance_embeds: bool = False, fp8: bool = False):
        super().__init__(state_dict, num_heads=num_heads, fp8=fp8)
        self.L, self.LS, self.LR = num_layers, num_single_layers, num_refiner_layers
        self.patch, self.out_channels, self.pooled_dim = tuple(patch_size), out_channels, pooled_dim
        self.rope_axes_dim, self.rope_theta = tuple(rope_axes_dim), float(rope_theta)
        self.guidance_embeds = guidance_embeds

    # ---- embedders ------------------------------------------------------------------------
    def mlp_silu(self, name: str, x):
        """MLP(act_type="silu") (layers/mlp.py:61-65)."""
        return self.lin(name + ".fc_out", F.silu(self.lin(name + ".fc_in", x)))

    def timestep_embedder(self, name: str, t):
        """TimestepEmbedder.forward + timestep_embedding (layers/visual_embedding.py:176-220)."""
        import math

        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        return self.mlp_silu(name + ".mlp", emb.to(torch.bfloat16))

    def patch_embed(self, x):
        """PatchEmbed.forward (layers/visual_embedding.py:81-86): Conv3d, flatten, transpose."""
        y = F.conv3d(x, self.sd["img_in.proj.weight"], self.sd["img_in.proj.bias"],
                     stride=self.patch)
        return y.flatten(2).transpose(1, 2)

    def refiner_block(self, pre: str, x, c):
        """IndividualTokenRefinerBlock.forward (hunyuanvideo.py:934-957)."""
        gate_msa, gate_mlp = self.modulate_projection(pre + ".adaLN_modulation", c).chunk(2, -1)
        D = x.shape[-1]
        ln = lambda n, v: F.layer_norm(v, (D,), self.sd[f"{pre}.{n}.weight"],  # noqa: E731
                                       self.sd[f"{pre}.{n}.bias"], 1e-6)
        qkv = self.lin(pre + ".self_attn_qkv", ln("norm1", x))
        B, L = qkv.shape[:2]
        qkv = qkv.view(B, L, 3, self.H, -1)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]).reshape(B, L, -1)
        x = x + self.lin(pre + ".self_attn_proj", a) * gate_msa.unsqueeze(1)
        return x + self.mlp_silu(pre + ".mlp", ln("norm2", x)) * gate_mlp.unsqueeze(1)

    def token_refiner(self, x, t):
        """SingleTokenRefiner.forward (hunyuanvideo.py:843-858)."""
        c = self.timestep_embedder("txt_in.t_embedder", t) + \
            self.mlp_silu("txt_in.c_embedder", torch.mean(x, dim=1))
        x = self.lin("txt_in.input_embedder", x)
        for i in range(self.LR):
            x = self.refiner_block(f"txt_in.refiner_blocks.{i}", x, c)
        return x

    def final_layer(self, x, c):
        """FinalLayer.forward (hunyuanvideo.py:994-999): (scale, shift) order."""
        scale, shift = self.modulate_projection("final_layer.adaLN_modulation", c).chunk(2, -1)
        x = F.layer_norm(x, (x.shape[-1],), eps=1e-6) * (1.0 + scale.unsqueeze(1)) + shift.unsqueeze(1)
        return self.lin("final_layer.linear", x)

    def rope(self, t: int, h: int, w: int):
        """get_rotary_pos_embed on one rank (layers/rotary_embedding/factory.py:85-171,
        mrope.py:96-125,296-392): per axis 1/theta^(arange(0,d,2)/d) in fp32, positions on the
        (t, h, w) grid flattened with w fastest."""
        grid = torch.stack(torch.meshgrid(torch.arange(t, dtype=torch.float32),
                                          torch.arange(h, dtype=torch.float32),
                                          torch.arange(w, dtype=torch.float32), indexing="ij"),
                           dim=-1).reshape(-1, 3)
        cos, sin = [], []
        for i, d in enumerate(self.rope_axes_dim):
            freqs = 1.0 / (self.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32)[: d // 2] / d))
            f = torch.outer(grid[:, i], freqs)
            cos.append(f.cos().float())
            sin.append(f.sin().float())
        return torch.cat(cos, 1), torch.cat(sin, 1)

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance=None):
        x = hidden_states
        if guidance is None:
            guidance = torch.tensor([6016.0], dtype=x.dtype)
        txt = encoder_hidden_states[:, 1:]
        text_states_2 = encoder_hidden_states[:, 0, : self.pooled_dim]
        _, _, ot, oh, ow = x.shape
        pt, ph, pw = self.patch
        tt, th, tw = ot // pt, oh // ph, ow // pw
        cos, sin = self.rope(tt, th, tw)
        vec = self.timestep_embedder("time_in", timestep)
        vec = vec + self.mlp_silu("vector_in", text_states_2)
        if self.guidance_embeds:
            vec = vec + self.timestep_embedder("guidance_in", guidance)
        img = self.patch_embed(x)
        txt = self.token_refiner(txt, timestep)
        txt_len, img_len = txt.shape[1], img.shape[1]
        for i in range(self.L):
            img, txt = self.double_block(f"double_blocks.{i}", img, txt, vec, cos, sin)
        x = torch.cat((img, txt), 1)
        for i in range(self.LS):
            x = self.single_block(f"single_blocks.{i}", x, vec, txt_len, cos, sin)
        img = self.final_layer(x[:, :img_len], vec)
        c = self.out_channels
        img = img.reshape(img.shape[0], tt, th, tw, c, pt, ph, pw)
        img = torch.einsum("nthwcopq->nctohpwq", img)
        return img.reshape(img.shape[0], c, tt * pt, th * ph, tw * pw)
