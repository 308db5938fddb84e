"""CPU oracle of the HunyuanVideo MM-DiT blocks -- TEST INFRASTRUCTURE ONLY.

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
                        and not k.endswith("modulation.linear.weight"):
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
