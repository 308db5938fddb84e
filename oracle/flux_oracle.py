"""CPU restatement of the reference's Flux DiT forward, scheduler and denoise loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-on-CPU eager ops in bf16, in
the order the reference executes them, consuming a state dict with the reference's parameter
names:

  FluxTransformer2DModel.forward   python/sglang/multimodal_gen/runtime/models/dits/flux.py:838-916
  FluxTransformerBlock.forward     flux.py:620-692
  FluxSingleTransformerBlock.forward  flux.py:499-559
  FluxAttention.forward            flux.py:343-421
  FluxPosEmbed / NDRotaryEmbedding flux.py:695-716, layers/rotary_embedding/mrope.py:96-125,259-294
  diffusers AdaLayerNormZero / ZeroSingle / Continuous, FeedForward, TimestepEmbedding,
  PixArtAlphaTextProjection, Combined*Embeddings (diffusers==0.36.0; not in the reference
  tree: restated from the call sites; parity unpinned by any reference fixture)
  FlowMatchEulerDiscreteScheduler  models/schedulers/scheduling_flow_match_euler_discrete.py:267-536
  denoise loop / CFG               pipelines_core/stages/denoising.py:966-1127,1350-1507
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


class FluxOracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], *, num_layers: int,
                 num_single_layers: int, num_heads: int, head_dim: int, guidance_embeds: bool,
                 fp8: bool = False, axes_dims_rope=(16, 56, 56), rope_bf16_cos_sin: bool = False,
                 fp8_block: bool = False):
        """state_dict holds bf16 weights (reference names).  fp8=True quantises the linears the
        reference builds with quant_config (attention projections, single-block proj_mlp /
        proj_out) per output channel at load time (layers/quantization/fp8.py:357-364).
        fp8_block=True instead takes them through the offline 128x128 block quantiser
        (tools/convert_hf_to_fp8.py:48-79) and the block linear (fp8_utils.py:518-548), the
        format of an fp8-serialised checkpoint."""
        self.sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        self.L, self.LS, self.H, self.dh = num_layers, num_single_layers, num_heads, head_dim
        self.D = num_heads * head_dim
        self.guidance_embeds = guidance_embeds
        self.fp8 = fp8
        self.axes = tuple(axes_dims_rope)
        self.rope_bf16 = rope_bf16_cos_sin
        self.q: Dict[str, tuple] = {}
        self.fp8_block = fp8_block
        if fp8 or fp8_block:
            for k, w in self.sd.items():
                if k.endswith(".weight") and self._is_fp8_layer(k[:-7]):
                    self.q[k[:-7]] = (ops.quantize_weight_block(w) if fp8_block
                                      else ops.quantize_weight_per_channel(w))

    @staticmethod
    def _is_fp8_layer(name: str) -> bool:
        leaf = name.split(".")[-1]
        if ".attn." in name and (leaf in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj",
                                          "add_v_proj", "to_add_out") or name.endswith("to_out.0")):
            return True
        return name.startswith("single_transformer_blocks.") and leaf in ("proj_mlp", "proj_out") \
            and ".attn." not in name

    # ---- building blocks -------------------------------------------------------------
    def lin(self, name: str, x, act=None):
        b = self.sd.get(name + ".bias")
        if name in self.q:
            wq, ws = self.q[name]
            y = ops.fp8_block_linear(x, wq, ws, b) if self.fp8_block else ops.fp8_linear(x, wq, ws, b)
        else:
            y = F.linear(x, self.sd[name + ".weight"], b)
        if act == "gelu":
            y = F.gelu(y, approximate="tanh")
        return y

    def temb(self, timestep, guidance, pooled):
        dt = pooled.dtype
        t = ops.timestep_embedding(timestep, 256, True, 0.0).to(dt)
        t = self.lin("time_text_embed.timestep_embedder.linear_2",
                     F.silu(self.lin("time_text_embed.timestep_embedder.linear_1", t)))
        p = self.lin("time_text_embed.text_embedder.linear_2",
                     F.silu(self.lin("time_text_embed.text_embedder.linear_1", pooled)))
        if self.guidance_embeds and guidance is not None:
            g = ops.timestep_embedding(guidance, 256, True, 0.0).to(dt)
            g = self.lin("time_text_embed.guidance_embedder.linear_2",
                         F.silu(self.lin("time_text_embed.guidance_embedder.linear_1", g)))
            return (t + g) + p
        return t + p

    def rope_freqs(self, ids: torch.Tensor):
        pos = ids.float()
        cos, sin = [], []
        for i, dim in enumerate(self.axes):
            pi = pos[:, i].to(torch.float64)
            freqs = 1.0 / (10000 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
            f = torch.outer(pi, freqs)
            cos.append(f.cos().float())
            sin.append(f.sin().float())
        return torch.cat(cos, 1).float(), torch.cat(sin, 1).float()

    def freqs_cis(self, text_len: int, height: int, width: int):
        h, w = height // 16, width // 16
        img = torch.zeros(h, w, 3)
        img[..., 1] += torch.arange(h)[:, None]
        img[..., 2] += torch.arange(w)[None, :]
        tc, ts = self.rope_freqs(torch.zeros(text_len, 3))
        ic, is_ = self.rope_freqs(img.reshape(h * w, 3))
        return torch.cat([tc, ic], 0), torch.cat([ts, is_], 0)

    def attention(self, pre: str, x, enc, freqs, has_added: bool, pre_only: bool, eps: float):
        H = self.H
        q = self.lin(pre + ".to_q", x).unflatten(-1, (H, -1))
        k = self.lin(pre + ".to_k", x).unflatten(-1, (H, -1))
        v = self.lin(pre + ".to_v", x).unflatten(-1, (H, -1))
        q = ops.rmsnorm(q, self.sd[pre + ".norm_q.weight"], eps)
        k = ops.rmsnorm(k, self.sd[pre + ".norm_k.weight"], eps)
        if has_added:
            eq = self.lin(pre + ".add_q_proj", enc).unflatten(-1, (H, -1))
            ek = self.lin(pre + ".add_k_proj", enc).unflatten(-1, (H, -1))
            ev = self.lin(pre + ".add_v_proj", enc).unflatten(-1, (H, -1))
            eq = ops.rmsnorm(eq, self.sd[pre + ".norm_added_q.weight"], eps)
            ek = ops.rmsnorm(ek, self.sd[pre + ".norm_added_k.weight"], eps)
            q, k, v = torch.cat([eq, q], 1), torch.cat([ek, k], 1), torch.cat([ev, v], 1)
        cos, sin = freqs
        if self.rope_bf16:
            cos, sin = cos.to(q.dtype), sin.to(q.dtype)
        q, k = ops.rope_interleaved(q, cos, sin), ops.rope_interleaved(k, cos, sin)
        o = ops.attention(q, k, v).flatten(2, 3)
        if has_added:
            tl = enc.shape[1]
            # The reference hands `split_with_sizes` views to the two output projections
            # (models/dits/flux.py:402-413).  At batch 1, the batch its scheduler runs, those
            # views are contiguous and F.linear fuses the bias into the matmul epilogue (one
            # rounding), and the fp8 path always fuses it.  At batch > 1 ATen's linear() falls
            # back to matmul + add_ for a non-contiguous 3-D input (two roundings) in bf16 mode
            # and the fp8 path's `input.view(-1, K)` raises; that batch-size artefact is not part
            # of the algorithm, so the oracle uses the batch-1 arithmetic at every batch size.
            eo, o = o[:, :tl].contiguous(), o[:, tl:].contiguous()
            o = self.lin(pre + ".to_out.0", o)
            eo = self.lin(pre + ".to_add_out", eo)
            return o, eo
        if not pre_only:
            o = self.lin(pre + ".to_out.0", o)
        return o

    def double_block(self, i: int, hid, enc, temb, freqs):
        p = f"transformer_blocks.{i}"
        st = F.silu(temb)
        e = self.lin(p + ".norm1.linear", st)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = e.chunk(6, dim=1)
        ce = self.lin(p + ".norm1_context.linear", st)
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = ce.chunk(6, 1)
        nh = ops.layernorm_modulate(hid, shift_msa, scale_msa)
        ne = ops.layernorm_modulate(enc, c_shift_msa, c_scale_msa)
        a, ca = self.attention(p + ".attn", nh, ne, freqs, True, False, 1e-6)
        hid = hid + gate_msa.unsqueeze(1) * a
        nh = F.layer_norm(hid, (self.D,), eps=1e-6)
        nh = nh * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        ff = self.lin(p + ".ff.net.2", self.lin(p + ".ff.net.0.proj", nh, act="gelu"))
        hid = hid + gate_mlp.unsqueeze(1) * ff
        enc = enc + c_gate_msa.unsqueeze(1) * ca
        ne = F.layer_norm(enc, (self.D,), eps=1e-6)
        ne = ne * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        cff = self.lin(p + ".ff_context.net.2", self.lin(p + ".ff_context.net.0.proj", ne, act="gelu"))
        enc = enc + c_gate_mlp.unsqueeze(1) * cff
        return enc, hid

    def single_block(self, i: int, hid, enc, temb, freqs):
        p = f"single_transformer_blocks.{i}"
        tl = enc.shape[1]
        x = torch.cat([enc, hid], dim=1)
        residual = x
        shift, scale, gate = self.lin(p + ".norm.linear", F.silu(temb)).chunk(3, dim=1)
        n = ops.layernorm_modulate(x, shift, scale)
        mlp = F.gelu(self.lin(p + ".proj_mlp", n), approximate="tanh")
        a = self.attention(p + ".attn", n, None, freqs, False, True, 1e-6)
        x = torch.cat([a, mlp], dim=2)
        x = gate.unsqueeze(1) * self.lin(p + ".proj_out", x)
        x = residual + x
        return x[:, :tl], x[:, tl:]

    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep,
                guidance, freqs_cis, return_intermediates: bool = False):
        hid = self.lin("x_embedder", hidden_states)
        temb = self.temb(timestep, guidance, pooled_projections)
        enc = self.lin("context_embedder", encoder_hidden_states)
        inter = {"temb": temb, "hid0": hid, "enc0": enc}
        for i in range(self.L):
            enc, hid = self.double_block(i, hid, enc, temb, freqs_cis)
            inter[f"d{i}"] = (enc, hid)
        for i in range(self.LS):
            enc, hid = self.single_block(i, hid, enc, temb, freqs_cis)
            inter[f"s{i}"] = (enc, hid)
        scale, shift = self.lin("norm_out.linear", F.silu(temb).to(hid.dtype)).chunk(2, dim=1)
        hid = F.layer_norm(hid, (self.D,), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
        out = self.lin("proj_out", hid)
        return (out, inter) if return_intermediates else out


# ----------------------------------------------------------------------------- scheduler --
def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5,
                    max_shift=1.15):
    """runtime/pipelines/flux.py:22-32."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flux_schedule(num_steps: int, height: int, width: int):
    """sigmas = linspace(1, 1/N, N) (configs/pipeline_configs/base.py:741-747), mu from
    (H/8)(W/8) (runtime/pipelines/flux.py:35-51), exponential time shift
    (scheduling_flow_match_euler_discrete.py:628-634), append 0 (:404-406).
    Returns (sigmas fp32 [N+1], timesteps fp32 [N])."""
    sig = np.linspace(1.0, 1 / num_steps, num_steps)
    sig = np.array(list(sig)).astype(np.float32)
    mu = calculate_shift((height // 8) * (width // 8))
    sig = np.exp(mu) / (np.exp(mu) + (1 / sig - 1) ** 1.0)
    st = torch.from_numpy(sig).to(torch.float32)
    ts = st * 1000
    return torch.cat([st, torch.zeros(1)]), ts


def denoise(model: FluxOracle, latents, prompt_embeds, pooled, *, height, width, num_steps,
            embedded_cfg_scale=3.5, neg_embeds=None, neg_pooled=None, guidance_scale=1.0,
            run_steps: Optional[int] = None, cfg_normalization: float = 0.0,
            guidance_rescale: float = 0.0):
    """DenoisingStage.forward with serial CFG (and its two optional post-combine steps,
    pipelines_core/stages/denoising.py:1490-1506)."""
    sigmas, timesteps = flux_schedule(num_steps, height, width)
    B = latents.shape[0]
    guidance = None
    if model.guidance_embeds:
        guidance = torch.full((B,), embedded_cfg_scale, dtype=torch.float32).to(torch.bfloat16) * 1000.0
    freqs = model.freqs_cis(prompt_embeds.shape[1], height, width)
    do_cfg = guidance_scale > 1.0 and neg_embeds is not None
    n = num_steps if run_steps is None else min(run_steps, num_steps)
    for i in range(n):
        x = latents.to(torch.bfloat16)
        t = timesteps[i].repeat(B)
        cond = model.forward(x, prompt_embeds, pooled, t, guidance, freqs)
        if do_cfg:
            uncond = model.forward(x, neg_embeds, neg_pooled, t, guidance, freqs)
            noise = ops.cfg_combine(cond, uncond, guidance_scale)
            noise = ops.cfg_normalize(noise, cond, cfg_normalization)
            if guidance_rescale > 0.0:
                noise = ops.rescale_noise_cfg(noise, cond, guidance_rescale)
        else:
            noise = cond
        latents = ops.euler_step(latents, noise, sigmas[i].item(), sigmas[i + 1].item())
    return latents
