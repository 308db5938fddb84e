"""Fused execution of one Flux DiT forward on MI355X.

Computes exactly what FluxTransformer2DModel.forward computes (reference:
python/sglang/multimodal_gen/runtime/models/dits/flux.py:838-916 and the block forwards at
:499-559,620-692), with the eager sequence's bf16 rounding points preserved, but laid out for
the hardware:

  * HBM layout.  The text and image streams live in ONE joint activation buffer
    [B, S_txt + S_img, D] from the first block on, so the `torch.cat` / split pairs of the
    reference (flux.py:380-382,401-408,508,555-558) never move data: GEMMs read and write
    their stream's rows through a row map.
  * q, k, v come out of one fused projection GEMM per stream ([3D, D] weight, per-channel
    scales concatenated) straight into a joint [B, S, 3, H, d] buffer; RMSNorm + RoPE run
    in place on its k slice, and on q inside the attention kernel as it loads its Q rows;
    the attention kernel reads the strided views.
  * LayerNorm + AdaLN modulation + dynamic per-token fp8 quantisation is one pass over the
    hidden state; `gate * y + residual` is the epilogue of the GEMM that produces y.
  * All 19*2 + 38 + 1 AdaLN modulation linears are one weight-streaming GEMM per step.

Block-quantised checkpoints (128x128 weight blocks, activations quantised per token and 128
channels) run the same schedule: the norm kernel emits group scales, the GEMM rescales every
K tile, and the product is rounded to bf16 before the bias as the layer path does.

fp8 scales, fp8 activations and all intermediates are caller-invisible workspace tensors
allocated once per (batch, sequence) shape.
"""
from __future__ import annotations

import contextlib
import functools
import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import _C
from ..layers.quantization import FP8_DTYPE, Fp8LinearMethod, UnquantizedLinearMethod


@functools.lru_cache(maxsize=None)
def _cu_count(device_index: int) -> int:
    return torch.cuda.get_device_properties(device_index).multi_processor_count


@functools.lru_cache(maxsize=None)
def _big_tile_rule(M: int, N: int, fp8: bool, cus: int) -> bool:
    tb = ((M + 255) // 256) * ((N + 255) // 256)
    ts = ((M + 127) // 128) * ((N + 127) // 128)
    if tb < cus * 3 // 4:
        return False
    eff_b = tb / (-(-tb // cus) * cus)
    eff_s = ts / (-(-ts // (2 * cus)) * (2 * cus))
    return eff_b >= eff_s * (0.8 if fp8 else 0.7)


def _uses_big_tile(M: int, N: int, fp8: bool, device) -> bool:
    """The tile choice of csrc/gemm.hip (use_big_tile), for labelling launches in traces."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _big_tile_rule(M, N, bool(fp8), _cu_count(idx))


def _is_block(lin) -> bool:
    qm = lin.quant_method
    return isinstance(qm, Fp8LinearMethod) and qm.block_quant


def _is_fp8(lin) -> bool:
    qm = lin.quant_method
    if isinstance(qm, Fp8LinearMethod):
        if lin.weight.dtype != FP8_DTYPE:
            raise RuntimeError("call process_weights_after_loading() before the fused path")
        return True
    assert isinstance(qm, UnquantizedLinearMethod)
    return False


class _Packed:
    """Several linears concatenated along N; the modules' parameters become views.

    fp8 with per-channel scales: `s` is [N].  fp8 from a block-quantised checkpoint
    (`block`): `s` is the [N/128, K/128] block-scale matrix, which concatenates along N like
    the weight as long as every member's N is a multiple of 128."""

    def __init__(self, lins: List):
        if any(getattr(l, "input_scale", None) is not None for l in lins):
            raise NotImplementedError(
                "the fused schedule quantises activations per token (dynamic scheme); a "
                "checkpoint with static activation scales runs on the layer path")
        self.fp8 = _is_fp8(lins[0])
        self.block = _is_block(lins[0])
        assert all(_is_fp8(l) == self.fp8 and _is_block(l) == self.block for l in lins)
        self.w = torch.cat([l.weight.data for l in lins], dim=0).contiguous()
        self.b = (torch.cat([l.bias.data for l in lins], dim=0).contiguous()
                  if lins[0].bias is not None else None)
        self.s = None
        if self.block:
            if len(lins) > 1 and any(l.weight.shape[0] % 128 for l in lins):
                raise NotImplementedError("fusing block-quantised projections needs output "
                                          "widths that are multiples of 128")
            self.s = torch.cat([l.weight_scale_inv.data for l in lins], dim=0).contiguous()
        elif self.fp8:
            self.s = torch.cat([l.weight_scale.data.reshape(-1) for l in lins]).contiguous()
        o = so = 0
        for l in lins:  # re-point the module parameters at the packed storage (no 2nd copy)
            n = l.weight.shape[0]
            l.weight.data = self.w[o:o + n]
            if self.b is not None:
                l.bias.data = self.b[o:o + n]
            if self.block:
                nb = l.weight_scale_inv.shape[0]
                l.weight_scale_inv.data = self.s[so:so + nb]
                so += nb
            elif self.fp8:
                l.weight_scale.data = self.s[o:o + n]
            o += n
        self.N, self.K = self.w.shape


class FluxEngine:
    def __init__(self, model):
        self.model = model
        c = model.config
        self.D = model.inner_dim
        self.H = c.num_attention_heads
        self.dh = c.attention_head_dim
        self.guidance_embeds = c.guidance_embeds
        D = self.D
        # ---- one modulation matrix for the whole network ---------------------------
        mods, self.mod_off = [], {}
        off = 0
        for i, blk in enumerate(model.transformer_blocks):
            for name, m in (("img", blk.norm1), ("txt", blk.norm1_context)):
                mods.append(m.linear)
                self.mod_off[("d", i, name)] = off
                off += 6 * D
        for i, blk in enumerate(model.single_transformer_blocks):
            mods.append(blk.norm.linear)
            self.mod_off[("s", i)] = off
            off += 3 * D
        mods.append(model.norm_out.linear)
        self.mod_off[("out",)] = off
        off += 2 * D
        self.n_mod = off
        self.mod_w = torch.cat([m.weight.data for m in mods], dim=0).contiguous()
        self.mod_b = torch.cat([m.bias.data for m in mods], dim=0).contiguous()
        o = 0
        for m in mods:
            n = m.weight.shape[0]
            m.weight.data = self.mod_w[o:o + n]
            m.bias.data = self.mod_b[o:o + n]
            o += n
        # ---- per-block packed projections ---------------------------------------------
        self.dbl = []
        for blk in model.transformer_blocks:
            a = blk.attn
            self.dbl.append(dict(
                qkv=_Packed([a.to_q, a.to_k, a.to_v]),
                add_qkv=_Packed([a.add_q_proj, a.add_k_proj, a.add_v_proj]),
                to_out=_Packed([a.to_out[0]]), to_add_out=_Packed([a.to_add_out]),
                ff1=blk.ff.net[0].proj, ff2=blk.ff.net[2],
                cff1=blk.ff_context.net[0].proj, cff2=blk.ff_context.net[2],
                nq=a.norm_q.weight.data, nk=a.norm_k.weight.data,
                naq=a.norm_added_q.weight.data, nak=a.norm_added_k.weight.data,
                eps=a.norm_q.variance_epsilon))
        self.sgl = []
        for blk in model.single_transformer_blocks:
            a = blk.attn
            self.sgl.append(dict(
                qkv=_Packed([a.to_q, a.to_k, a.to_v]), mlp=_Packed([blk.proj_mlp]),
                out=_Packed([blk.proj_out]), nq=a.norm_q.weight.data, nk=a.norm_k.weight.data,
                eps=a.norm_q.variance_epsilon))
        packs = [pk for blk in self.dbl + self.sgl for pk in blk.values()
                 if isinstance(pk, _Packed) and pk.fp8]
        self.block = bool(packs) and packs[0].block
        if any(pk.block != self.block for pk in packs):
            raise NotImplementedError("per-channel and block-quantised layers in one model")
        if self.block and (D % 128 or (D + (model.single_transformer_blocks[0].mlp_hidden_dim
                                            if self.sgl else 0)) % 128):
            raise NotImplementedError("block-quantised fused path needs widths % 128 == 0")
        self._ws: Dict[Tuple, dict] = {}
        # optional per-launch timing: list of (kind, flops, bytes, start_event, end_event)
        self.trace: Optional[list] = None
        # run the double blocks' text chain on a second stream: same bits, capturable in a HIP
        # graph, -2 ms (-1 %) per step at B = 4 in same-box A/Bs.  Per-launch timing (self.trace)
        # and probes fall back to one stream, where a launch's events mean something.
        # MI355_FLUX_OVERLAP_TEXT=0 turns it off.
        self.overlap_text = os.environ.get("MI355_FLUX_OVERLAP_TEXT", "1") == "1"
        # single blocks: proj_mlp beside the attention branch on the second stream
        # (MI355_FLUX_OVERLAP_MLP=0 turns it off)
        self.overlap_mlp = os.environ.get("MI355_FLUX_OVERLAP_MLP", "1") == "1"
        self._side_streams: Dict[str, torch.cuda.Stream] = {}

    def _side(self, dev) -> torch.cuda.Stream:
        key = str(dev)
        if key not in self._side_streams:
            self._side_streams[key] = torch.cuda.Stream(device=dev)
        return self._side_streams[key]

    def _run(self, kind: str, flops: float, nbytes: float, fn, *args, **kw):
        """Launch `fn`; when tracing, bracket it with events on the launch stream."""
        if self.trace is None:
            return fn(*args, **kw)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*args, **kw)
        e1.record()
        self.trace.append((kind, flops, nbytes, e0, e1))

    # ------------------------------------------------------------------ workspace --
    def _workspace(self, B: int, Ti: int, Tt: int, dev) -> dict:
        key = (B, Ti, Tt, str(dev))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        D, S = self.D, Ti + Tt
        Mi, Mt, Ms = B * Ti, B * Tt, B * S
        bf, f8, f32 = torch.bfloat16, FP8_DTYPE, torch.float32
        mlp = self.model.single_transformer_blocks[0].mlp_hidden_dim \
            if len(self.model.single_transformer_blocks) else 4 * D
        e = lambda *shape, dtype=bf: torch.empty(shape, dtype=dtype, device=dev)  # noqa: E731
        # activation scales: one per token, or one per (token, 128 channels) for block checkpoints
        sc = (lambda width: e(Ms, width // 128, dtype=f32)) if self.block \
            else (lambda width: e(Ms, dtype=f32))
        ws = dict(
            joint=e(Ms, D), mod=e(B, self.n_mod), silu_temb=e(B, D),
            xq=e(Ms, D, dtype=f8), xs=sc(D), xn=e(Ms, D),
            qkv=e(Ms, 3 * D), attn=e(Ms, D), aq=e(Ms, D, dtype=f8), as_=sc(D),
            ffh=e(max(Mi, Mt), 4 * D), cat=e(Ms, D + mlp), cq=e(Ms, D + mlp, dtype=f8),
            cs=sc(D + mlp), out=e(Mi, self.model.proj_out.weight.shape[0]),
            # the text stream's own staging buffers (it may run beside the image stream)
            xq_t=e(Mt, D, dtype=f8), xs_t=sc(D)[:Mt], xn_t=e(Mt, D), ffh_t=e(Mt, 4 * D),
        )
        self._ws[key] = ws
        return ws

    # ---------------------------------------------------------------------- helpers --
    def _prob(self, pk: _Packed, a, out, *, M, sa=None, a_map=None, o_map=None, residual=None,
              gate=None, gate_rows=0, act=0, ldo=None, lda=None, f16_cols=None):
        """One quantised (or packed) projection as a linear problem: (kind, flops, bytes, a, w,
        out, kwargs of _C.linear_ex)."""
        kind = "gemm_fp8blk" if pk.block else "gemm_fp8" if pk.fp8 else "gemm_bf16"
        kw = dict(M=M, N=pk.N, K=pk.K, lda=lda or a.stride(0), ldw=pk.w.stride(0),
                  ldo=ldo or out.stride(0), sa=sa, sb=pk.s, bias=pk.b, residual=residual, gate=gate,
                  ldg=gate.stride(0) if gate is not None else 0, gate_rows=gate_rows, a_map=a_map,
                  o_map=o_map, fp8=pk.fp8, act=act, f16_cols=f16_cols, block_scale=pk.block)
        return (kind, 2.0 * M * pk.N * pk.K,
                (M * pk.K + pk.N * pk.K) * (1 if pk.fp8 else 2) + 2.0 * M * pk.N, a, pk.w, out, kw)

    def _prob_mod(self, mod, a, out, *, M, o_map=None, residual=None, gate=None, gate_rows=0, act=0):
        """Unquantised nn.Linear-style module (FF, embedders) as a linear problem."""
        w = mod.weight.data
        N, K = w.shape
        kw = dict(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldo=out.stride(0),
                  bias=mod.bias.data if mod.bias is not None else None, residual=residual, gate=gate,
                  ldg=gate.stride(0) if gate is not None else 0, gate_rows=gate_rows, o_map=o_map,
                  fp8=False, act=act)
        return ("gemm_bf16", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N), a, w, out, kw)

    def _launch(self, prob):
        kind, flops, nbytes, a, w, out, kw = prob
        big = _uses_big_tile(kw["M"], kw["N"], kw["fp8"], a.device)
        self._run(kind + ("_256" if big else "_128"), flops, nbytes, _C.linear_ex, a, w, out, **kw)

    def _lin(self, pk: _Packed, a, out, **kw):
        self._launch(self._prob(pk, a, out, **kw))

    def _lin_mod(self, mod, a, out, **kw):
        self._launch(self._prob_mod(mod, a, out, **kw))

    # --------------------------------------------------------------------------- step --
    def step(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance,
             freqs_cis, probe=None) -> torch.Tensor:
        """`probe` (tests only): an object whose optional methods see the run from inside --
        `temb(t) -> t` may replace the conditioning vector, `before(kind, i, joint)` /
        `after(kind, i, joint)` are called around block i of kind "d" (double) / "s" (single)
        with the joint hidden state viewed as [B, S_txt + S_img, D] and may overwrite or copy
        it.  That is how a single block is compared with the CPU oracle on identical inputs."""
        m = self.model
        D, H, dh = self.D, self.H, self.dh
        B, Ti, Cin = hidden_states.shape
        Tt = encoder_hidden_states.shape[1]
        S = Ti + Tt
        Mi, Mt, Ms = B * Ti, B * Tt, B * S
        dev = hidden_states.device
        ws = self._workspace(B, Ti, Tt, dev)
        joint, mod = ws["joint"], ws["mod"]
        map_i, map_t = (Ti, S, Tt), (Tt, S, 0)
        cos, sin = freqs_cis
        cos = cos.to(torch.float32).contiguous()
        sin = sin.to(torch.float32).contiguous()

        # ---- conditioning: temb, then every AdaLN modulation in one GEMM ------------------
        if self.guidance_embeds and guidance is not None:
            temb = m.time_text_embed(timestep, guidance, pooled_projections)
        else:
            temb = m.time_text_embed(timestep, pooled_projections)
        if probe is not None and hasattr(probe, "temb"):
            temb = probe.temb(temb)
        _C.activation(ws["silu_temb"], temb.contiguous(), 0)
        self._run("modulation", 2.0 * B * self.n_mod * D, 2.0 * self.n_mod * D, _C.linear_bf16,
                  ws["silu_temb"], self.mod_w, self.mod_b, mod, 0)

        def chunk(off, j):
            return mod[:, off + j * D: off + (j + 1) * D]

        # ---- embedders write their stream's rows of the joint buffer ----------------------
        self._lin_mod(m.x_embedder, hidden_states.reshape(Mi, Cin), joint, M=Mi, o_map=map_i)
        self._lin_mod(m.context_embedder,
                      encoder_hidden_states.reshape(Mt, encoder_hidden_states.shape[-1]), joint,
                      M=Mt, o_map=map_t)

        qkv = ws["qkv"]
        qkv5 = qkv.view(B, S, 3, H, dh)
        q, k, v = qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2]
        attn = ws["attn"]
        attn4 = attn.view(B, S, H, dh)
        scale = dh ** -0.5
        # the V third of every fused q|k|v projection is written in half format: the attention
        # kernel multiplies half probabilities with half values (bf16 V and bf16 probabilities
        # where a caller asked for it, see _C.value_half)
        v_half = _C.value_half()
        vcols = (2 * D, 3 * D) if v_half else None
        if self.block:
            def quant(x, q, s_):  # per (token, 128 channels), fp8.py:303-352 on a block checkpoint
                _C.per_token_group_quant_fp8(x, q, s_, 128, 1e-10, -448.0, 448.0, False)
        else:
            quant = _C.per_token_quant_fp8

        def norm_in(pk_fp8, shift, scl, T, rmap, M, sfx=""):
            """LayerNorm + modulation (+ fp8 quant) of one stream -> (activation, scales)."""
            if pk_fp8:
                fn = (_C.layernorm_modulate_group_quant if self.block
                      else _C.layernorm_modulate_quant)
                self._run("elementwise", 0.0, 3.0 * M * D, fn, ws["xq" + sfx], ws["xs" + sfx],
                          joint, shift, scl, B, T, 1e-6, x_map=rmap)
                return ws["xq" + sfx], ws["xs" + sfx]
            self._run("elementwise", 0.0, 4.0 * M * D, _C.layernorm_modulate, ws["xn" + sfx],
                      joint, shift, scl, B, T, 1e-6, x_map=rmap)
            return ws["xn" + sfx], None

        # The text rows of a double block depend on the image rows only through the joint
        # attention, so their chain can run on a second stream beside the image chain.
        main = torch.cuda.current_stream(dev)
        side = (self._side(dev) if (self.overlap_text and self.trace is None and probe is None
                                    and self.dbl) else None)
        if side is not None:
            side.wait_event(main.record_event())

        def on_text_stream():
            return torch.cuda.stream(side) if side is not None else contextlib.nullcontext()

        # ------------------------------------------------------------------ double blocks --
        joint3 = joint.view(B, S, D)

        def tell(when, kind, i):
            fn = getattr(probe, when, None) if probe is not None else None
            if fn is not None:
                fn(kind, i, joint3)

        def pair(img, txt):
            """The image- and the text-stream problem of one projection (the text one on the
            second stream when that is on).  Sharing ONE launch was measured and buys nothing:
            the image problem fills whole rounds of the persistent grid by itself."""
            self._launch(img)
            with on_text_stream():
                self._launch(txt)

        for i, blk in enumerate(self.dbl):
            tell("before", "d", i)
            oi, ot = self.mod_off[("d", i, "img")], self.mod_off[("d", i, "txt")]
            a, sa = norm_in(blk["qkv"].fp8, chunk(oi, 0), chunk(oi, 1), Ti, map_i, Mi)
            with on_text_stream():
                at, sat = norm_in(blk["add_qkv"].fp8, chunk(ot, 0), chunk(ot, 1), Tt, map_t, Mt, "_t")
            pair(self._prob(blk["qkv"], a, qkv, M=Mi, sa=sa, o_map=map_i, f16_cols=vcols),
                 self._prob(blk["add_qkv"], at, qkv, M=Mt, sa=sat, o_map=map_t, f16_cols=vcols))
            if side is not None:
                main.wait_event(side.record_event())
            # RMSNorm + RoPE: in place on k; the attention kernel does q's as it loads Q.  (The
            # key side inside the q|k|v projection's epilogue was built and measured: same bits,
            # but the exposed epilogue work added 5.1 ms per step where this pass costs 2.65.)
            self._run("elementwise", 0.0, 4.0 * Ms * D, _C.qk_norm_rope, None, k, blk["naq"],
                      blk["nak"], blk["nq"], blk["nk"], Tt, blk["eps"], cos, sin, True)
            self._run("attention", 4.0 * B * S * S * D, 8.0 * Ms * D, _C.flash_attn_fwd, q, k, v,
                      attn4, None, scale, False, v_half,
                      q_norm=(blk["naq"], blk["nq"], Tt, blk["eps"]), q_rope=(cos, sin))
            if blk["to_out"].fp8:
                self._run("elementwise", 0.0, 3.0 * Ms * D, quant, attn, ws["aq"], ws["as_"])
                a_src, a_s = ws["aq"], ws["as_"]
            else:
                a_src, a_s = attn, None
            if side is not None:
                side.wait_event(main.record_event())
            # output projections, then the feed-forward (never quantised); gate and residual
            # are folded into the GEMM that ends each half
            pair(self._prob(blk["to_out"], a_src, joint, M=Mi, sa=a_s, a_map=map_i, o_map=map_i,
                            residual=joint, gate=chunk(oi, 2), gate_rows=Ti),
                 self._prob(blk["to_add_out"], a_src, joint, M=Mt, sa=a_s, a_map=map_t, o_map=map_t,
                            residual=joint, gate=chunk(ot, 2), gate_rows=Tt))
            self._run("elementwise", 0.0, 4.0 * Mi * D, _C.layernorm_modulate, ws["xn"], joint,
                      chunk(oi, 3), chunk(oi, 4), B, Ti, 1e-6, x_map=map_i)
            with on_text_stream():
                self._run("elementwise", 0.0, 4.0 * Mt * D, _C.layernorm_modulate, ws["xn_t"], joint,
                          chunk(ot, 3), chunk(ot, 4), B, Tt, 1e-6, x_map=map_t)
            ffh, ffh_t = ws["ffh"][:Mi], ws["ffh_t"][:Mt]
            pair(self._prob_mod(blk["ff1"], ws["xn"][:Mi], ffh, M=Mi, act=1),
                 self._prob_mod(blk["cff1"], ws["xn_t"][:Mt], ffh_t, M=Mt, act=1))
            pair(self._prob_mod(blk["ff2"], ffh, joint, M=Mi, o_map=map_i, residual=joint,
                                gate=chunk(oi, 5), gate_rows=Ti),
                 self._prob_mod(blk["cff2"], ffh_t, joint, M=Mt, o_map=map_t, residual=joint,
                                gate=chunk(ot, 5), gate_rows=Tt))
            tell("after", "d", i)
        if side is not None:
            main.wait_event(side.record_event())

        # ------------------------------------------------------------------ single blocks --
        cat = ws["cat"]
        catw = cat.shape[1]
        side_single = (self._side(dev) if (self.overlap_mlp and self.trace is None
                                           and probe is None) else None)
        attn_in_cat = cat.view(B, S, catw)[:, :, :D].unflatten(-1, (H, dh))
        mlp_cols = cat[:, D:]
        for i, blk in enumerate(self.sgl):
            tell("before", "s", i)
            off = self.mod_off[("s", i)]
            a, sa = norm_in(blk["qkv"].fp8, chunk(off, 0), chunk(off, 1), S, None, Ms)
            # proj_mlp (+ GELU) does not depend on the attention branch: on the second stream
            # its tiles fill the CUs that the q|k|v projection's and the attention's partial
            # last rounds leave idle (same bits; joined before the concatenation is quantised)
            if side_single is not None:
                side_single.wait_event(main.record_event())
                with torch.cuda.stream(side_single):
                    self._lin(blk["mlp"], a, mlp_cols, M=Ms, sa=sa, act=1, ldo=catw)
            self._lin(blk["qkv"], a, qkv, M=Ms, sa=sa, f16_cols=vcols)
            if side_single is None:
                self._lin(blk["mlp"], a, mlp_cols, M=Ms, sa=sa, act=1, ldo=catw)
            self._run("elementwise", 0.0, 4.0 * Ms * D, _C.qk_norm_rope, None, k, blk["nq"],
                      blk["nk"], blk["nq"], blk["nk"], 0, blk["eps"], cos, sin, True)
            self._run("attention", 4.0 * B * S * S * D, 8.0 * Ms * D, _C.flash_attn_fwd, q, k, v,
                      attn_in_cat, None, scale, False, v_half,
                      q_norm=(blk["nq"], blk["nq"], 0, blk["eps"]), q_rope=(cos, sin))
            if side_single is not None:
                main.wait_event(side_single.record_event())
            if blk["out"].fp8:
                self._run("elementwise", 0.0, 3.0 * Ms * catw, quant, cat, ws["cq"], ws["cs"])
                c_src, c_s = ws["cq"], ws["cs"]
            else:
                c_src, c_s = cat, None
            self._lin(blk["out"], c_src, joint, M=Ms, sa=c_s, residual=joint, gate=chunk(off, 2),
                      gate_rows=S)
            tell("after", "s", i)

        # ---- output head: AdaLayerNormContinuous (scale first) + proj_out on image rows ------
        oo = self.mod_off[("out",)]
        self._run("elementwise", 0.0, 4.0 * Mi * D, _C.layernorm_modulate, ws["xn"], joint,
                  chunk(oo, 1), chunk(oo, 0), B, Ti, 1e-6, x_map=map_i)
        out = ws["out"]
        self._lin_mod(m.proj_out, ws["xn"][:Mi], out, M=Mi)
        return out.view(B, Ti, -1)
