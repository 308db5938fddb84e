"""Parity at the REAL Flux.1 width: one double and one single block with D = 3072, 24 heads of
128, 512 text + 4096 image tokens, against the CPU oracle on identical block inputs
("teacher-forced": every block, and every operator in the budget test, gets the oracle's input,
so rounding differences cannot compound through the residual stream).

What only shows up at this size and is therefore checked here and nowhere else: the K = 15360
proj_out GEMM, the 24-head strided q|k|v views with half-format V columns, the joint-buffer row
maps at S = 4608, and the one-launch modulation GEMM.

Reference block code: python/sglang/multimodal_gen/runtime/models/dits/flux.py:499-559 (single),
:620-692 (double).
"""
import pytest
import torch
import torch.nn.functional as F

from tests.flux_small import build_oracle, build_small
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu

D, H, DH, TT, TI = 3072, 24, 128, 512, 4096
S = TT + TI

# ---- per-block bars (relative L2 per output stream, teacher-forced) ---------------------------
# A block's output stream is residual + gate * branch.  What differs between the kernels and
# the oracle is (a) the fp32 summation order inside GEMMs / norms, which flips a bf16 rounding
# now and then (each flip is 2^-8 relative on that element), and (b) the attention kernel's
# probabilities rounded to 16 bits before P.V against the oracle's fp32 softmax.  The budget
# test below measures each operator alone at this size.  The bars (tests/real_width_bars.py)
# are read against a MEASURED floor: tests/test_oracle_sensitivity.py evaluates the oracle
# itself with another GEMM summation order and 16-bit probabilities at this very size and
# lands at 1.63e-3 (double) / 3.1e-4 (single) in bf16 -- where the kernels land too (1.67e-3 /
# 3.5e-4) -- and at 1.65e-3 / 4.5e-4 under W8A8, where the kernels are at 2.2e-3 / 1.2e-3: the
# fp8 MFMA's own accumulation (the W8A8 q|k|v GEMM alone is 2.5e-4 from an exact fp32 sum)
# leaves one-ulp differences in front of the next quantiser, and each can move an e4m3 code.
from tests.real_width_bars import BAR  # noqa: E402  (shared with the CPU floor measurement)


def _real_width(device, fp8):
    model, state, cfg = build_small(device, fp8=fp8, num_layers=1, num_single_layers=1,
                                    num_attention_heads=H, joint_attention_dim=4096,
                                    pooled_projection_dim=768)
    return model, build_oracle(state, cfg, fp8), cfg


def _block_inputs(seed=3):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(1, TI, D, generator=g).bfloat16()
    enc = torch.randn(1, TT, D, generator=g).bfloat16()
    temb = torch.randn(1, D, generator=g).bfloat16()
    return hid, enc, temb


class _TeacherForcing:
    """Engine probe: overwrite the joint state before a block, keep a copy after it."""

    def __init__(self, temb, feeds):
        self._temb, self.feeds, self.got = temb, feeds, {}

    def temb(self, t):
        return self._temb.to(t.device)

    def before(self, kind, i, joint):
        enc, hid = self.feeds[(kind, i)]
        joint[:, :enc.shape[1]].copy_(enc)
        joint[:, enc.shape[1]:].copy_(hid)

    def after(self, kind, i, joint):
        self.got[(kind, i)] = joint.clone()


@pytest.mark.parametrize("fp8", [False, True])
def test_blocks_at_real_width_vs_oracle(device, fp8):
    model, oracle, cfg = _real_width(device, fp8)
    hid, enc, temb = _block_inputs()
    freqs = oracle.freqs_cis(TT, 1024, 1024)
    with torch.no_grad():
        d_enc, d_hid = oracle.double_block(0, hid, enc, temb, freqs)
        s_enc, s_hid = oracle.single_block(0, d_hid, d_enc, temb, freqs)
    dev = lambda x: x.to(device)  # noqa: E731
    fd = (dev(freqs[0]), dev(freqs[1]))
    errs = {}
    with torch.no_grad():
        # layer path: the block modules, called as the model's forward calls them
        e1, h1 = model.transformer_blocks[0](hidden_states=dev(hid), encoder_hidden_states=dev(enc),
                                             temb=dev(temb), freqs_cis=fd)
        e2, h2 = model.single_transformer_blocks[0](hidden_states=dev(d_hid),
                                                    encoder_hidden_states=dev(d_enc),
                                                    temb=dev(temb), freqs_cis=fd)
        errs["layer"] = {"d": (rel_l2(e1, d_enc), rel_l2(h1, d_hid)),
                         "s": (rel_l2(e2, s_enc), rel_l2(h2, s_hid))}
        # fused engine: the whole 1 + 1 block model runs, its blocks fed the oracle's inputs
        probe = _TeacherForcing(dev(temb), {("d", 0): (dev(enc), dev(hid)),
                                            ("s", 0): (dev(d_enc), dev(d_hid))})
        g = torch.Generator().manual_seed(1)
        lat = torch.randn(1, TI, cfg.in_channels, generator=g).bfloat16()
        txt = torch.randn(1, TT, cfg.joint_attention_dim, generator=g).bfloat16()
        pooled = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16()
        model.forward_fused(dev(lat), dev(txt), dev(pooled), dev(torch.full((1,), 500.0)),
                            dev(torch.full((1,), 3500.0).bfloat16()), fd)      # builds the engine
        model._engine.step(dev(lat), dev(txt), dev(pooled), dev(torch.full((1,), 500.0)),
                           dev(torch.full((1,), 3500.0).bfloat16()), fd, probe=probe)
        jd, js = probe.got[("d", 0)], probe.got[("s", 0)]
        errs["fused"] = {"d": (rel_l2(jd[:, :TT], d_enc), rel_l2(jd[:, TT:], d_hid)),
                         "s": (rel_l2(js[:, :TT], s_enc), rel_l2(js[:, TT:], s_hid))}
        fused_vs_layer = max(rel_l2(jd[:, :TT], e1), rel_l2(jd[:, TT:], h1),
                             rel_l2(js[:, :TT], e2), rel_l2(js[:, TT:], h2))
    for path, e in errs.items():
        print(f"real width fp8={fp8} {path}: double (txt {e['d'][0]:.2e}, img {e['d'][1]:.2e})  "
              f"single (txt {e['s'][0]:.2e}, img {e['s'][1]:.2e})")
    print(f"real width fp8={fp8}: fused vs layer path {fused_vs_layer:.2e}")
    for path, e in errs.items():
        for kind in ("d", "s"):
            for stream, v in zip(("text", "image"), e[kind]):
                assert v <= BAR[(kind, fp8)], (path, kind, stream, v)
    # the two product paths are the same arithmetic
    assert fused_vs_layer == 0.0


def test_modulation_gemm_at_real_width(device):
    """All AdaLN modulation linears of the network as ONE skinny GEMM (here 1 double + 1 single
    block + the output norm: N = 12 D + 3 D + 2 D): every chunk equals the oracle's
    Linear(SiLU(temb)) of the module it belongs to."""
    model, oracle, cfg = _real_width(device, False)
    _, _, temb = _block_inputs()
    temb = temb.repeat(4, 1) * torch.tensor([1.0, 0.5, -1.0, 2.0]).view(4, 1).bfloat16()
    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.models.flux_engine import FluxEngine

    eng = model._engine or FluxEngine(model)
    st = torch.empty(4, D, dtype=torch.bfloat16, device=device)
    _C.activation(st, temb.to(device).contiguous(), 0)
    mod = torch.empty(4, eng.n_mod, dtype=torch.bfloat16, device=device)
    _C.linear_bf16(st, eng.mod_w, eng.mod_b, mod, 0)
    for key, name, width in ((("d", 0, "img"), "transformer_blocks.0.norm1.linear", 6 * D),
                             (("d", 0, "txt"), "transformer_blocks.0.norm1_context.linear", 6 * D),
                             (("s", 0), "single_transformer_blocks.0.norm.linear", 3 * D),
                             (("out",), "norm_out.linear", 2 * D)):
        ref = oracle.lin(name, F.silu(temb))
        off = eng.mod_off[key]
        e = rel_l2(mod[:, off:off + width], ref)
        print(f"modulation {name}: {e:.2e}")
        assert e <= 1e-3, (name, e)


# ---- operator budget of the double block's image stream, each operator on the oracle's input --
OP_BAR = {
    # bf16 in, bf16 out, fp32 inside; only the summation order differs -> rare one-ulp flips
    # (measured 5e-6 .. 5e-5; W8A8 q|k|v 2.5e-4)
    "layernorm_modulate": 1e-4, "qkv_gemm": 5e-4, "qk_norm_rope": 1e-4,
    # 16-bit probabilities before P.V (half here, bf16 in every flash kernel the reference
    # calls); measured 8.6e-4: this operator carries the block's deviation
    "attention": 1e-3,
    "to_out_gate_residual": 1e-4, "ff1_gelu": 1e-4, "ff2_gate_residual": 1e-4,
}


@pytest.mark.parametrize("fp8", [False, True])
def test_operator_error_budget_at_real_width(device, fp8):
    """Each operator of a double block (image stream; attention over the joint sequence) run
    alone at the real shapes on the ORACLE's input: where a block's deviation comes from."""
    from oracle import ops as O
    from sglang_diffusion_amd import _C
    from sglang_diffusion_amd.models.flux_engine import FluxEngine

    model, oracle, cfg = _real_width(device, fp8)
    eng = model._engine or FluxEngine(model)
    blk = eng.dbl[0]
    p = "transformer_blocks.0"
    hid, enc, temb = _block_inputs(seed=5)
    freqs = oracle.freqs_cis(TT, 1024, 1024)
    cos, sin = freqs[0].to(device), freqs[1].to(device)
    dev = lambda x: x.to(device)  # noqa: E731
    bf = torch.bfloat16
    got = {}
    with torch.no_grad():
        mods = oracle.lin(p + ".norm1.linear", F.silu(temb)).chunk(6, dim=1)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mods
        # 1. LayerNorm + modulation
        r_ln = O.layernorm_modulate(hid, shift_msa, scale_msa)
        o = torch.empty(TI, D, dtype=bf, device=device)
        _C.layernorm_modulate(o, dev(hid).view(TI, D), dev(shift_msa), dev(scale_msa), 1, TI, 1e-6)
        got["layernorm_modulate"] = rel_l2(o, r_ln[0])
        # 2. fused q|k|v projection (W8A8: per-token quantisation of the oracle's input first)
        x = dev(r_ln[0]).contiguous()
        r_qkv = torch.cat([oracle.lin(f"{p}.attn.{n}", r_ln) for n in ("to_q", "to_k", "to_v")], -1)[0]
        qkv = torch.empty(TI, 3 * D, dtype=bf, device=device)
        if fp8:
            xq = torch.empty(TI, D, dtype=torch.float8_e4m3fn, device=device)
            xs = torch.empty(TI, dtype=torch.float32, device=device)
            _C.per_token_quant_fp8(x, xq, xs)
            rq, rs = O.per_token_quant_fp8(r_ln[0])
            assert torch.equal(xq.cpu().view(torch.uint8), rq.view(torch.uint8))
            assert torch.equal(xs.cpu(), rs.view(-1))
            eng._lin(blk["qkv"], xq, qkv, M=TI, sa=xs)
        else:
            eng._lin(blk["qkv"], x, qkv, M=TI)
        got["qkv_gemm"] = rel_l2(qkv, r_qkv)
        # 3. per-head RMSNorm + interleaved RoPE on a [1, S, H, d] tensor (image rows rotate
        #    with the image part of the table)
        g = torch.Generator().manual_seed(11)
        q_in = torch.randn(1, S, H, DH, generator=g).bfloat16()
        k_in = torch.randn(1, S, H, DH, generator=g).bfloat16()
        w1 = oracle.sd[p + ".attn.norm_q.weight"]
        rq_, rk_ = O.qk_norm_rope(q_in, k_in, w1, oracle.sd[p + ".attn.norm_k.weight"], 1e-6,
                                  freqs[0], freqs[1])
        qd, kd = dev(q_in).clone(), dev(k_in).clone()
        _C.qk_norm_rope(qd, kd, blk["nq"], blk["nk"], blk["nq"], blk["nk"], 0, 1e-6, cos, sin, True)
        got["qk_norm_rope"] = max(rel_l2(qd, rq_), rel_l2(kd, rk_))
        # 4. joint attention, all 24 heads, S = 4608, from strided q|k|v views with V in half
        v_in = torch.randn(1, S, H, DH, generator=g).bfloat16()
        r_att = O.attention(rq_, rk_, v_in)
        buf = torch.empty(1, S, 3, H, DH, dtype=bf, device=device)
        buf[:, :, 0], buf[:, :, 1] = dev(rq_), dev(rk_)
        _C.bf16_to_f16(dev(v_in).view(S, D), buf.view(S, 3 * D)[:, 2 * D:])
        att = torch.empty(1, S, H, DH, dtype=bf, device=device)
        _C.flash_attn_fwd(buf[:, :, 0], buf[:, :, 1], buf[:, :, 2], att, None, DH ** -0.5, False, True)
        got["attention"] = rel_l2(att, r_att)
        # 5. to_out + gate * y + residual on the image rows
        a_img = r_att[:, TT:].flatten(2, 3)
        r_out = hid + gate_msa.unsqueeze(1) * oracle.lin(p + ".attn.to_out.0", a_img)
        a_dev = dev(a_img[0]).contiguous()
        res = dev(hid[0]).clone()
        if fp8:
            _C.per_token_quant_fp8(a_dev, xq, xs)
            eng._lin(blk["to_out"], xq, res, M=TI, sa=xs, residual=res, gate=dev(gate_msa), gate_rows=TI)
        else:
            eng._lin(blk["to_out"], a_dev, res, M=TI, residual=res, gate=dev(gate_msa), gate_rows=TI)
        got["to_out_gate_residual"] = rel_l2(res, r_out[0])
        # 6. feed-forward: FF1 + tanh-GELU, FF2 + gate * y + residual (never quantised)
        nh = F.layer_norm(r_out, (D,), eps=1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        r_ff1 = oracle.lin(p + ".ff.net.0.proj", nh, act="gelu")
        ffh = torch.empty(TI, 4 * D, dtype=bf, device=device)
        eng._lin_mod(blk["ff1"], dev(nh[0]).contiguous(), ffh, M=TI, act=1)
        got["ff1_gelu"] = rel_l2(ffh, r_ff1[0])
        r_ff2 = r_out + gate_mlp.unsqueeze(1) * oracle.lin(p + ".ff.net.2", r_ff1)
        res2 = dev(r_out[0]).clone()
        eng._lin_mod(blk["ff2"], dev(r_ff1[0]).contiguous(), res2, M=TI, residual=res2,
                     gate=dev(gate_mlp), gate_rows=TI)
        got["ff2_gate_residual"] = rel_l2(res2, r_ff2[0])
    print(f"operator budget at real width (fp8={fp8}): "
          + ", ".join(f"{k} {v:.2e}" for k, v in got.items()))
    for k, v in got.items():
        assert v <= OP_BAR[k], (k, v)
