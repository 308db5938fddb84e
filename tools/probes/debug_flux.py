import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
import torch.nn.functional as F
from tests.flux_small import build_oracle, build_small, small_inputs
from tests.helpers import rel_l2, mismatch_frac
from oracle import ops
dev = torch.device("cuda:0")
fp8 = len(sys.argv) > 1 and sys.argv[1] == "fp8"
model, state, cfg = build_small(dev, fp8=fp8)
oracle = build_oracle(state, cfg, fp8)
B, Hpx, Wpx, Tt = 2, 128, 96, 24
Ti = (Hpx // 16) * (Wpx // 16)
hid, enc, pooled, t, guid = small_inputs(B, Ti, Tt, cfg)
freqs = oracle.freqs_cis(Tt, Hpx, Wpx)
ref, inter = oracle.forward(hid, enc, pooled, t, guid, freqs, return_intermediates=True)
d = lambda x: x.to(dev)
fd = (d(freqs[0]), d(freqs[1]))
with torch.no_grad():
    h, _ = model.x_embedder(d(hid))
    temb = model.time_text_embed(d(t), d(guid), d(pooled))
    e, _ = model.context_embedder(d(enc))
    print("temb", rel_l2(temb, inter["temb"]), mismatch_frac(temb, inter["temb"]))
    print("hid0", rel_l2(h, inter["hid0"]), mismatch_frac(h, inter["hid0"]))
    print("enc0", rel_l2(e, inter["enc0"]), mismatch_frac(e, inter["enc0"]))
    # feed ORACLE intermediates into each block to isolate per-block error
    prev_e, prev_h = inter["enc0"], inter["hid0"]
    for i, blk in enumerate(model.transformer_blocks):
        eo, ho = blk(hidden_states=d(prev_h), encoder_hidden_states=d(prev_e), temb=d(inter["temb"]), freqs_cis=fd)
        re, rh = inter[f"d{i}"]
        print(f"double {i}: enc {rel_l2(eo, re):.2e} hid {rel_l2(ho, rh):.2e}  mismatch {mismatch_frac(ho, rh):.3f}")
        prev_e, prev_h = re, rh
    for i, blk in enumerate(model.single_transformer_blocks):
        eo, ho = blk(hidden_states=d(prev_h), encoder_hidden_states=d(prev_e), temb=d(inter["temb"]), freqs_cis=fd)
        re, rh = inter[f"s{i}"]
        print(f"single {i}: enc {rel_l2(eo, re):.2e} hid {rel_l2(ho, rh):.2e}  mismatch {mismatch_frac(ho, rh):.3f}")
        prev_e, prev_h = re, rh
    # inside double block 0: step by step with oracle inputs
    p = "transformer_blocks.0"
    blk = model.transformer_blocks[0]
    tembo = inter["temb"]; hido = inter["hid0"]; enco = inter["enc0"]
    st = F.silu(tembo)
    eo = oracle.lin(p + ".norm1.linear", st)
    from sglang_diffusion_amd.layers.layernorm import _silu
    eg = blk.norm1.linear(_silu(d(tembo)))
    print("norm1.linear", rel_l2(eg, eo), mismatch_frac(eg, eo))
    sh, sc, gm, shm, scm, gmlp = eo.chunk(6, dim=1)
    nh_o = ops.layernorm_modulate(hido, sh, sc)
    nh_g, *_ = blk.norm1(d(hido), emb=d(tembo))
    print("norm1 out", rel_l2(nh_g, nh_o), mismatch_frac(nh_g, nh_o))
    q_o = oracle.lin(p + ".attn.to_q", nh_o)
    q_g, _ = blk.attn.to_q(d(nh_o))
    print("to_q", rel_l2(q_g, q_o), mismatch_frac(q_g, q_o))
    ff_in = torch.randn(2, 48, 256).bfloat16()
    f_o = oracle.lin(p + ".ff.net.0.proj", ff_in, act="gelu")
    f_g = blk.ff.net[0](d(ff_in))
    print("ff1+gelu", rel_l2(f_g, f_o), mismatch_frac(f_g, f_o))
    f2_o = oracle.lin(p + ".ff.net.2", f_o)
    f2_g = blk.ff.net[2](d(f_o))
    print("ff2", rel_l2(f2_g, f2_o), mismatch_frac(f2_g, f2_o))
    # oracle linear vs fp32 reference
    w = state[p + ".ff.net.2.weight"]; b = state[p + ".ff.net.2.bias"]
    f2_ref = (f_o.float() @ w.float().t() + b.float()).bfloat16()
    print("ff2 oracle(F.linear bf16 cpu) vs fp32-matmul-rounded:", rel_l2(f2_o, f2_ref), mismatch_frac(f2_o, f2_ref))
    print("ff2 gpu vs fp32-matmul-rounded:", rel_l2(f2_g, f2_ref), mismatch_frac(f2_g, f2_ref))
