import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
import torch.nn.functional as F
from tests.flux_small import build_oracle, build_small, small_inputs
from tests.helpers import rel_l2, mismatch_frac
from oracle import ops
from sglang_diffusion_amd import _C
dev = torch.device("cuda:0")
model, state, cfg = build_small(dev, fp8=False)
orc = build_oracle(state, cfg, False)
B, Hpx, Wpx, Tt = 2, 128, 96, 24
Ti = (Hpx // 16) * (Wpx // 16)
hid, enc, pooled, t, guid = small_inputs(B, Ti, Tt, cfg)
freqs = orc.freqs_cis(Tt, Hpx, Wpx)
ref, inter = orc.forward(hid, enc, pooled, t, guid, freqs, return_intermediates=True)
d = lambda x: x.to(dev)
def rep(name, g, o): print(f"{name:28s} rel {rel_l2(g, o):.2e} mismatch {mismatch_frac(g, o):.4f}")
p = "transformer_blocks.0"; blk = model.transformer_blocks[0]; H = orc.H
temb, hid0, enc0 = inter["temb"], inter["hid0"], inter["enc0"]
st = F.silu(temb)
e = orc.lin(p + ".norm1.linear", st); ce = orc.lin(p + ".norm1_context.linear", st)
sh, sc, gm, shm, scm, gmlp = e.chunk(6, 1); csh, csc, cgm, cshm, cscm, cgmlp = ce.chunk(6, 1)
nh = ops.layernorm_modulate(hid0, sh, sc); ne = ops.layernorm_modulate(enc0, csh, csc)
pre = p + ".attn"
q = orc.lin(pre + ".to_q", nh).unflatten(-1, (H, -1)); k = orc.lin(pre + ".to_k", nh).unflatten(-1, (H, -1))
v = orc.lin(pre + ".to_v", nh).unflatten(-1, (H, -1))
eq = orc.lin(pre + ".add_q_proj", ne).unflatten(-1, (H, -1)); ek = orc.lin(pre + ".add_k_proj", ne).unflatten(-1, (H, -1))
ev = orc.lin(pre + ".add_v_proj", ne).unflatten(-1, (H, -1))
qn = ops.rmsnorm(q, orc.sd[pre + ".norm_q.weight"], 1e-6); kn = ops.rmsnorm(k, orc.sd[pre + ".norm_k.weight"], 1e-6)
eqn = ops.rmsnorm(eq, orc.sd[pre + ".norm_added_q.weight"], 1e-6); ekn = ops.rmsnorm(ek, orc.sd[pre + ".norm_added_k.weight"], 1e-6)
Q, K, V = torch.cat([eqn, qn], 1), torch.cat([ekn, kn], 1), torch.cat([ev, v], 1)
cos, sin = freqs
Qr, Kr = ops.rope_interleaved(Q, cos, sin), ops.rope_interleaved(K, cos, sin)
O = ops.attention(Qr, Kr, V)
with torch.no_grad():
    # GPU: norm+rope on oracle q/k
    Qg, Kg = d(torch.cat([eq, q], 1)).contiguous(), d(torch.cat([ek, k], 1)).contiguous()
    a = blk.attn
    _C.qk_norm_rope(Qg, Kg, a.norm_added_q.weight.data, a.norm_added_k.weight.data, a.norm_q.weight.data,
                    a.norm_k.weight.data, Tt, 1e-6, d(cos).contiguous(), d(sin).contiguous(), True)
    rep("qk_norm_rope q", Qg, Qr); rep("qk_norm_rope k", Kg, Kr)
    Og = torch.empty_like(Qg); _C.attention(d(Qr), d(Kr), d(V), Og, None, Q.shape[-1] ** -0.5)
    rep("attention (half P)", Og, O)
    Og0 = torch.empty_like(Qg); _C.flash_attn_fwd(d(Qr), d(Kr), d(V), Og0, None, Q.shape[-1] ** -0.5)
    rep("attention (bf16 P)", Og0, O)
    o = O.flatten(2, 3); eo, ho = o[:, :Tt], o[:, Tt:]
    to = orc.lin(pre + ".to_out.0", ho); tg, _ = a.to_out[0](d(ho)); rep("to_out", tg, to)
    hid1 = hid0 + gm.unsqueeze(1) * to
    g1 = torch.empty_like(d(hid0)); _C.gate_residual(g1, d(hid0), d(gm).contiguous(), d(to).contiguous(), B, Ti)
    rep("gate_residual", g1, hid1)
    n2 = F.layer_norm(hid1, (orc.D,), eps=1e-6); n2 = n2 * (1 + scm[:, None]) + shm[:, None]
    n2g = torch.empty_like(g1); _C.layernorm_modulate(n2g, d(hid1), d(shm).contiguous(), d(scm).contiguous(), B, Ti, 1e-6)
    rep("norm2+mod", n2g, n2)
    f1 = orc.lin(p + ".ff.net.0.proj", n2, act="gelu"); f1g = blk.ff.net[0](d(n2)); rep("ff1+gelu", f1g, f1)
    f2 = orc.lin(p + ".ff.net.2", f1); f2g, _ = blk.ff.net[2](d(f1)); rep("ff2", f2g, f2)
    hid2 = hid1 + gmlp.unsqueeze(1) * f2
    g2 = torch.empty_like(g1); _C.gate_residual(g2, d(hid1), d(gmlp).contiguous(), d(f2).contiguous(), B, Ti)
    rep("gate_residual 2", g2, hid2)
    print("oracle d0 hid == recomputed:", rel_l2(hid2, inter["d0"][1]))
    eo_, ho_ = blk(hidden_states=d(hid0), encoder_hidden_states=d(enc0), temb=d(temb), freqs_cis=(d(cos), d(sin)))
    rep("whole block hid", ho_, inter["d0"][1])
