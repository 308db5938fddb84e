"""Accuracy and speed of the two attention forms against an fp32 softmax reference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
torch.manual_seed(0)
dev = "cuda"
B, S, H, D = 1, 1024, 4, 128
q = torch.randn(B, S, H, D, device=dev).bfloat16(); k = torch.randn(B, S, H, D, device=dev).bfloat16()
v = torch.randn(B, S, H, D, device=dev).bfloat16()
qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
ref = (torch.softmax(qf @ kf.transpose(-1, -2) * D ** -0.5, -1) @ vf).transpose(1, 2)
rel = lambda a: ((a.float() - ref).norm() / ref.norm()).item()
o0 = torch.empty_like(q); _C.flash_attn_fwd(q, k, v, o0, None, D ** -0.5)
o1 = torch.empty_like(q); _C.attention(q, k, v, o1, None, D ** -0.5)
print(f"bf16 P: rel {rel(o0):.3e}   half P: rel {rel(o1):.3e}   ref rounded to bf16: {rel(ref.bfloat16()):.3e}")
B, S, H = 4, 4608, 24
q = torch.randn(B, S, H, D, device=dev).bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q)
v16 = _C.value_to_f16(v); o = torch.empty_like(q)
for name, fn in (("bf16", lambda: _C.flash_attn_fwd(q, k, v, o, None, D ** -0.5)),
                 ("half", lambda: _C.flash_attn_fwd(q, k, v16, o, None, D ** -0.5, False, True)),
                 ("cvt ", lambda: _C.value_to_f16(v))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    print(f"{name}: {us:8.1f} us  {4.0*B*S*S*H*D/us/1e6:7.1f} TF/s")
