import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sglang-diffusion_amd"))
from sglang_diffusion_amd import _C
dev = torch.device("cuda:0")
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
B, S, H, D = 1, 64, 1, 128
# (a) uniform attention, V[key][dv] = dv
q = torch.zeros(B, S, H, D, device=dev, dtype=torch.bfloat16)
k = torch.zeros_like(q)
v = torch.arange(D, device=dev).float().view(1, 1, 1, D).expand(B, S, H, D).contiguous().bfloat16()
out = torch.empty_like(q)
_C.flash_attn_fwd(q, k, v, out, None, D ** -0.5)
print("(a) out[0,0,0,:] should be 0..127:\n", out[0, 0, 0].float())
print("(a) out[0,5,0,:16]:", out[0, 5, 0, :16].float())
# (b) one-hot attention, V[key][dv] = key
q = torch.zeros(B, S, H, D, device=dev)
k = torch.zeros(B, S, H, D, device=dev)
for i in range(S):
    q[0, i, 0, i] = 30.0
    k[0, i, 0, i] = 30.0
v = torch.arange(S, device=dev).float().view(1, S, 1, 1).expand(B, S, H, D).contiguous()
q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
out = torch.empty_like(q)
_C.flash_attn_fwd(q, k, v, out, None, D ** -0.5)
print("(b) out[0,q,0,0] should be q:\n", out[0, :, 0, 0].float())
print("(b) out[0,q,0,77]:\n", out[0, :, 0, 77].float())
