"""In-kernel split of the two-phase attention loop (MI355_ATTN_PINGPONG=1 MI355_ATTN_ABLATE=2): per
wave of workgroup 0, shader clocks (s_memtime) spent in MFMA blocks, in vector (softmax) blocks and
waiting at the two barriers of a tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
v16 = _C.value_to_f16(v); o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, H, S, device=dev)
for _ in range(3):
    _C.flash_attn_fwd(q, k, v16, o, lse, D ** -0.5, False, True)
torch.cuda.synchronize()
nt = S // 64
x = lse.view(-1)[:24].cpu().view(8, 3)
print(f"per wave, shader clocks per 64-key tile ({nt} tiles): MFMA block, vector block, barrier waits")
for w in range(8):
    print(f"wave {w}: mfma {x[w,0]/nt:7.0f}  vector {x[w,1]/nt:7.0f}  barriers {x[w,2]/nt:7.0f}  sum {x[w].sum()/nt:7.0f}")
