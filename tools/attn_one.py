"""The Flux-shaped attention call a few times: target of rocprofv3 counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
v16 = _C.value_to_f16(v); o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    _C.flash_attn_fwd(q, k, v16, o, None, D ** -0.5, False, True)
torch.cuda.synchronize()
