"""Times qk_norm_rope on the key slice of the Flux joint q|k|v buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
w = [torch.rand(D, device=dev) + 0.5 for _ in range(4)]
ang = torch.rand(S, D // 2, device=dev) * 6.28; cos, sin = torch.cos(ang), torch.sin(ang)
for q in (None, qkv[:, :, 0]):
    f = lambda: _C.qk_norm_rope(q, qkv[:, :, 1], w[0], w[1], w[2], w[3], 512, 1e-6, cos, sin, True)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    nb = (1 if q is None else 2) * B * S * H * D * 4
    print(f"qk_norm_rope {'k only' if q is None else 'q and k'}: {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s", flush=True)
