"""Times the Flux-shaped attention call (random data).  python tools/attn_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
v16 = _C.value_to_f16(v); o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
for half in (True, False, True, False):
    vv = v16 if half else v
    for _ in range(15):
        _C.flash_attn_fwd(q, k, vv, o, None, D ** -0.5, False, half)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _C.flash_attn_fwd(q, k, vv, o, None, D ** -0.5, False, half)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"attention B={B} S={S} H={H} D={D} {'half' if half else 'bf16'} PV: {us:8.1f} us  "
          f"{4.0 * B * H * S * S * D / us / 1e6:7.1f} TF/s", flush=True)
