"""Times the per-token fp8 quantiser on the Flux shapes.  python tools/quant_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"
for (M, K) in [(18432, 15360), (18432, 3072), (16384, 3072)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    q = torch.empty(M, K, device=dev, dtype=torch.float8_e4m3fn); s = torch.empty(M, device=dev)
    for _ in range(10): _C.per_token_quant_fp8(x, q, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): _C.per_token_quant_fp8(x, q, s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"per_token_quant_fp8 M={M} K={K}: {us:7.1f} us  {3.0 * M * K / us / 1e6:6.2f} TB/s", flush=True)
