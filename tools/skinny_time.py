"""Times the small-M weight-streaming linear on the modulation shape of the Flux engine."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"
for (M, N, K) in [(4, 19 * 12 * 3072 + 38 * 3 * 3072 + 2 * 3072, 3072), (1, 3072 * 6, 3072), (8, 3072, 768)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): _C.linear_bf16(x, w, b, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _C.linear_bf16(x, w, b, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"skinny M={M} N={N} K={K}: {us:8.1f} us  {2.0 * N * K / us / 1e6:6.2f} TB/s  sum={float(out.float().sum()):.3f}", flush=True)
