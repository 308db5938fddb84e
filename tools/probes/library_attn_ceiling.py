"""torch SDPA (the vendor flash-attention path on ROCm) on the Flux attention shape: a ceiling reference."""
import torch, torch.nn.functional as F
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
q = torch.randn(B, H, S, D, device=dev).bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q)
def t(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(be):
            us = t(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"{be}: {us:8.1f} us {4.0*B*S*S*H*D/us/1e6:7.1f} TF/s", flush=True)
    except Exception as e:
        print(be, "failed:", str(e)[:200])
