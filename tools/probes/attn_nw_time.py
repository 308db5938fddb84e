"""Times the attention kernel on the Flux and the video-rank shapes (random data)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"
for (B, S, H) in [(4, 4608, 24), (1, 119056, 3), (1, 15106, 24)]:
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = (torch.randn(B, S, H, 128, generator=g, device=dev).bfloat16() for _ in range(3))
    vh = v.to(torch.float16)
    out = torch.empty_like(q)
    run = lambda: _C.flash_attn_fwd(q, k, vh, out, None, 128 ** -0.5, v_is_f16=True)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10 if S < 50000 else 3
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"B={B} S={S} H={H}: {us:9.1f} us  {4.0*B*S*S*128*H/us/1e6:7.1f} TF/s  chk={out.float().abs().sum().item():.6e}", flush=True)
