"""Times LayerNorm + modulation (+ per-token fp8 quantisation) on the Flux joint sequence."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, T, D = 4, 4608, 3072
x = torch.randn(B * T, D, device=dev).bfloat16()
sh = torch.randn(B, D, device=dev).bfloat16(); sc = torch.randn(B, D, device=dev).bfloat16()
q = torch.empty(B * T, D, device=dev, dtype=torch.float8_e4m3fn); s = torch.empty(B * T, device=dev)
o = torch.empty_like(x)
def t(f, nbytes, name):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name}: {us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s  chk={float(o.float().abs().sum()) + float(s.sum()):.4e}", flush=True)
t(lambda: _C.layernorm_modulate_quant(q, s, x, sh, sc, B, T, 1e-6), 3 * B * T * D, "LN + modulate + quant")
t(lambda: _C.layernorm_modulate(o, x, sh, sc, B, T, 1e-6), 4 * B * T * D, "LN + modulate        ")
