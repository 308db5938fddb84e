"""What the gate * y + residual epilogue costs: the gated Flux GEMMs (FF2, to_out, proj_out) timed
with and without gate / residual, alternating.  python tools/gemm_gate_cost.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

dev = "cuda"
for fp8, M, N, K in [(False, 16384, 3072, 12288), (True, 16384, 3072, 3072), (True, 18432, 3072, 15360)]:
    g = torch.Generator(device=dev).manual_seed(0)
    if fp8:
        a = (torch.randn(M, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 0.01
        sb = torch.rand(N, device=dev, generator=g) * 0.01
    else:
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        sa = sb = None
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    gate = torch.randn(4, N, device=dev, generator=g).bfloat16()
    out = torch.randn(M, N, device=dev, generator=g).bfloat16()
    plain = lambda: _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8)
    gated = lambda: _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8,
                                 gate=gate, ldg=N, gate_rows=M // 4, residual=out)
    res = {"plain": [], "gated": []}
    for rep in range(3):
        for name, fn in (("plain", plain), ("gated", gated)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{'fp8' if fp8 else 'bf16'} M={M} N={N} K={K}: plain " + " ".join(f"{u:7.1f}" for u in res["plain"]) +
          "   gate + residual " + " ".join(f"{u:7.1f}" for u in res["gated"]) +
          f"   +{min(res['gated']) - min(res['plain']):.1f} us ({(min(res['gated']) / min(res['plain']) - 1) * 100:.1f} %)",
          flush=True)
