"""Runs one hot kernel back to back for a few seconds and prints its average time, so that a
clock / power sampler running beside it (tools/probes/sample_clocks.sh) sees a steady state.
  python tools/probes/sustained_load.py attn|attn_bf16|gemm_fp8|gemm_bf16 [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

what = sys.argv[1] if len(sys.argv) > 1 else "attn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
dev = "cuda"
if what.startswith("attn"):
    B, S, H, D = 4, 4608, 24, 128
    qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    half = what == "attn"
    vv = _C.value_to_f16(v) if half else v
    o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
    call = lambda: _C.flash_attn_fwd(q, k, vv, o, None, D ** -0.5, False, half)
    flop = 4.0 * B * H * S * S * D
else:
    M, N, K = 16384, 9216, 3072
    fp8 = what == "gemm_fp8"
    if fp8:
        a = (torch.randn(M, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev) * 0.01; sb = torch.rand(N, device=dev) * 0.01
    else:
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16() * 0.02
        sa = sb = None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    call = lambda: _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, fp8=fp8)
    flop = 2.0 * M * N * K
for _ in range(5):
    call()
torch.cuda.synchronize()
print(f"{time.time():.2f} start {what}", flush=True)
t_end = time.time() + secs
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
first = last = None
while time.time() < t_end:
    e0.record()
    for _ in range(50):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    first = us if first is None else first
    last = us
    n += 50
print(f"{time.time():.2f} end {what}: {n} launches, first 50 avg {first:.1f} us, last 50 avg {last:.1f} us, "
      f"{flop / last / 1e6:.0f} TF/s", flush=True)
