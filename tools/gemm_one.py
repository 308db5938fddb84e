"""One fp8 (or bf16) GEMM shape a few times: the target of rocprofv3 counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
fp8 = (sys.argv[1] if len(sys.argv) > 1 else "fp8") == "fp8"
M, N, K = 16384, 9216, 3072
dev = "cuda"
if fp8:
    a = (torch.randn(M, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
    w = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=dev) * 0.01; sb = torch.rand(N, device=dev) * 0.01
else:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16() * 0.02
    sa = sb = None
bias = torch.randn(N, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8)
torch.cuda.synchronize()
