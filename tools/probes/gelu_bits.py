"""tanh-GELU of EVERY bf16 value, through the stand-alone activation kernel and through a GEMM
epilogue (identity weights), as raw 16-bit results in a file: run once per build of the library
(MI355_DIT_LIB) and compare the files to show that a rewrite of the formula changed no bit.

    python tools/probes/gelu_bits.py out.pt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch  # noqa: E402

from sglang_diffusion_amd import _C  # noqa: E402

dev = "cuda"
allv = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).to(dev)   # every bit pattern
x = allv.view(512, 128).contiguous()
out = torch.empty_like(x)
_C.activation(out, x, 1)
# through the epilogue of the bf16 GEMM: x @ I + 0 with tanh-GELU (finite values only: inf * 0 is NaN)
fin = torch.where(torch.isfinite(x.float()), x, torch.zeros_like(x))
eye = torch.eye(128, device=dev).bfloat16()
g = torch.empty(512, 128, device=dev, dtype=torch.bfloat16)
_C.linear_ex(fin, eye, g, M=512, N=128, K=128, lda=128, ldw=128, ldo=128, bias=torch.zeros(128, device=dev).bfloat16(), act=1)
torch.cuda.synchronize()
torch.save({"act": out.view(torch.int16).cpu(), "gemm": g.view(torch.int16).cpu()}, sys.argv[1])
print("saved", sys.argv[1])
