"""Second probe: fp8 conversion corner cases and ds_read_b64_tr_b16 semantics."""
import os, sys, subprocess, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sglang-diffusion_amd"))
from sglang_diffusion_amd import _C
dev = torch.device("cuda:0")

# ---- fp8 conversion: one row, max element 448 so scale == 1 -------------------
K = 8192
vals = torch.linspace(-0.07, 0.07, K - 8, device=dev)
x = torch.cat([vals, torch.tensor([448.0, 1.0, 0.001953125, 0.0009765625, 0.0029296875, 17.0, 19.0, 21.0], device=dev)]).bfloat16().view(1, K)
q = torch.empty(1, K, device=dev, dtype=torch.float8_e4m3fn)
s = torch.empty(1, device=dev)
_C.per_token_quant_fp8(x, q, s)
ref = x.float().clamp(-448, 448).to(torch.float8_e4m3fn)
mis = (q.view(torch.uint8) != ref.view(torch.uint8)).nonzero()
print("scale", s.item(), "mismatches", mis.shape[0])
for idx in mis[:12]:
    i = idx[1].item()
    print("  x=%r got=%r ref=%r" % (x[0, i].item(), q[0, i].float().item(), ref[0, i].float().item()))

# ---- random data: classify mismatches ------------------------------------------
torch.manual_seed(0)
M, K = 1000, 3072
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 3
q = torch.empty(M, K, device=dev, dtype=torch.float8_e4m3fn)
s = torch.empty(M, device=dev)
_C.per_token_quant_fp8(x, q, s)
xf = x.float()
sr = xf.abs().amax(-1) / 448.0
print("rows with scale diff", (s != sr).sum().item())
inv = torch.where(s == 0, torch.zeros_like(s), 1.0 / s)
qr = (xf * inv[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
mm = (q.view(torch.uint8) != qr.view(torch.uint8))
print("mismatch using kernel scale:", mm.sum().item())
ii = mm.nonzero()[:8]
for r, c in ii.tolist():
    print("  x*inv=%r got=%r ref=%r" % ((xf[r, c] * inv[r]).item(), q[r, c].float().item(), qr[r, c].float().item()))
