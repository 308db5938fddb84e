"""Shapes, inputs and per-block parity bars of the HunyuanVideo blocks at their real width,
shared by tests/test_hunyuan_real_width_gpu.py (HIP path against the oracle) and
tests/test_hunyuan_oracle_sensitivity.py (the oracle against itself: the floor the bars are
read against).  Keys: (block kind "d" double / "s" single, W8A8?)."""
import torch

HID, HEADS, TI, TT = 3072, 24, 1536, 128

# measured on MI355X: double bf16 1.77e-3, single bf16 5.4e-4, double W8A8 8.6-8.8e-3, single W8A8 1.3e-3;
# floor (oracle re-evaluated with another GEMM order and half-precision probabilities): 1.75e-3, 3.6e-4,
# 6.2-6.8e-3, 5.4e-4
BAR = {("d", False): 2e-3, ("s", False): 7e-4, ("d", True): 1e-2, ("s", True): 1.6e-3}


def block_inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, TI, HID, generator=g).bfloat16()
    txt = torch.randn(1, TT, HID, generator=g).bfloat16()
    vec = torch.randn(1, HID, generator=g).bfloat16()
    ang = torch.rand(TI, HID // HEADS // 2, generator=g) * 6.28
    return img, txt, vec, torch.cos(ang), torch.sin(ang)
