"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def fp8(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.copy()).view(torch.float8_e4m3fn)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def mismatch_frac(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype == torch.float8_e4m3fn:
        a, b = a.view(torch.uint8), b.view(torch.uint8)
    return (a != b).float().mean().item()


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors."""
    def key(t):
        i = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a) - key(b)).abs()


def assert_close_bf16(out: torch.Tensor, ref: torch.Tensor, rel: float = 1e-3, max_ulps: float = 2.0,
                      what: str = ""):
    """Two bf16 tensors computed with the same rounding points but a different fp32 operation
    order: relative L2 within `rel`, and no element further apart than `max_ulps` bf16 ulps of
    the LARGER of its own magnitude and the tensor's RMS (so cancellation near zero, where a
    ulp is tiny, is judged on the scale of the operands that produced it)."""
    o, r = out.detach().float().cpu(), ref.detach().float().cpu()
    e = rel_l2(o, r)
    assert e <= rel, f"{what}: relative L2 {e:.3e} > {rel:.1e}"
    rms = r.pow(2).mean().sqrt().item()
    scale = torch.maximum(r.abs(), torch.tensor(rms))
    bad = ((o - r).abs() > max_ulps * 2.0 ** -7 * scale)
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond {max_ulps} bf16 ulps"
