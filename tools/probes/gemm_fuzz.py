"""Random shapes through mi355_linear_ex: every tile class (64 / 128 / 256-row tiles, the ring,
its half-tile tail), ragged M and N, joint-buffer row maps with random batch geometry, and the
four epilogue feature sets (plain, GELU, half-format columns, gate + residual), fp8 and bf16.
Each case is compared with an fp32 torch rendering on the GPU (relative L2 <= 2e-3: this is
a search for indexing errors, not a rounding study) and the output buffer carries sentinel rows
in every gap the kernel must not touch.

    python tools/probes/gemm_fuzz.py [cases] [seed]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sglang_diffusion_amd import _C  # noqa: E402

dev = "cuda"
FP8 = torch.float8_e4m3fn
SENT = 12345.0


def one(rng, idx):
    fp8 = rng.random() < 0.6
    big = rng.random() < 0.4
    B = rng.choice([1, 1, 2, 3, 4])
    rows = rng.randint(1, 2600) if big else rng.randint(1, 400)       # rows of a batch the GEMM sees
    N = 8 * (rng.randint(100, 500) if big else rng.randint(1, 80))
    K = 128 * rng.randint(1, 6) if fp8 else 64 * rng.randint(1, 8)
    mode = rng.choice(["plain", "act", "f16", "gate", "act+f16"])
    mapped = rng.random() < 0.6
    gap = rng.randint(0, 70) if mapped else 0                           # rows of a batch the GEMM skips
    off = rng.randint(0, gap) if mapped else 0
    brows = rows + gap
    M = B * rows
    g = torch.Generator(device=dev).manual_seed(idx)
    x = torch.randn(B * brows, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    bias = torch.randn(N, device=dev, generator=g).bfloat16()
    if fp8:
        xq = (x * 4).clamp(-448, 448).to(FP8)
        wq = (w * 16).clamp(-448, 448).to(FP8)
        sa = torch.rand(B * brows, device=dev, generator=g) * 0.2 + 0.05
        sb = torch.rand(N, device=dev, generator=g) * 0.05 + 0.01
        ref = (xq.float() @ wq.float().t()) * sa[:, None] * sb[None, :] + bias.float()
        a_op, w_op = xq, wq
    else:
        xb, wb = x.bfloat16(), w.bfloat16()
        sa = sb = None
        ref = xb.float() @ wb.float().t() + bias.float()
        a_op, w_op = xb, wb
    ref = ref.bfloat16().float()
    kw = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8)
    if mapped:
        kw.update(a_map=(rows, brows, off), o_map=(rows, brows, off))
    else:
        assert brows == rows
    out = torch.full((B * brows, N), SENT, device=dev, dtype=torch.bfloat16)
    res = None
    f16 = None
    if "act" in mode:
        kw.update(act=1)
        ref = F.gelu(ref, approximate="tanh").bfloat16().float()
    if "f16" in mode and N >= 64 and (fp8 or M > 16):
        b32 = 32 * rng.randint(0, N // 32 - 1)
        e32 = 32 * rng.randint(b32 // 32 + 1, N // 32)
        f16 = (b32, e32)
        kw.update(f16_cols=f16)
    if mode == "gate":
        gate = torch.randn(B, N, device=dev, generator=g).bfloat16()
        res = torch.randn(B * brows, N, device=dev, generator=g).bfloat16()
        out = res.clone()
        gate_rows = rows
        kw.update(residual=out, gate=gate, ldg=gate.stride(0), gate_rows=gate_rows)
    _C.linear_ex(a_op, w_op, out, **kw)
    torch.cuda.synchronize()
    sel = torch.zeros(B * brows, dtype=torch.bool, device=dev)
    for b in range(B):
        sel[b * brows + off:b * brows + off + rows] = True
    got = out.float()
    if f16 is not None:      # the rows the GEMM wrote hold IEEE half bits in these columns
        half = out[:, f16[0]:f16[1]].contiguous().view(torch.float16).float()
        got[:, f16[0]:f16[1]] = torch.where(sel[:, None], half, got[:, f16[0]:f16[1]])
        _C.f16_range_status(reset=True)
    want = ref[sel] if mapped else ref
    if mode == "gate":
        gsel = gate.float().repeat_interleave(rows, dim=0)
        want = res.float()[sel] + (gsel * want).bfloat16().float()
        want = want.bfloat16().float()
    err = ((got[sel] - want).norm() / want.norm().clamp_min(1e-20)).item()
    untouched = got[~sel]
    keep = res.float()[~sel] if mode == "gate" else torch.full_like(untouched, SENT)
    clean = torch.equal(untouched, keep.bfloat16().float())
    tag = f"{'fp8 ' if fp8 else 'bf16'} M={M:5d} (B={B} rows={rows} gap={gap} off={off}) N={N:5d} K={K:4d} {mode:8s} f16={f16}"
    ok = err <= 2e-3 and clean and bool(torch.isfinite(got[sel]).all())
    print(f"{'ok  ' if ok else 'FAIL'} {tag}: rel {err:.2e}{'' if clean else '  ROWS OUTSIDE THE MAP TOUCHED'}", flush=True)
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = sum(0 if one(rng, i) else 1 for i in range(cases))
    print(f"{cases - bad} of {cases} cases ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
