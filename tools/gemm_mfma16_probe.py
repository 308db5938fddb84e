"""A/B of the GEMM kernel's MFMA shape on the Flux shapes: 32x32 (shipped) against 16x16
(MI355_GEMM_MFMA16, bit 0 bf16 / bit 1 fp8), alternating in one process, with the largest
difference between their outputs (the K sum is taken in another order, so a last-bit rounding
of the bf16 result may flip; anything larger is a bug).
python tools/gemm_mfma16_probe.py [check|time|both]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

dev = "cuda"


def operands(fp8, M, N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
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
    return a, w, sa, sb, bias


def run_one(fp8, a, w, sa, sb, bias, out, mode, **kw):
    os.environ["MI355_GEMM_MFMA16"] = str(mode)
    M, K = a.shape
    N = w.shape[0]
    _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8, **kw)


def check():
    ok = True
    cases = [(False, 300, 264, 192, {}), (True, 300, 264, 256, {}), (False, 1024, 512, 3072, {"act": 1}),
             (True, 2048, 9216, 3072, {}), (False, 16384, 3072, 1024, {}), (True, 18432, 3072, 15360, {}),
             (True, 4608, 9216, 3072, {"f16_cols": (6144, 9216)})]
    for fp8, M, N, K, kw in cases:
        a, w, sa, sb, bias = operands(fp8, M, N, K)
        o0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        o1 = torch.empty_like(o0)
        run_one(fp8, a, w, sa, sb, bias, o0, 0, **kw)
        run_one(fp8, a, w, sa, sb, bias, o1, 3, **kw)
        torch.cuda.synchronize()
        if "f16_cols" in kw:
            b, e = kw["f16_cols"]
            f0 = torch.cat([o0[:, :b].float(), o0[:, b:e].contiguous().view(torch.float16).float()], 1)
            f1 = torch.cat([o1[:, :b].float(), o1[:, b:e].contiguous().view(torch.float16).float()], 1)
        else:
            f0, f1 = o0.float(), o1.float()
        # fp32 reference of the same sum
        if fp8:
            ref = (a.float() @ w.float().t()) * sa[:, None] * sb[None, :] + bias.float()
        else:
            ref = a.float() @ w.float().t() + bias.float()
        if kw.get("act"):
            ref = torch.nn.functional.gelu(ref.bfloat16().float(), approximate="tanh")
        rel = lambda x: ((x - ref).norm() / ref.norm()).item()
        diff = (f0 - f1).abs()
        nbad = (diff > 0).sum().item()
        ulp = (diff / f0.abs().clamp_min(1e-30)).max().item()
        good = rel(f1) < 3e-3 and ulp <= 2 ** -7 * 1.01
        ok = ok and good
        print(f"{'fp8' if fp8 else 'bf16'} M={M} N={N} K={K} {kw}: rel-L2 vs fp32 32x32 {rel(f0):.2e} 16x16 {rel(f1):.2e}; "
              f"elements differing {nbad} of {f0.numel()}, largest relative difference {ulp:.2e} {'OK' if good else 'BAD'}",
              flush=True)
    return ok


def time_ab(iters=20):
    shapes = [(True, 16384, 9216, 3072), (True, 18432, 12288, 3072), (True, 18432, 3072, 15360),
              (True, 16384, 3072, 3072), (True, 2048, 9216, 3072), (False, 16384, 12288, 3072),
              (False, 16384, 3072, 12288), (False, 2048, 12288, 3072), (False, 8192, 8192, 8192)]
    for fp8, M, N, K in shapes:
        a, w, sa, sb, bias = operands(fp8, M, N, K)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {0: [], 3: []}
        for rep in range(3):
            for mode in (0, 3):
                for _ in range(5):
                    run_one(fp8, a, w, sa, sb, bias, out, mode)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run_one(fp8, a, w, sa, sb, bias, out, mode)
                e1.record()
                torch.cuda.synchronize()
                res[mode].append(e0.elapsed_time(e1) * 1e3 / iters)
        f = 2.0 * M * N * K / 1e6
        print(f"{'fp8' if fp8 else 'bf16'} M={M} N={N} K={K}: 32x32 " + " ".join(f"{u:7.1f}" for u in res[0]) +
              f" us ({f / min(res[0]):6.0f} TF/s)   16x16 " + " ".join(f"{u:7.1f}" for u in res[3]) +
              f" us ({f / min(res[3]):6.0f} TF/s)   ratio {min(res[3]) / min(res[0]):.3f}", flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else "both"
ok = True
if mode in ("check", "both"):
    ok = check()
if mode in ("time", "both") and ok:
    time_ab()
sys.exit(0 if ok else 1)
