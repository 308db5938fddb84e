"""A/B of one environment switch of the GEMM kernels on the Flux shapes, alternating in one
process; the outputs of both settings are compared bit for bit first.
python tools/gemm_env_ab.py VAR [A_VALUE B_VALUE [block]]     ('-' = unset; default: unset against 1)
With `block` the block-scaled fp8 form runs instead (scales per token x 128 K and per 128 x 128 of
the weight); there the two settings may sum a block's 128 products in another order, so the
outputs are compared to one bf16 rounding step instead of bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

dev = "cuda"
var = sys.argv[1]
vals = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("-", "1")
block = len(sys.argv) > 4 and sys.argv[4] == "block"


def setv(v):
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v


def operands(fp8, M, N, K):
    g = torch.Generator(device=dev).manual_seed(0)
    if fp8 and block:
        a = (torch.randn(M, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, K // 128, device=dev, generator=g) * 0.01
        sb = torch.rand((N + 127) // 128, K // 128, device=dev, generator=g) * 0.01
    elif fp8:
        a = (torch.randn(M, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev, generator=g) * 0.01
        sb = torch.rand(N, device=dev, generator=g) * 0.01
    else:
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        sa = sb = None
    return a, w, sa, sb, torch.randn(N, device=dev, generator=g).bfloat16()


shapes = [(True, 16384, 9216, 3072), (True, 18432, 12288, 3072), (True, 18432, 3072, 15360),
          (True, 16384, 3072, 3072), (True, 2048, 9216, 3072), (True, 300, 264, 256), (False, 16384, 12288, 3072),
          (False, 16384, 3072, 12288), (False, 2048, 12288, 3072), (False, 1000, 520, 192)]
if block:
    shapes = [s for s in shapes if s[0]] + [(True, 130, 384, 1024)]
bad = 0
for fp8, M, N, K in shapes:
    a, w, sa, sb, bias = operands(fp8, M, N, K)
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    call = lambda o: _C.linear_ex(a, w, o, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8,
                                  block_scale=block)
    for i in range(2):
        setv(vals[i])
        call(outs[i])
    torch.cuda.synchronize()
    ndiff = (outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum().item()
    if block:
        f0, f1 = outs[0].float(), outs[1].float()
        step = ((f0 - f1).abs() / f0.abs().clamp_min(1e-30)).max().item()
        bad += 0 if step <= 2 ** -7 * 1.01 and bool(torch.isfinite(f1).all()) else 1
    else:
        bad += ndiff
    res = [[], []]
    for rep in range(3):
        for i in range(2):
            setv(vals[i])
            for _ in range(5):
                call(outs[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call(outs[i])
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 1e3 / 20)
    f = 2.0 * M * N * K / 1e6
    print(f"{'fp8' if fp8 else 'bf16'} M={M} N={N} K={K}: {var}={vals[0]} " + " ".join(f"{u:7.1f}" for u in res[0]) +
          f" us ({f / min(res[0]):6.0f} TF/s)   {var}={vals[1]} " + " ".join(f"{u:7.1f}" for u in res[1]) +
          f" us ({f / min(res[1]):6.0f} TF/s)   ratio {min(res[1]) / min(res[0]):.3f}   differing elements {ndiff}",
          flush=True)
sys.exit(1 if bad else 0)
