"""Times the linear kernels on the Flux shapes (random data).
python tools/gemm_probe.py [fp8|bf16|block|both]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

def bench(fp8, M, N, K, iters=20, block=False, **kw):
    dev = "cuda"
    if block:
        a = (torch.randn(M, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, K // 128, device=dev) * 0.01
        sb = torch.rand((N + 127) // 128, K // 128, device=dev) * 0.01
        kw["block_scale"] = True
    elif fp8:
        a = (torch.randn(M, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev) * 0.01; sb = torch.rand(N, device=dev) * 0.01
    else:
        a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16() * 0.02
        sa = sb = None
    bias = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: _C.linear_ex(a, w, out, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8, **kw)
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{'fp8-block' if block else 'fp8' if fp8 else 'bf16'} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("fp8", "both"):
    for (M, N, K) in [(16384, 9216, 3072), (18432, 12288, 3072), (18432, 3072, 15360), (8192, 8192, 8192)]:
        bench(True, M, N, K)
if mode == "block":
    for (M, N, K) in [(16384, 9216, 3072), (18432, 12288, 3072), (18432, 3072, 15360), (8192, 8192, 8192),
                      (18432, 3072, 3072), (2048, 9216, 3072)]:
        bench(True, M, N, K, block=True)
if mode == "tail":
    # same N, K with and without a partial last round of 256x256 tiles over 256 CUs
    for (M, N, K) in [(16384, 3072, 15360), (18432, 3072, 15360), (21504, 3072, 15360),
                      (16384, 9216, 3072), (18432, 9216, 3072), (2048, 9216, 3072), (1792, 9216, 3072),
                      (18432, 12288, 3072), (20480, 12288, 3072)]:
        t = (M // 256) * (N // 256)
        print(f"tiles {t} = {t / 256:.3f} rounds", flush=True)
        bench(True, M, N, K)
if mode == "txt":
    # the text-stream GEMMs of a double block (M = 4 x 512 rows) and the image-stream ones
    for (fp8, M, N, K) in [(False, 2048, 3072, 12288), (False, 2048, 12288, 3072), (True, 2048, 9216, 3072),
                           (True, 2048, 3072, 3072), (False, 16384, 3072, 12288), (True, 18432, 3072, 15360)]:
        bench(fp8, M, N, K)
if mode == "narrow":
    # N = 3072 outputs (12 tile columns): proj_out, to_out, FF2
    for (fp8, M, N, K) in [(True, 18432, 3072, 15360), (True, 16384, 3072, 3072), (False, 16384, 3072, 12288),
                           (True, 16384, 9216, 3072)]:
        bench(fp8, M, N, K)
if mode in ("bf16", "both"):
    for (M, N, K) in [(16384, 12288, 3072), (16384, 3072, 12288), (8192, 8192, 8192)]:
        bench(False, M, N, K)
if mode == "gelu":
    # what the tanh-GELU in the epilogue costs: FF1 (bf16) and proj_mlp (fp8), with and without
    for (fp8, M, N, K) in [(False, 16384, 12288, 3072), (True, 18432, 12288, 3072)]:
        for act in (0, 1):
            print(f"act={act}", end="  ")
            bench(fp8, M, N, K, act=act)
if mode == "shape":
    # K = 3072 (24 K tiles per output tile): does the time per round of 256 tiles depend on M, N?
    for M in (16384, 16640, 18432, 20480):
        for N in (8192, 9216, 9472, 12288):
            t = (M // 256) * (N // 256)
            print(f"tiles {t} = {t / 256:.3f} rounds", end="  ", flush=True)
            bench(True, M, N, 3072)
if mode == "ksweep":
    # 1024 big tiles = 4 full rounds on 256 CUs; time is linear in K: the slope is the K loop's
    # cost per 128-byte K tile, the intercept what a tile costs outside it (epilogue, seams)
    for fp8 in (True, False):
        for K in (1024, 2048, 4096, 8192):
            bench(fp8, 16384, 4096, K)
