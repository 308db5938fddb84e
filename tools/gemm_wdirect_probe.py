"""A/B of the linear kernels with the weight through LDS (w_shuffled=None) and straight into
registers (w_shuffled given), same process, interleaved rounds; checks bit-identity first.
python tools/gemm_wdirect_probe.py [fp8|bf16|group]"""
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


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def ab(fp8, M, N, K, iters=10, rounds=5, **kw):
    a, w, sa, sb, bias = operands(fp8, M, N, K)
    wsh = _C.shuffle_weight(w)
    o0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    o1 = torch.empty_like(o0)
    base = dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8, **kw)
    run0 = lambda: _C.linear_ex(a, w, o0, **base)
    run1 = lambda: _C.linear_ex(a, w, o1, w_shuffled=wsh, **base)
    run0(); run1(); torch.cuda.synchronize()
    same = torch.equal(o0, o1)
    t0, t1 = [], []
    for _ in range(rounds):
        t0.append(timed(run0, iters)); t1.append(timed(run1, iters))
    f = 2.0 * M * N * K / 1e6
    m0, m1 = sorted(t0)[len(t0) // 2], sorted(t1)[len(t1) // 2]
    print(f"{'fp8 ' if fp8 else 'bf16'} M={M} N={N} K={K} {kw or ''}: LDS weight {m0:7.1f} us {f / m0:6.0f} TF | "
          f"register weight {m1:7.1f} us {f / m1:6.0f} TF | {m0 / m1:5.3f}x | bit-identical {same}", flush=True)
    return same


mode = sys.argv[1] if len(sys.argv) > 1 else "fp8"
ok = True
if mode in ("fp8", "all"):
    for shape in [(16384, 9216, 3072), (18432, 12288, 3072), (18432, 3072, 15360), (16384, 3072, 3072),
                  (8192, 8192, 8192), (2000, 9216, 3072)]:
        ok &= ab(True, *shape)
    ok &= ab(True, 18432, 12288, 3072, act=1)
if mode in ("bf16", "all"):
    for shape in [(16384, 12288, 3072), (16384, 3072, 12288), (8192, 8192, 8192)]:
        ok &= ab(False, *shape)
if mode in ("group", "all"):
    # text + image problems of a double block in one launch vs two launches
    for fp8, N, K in ((True, 9216, 3072), (True, 3072, 3072), (False, 12288, 3072), (False, 3072, 12288)):
        A = [operands(fp8, M, N, K, seed=i) for i, M in enumerate((16384, 2048))]
        sh = [_C.shuffle_weight(x[1]) for x in A]
        outs_sep = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for M in (16384, 2048)]
        outs_grp = [torch.empty_like(o) for o in outs_sep]
        def kw(i, M):
            a, w, sa, sb, bias = A[i]
            return dict(M=M, N=N, K=K, lda=K, ldw=K, ldo=N, sa=sa, sb=sb, bias=bias, fp8=fp8, w_shuffled=sh[i])
        def sep_old():
            for i, M in enumerate((16384, 2048)):
                k = kw(i, M); k.pop("w_shuffled")
                _C.linear_ex(A[i][0], A[i][1], outs_sep[i], **k)
        def grp():
            _C.linear_grouped([(A[i][0], A[i][1], outs_grp[i], kw(i, M)) for i, M in enumerate((16384, 2048))])
        sep_old(); grp(); torch.cuda.synchronize()
        same = all(torch.equal(x, y) for x, y in zip(outs_sep, outs_grp))
        ok &= same
        ts, tg = [], []
        for _ in range(5):
            ts.append(timed(sep_old, 10)); tg.append(timed(grp, 10))
        ms, mg = sorted(ts)[2], sorted(tg)[2]
        print(f"{'fp8 ' if fp8 else 'bf16'} img 16384 + txt 2048 rows, N={N} K={K}: two launches (LDS weight) {ms:7.1f} us | "
              f"one grouped launch {mg:7.1f} us | {ms / mg:5.3f}x | bit-identical {same}", flush=True)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
