"""Quick on-GPU sanity probe of the raw kernels against inline torch formulas.

Not collected by pytest (no test_ prefix); run as `python tests/gpu_probe.py`.
Prints one line per kernel with the relative error and a rough timing.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "sglang-diffusion_amd"))
from sglang_diffusion_amd import _C  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def probe_quant():
    M, K = 1000, 3072
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 3
    q = torch.empty(M, K, device=dev, dtype=torch.float8_e4m3fn)
    s = torch.empty(M, device=dev, dtype=torch.float32)
    _C.per_token_quant_fp8(x, q, s)
    xf = x.float()
    sr = xf.abs().amax(-1) / 448.0
    qr = (xf * (1.0 / sr)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    print("per_token_quant: scale max abs diff", (s - sr).abs().max().item(),
          "q mismatches", (q.view(torch.uint8) != qr.view(torch.uint8)).sum().item(), "of", q.numel())
    M, K = 16384, 3072
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    q = torch.empty(M, K, device=dev, dtype=torch.float8_e4m3fn)
    s = torch.empty(M, device=dev, dtype=torch.float32)
    t = timeit(lambda: _C.per_token_quant_fp8(x, q, s))
    print(f"  16384x3072: {t*1e6:.1f} us, {M*K*3/t/1e9:.0f} GB/s")


def probe_gemm(fp8, M, N, K, act=0):
    if fp8:
        a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e4m3fn)
        w = (torch.randn(N, K, device=dev) * 2).to(torch.float8_e4m3fn)
        sa = torch.rand(M, device=dev) * 0.01 + 0.001
        sb = torch.rand(N, device=dev) * 0.01 + 0.001
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        _C.fp8_scaled_mm(a, w, sa, sb, bias, out)
        ref = (a.float() @ w.float().t()) * sa[:, None] * sb[None, :] + bias.float()
        e = rel(out, ref)
        t = timeit(lambda: _C.fp8_scaled_mm(a, w, sa, sb, bias, out))
    else:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        _C.linear_bf16(a, w, bias, out, act)
        ref = a.float() @ w.float().t() + bias.float()
        if act:
            ref = torch.nn.functional.gelu(ref.bfloat16().float(), approximate="tanh")
        e = rel(out, ref)
        t = timeit(lambda: _C.linear_bf16(a, w, bias, out, act))
    print(f"gemm fp8={fp8} act={act} M={M} N={N} K={K}: rel err {e:.2e}  {t*1e6:.1f} us  "
          f"{2*M*N*K/t/1e12:.1f} TFLOP/s")


def probe_attn(B, S, H, D):
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    scale = D ** -0.5
    _C.flash_attn_fwd(q, k, v, out, lse, scale)
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    sc = (qf @ kf.transpose(-1, -2)) * scale
    ref = (torch.softmax(sc, -1) @ vf).transpose(1, 2)
    lref = torch.logsumexp(sc, -1)
    e = rel(out, ref)
    t = timeit(lambda: _C.flash_attn_fwd(q, k, v, out, lse, scale), n=5)
    print(f"attn B={B} S={S} H={H} D={D}: rel err {e:.2e}  lse max diff {(lse-lref).abs().max().item():.2e} "
          f"{t*1e6:.1f} us  {4*B*H*S*S*D/t/1e12:.1f} TFLOP/s")


def probe_elementwise():
    B, T, D = 2, 300, 3072
    x = torch.randn(B * T, D, device=dev, dtype=torch.bfloat16)
    shift = torch.randn(B, D, device=dev, dtype=torch.bfloat16)
    scale = torch.randn(B, D, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(x)
    _C.layernorm_modulate(out, x, shift, scale, B, T, 1e-6)
    ln = torch.nn.functional.layer_norm(x.view(B, T, D), (D,), eps=1e-6)
    ref = ln * (1 + scale[:, None]) + shift[:, None]
    print("layernorm_modulate rel err", rel(out.view(B, T, D), ref),
          "mismatch", (out.view(B, T, D) != ref).sum().item(), "of", out.numel())
    gate = torch.randn(B, D, device=dev, dtype=torch.bfloat16)
    res = torch.randn(B * T, D, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty_like(x)
    _C.gate_residual(o2, res, gate, x, B, T)
    ref2 = res.view(B, T, D) + gate[:, None] * x.view(B, T, D)
    print("gate_residual mismatch", (o2.view(B, T, D) != ref2).sum().item())
    # qk norm + rope
    Bq, S, H, Dh = 2, 77, 24, 128
    q = torch.randn(Bq, S, H, Dh, device=dev, dtype=torch.bfloat16)
    k = torch.randn(Bq, S, H, Dh, device=dev, dtype=torch.bfloat16)
    wq = torch.rand(Dh, device=dev) + 0.5
    wk = torch.rand(Dh, device=dev) + 0.5
    ang = torch.rand(S, Dh // 2, device=dev) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()

    def ref_norm_rope(x, w):
        xf = x.float()
        y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w).to(torch.bfloat16)
        yf = y.float()
        x1, x2 = yf[..., 0::2], yf[..., 1::2]
        c, s_ = cos[None, :, None, :], sin[None, :, None, :]
        o1 = x1 * c - x2 * s_
        o2_ = x2 * c + x1 * s_
        return torch.stack([o1, o2_], -1).flatten(-2).to(torch.bfloat16)

    rq, rk = ref_norm_rope(q, wq), ref_norm_rope(k, wk)
    _C.qk_norm_rope(q, k, wq, wk, wq, wk, 0, 1e-6, cos, sin, True)
    print("qk_norm_rope rel err", rel(q, rq), rel(k, rk), "mismatch", (q != rq).sum().item(), "of", q.numel())
    # euler + cfg
    n = 4 * 4096 * 64
    lat = torch.randn(n, device=dev, dtype=torch.bfloat16)
    vv = torch.randn(n, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(lat)
    sig, sign = 0.9, 0.85
    _C.euler_step(o, lat, vv, sig, sign)
    dt = torch.tensor(sign, device=dev) - torch.tensor(sig, device=dev)
    ref = (lat.float() + dt * vv).to(torch.bfloat16)
    print("euler mismatch", (o != ref).sum().item())
    _C.cfg_combine(o, lat, vv, 3.5)
    ref = vv + 3.5 * (lat - vv)
    print("cfg mismatch", (o != ref).sum().item())
    # timestep embedding
    t = torch.tensor([999.0, 500.5, 1.0, 3504.0], device=dev)
    emb = torch.empty(4, 256, device=dev, dtype=torch.float32)
    _C.timestep_embedding(t, emb, True, 0.0, 1.0, 10000.0)
    half = 128
    ex = -torch.log(torch.tensor(10000.0, device=dev)) * torch.arange(half, device=dev, dtype=torch.float32) / half
    e = t[:, None] * torch.exp(ex)[None]
    ref = torch.cat([torch.cos(e), torch.sin(e)], -1)
    print("timestep_embedding max abs diff", (emb - ref).abs().max().item())


if __name__ == "__main__":
    which = sys.argv[1:] or ["quant", "gemm", "attn", "elem"]
    if "quant" in which:
        probe_quant()
    if "gemm" in which:
        probe_gemm(True, 256, 256, 512)
        probe_gemm(True, 300, 136, 1024)
        probe_gemm(False, 256, 256, 512)
        probe_gemm(False, 300, 136, 1024, act=1)
        probe_gemm(False, 4, 9216, 3072)
        probe_gemm(True, 16384, 3072, 3072)
        probe_gemm(True, 16384, 12288, 3072)
        probe_gemm(True, 18432, 3072, 15360)
        probe_gemm(True, 2048, 3072, 3072)
        probe_gemm(False, 16384, 12288, 3072, act=1)
        probe_gemm(False, 16384, 3072, 12288)
    if "attn" in which:
        probe_attn(1, 128, 2, 128)
        probe_attn(1, 333, 3, 128)
        probe_attn(1, 200, 2, 64)
        probe_attn(2, 4608, 24, 128)
    if "elem" in which:
        probe_elementwise()
