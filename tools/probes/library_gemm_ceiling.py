"""What the vendor library reaches on the same GEMM shapes (a ceiling reference, not the product)."""
import torch
dev = "cuda"
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in [(16384, 9216, 3072), (18432, 12288, 3072), (18432, 3072, 15360), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    us = t(lambda: torch.mm(a, w.t()))
    print(f"bf16 torch.mm       M={M} N={N} K={K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    a8 = (torch.randn(M, K, device=dev) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
    w8 = (torch.randn(N, K, device=dev) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
    one = torch.ones((), device=dev)
    try:
        us = t(lambda: torch._scaled_mm(a8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
        print(f"fp8 _scaled_mm tensorwise                  : {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    except Exception as e:
        print("fp8 tensorwise failed:", str(e)[:200])
    try:
        sa = torch.ones(M, 1, device=dev); sb = torch.ones(1, N, device=dev)
        us = t(lambda: torch._scaled_mm(a8, w8.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16))
        print(f"fp8 _scaled_mm rowwise                     : {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    except Exception as e:
        print("fp8 rowwise failed:", str(e)[:200])
