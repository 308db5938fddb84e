"""layernorm_modulate(_quant): the whole-chunk kernel against the general one (MI355_LN_GENERAL=1 in a
second process): prints times and a hash of every output so the two runs can be compared for bit equality."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"
def h(t): return hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]
for (B, T, D, maps) in [(4, 4608, 3072, False), (4, 4096, 3072, True), (3, 333, 3072, False), (2, 100, 2048, False), (2, 77, 1024, True)]:
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    S = T + 64
    xfull = (torch.randn(B * S, D, generator=g, device=dev) * 3 + 0.5).bfloat16()
    sh = torch.randn(B, D, generator=g, device=dev).bfloat16(); sc = torch.randn(B, D, generator=g, device=dev).bfloat16()
    x_map = (T, S, 64) if maps else None
    x = xfull if maps else xfull[: B * T]
    q = torch.empty(B * T, D, device=dev, dtype=torch.float8_e4m3fn); s = torch.empty(B * T, device=dev)
    o = torch.empty(B * T, D, device=dev, dtype=torch.bfloat16)
    _C.layernorm_modulate_quant(q, s, x, sh, sc, B, T, 1e-6, x_map=x_map)
    _C.layernorm_modulate(o, x, sh, sc, B, T, 1e-6, x_map=x_map)
    torch.cuda.synchronize()
    line = f"B={B} T={T} D={D} maps={maps}: q {h(q)} s {h(s)} o {h(o)}"
    if T >= 4096:
        for name, f in (("quant", lambda: _C.layernorm_modulate_quant(q, s, x, sh, sc, B, T, 1e-6, x_map=x_map)),
                        ("bf16", lambda: _C.layernorm_modulate(o, x, sh, sc, B, T, 1e-6, x_map=x_map))):
            for _ in range(10): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            line += f"  {name} {e0.elapsed_time(e1) * 20:.1f} us"
    print(line, flush=True)
