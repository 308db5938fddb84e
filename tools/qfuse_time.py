"""Same-process A/B: (qk_norm_rope on q and k, attention) against (qk_norm_rope on k only,
attention that normalises and rotates Q while loading it), at the Flux.1 joint-attention shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C

B, S, H, D, Tt = 4, 4608, 24, 128, 512
dev = "cuda:0"
torch.manual_seed(0)
qkv = torch.randn(B, S, 3, H, D, device=dev).bfloat16()
w = [torch.rand(D, device=dev) + 0.5 for _ in range(4)]
ang = torch.rand(S, D // 2, device=dev) * 6.28
cos, sin = torch.cos(ang), torch.sin(ang)
v16 = _C.value_to_f16(qkv[:, :, 2])
out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=dev)
scale = D ** -0.5


def a():
    _C.qk_norm_rope(qkv[:, :, 0], qkv[:, :, 1], w[0], w[1], w[2], w[3], Tt, 1e-6, cos, sin, True)
    _C.flash_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], v16, out, None, scale, False, True)


def b():
    _C.qk_norm_rope(None, qkv[:, :, 1], w[0], w[1], w[2], w[3], Tt, 1e-6, cos, sin, True)
    _C.flash_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], v16, out, None, scale, False, True,
                      q_norm=(w[0], w[2], Tt, 1e-6), q_rope=(cos, sin))


def t(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for r in range(3):
    print(f"separate {t(a):8.1f} us   fused-into-Q-load {t(b):8.1f} us", flush=True)
