"""What one workgroup of the attention kernel costs outside its key loop: the Flux-shaped launch
(B=4, H=24, 4608 query rows = 1728 workgroups on 256 CUs) with 64, 128, 256 and 4608 keys.  The
intercept of time over key tiles is the per-launch cost of prologues (Q load with the fused norm
and rotation when given), first-tile wait, epilogues and workgroup turnover.
python tools/attn_seam_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
dev = "cuda"; B, S, H, D = 4, 4608, 24, 128
q = torch.randn(B, S, H, D, device=dev).bfloat16()
o = torch.empty(B, S, H, D, device=dev, dtype=torch.bfloat16)
w0 = torch.rand(D, device=dev) + 0.5
ang = torch.rand(S, D // 2, device=dev) * 6.28
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
for fused in (False, True):
    for Sk in (64, 128, 256, 1024, 4608):
        k = torch.randn(B, Sk, H, D, device=dev).bfloat16()
        v = _C.value_to_f16(torch.randn(B, Sk, H, D, device=dev).bfloat16())
        kw = dict(q_norm=(w0, w0, 512, 1e-6), q_rope=(cos, sin)) if fused else {}
        run = lambda: _C.flash_attn_fwd(q, k, v, o, None, D ** -0.5, False, True, **kw)
        for _ in range(10): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"query transform {'fused' if fused else 'none '}  keys {Sk:5d} ({Sk // 64:3d} tiles): {us:8.1f} us", flush=True)
