"""Per-rank work of one HunyuanVideo 720p x 129-frame denoise step under 8-way sequence
parallelism, timed on ONE GPU (the collectives are not part of it): a rank holds 118800 / 8
video tokens for the linears and, after the Ulysses exchange, all tokens with 24 / 8 heads for
attention.  python tools/hunyuan_rank_shape.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))
import torch
from sglang_diffusion_amd import _C
from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.hunyuanvideo import (MMDoubleStreamBlock, MMSingleStreamBlock,
                                                      process_weights_after_loading)
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

dev = torch.device("cuda")
HID, HEADS, P = 3072, 24, 8
TI, TT = 33 * 45 * 80, 256          # (129-1)/4+1 latent frames x 720/16 x 1280/16; text tokens
ti = TI // P


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# attention after the exchange: every token, 3 heads
S = TI + TT
qkv = torch.randn(1, S, 3, HEADS // P, 128, device=dev).bfloat16()
o = torch.empty(1, S, HEADS // P, 128, device=dev, dtype=torch.bfloat16)
ms = timed(lambda: _C.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, None, 128 ** -0.5), 3)
print(f"attention S={S} H={HEADS // P}: {ms:.2f} ms  {4.0 * S * S * (HEADS // P) * 128 / ms / 1e9:.0f} TF/s")

for fp8 in (False, True):
    qc = Fp8Config(is_checkpoint_fp8_serialized=False) if fp8 else None
    for cls, name in ((MMDoubleStreamBlock, "double"), (MMSingleStreamBlock, "single")):
        with torch.device(dev):
            blk = cls(HID, HEADS, 4.0, dtype=torch.bfloat16, prefix="b", quant_config=qc)
        init_synthetic_weights(blk, seed=0, host_rng=False)
        process_weights_after_loading(blk)
        img = torch.randn(1, ti, HID, device=dev).bfloat16()
        txt = torch.randn(1, TT, HID, device=dev).bfloat16()
        vec = torch.randn(1, HID, device=dev).bfloat16()
        cs = (torch.ones(ti, 64, device=dev), torch.zeros(ti, 64, device=dev))
        with torch.no_grad():
            if cls is MMDoubleStreamBlock:
                ms = timed(lambda: blk(img, txt, vec, cs))
            else:
                x = torch.cat([img, txt], 1)
                ms = timed(lambda: blk(x, vec, TT, cs))
        print(f"{name} block, {ti}+{TT} tokens, {'fp8' if fp8 else 'bf16'} (local attention over the "
              f"rank's own tokens, all heads): {ms:.2f} ms")
        del blk
