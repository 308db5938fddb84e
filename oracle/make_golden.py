"""Mint the fixtures under tests/golden/ from the reference's own compiled CPU kernels.

Run in the build container only (needs /root/reference and an AVX512 host):
    make -C oracle ref && python oracle/make_golden.py
Each fixture stores seeded inputs and the REFERENCE kernel's output (bf16 as uint16 bit
patterns, fp8 as uint8); tests/test_oracle_pins.py checks the oracle's restatements against
them everywhere (no reference needed at test time).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402

LIB = os.path.join(ROOT, "oracle", "_ref", "libsgl_ref_cpu.so")
OUT = os.path.join(ROOT, "tests", "golden")


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    if t.dtype == torch.float8_e4m3fn:
        return t.contiguous().view(torch.uint8).numpy()
    return t.contiguous().numpy()


def main():
    torch.ops.load_library(LIB)
    R = torch.ops.sgl_ref_cpu
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731

    # attention: reference flash_attn_varlen_func (sgl-kernel/csrc/cpu/flash_attn.cpp:423)
    for name, (S, H, D) in {"attn_s200_h3_d128": (200, 3, 128), "attn_s96_h2_d64": (96, 2, 64)}.items():
        q, k, v = (rn(S, H, D).bfloat16() for _ in range(3))
        cu = torch.tensor([0, S], dtype=torch.int32)
        o = R.flash_attn_varlen_func(q, k, v, cu, cu, S, S, False)
        np.savez_compressed(os.path.join(OUT, f"ref_{name}.npz"), q=bits(q), k=bits(k), v=bits(v),
                            out=bits(o))
    # rmsnorm / fused_add_rmsnorm (csrc/cpu/norm.cpp)
    x = rn(37, 128).bfloat16()
    w = (torch.rand(128, generator=g) + 0.5).bfloat16()
    o = R.rmsnorm_cpu(x.clone(), w, 1e-6)
    xr, res = x.clone(), rn(37, 128).bfloat16()
    res0 = res.clone()
    R.fused_add_rmsnorm_cpu(xr, res, w, 1e-6)
    np.savez_compressed(os.path.join(OUT, "ref_rmsnorm.npz"), x=bits(x), w=bits(w), out=bits(o),
                        res_in=bits(res0), fa_out=bits(xr), fa_res=bits(res))
    # rotary embedding, interleaved (is_neox=False) and neox (csrc/cpu/rope.cpp)
    T, Hh, Dh = 24, 4, 128
    pos = torch.arange(T, dtype=torch.int64)
    ang = torch.rand(T, Dh // 2, generator=g) * 6.28
    cache = torch.cat([ang.cos(), ang.sin()], dim=-1).float()
    for neox in (False, True):
        q = rn(T, Hh * Dh).bfloat16()
        k = rn(T, Hh * Dh).bfloat16()
        qo, ko = R.rotary_embedding_cpu(pos, q.clone(), k.clone(), Dh, cache.bfloat16(), neox)
        np.savez_compressed(os.path.join(OUT, f"ref_rope_neox{int(neox)}.npz"), q=bits(q), k=bits(k),
                            cache=bits(cache.bfloat16()), qo=bits(qo), ko=bits(ko))
    # tanh-GELU (gelu_tanh_and_mul with a ones gate; csrc/cpu/activation.cpp)
    x = (rn(16, 256) * 2).bfloat16()
    xin = torch.cat([x, torch.ones_like(x)], dim=-1)
    o = R.gelu_tanh_and_mul_cpu(xin)
    np.savez_compressed(os.path.join(OUT, "ref_gelu_tanh.npz"), x=bits(x), out=bits(o))
    # block-fp8 weight GEMM, W8A16 (csrc/cpu/gemm_fp8.cpp:475; test/srt/cpu/test_gemm.py:176-218)
    M, N, K = 32, 256, 512
    a = (rn(M, K) * 0.5).bfloat16()
    wq, ws = ops.quantize_weight_block((rn(N, K) * 0.3).bfloat16(), 128)
    bias = rn(N).float()
    packed = R.convert_weight_packed(wq)
    o = R.fp8_scaled_mm_cpu(a, packed, ws, [128, 128], bias, torch.bfloat16, True)
    np.savez_compressed(os.path.join(OUT, "ref_fp8_block_w8a16.npz"), a=bits(a), wq=bits(wq),
                        ws=bits(ws), bias=bits(bias), out=bits(o))
    print("fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(" ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
