"""The HIP kernels DIRECTLY against outputs of the reference's own compiled kernels.

tests/golden/ref_*.npz hold seeded inputs and what the reference's CPU kernels
(sgl-kernel/csrc/cpu/{flash_attn,norm,rope,activation,gemm_fp8}.cpp, compiled in place by
oracle/Makefile, run by oracle/make_golden.py) returned for them.  tests/test_oracle_pins.py pins
the ORACLE with these files; here the same stored inputs go through the C ABI on the GPU and the
results are held to the reference's own test tolerance for each kernel --
`torch.testing.assert_close(ref, out, atol=1e-2, rtol=1e-2)` for bf16
(test/srt/cpu/utils.py:7-11, used by test_flash_attn.py:110-111, test_norm.py:79-80,
test_rope.py:140-143, test_activation.py:26-27, test_gemm.py:217-218) -- and to a tighter
relative-L2 figure of our own beside it.
"""
import pytest
import torch

from sglang_diffusion_amd import _C
from tests.helpers import assert_close_bf16, bf16, bf16_ulp_diff, fp8, load_golden, rel_l2

pytestmark = pytest.mark.gpu

REF_TOL = dict(atol=1e-2, rtol=1e-2)   # precision[torch.bfloat16] of the reference's CPU tests


def _close(out, ref):
    torch.testing.assert_close(out.float().cpu(), ref.float(), **REF_TOL)


@pytest.mark.parametrize("half_v", [False, True])
@pytest.mark.parametrize("name,D", [("ref_attn_s200_h3_d128.npz", 128), ("ref_attn_s96_h2_d64.npz", 64)])
def test_attention_vs_reference_flash_attn_varlen_func(device, name, D, half_v):
    """flash_attn_varlen_func (sgl-kernel/csrc/cpu/flash_attn.cpp:423) on one sequence, non
    causal.  Both forms of the HIP kernel: bf16 V with bf16 probabilities (what the reference
    kernel itself does) and half V with half probabilities (what the DiT path runs)."""
    z = load_golden(name)
    q, k, v, ref = (bf16(z[n]) for n in ("q", "k", "v", "out"))
    qd, kd, vd = (t[None].to(device) for t in (q, k, v))
    out = torch.full_like(qd, float("nan"))
    vv = _C.value_to_f16(vd) if half_v else vd
    _C.flash_attn_fwd(qd, kd, vv, out, None, D ** -0.5, v_is_f16=half_v)
    torch.cuda.synchronize()
    _close(out[0], ref)
    e = rel_l2(out[0], ref)
    print(f"{name} half_v={half_v}: rel-L2 vs the reference kernel's output {e:.2e}")
    # two bf16-probability kernels differ in the rounding of P: a few 1e-3
    assert e <= 4e-3


def test_rmsnorm_vs_reference_rmsnorm_cpu(device):
    """rmsnorm_cpu / fused_add_rmsnorm_cpu (sgl-kernel/csrc/cpu/norm.cpp), bf16 weight."""
    z = load_golden("ref_rmsnorm.npz")
    x, w = bf16(z["x"]).to(device), bf16(z["w"]).to(device)
    out = torch.empty_like(x)
    _C.rmsnorm(out, x, w, 1e-6)
    _close(out, bf16(z["out"]))
    assert bf16_ulp_diff(out, bf16(z["out"])).max() <= 1
    xr, res = x.clone(), bf16(z["res_in"]).to(device)
    _C.fused_add_rmsnorm(xr, res, w, 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(res.cpu(), bf16(z["fa_res"]))        # the bf16 sum: same bits
    _close(xr, bf16(z["fa_out"]))
    assert bf16_ulp_diff(xr, bf16(z["fa_out"])).max() <= 1


@pytest.mark.parametrize("neox", [0, 1])
def test_rotary_embedding_vs_reference_rotary_embedding_cpu(device, neox):
    """rotary_embedding_cpu (sgl-kernel/csrc/cpu/rope.cpp) with a bf16 cos|sin cache."""
    z = load_golden(f"ref_rope_neox{neox}.npz")
    q, k, cache = (bf16(z[n]).to(device) for n in ("q", "k", "cache"))
    T = q.shape[0]
    pos = torch.arange(T, dtype=torch.int64, device=device)
    _C.rotary_embedding(pos, q, k, 128, cache, bool(neox))
    torch.cuda.synchronize()
    _close(q, bf16(z["qo"]))
    _close(k, bf16(z["ko"]))
    assert bf16_ulp_diff(q, bf16(z["qo"])).max() <= 1
    assert bf16_ulp_diff(k, bf16(z["ko"])).max() <= 1


def test_gelu_tanh_vs_reference_gelu_tanh_and_mul_cpu(device):
    """gelu_tanh_and_mul_cpu (sgl-kernel/csrc/cpu/activation.cpp) with a gate of ones."""
    z = load_golden("ref_gelu_tanh.npz")
    x = bf16(z["x"]).to(device)
    out = torch.empty_like(x)
    _C.activation(out, x, 1)
    torch.cuda.synchronize()
    _close(out, bf16(z["out"]))
    # one bf16 ulp, judged on the scale of the operands: for x << 0 the sum 1 + tanh(u) cancels
    # and the two implementations of tanh leave different dust (-0.0 here, -1e-8 there), which
    # an ulp count of the RESULT would call thousands of ulps
    assert_close_bf16(out, bf16(z["out"]), rel=1e-3, max_ulps=1.0, what="gelu_tanh")


def test_block_fp8_weight_gemm_vs_reference_fp8_scaled_mm_cpu(device):
    """fp8_scaled_mm_cpu (sgl-kernel/csrc/cpu/gemm_fp8.cpp:475): bf16 activations times a
    128 x 128-block-quantised fp8 weight.  The library has no W8A16 form (the GPU path of the
    reference quantises the activations too), so two results meet the stored output:
      * the bf16 linear on the dequantised weight -- exactly the comparison value the
        reference's own test builds (test/srt/cpu/test_gemm.py:176-218), at its tolerance;
      * the W8A8 block linear (group-quantised activations, mi355_fp8_blockwise_scaled_mm),
        which carries the activations' 3-mantissa-bit rounding on top."""
    z = load_golden("ref_fp8_block_w8a16.npz")
    a, wq = bf16(z["a"]), fp8(z["wq"])
    ws = torch.from_numpy(z["ws"])
    bias = torch.from_numpy(z["bias"]).bfloat16()
    ref = bf16(z["out"])
    M, K = a.shape
    N = wq.shape[0]
    dq = (wq.float() * ws.repeat_interleave(128, 0).repeat_interleave(128, 1)).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    _C.linear_bf16(a.to(device), dq.to(device), bias.to(device), out, 0)
    torch.cuda.synchronize()
    _close(out, ref)
    assert rel_l2(out, ref) <= 5e-3
    # W8A8 block path
    ad = a.to(device)
    aq = torch.empty(M, K, dtype=torch.float8_e4m3fn, device=device)
    asc = torch.empty(M, K // 128, dtype=torch.float32, device=device)
    _C.per_token_group_quant_fp8(ad, aq, asc, 128, 1e-10, -448.0, 448.0, False)
    out8 = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    wsd = ws.to(device).contiguous()
    _C.fp8_blockwise_scaled_mm(aq, wq.to(device), asc, wsd, out8, (asc.stride(0), asc.stride(1)),
                               (wsd.stride(1), wsd.stride(0)))
    out8 += bias.to(device)
    torch.cuda.synchronize()
    e8 = rel_l2(out8, ref)
    print(f"W8A8 block linear vs the reference's W8A16 kernel: rel-L2 {e8:.2e}")
    assert e8 <= 4e-2
