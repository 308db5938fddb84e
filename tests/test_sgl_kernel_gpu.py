"""`torch.ops.sgl_kernel.*` drop-in registry on MI355X.

Each case restates the check the reference's own kernel tests make for the op
(sgl-kernel/tests/test_fp8_gemm.py:6-45, test_fp8_blockwise_gemm.py:19-89,
test_per_token_quant_fp8.py:14-53, test_norm.py, test_rotary_embedding.py,
test_flash_attention.py) on bf16 inputs, with the reference test's tolerance, and adds the
error behaviour: unsupported arguments raise RuntimeError, CPU tensors are not dispatched.
"""
import math

import pytest
import torch

import sgl_kernel
from oracle import ops as O

pytestmark = pytest.mark.gpu
FP8 = torch.float8_e4m3fn
DEV = "cuda"


def _rand_fp8(rows, cols, g):
    x = (torch.rand(rows, cols, generator=g) - 0.5) * 2 * 448.0
    return x.clamp(-448, 448).to(FP8)


@pytest.mark.parametrize("M", [1, 128, 1024])
@pytest.mark.parametrize("N", [16, 512, 4096])
@pytest.mark.parametrize("K", [512, 4096])
@pytest.mark.parametrize("with_bias", [True, False])
def test_fp8_scaled_mm(M, N, K, with_bias):
    """sgl-kernel/tests/test_fp8_gemm.py: rtol 0.02, atol 1 on +-448-range operands."""
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    a = _rand_fp8(M, K, g).to(DEV)
    b = _rand_fp8(N, K, g).to(DEV).t()
    sa = (torch.randn(M, generator=g) * 0.001).to(DEV)
    sb = (torch.randn(N, generator=g) * 0.001).to(DEV)
    bias = torch.randn(N, generator=g).bfloat16().to(DEV) if with_bias else None
    out = sgl_kernel.fp8_scaled_mm(a, b, sa, sb, torch.bfloat16, bias)
    ref = torch.matmul(a.float(), b.float()) * sa.view(-1, 1) * sb.view(1, -1)
    ref = ref.to(torch.bfloat16)
    if bias is not None:
        ref = ref + bias.view(1, -1)
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    torch.testing.assert_close(ref, out, rtol=0.02, atol=1)


@pytest.mark.parametrize("M", [1, 3, 127, 512])
@pytest.mark.parametrize("N,K", [(128, 512), (1024, 1024), (14080, 512), (512, 14080)])
def test_fp8_blockwise_scaled_mm(M, N, K):
    """sgl-kernel/tests/test_fp8_blockwise_gemm.py: column-major scale tensors, rtol 0.02 atol 1."""
    g = torch.Generator().manual_seed(M + N + K)
    a = _rand_fp8(M, K, g).to(DEV)
    b = _rand_fp8(N, K, g).to(DEV).t()
    sa = (torch.randn(M, K // 128, generator=g) * 0.001).to(DEV).t().contiguous().t()
    sb = (torch.randn(K // 128, N // 128, generator=g) * 0.001).to(DEV).t().contiguous().t()
    out = sgl_kernel.fp8_blockwise_scaled_mm(a, b, sa, sb, torch.bfloat16)
    A = a.float() * sa.repeat_interleave(128, dim=1)
    B = b.float() * sb.repeat_interleave(128, dim=0).repeat_interleave(128, dim=1)
    ref = torch.mm(A, B).to(torch.bfloat16)
    torch.testing.assert_close(ref, out, rtol=0.02, atol=1)


@pytest.mark.parametrize("num_tokens", [128, 512])
@pytest.mark.parametrize("hidden_dim", [512, 1368, 4096])
def test_per_token_quant_fp8(num_tokens, hidden_dim):
    """sgl-kernel/tests/test_per_token_quant_fp8.py: quantising with the returned scale
    reproduces the output (1e-3); plus bit-exactness against the oracle."""
    g = torch.Generator().manual_seed(hidden_dim)
    x = torch.rand(num_tokens, hidden_dim, generator=g).bfloat16()
    xd = x.to(DEV)
    s = torch.zeros(num_tokens, device=DEV)
    q = torch.empty_like(xd, dtype=FP8)
    sgl_kernel.sgl_per_token_quant_fp8(xd, q, s)
    requant = (xd.float() * s.view(-1, 1).reciprocal()).clamp(-448, 448).to(FP8)
    torch.testing.assert_close(q.float(), requant.float(), rtol=1e-3, atol=1e-3)
    q_ref, s_ref = O.per_token_quant_fp8(x)
    assert torch.equal(s.cpu(), s_ref.view(-1))
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))


@pytest.mark.parametrize("column_major", [False, True])
def test_per_token_group_quant_8bit(column_major):
    g = torch.Generator().manual_seed(3)
    M, K, G = 96, 1024, 128
    x = torch.randn(M, K, generator=g).bfloat16()
    xd = x.to(DEV)
    q = torch.empty(M, K, dtype=FP8, device=DEV)
    s = (torch.empty(K // G, M, device=DEV).t() if column_major else torch.empty(M, K // G, device=DEV))
    sgl_kernel.sgl_per_token_group_quant_8bit(xd, q, s, G, 1e-10, -448.0, 448.0, False)
    q_ref, s_ref = O.per_token_group_quant_fp8(x, G)
    assert torch.equal(s.cpu().contiguous(), s_ref)
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    with pytest.raises(RuntimeError):
        sgl_kernel.sgl_per_token_group_quant_8bit(xd, q, s, G, 1e-10, -448.0, 448.0, True)


def test_per_tensor_quant_fp8():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(77, 640, generator=g).bfloat16()
    xd = x.to(DEV)
    q = torch.empty_like(xd, dtype=FP8)
    s = torch.zeros(1, device=DEV)
    sgl_kernel.sgl_per_tensor_quant_fp8(xd, q, s, False)
    q_ref, s_ref = O.per_tensor_quant_fp8(x)
    assert torch.equal(s.cpu(), s_ref.view(1))
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    s2 = torch.full((1,), 0.05, device=DEV)
    sgl_kernel.sgl_per_tensor_quant_fp8(xd, q, s2, True)
    q_ref2, _ = O.per_tensor_quant_fp8(x, scale=torch.tensor([0.05]))
    assert torch.equal(q.cpu().view(torch.uint8), q_ref2.view(torch.uint8))
    assert s2.item() == pytest.approx(0.05)


@pytest.mark.parametrize("rows,dim", [(1, 128), (19, 4096), (99, 3072), (4608 * 2, 128)])
def test_rmsnorm_and_fused_add(rows, dim):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, dim, generator=g).bfloat16()
    r = torch.randn(rows, dim, generator=g).bfloat16()
    w = torch.randn(dim, generator=g).bfloat16()
    out = sgl_kernel.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    ref = O.rmsnorm(x, w, 1e-6)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-2, atol=1e-2)
    xd, rd = x.to(DEV), r.to(DEV)
    sgl_kernel.fused_add_rmsnorm(xd, rd, w.to(DEV), 1e-6)
    x_ref, r_ref = O.fused_add_rmsnorm(x, r, w, 1e-6)
    torch.testing.assert_close(rd.cpu().float(), r_ref.float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(xd.cpu().float(), x_ref.float(), rtol=1e-2, atol=1e-2)


def _cos_sin_cache(rotary_dim, max_pos, base):
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float32) / rotary_dim))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return torch.cat([f.cos(), f.sin()], dim=-1)


@pytest.mark.parametrize("head_size,rotary_dim,max_pos,base,is_neox,bs,seq,hq,hk", [
    (64, 64, 4096, 8000, True, 2, 512, 64, 8),
    (256, 128, 4096, 10000, True, 2, 512, 4, 2),
    (512, 128, 311, 10000, True, 3, 39, 4, 2),
    (128, 128, 2048, 10000, False, 2, 512, 32, 8),
    (512, 128, 311, 10000, False, 3, 39, 4, 2),
])
def test_rotary_embedding(head_size, rotary_dim, max_pos, base, is_neox, bs, seq, hq, hk):
    """Cases of sgl-kernel/tests/test_rotary_embedding.py:15-57 (bf16), atol = rtol = 1e-2,
    through both entry points of the registry."""
    g = torch.Generator().manual_seed(head_size + seq)
    cache = _cos_sin_cache(rotary_dim, max_pos, base)
    pos = torch.randint(0, max_pos, (bs * seq,), generator=g)
    q = torch.randn(bs * seq, hq * head_size, generator=g).bfloat16()
    k = torch.randn(bs * seq, hk * head_size, generator=g).bfloat16()
    q_ref, k_ref = O.rotary_embedding_cache(pos, q, k, head_size, cache, is_neox)
    qd, kd = q.to(DEV), k.to(DEV)
    sgl_kernel.rotary_embedding(pos.to(DEV), qd, kd, head_size, cache.to(DEV), is_neox)
    torch.testing.assert_close(qd.cpu().float(), q_ref.float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(kd.cpu().float(), k_ref.float(), atol=1e-2, rtol=1e-2)
    qd, kd = q.to(DEV), k.to(DEV)
    sgl_kernel.apply_rope_with_cos_sin_cache_inplace(pos.to(DEV), qd, kd, head_size, cache.to(DEV), is_neox)
    torch.testing.assert_close(qd.cpu().float(), q_ref.float(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(kd.cpu().float(), k_ref.float(), atol=1e-2, rtol=1e-2)


def _attn_ref(q, k, v, scale):
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    rep = qf.shape[1] // kf.shape[1]
    kf, vf = kf.repeat_interleave(rep, 1), vf.repeat_interleave(rep, 1)
    s = qf @ kf.transpose(-1, -2) * scale
    lse = torch.logsumexp(s, -1)
    return (torch.softmax(s, -1) @ vf).transpose(1, 2), lse


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,d", [(2, 333, 333, 4, 4, 128), (1, 1024, 777, 6, 2, 64),
                                            (3, 64, 200, 2, 1, 128)])
def test_flash_attn_varlen_func(B, Sq, Sk, H, Hk, d):
    g = torch.Generator().manual_seed(Sq)
    q = torch.randn(B, Sq, H, d, generator=g).bfloat16()
    k = torch.randn(B, Sk, Hk, d, generator=g).bfloat16()
    v = torch.randn(B, Sk, Hk, d, generator=g).bfloat16()
    ref, lse_ref = _attn_ref(q, k, v, d ** -0.5)
    out, lse, *_ = sgl_kernel.flash_attn_varlen_func(
        q.to(DEV), k.to(DEV), v.to(DEV), None, None, max_seqlen_q=Sq, max_seqlen_k=Sk,
        return_softmax_lse=True)
    assert out.shape == q.shape and lse.shape == (B, H, Sq)
    torch.testing.assert_close(out.cpu().float(), ref, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(lse.cpu(), lse_ref, atol=1e-3, rtol=1e-3)
    # packed layout with all sequences at the maximum length
    cu_q = torch.arange(0, (B + 1) * Sq, Sq, dtype=torch.int32, device=DEV)
    cu_k = torch.arange(0, (B + 1) * Sk, Sk, dtype=torch.int32, device=DEV)
    out2, lse2, *_ = sgl_kernel.flash_attn_varlen_func(
        q.view(B * Sq, H, d).to(DEV), k.view(B * Sk, Hk, d).to(DEV), v.view(B * Sk, Hk, d).to(DEV),
        cu_q, cu_k, max_seqlen_q=Sq, max_seqlen_k=Sk, return_softmax_lse=True)
    assert out2.shape == (B * Sq, H, d) and lse2.shape == (H, B * Sq)
    assert torch.equal(out2.view(B, Sq, H, d), out)
    assert torch.equal(lse2.view(H, B, Sq).permute(1, 0, 2), lse)


def test_errors_are_loud():
    q = torch.randn(1, 64, 2, 128, device=DEV).bfloat16()
    with pytest.raises(RuntimeError, match="causal"):
        sgl_kernel.flash_attn_varlen_func(q, q, q, None, None, 64, 64, causal=True)
    with pytest.raises(RuntimeError, match="sliding"):
        sgl_kernel.flash_attn_varlen_func(q, q, q, None, None, 64, 64, window_size=(8, 8))
    with pytest.raises(RuntimeError, match="not supported"):
        sgl_kernel.flash_attn_varlen_func(q, q, q, None, None, 64, 64,
                                          q_descale=torch.ones(1, device=DEV))
    with pytest.raises(ValueError):
        sgl_kernel.flash_attn_varlen_func(q, q, q, None, None)
    a = torch.zeros(8, 128, device=DEV).to(FP8)
    with pytest.raises(RuntimeError, match="column major"):
        sgl_kernel.fp8_scaled_mm(a, torch.zeros(128, 16, device=DEV).to(FP8),
                                 torch.ones(8, device=DEV), torch.ones(16, device=DEV), torch.bfloat16)
    with pytest.raises(RuntimeError, match="bfloat16"):
        sgl_kernel.fp8_scaled_mm(a, torch.zeros(16, 128, device=DEV).to(FP8).t(),
                                 torch.ones(8, device=DEV), torch.ones(16, device=DEV), torch.float32)
    with pytest.raises(NotImplementedError):  # no CPU kernel is registered
        sgl_kernel.rmsnorm(torch.randn(4, 64).bfloat16(), torch.ones(64))
    # empty batch: functional ops still return a correctly shaped tensor
    out = sgl_kernel.fp8_scaled_mm(a[:0], torch.zeros(16, 128, device=DEV).to(FP8).t(),
                                   torch.ones(0, device=DEV), torch.ones(16, device=DEV), torch.bfloat16)
    assert out.shape == (0, 16)
