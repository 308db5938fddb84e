"""Parity of every C-ABI kernel against the CPU oracle on seeded inputs (MI355X only).

Bars: bit-exact for the quantisers' fp8 bytes, scales and the pure elementwise bf16 ops;
for ops with fp32 accumulation (GEMM, attention, norms) the results are rounded to bf16 at
the same point as the oracle, so they may differ by accumulation order only: relative L2
<= 1e-3 (the project tolerance) and a tiny fraction of 1-ulp bf16 flips.
"""
import pytest
import torch

from oracle import ops
from sglang_diffusion_amd import _C
from tests.helpers import assert_close_bf16, bf16_ulp_diff, mismatch_frac, rel_l2

pytestmark = pytest.mark.gpu
FP8 = torch.float8_e4m3fn


def g(seed=0):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------- quant --
@pytest.mark.parametrize("M,K,scale", [(1, 128, 1.0), (7, 3072, 3.0), (130, 15360, 0.02),
                                        (33, 24576, 1.0), (5, 8, 1.0)])
def test_per_token_quant_bit_exact(device, M, K, scale):
    x = (torch.randn(M, K, generator=g(M + K)) * scale).bfloat16()
    x[0, 0] = 0
    if M > 2:
        x[2] = 0  # an all-zero row: scale 0, q 0
    qr, sr = ops.per_token_quant_fp8(x)
    xd = x.to(device)
    q = torch.empty(M, K, dtype=FP8, device=device)
    s = torch.empty(M, dtype=torch.float32, device=device)
    _C.per_token_quant_fp8(xd, q, s)
    assert torch.equal(s.cpu(), sr)
    assert mismatch_frac(q, qr) == 0.0


def test_per_token_quant_strided_rows_and_empty(device):
    big = torch.randn(16, 512, generator=g(3)).bfloat16()
    x = big[:, :256]  # row stride 512
    qr, sr = ops.per_token_quant_fp8(x.contiguous())
    q = torch.empty(16, 256, dtype=FP8, device=device)
    s = torch.empty(16, dtype=torch.float32, device=device)
    _C.per_token_quant_fp8(big.to(device)[:, :256], q, s)
    assert torch.equal(s.cpu(), sr) and mismatch_frac(q, qr) == 0.0
    e = torch.empty(0, 256, dtype=torch.bfloat16, device=device)
    _C.per_token_quant_fp8(e, torch.empty(0, 256, dtype=FP8, device=device),
                           torch.empty(0, device=device))


def test_per_token_quant_rejects_bad_shapes(device):
    x = torch.zeros(4, 100, dtype=torch.bfloat16, device=device)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        _C.per_token_quant_fp8(x, torch.empty(4, 100, dtype=FP8, device=device),
                               torch.empty(4, device=device))


@pytest.mark.parametrize("M,K,G,colmajor", [(9, 512, 128, False), (64, 3072, 128, True),
                                             (3, 1024, 256, False), (10, 3072, 3072, False)])
def test_per_token_group_quant_bit_exact(device, M, K, G, colmajor):
    x = (torch.randn(M, K, generator=g(K)) * 2).bfloat16()
    x[0, :G] = 0  # a zero group: eps floor
    qr, sr = ops.per_token_group_quant_fp8(x, G)
    q = torch.empty(M, K, dtype=FP8, device=device)
    if colmajor:
        s = torch.empty(K // G, M, dtype=torch.float32, device=device).t()
    else:
        s = torch.empty(M, K // G, dtype=torch.float32, device=device)
    _C.per_token_group_quant_fp8(x.to(device), q, s, G, 1e-10, -448.0, 448.0, False)
    assert torch.equal(s.cpu(), sr)
    assert mismatch_frac(q, qr) == 0.0


def test_group_quant_ue8m0_raises(device):
    x = torch.zeros(4, 128, dtype=torch.bfloat16, device=device)
    with pytest.raises(RuntimeError, match="UE8M0"):
        _C.per_token_group_quant_fp8(x, torch.empty(4, 128, dtype=FP8, device=device),
                                     torch.empty(4, 1, device=device), 128, 1e-10, -448, 448, True)


@pytest.mark.parametrize("n", [8, 4096 * 3 + 8, 1000003])
def test_per_tensor_quant_bit_exact(device, n):
    x = (torch.randn(n, generator=g(n)) * 4).bfloat16()
    qr, sr = ops.per_tensor_quant_fp8(x)
    q = torch.empty(n, dtype=FP8, device=device)
    s = torch.zeros(1, dtype=torch.float32, device=device)
    _C.per_tensor_quant_fp8(x.to(device), q, s, False)
    assert torch.equal(s.cpu(), sr)
    assert mismatch_frac(q, qr) == 0.0
    # static: reuse the scale
    q2 = torch.empty(n, dtype=FP8, device=device)
    _C.per_tensor_quant_fp8(x.to(device), q2, s, True)
    assert torch.equal(q2.view(torch.uint8), q.view(torch.uint8))


# -------------------------------------------------------------------------------- GEMM --
def _gemm_close(out, ref, tol=1e-3):
    assert_close_bf16(out, ref, rel=tol, what="gemm")
    assert mismatch_frac(out, ref) < 0.02  # only accumulation-order rounding flips


@pytest.mark.parametrize("M,N,K,bias", [(1, 16, 512, True), (17, 136, 128, False),
                                         (257, 264, 1024, True), (512, 3072, 3072, True),
                                         (300, 64, 2048, True), (2048, 512, 15360, False)])
def test_fp8_scaled_mm_vs_oracle(device, M, N, K, bias):
    """shapes after sgl-kernel/tests/test_fp8_gemm.py:17-35 (M 1..4096, N 16.., K 512..16384)."""
    a = (torch.randn(M, K, generator=g(1)) * 3).clamp(-448, 448).to(FP8)
    w = (torch.randn(N, K, generator=g(2)) * 3).clamp(-448, 448).to(FP8)
    sa = torch.rand(M, generator=g(3)) * 0.02 + 1e-3
    sb = torch.rand(N, generator=g(4)) * 0.02 + 1e-3
    b = torch.randn(N, generator=g(5)).bfloat16() if bias else None
    ref = ops.fp8_scaled_mm(a, w, sa, sb, b)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    _C.fp8_scaled_mm(a.to(device), w.to(device), sa.to(device), sb.to(device),
                     b.to(device) if bias else None, out)
    _gemm_close(out, ref)


@pytest.mark.parametrize("M,N,K,bias", [(1, 16, 512, True), (128, 512, 1024, False),
                                         (1024, 1024, 2048, True), (300, 136, 16384, True)])
def test_fp8_scaled_mm_float16_output(device, M, N, K, bias):
    """out_dtype = float16 through the op registry, inputs and formula of
    sgl-kernel/tests/test_fp8_gemm.py:6-35 (full +-448 range, scales ~1e-3): the fp32 accumulator
    times the scales, plus the half bias, rounded to half.  Checked to the reference test's own
    rtol 0.02 / atol 1 and, tighter, to 1e-3 relative against the same formula in float64."""
    import sgl_kernel  # noqa: F401  (registers torch.ops.sgl_kernel)

    a = ((torch.rand(M, K, generator=g(41)) - 0.5) * 2 * 448).to(FP8)
    w = ((torch.rand(N, K, generator=g(42)) - 0.5) * 2 * 448).to(FP8)
    sa = torch.randn(M, generator=g(43)) * 0.001
    sb = torch.randn(N, generator=g(44)) * 0.001
    b = torch.randn(N, generator=g(45)).half() if bias else None
    ref = (a.double() @ w.double().t()) * sa.double().view(-1, 1) * sb.double().view(1, -1)
    if bias:
        ref = ref + b.double().view(1, -1)
    out = torch.ops.sgl_kernel.fp8_scaled_mm(a.to(device), w.to(device).t(), sa.to(device),
                                             sb.to(device), torch.float16,
                                             b.to(device) if bias else None)
    assert out.dtype == torch.float16 and tuple(out.shape) == (M, N)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=0.02, atol=1)
    assert rel_l2(out.cpu().float(), ref.float()) <= 1e-3
    # block-scaled op, same dtype
    if K % 128 == 0 and N % 128 == 0 and not bias:
        x = torch.randn(M, K, generator=g(46)).bfloat16()
        wq, ws = ops.quantize_weight_block((torch.randn(N, K, generator=g(47)) * 0.2).bfloat16())
        xq, xs = ops.per_token_group_quant_fp8(x, 128)
        refb = ops.fp8_blockwise_scaled_mm(xq, wq, xs, ws)       # bf16-rounded oracle
        outb = torch.ops.sgl_kernel.fp8_blockwise_scaled_mm(
            xq.to(device), wq.to(device).t(), xs.to(device), ws.to(device).t(), torch.float16)
        assert outb.dtype == torch.float16
        assert rel_l2(outb.cpu().float(), refb.float()) <= 3e-3   # the oracle's own bf16 rounding


def test_fp8_scaled_mm_exact_on_integer_data(device):
    """Small integers: every product and partial sum is exact, so any operand-layout or
    indexing slip shows as a hard mismatch.  Asymmetric operands catch transposes."""
    M, N, K = 256, 256, 256
    a = torch.randint(-2, 3, (M, K), generator=g(7)).float()
    w = torch.randint(-2, 3, (N, K), generator=g(8)).float()
    a[:, 0] = torch.arange(M).float() % 3
    w[:, 1] = torch.arange(N).float() % 5 - 2
    ones_m, ones_n = torch.ones(M), torch.ones(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    _C.fp8_scaled_mm(a.to(FP8).to(device), w.to(FP8).to(device), ones_m.to(device),
                     ones_n.to(device), None, out)
    assert torch.equal(out.float().cpu(), (a @ w.t()).bfloat16().float())


@pytest.mark.parametrize("K", [256, 384, 3072])
def test_fp8_scaled_mm_big_tiles_through_the_ring(device, K):
    """Outputs of >= 192 tiles of 256 x 256 take the five-slot-ring kernel (K >= 256).  Ragged
    M and N (partial last tiles), two, three and 24 K tiles: against the oracle, and -- same
    MFMAs in the same K order -- bit for bit against the small-tile kernel run on a slice."""
    M, N = 17 * 256 + 19, 12 * 256 + 8
    a = (torch.randn(M, K, generator=g(51)) * 3).clamp(-448, 448).to(FP8)
    w = (torch.randn(N, K, generator=g(52)) * 3).clamp(-448, 448).to(FP8)
    sa = torch.rand(M, generator=g(53)) * 0.02 + 1e-3
    sb = torch.rand(N, generator=g(54)) * 0.02 + 1e-3
    b = torch.randn(N, generator=g(55)).bfloat16()
    ad, wd, sad, sbd, bd = (t.to(device) for t in (a, w, sa, sb, b))
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=device)
    _C.fp8_scaled_mm(ad, wd, sad, sbd, bd, out)
    _gemm_close(out, ops.fp8_scaled_mm(a, w, sa, sb, b))
    for r0, rows in ((0, 256), (1000, 130), (M - 19, 19)):
        part = torch.empty(rows, N, dtype=torch.bfloat16, device=device)
        _C.fp8_scaled_mm(ad[r0:r0 + rows], wd, sad[r0:r0 + rows], sbd, bd, part)
        assert torch.equal(part, out[r0:r0 + rows]), (K, r0)


@pytest.mark.parametrize("M,N", [(22 * 256, 3072), (22 * 256 + 5, 3080), (18432, 3072)])
def test_ring_gemm_partial_last_round_as_half_tiles(device, M, N):
    """More tiles than workgroups with a short last round (264, 299 and 864 tiles on 256 CUs):
    the tail runs as 256 x 128 half tiles.  Same MFMAs per output element: bit-equal to the
    small-tile kernel on row slices, and right against the oracle."""
    K = 384
    a = (torch.randn(M, K, generator=g(71)) * 3).clamp(-448, 448).to(FP8)
    w = (torch.randn(N, K, generator=g(72)) * 3).clamp(-448, 448).to(FP8)
    sa = torch.rand(M, generator=g(73)) * 0.02 + 1e-3
    sb = torch.rand(N, generator=g(74)) * 0.02 + 1e-3
    b = torch.randn(N, generator=g(75)).bfloat16()
    ad, wd, sad, sbd, bd = (t.to(device) for t in (a, w, sa, sb, b))
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=device)
    _C.fp8_scaled_mm(ad, wd, sad, sbd, bd, out)
    if M <= 6000:
        _gemm_close(out, ops.fp8_scaled_mm(a, w, sa, sb, b))
    for r0 in range(0, M, 2048):             # <= 8 x 13 tiles per slice: the small-tile path
        rows = min(2048, M - r0)
        part = torch.empty(rows, N, dtype=torch.bfloat16, device=device)
        _C.fp8_scaled_mm(ad[r0:r0 + rows], wd, sad[r0:r0 + rows], sbd, bd, part)
        assert torch.equal(part.view(torch.int16), out[r0:r0 + rows].view(torch.int16)), (M, N, r0)


@pytest.mark.parametrize("gelu", [False, True])
def test_ring_gemm_with_row_maps_gate_and_half_columns(device, gelu):
    """The ring kernel behind the fused epilogues: joint-buffer row maps whose batches do not end
    on tile borders, then either gate / residual or tanh-GELU with a range of half-format output
    columns -- bit for bit what the small-tile kernel gives when each batch is a call of its own."""
    B, Ti, Tt, K, N = 2, 2300, 60, 512, 3072
    S = Ti + Tt
    x = torch.randn(B * S, K, generator=g(61)).bfloat16()
    w = (torch.randn(N, K, generator=g(62)) * K ** -0.5).bfloat16()
    bias = (torch.randn(N, generator=g(63)) * 0.1).bfloat16().to(device)
    gate = torch.randn(B, N, generator=g(64)).bfloat16().to(device)
    res = torch.randn(B * S, N, generator=g(65)).bfloat16()
    xq, xs = ops.per_token_quant_fp8(x)
    wq, wsc = ops.quantize_weight_per_channel(w)
    xq, xs, wq, wsc = xq.to(device), xs.to(device), wq.to(device), wsc.to(device)
    kw = dict(N=N, K=K, lda=K, ldw=K, ldo=N, sb=wsc, bias=bias, fp8=True)
    if gelu:
        kw.update(act=1, f16_cols=(1024, 2048))
    whole = res.to(device).clone()
    extra = {} if gelu else dict(residual=whole, gate=gate, ldg=gate.stride(0), gate_rows=Ti)
    _C.linear_ex(xq, wq, whole, M=B * Ti, sa=xs, a_map=(Ti, S, Tt), o_map=(Ti, S, Tt), **kw, **extra)
    parts = res.to(device).clone()
    for b in range(B):                      # 9 x 12 tiles each: the small-tile path
        pb = parts[b * S:(b + 1) * S]
        extra = {} if gelu else dict(residual=pb, gate=gate[b:b + 1], ldg=gate.stride(0), gate_rows=Ti)
        _C.linear_ex(xq[b * S:(b + 1) * S], wq, pb, M=Ti, sa=xs[b * S:(b + 1) * S],
                     a_map=(Ti, S, Tt), o_map=(Ti, S, Tt), **kw, **extra)
    torch.cuda.synchronize()
    assert torch.equal(whole.view(torch.int16), parts.view(torch.int16))
    assert torch.equal(whole.view(B, S, N)[:, :Tt].cpu(), res.view(B, S, N)[:, :Tt]), "text rows untouched"
    _C.f16_range_status(reset=True)


def test_fp8_scaled_mm_rejects_unaligned(device):
    a = torch.zeros(8, 100, dtype=FP8, device=device)
    w = torch.zeros(16, 100, dtype=FP8, device=device)
    with pytest.raises(RuntimeError, match="multiple of 128"):
        _C.fp8_scaled_mm(a, w, torch.ones(8, device=device), torch.ones(16, device=device), None,
                         torch.empty(8, 16, dtype=torch.bfloat16, device=device))


@pytest.mark.parametrize("M,N,K", [(5, 128, 256), (130, 384, 1024), (256, 3072, 3072)])
def test_fp8_blockwise_scaled_mm_vs_oracle(device, M, N, K):
    """python/sglang/test/test_block_fp8.py:291-343 formula."""
    x = (torch.randn(M, K, generator=g(11))).bfloat16()
    wq, ws = ops.quantize_weight_block((torch.randn(N, K, generator=g(12)) * 0.2).bfloat16())
    xq, xs = ops.per_token_group_quant_fp8(x, 128)
    ref = ops.fp8_blockwise_scaled_mm(xq, wq, xs, ws)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    xsd, wsd = xs.to(device), ws.to(device)
    _C.fp8_blockwise_scaled_mm(xq.to(device), wq.to(device), xsd, wsd, out,
                               (xsd.stride(0), xsd.stride(1)), (wsd.stride(1), wsd.stride(0)))
    _gemm_close(out, ref)


@pytest.mark.parametrize("M,N,K,act", [(4, 9216, 3072, 0), (1, 64, 256, 0), (16, 3072, 768, 1),
                                        (100, 3072, 64, 0), (257, 264, 192, 1),
                                        (1024, 12288, 3072, 1), (700, 64, 3072, 0),
                                        # small M, many channels: four channels per wave, ragged tail
                                        (3, 16392 + 8, 256, 0), (16, 18432, 512, 1)])
def test_linear_bf16_vs_oracle(device, M, N, K, act):
    x = torch.randn(M, K, generator=g(21)).bfloat16()
    w = (torch.randn(N, K, generator=g(22)) * K ** -0.5).bfloat16()
    b = (torch.randn(N, generator=g(23)) * 0.1).bfloat16()
    ref = ops.linear_bf16(x, w, b, "gelu_tanh" if act else None)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    _C.linear_bf16(x.to(device), w.to(device), b.to(device), out, act)
    # the CPU bf16 GEMM (oneDNN) accumulates in a different order: same tolerance class
    assert_close_bf16(out, ref, rel=1e-3, what="linear_bf16")


@pytest.mark.parametrize("B,Ti", [(2, 96), (1, 96), (1, 100), (3, 45)])
def test_linear_ex_rowmaps_gate_residual(device, B, Ti):
    """Joint-buffer row maps + fused gate/residual epilogue == separate eager ops.  One batch
    whose rows end inside a tile: the tile's row blocks past the end take clamped addresses for
    the residual and gate they prefetch, in the batch of the LAST row; batches shorter than a
    32-row block take the per-row division."""
    Tt, D, N = 32, 256, 384
    S = Ti + Tt
    x = torch.randn(B * S, D, generator=g(31)).bfloat16()       # joint activations
    w = (torch.randn(N, D, generator=g(32)) * D ** -0.5).bfloat16()
    bias = (torch.randn(N, generator=g(33)) * 0.1).bfloat16()
    gate = torch.randn(B, 3 * N, generator=g(34)).bfloat16()
    res = torch.randn(B * S, N, generator=g(35)).bfloat16()      # joint residual/out
    xq, xs = ops.per_token_quant_fp8(x)
    wq, wsc = ops.quantize_weight_per_channel(w)
    # oracle: image rows only
    x3 = x.view(B, S, D)[:, Tt:]
    y = ops.fp8_linear(x3, wq, wsc, bias)
    exp = res.view(B, S, N).clone()
    exp[:, Tt:] = ops.gate_residual(exp[:, Tt:], gate[:, N:2 * N], y)
    out = res.to(device).clone()
    gd = gate.to(device)
    _C.linear_ex(xq.to(device), wq.to(device), out, M=B * Ti, N=N, K=D, lda=D, ldw=D, ldo=N,
                 sa=xs.to(device), sb=wsc.to(device), bias=bias.to(device), residual=out,
                 gate=gd[:, N:2 * N], ldg=gd.stride(0), gate_rows=Ti, a_map=(Ti, S, Tt),
                 o_map=(Ti, S, Tt), fp8=True)
    got = out.cpu().view(B, S, N)
    assert torch.equal(got[:, :Tt], res.view(B, S, N)[:, :Tt]), "text rows must be untouched"
    assert rel_l2(got[:, Tt:], exp[:, Tt:]) <= 1e-3


def test_linear_ex_random_shapes(device):
    """tools/probes/gemm_fuzz.py: random M / N / K, batch geometry of the row maps and epilogue
    feature sets through every tile class, against an fp32 rendering, with sentinel rows in the
    gaps of the joint buffer."""
    import importlib.util
    import os
    import random
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes",
                        "gemm_fuzz.py")
    spec = importlib.util.spec_from_file_location("gemm_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = random.Random(11)
    bad = [i for i in range(80) if not mod.one(rng, i)]
    assert not bad, bad


def test_linear_ex_block_scaled_epilogues(device):
    """The block-scaled GEMM behind the fused epilogues (bias after the bf16 cast, GELU, joint
    row maps with group scales indexed by the mapped row, gate/residual, half columns) gives
    the bits of mi355_fp8_blockwise_scaled_mm followed by the separate eager ops."""
    B, Ti, Tt, D, N = 2, 96, 32, 256, 384
    S = Ti + Tt
    x = torch.randn(B * S, D, generator=g(61)).bfloat16()
    w = (torch.randn(N, D, generator=g(62)) * D ** -0.5).bfloat16()
    bias = (torch.randn(N, generator=g(63)) * 0.1).bfloat16().to(device)
    gate = torch.randn(B, 3 * N, generator=g(64)).bfloat16().to(device)
    res = torch.randn(B * S, N, generator=g(65)).bfloat16().to(device)
    xq, xs = ops.per_token_group_quant_fp8(x, 128)
    wq, wsc = ops.quantize_weight_block(w)
    xq, xs, wq, wsc = xq.to(device), xs.to(device), wq.to(device), wsc.to(device)

    def plain(rows_q, rows_s):
        out = torch.empty(rows_q.shape[0], N, dtype=torch.bfloat16, device=device)
        _C.fp8_blockwise_scaled_mm(rows_q, wq, rows_s, wsc, out, (rows_s.stride(0), rows_s.stride(1)),
                                   (wsc.stride(1), wsc.stride(0)))
        return out

    base = plain(xq, xs)
    # bias, then GELU
    got = torch.empty_like(base)
    _C.linear_ex(xq, wq, got, M=B * S, N=N, K=D, lda=D, ldw=D, ldo=N, sa=xs, sb=wsc, bias=bias,
                 fp8=True, block_scale=True)
    want = base + bias
    assert torch.equal(got, want)
    _C.linear_ex(xq, wq, got, M=B * S, N=N, K=D, lda=D, ldw=D, ldo=N, sa=xs, sb=wsc, bias=bias,
                 fp8=True, block_scale=True, act=1)
    gelu = torch.empty_like(want)
    _C.activation(gelu, want, 1)
    assert torch.equal(got, gelu)
    # image rows of the joint buffer, gate * y + residual written back in place
    out = res.clone()
    _C.linear_ex(xq, wq, out, M=B * Ti, N=N, K=D, lda=D, ldw=D, ldo=N, sa=xs, sb=wsc, bias=bias,
                 residual=out, gate=gate[:, N:2 * N], ldg=gate.stride(0), gate_rows=Ti,
                 a_map=(Ti, S, Tt), o_map=(Ti, S, Tt), fp8=True, block_scale=True)
    exp = res.clone().view(B, S, N)
    y = want.view(B, S, N)[:, Tt:]
    exp[:, Tt:] = exp[:, Tt:] + gate[:, None, N:2 * N] * y
    assert torch.equal(out.view(B, S, N), exp)
    # half columns
    mixed = torch.empty_like(base)
    _C.linear_ex(xq, wq, mixed, M=B * S, N=N, K=D, lda=D, ldw=D, ldo=N, sa=xs, sb=wsc, bias=bias,
                 fp8=True, block_scale=True, f16_cols=(256, 384))
    assert torch.equal(mixed[:, :256], want[:, :256])
    assert torch.equal(_f16_bits_to_float(mixed[:, 256:].contiguous()), want[:, 256:].float().cpu())


@pytest.mark.parametrize("B,T,dim,mapped", [(2, 50, 256, False), (3, 33, 3072, True), (1, 7, 128, False)])
def test_layernorm_modulate_group_quant(device, B, T, dim, mapped):
    """LayerNorm + modulation + per-(token, 128 channels) quantisation in one pass == the
    bf16 kernel followed by mi355_per_token_group_quant_fp8, bit for bit."""
    extra = 5 if mapped else 0
    x = (torch.randn(B * (T + extra), dim, generator=g(71)) * 2).bfloat16().to(device)
    if B > 1:
        x[1, 128:256] = 0  # an all-zero group after modulation is still possible only via scale
    shift = torch.randn(B, dim, generator=g(72)).bfloat16().to(device)
    scale = (torch.randn(B, dim, generator=g(73)) * 0.3).bfloat16().to(device)
    xm = (T, T + extra, extra) if mapped else None
    y = torch.empty(B * T, dim, dtype=torch.bfloat16, device=device)
    _C.layernorm_modulate(y, x, shift, scale, B, T, 1e-6, x_map=xm)
    q0 = torch.empty(B * T, dim, dtype=torch.float8_e4m3fn, device=device)
    s0 = torch.empty(B * T, dim // 128, dtype=torch.float32, device=device)
    _C.per_token_group_quant_fp8(y, q0, s0, 128, 1e-10, -448.0, 448.0, False)
    q1 = torch.empty_like(q0)
    s1 = torch.empty_like(s0)
    _C.layernorm_modulate_group_quant(q1, s1, x, shift, scale, B, T, 1e-6, x_map=xm)
    assert torch.equal(s1, s0)
    assert torch.equal(q1.view(torch.uint8), q0.view(torch.uint8))
    with pytest.raises(RuntimeError):
        _C.layernorm_modulate_group_quant(q1[:, :64], s1[:, :1].contiguous(), x[:, :64].contiguous(),
                                          None, None, B, T, 1e-6)


# --------------------------------------------------------------------------- attention --
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D", [(1, 128, 128, 2, 2, 128), (2, 333, 333, 3, 3, 128),
                                             (1, 1, 65, 2, 1, 128), (1, 200, 77, 4, 2, 64),
                                             (1, 640, 640, 24, 24, 128)])
def test_flash_attention_vs_oracle(device, B, Sq, Sk, H, Hk, D):
    q = torch.randn(B, Sq, H, D, generator=g(41)).bfloat16()
    k = torch.randn(B, Sk, Hk, D, generator=g(42)).bfloat16()
    v = torch.randn(B, Sk, Hk, D, generator=g(43)).bfloat16()
    ref, lse_ref = ops.attention(q, k, v, return_lse=True)
    out = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device=device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=device)
    _C.flash_attn_fwd(q.to(device), k.to(device), v.to(device), out, lse, D ** -0.5)
    # tolerance of the reference's own FA test: 2x the bf16 round-trip error of the fp32
    # reference (sgl-kernel/tests/test_flash_attention.py:1273-1308); P is bf16 inside the kernel
    assert rel_l2(out, ref) <= 4e-3
    assert (out.float().cpu() - ref.float()).abs().max() <= 2e-2
    assert (lse.cpu() - lse_ref).abs().max() <= 1e-4


def test_flash_attention_online_softmax_rescale(device):
    """A spiked score late in the key sequence forces the running-max rescale branch."""
    B, S, H, D = 1, 256, 1, 128
    q = torch.randn(B, S, H, D, generator=g(44)).bfloat16()
    k = torch.randn(B, S, H, D, generator=g(45)).bfloat16()
    v = torch.randn(B, S, H, D, generator=g(46)).bfloat16()
    k[0, 200, 0] = q[0, 17, 0] * 4  # huge score for (q=17, key=200), in the 4th KV tile
    ref = ops.attention(q, k, v)
    out = torch.empty_like(q, device=device)
    _C.flash_attn_fwd(q.to(device), k.to(device), v.to(device), out, None, D ** -0.5)
    assert rel_l2(out, ref) <= 4e-3
    assert rel_l2(out[0, 17], ref[0, 17]) <= 1e-2


@pytest.mark.parametrize("half_v", [False, True])
def test_flash_attention_reference_moves_on_overflowing_sums(device, half_v):
    """The reference point of the exponentials moves when a lane's sum of them is not <= 2^13.
    Spikes whose exponentials overflow fp32 against the old reference (inf, and NaN beside masked
    keys of a ragged last tile), spikes in every tile of some rows, and a plateau of scores that
    sits just under the threshold for many tiles must all match the oracle."""
    B, S, H, D = 1, 700, 2, 128  # 700 keys: the last 64-key tile is ragged
    q = torch.randn(B, S, H, D, generator=g(91)).bfloat16()
    k = torch.randn(B, S, H, D, generator=g(92)).bfloat16()
    v = torch.randn(B, S, H, D, generator=g(93)).bfloat16()
    k[0, 650, 0] = q[0, 5, 0] * 12       # scaled score ~135: 2^195 against the old reference -> inf
    k[0, 699, 1] = q[0, 300, 1] * 12     # the same in the ragged tile, beside masked keys
    for t in range(1, 10):               # row 40 of head 0: a growing spike in every tile
        k[0, 64 * t + 7, 0] = q[0, 40, 0] * (0.5 * t)
    # rows 100..131 of head 1: all keys from 128 on score ~ +5.4 (2^7.8) above the first tiles
    u = q[0, 100:132, 1].float().mean(0)
    u = u / u.norm()
    k[0, 128:, 1] = (k[0, 128:, 1].float() + u * 6.0).bfloat16()
    ref, lse_ref = ops.attention(q, k, v, return_lse=True)
    out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=device)
    fn = _C.attention if half_v else _C.flash_attn_fwd
    fn(q.to(device), k.to(device), v.to(device), out, lse, D ** -0.5)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    assert rel_l2(out, ref) <= (1.5e-3 if half_v else 4e-3)
    for (row, head) in ((5, 0), (300, 1), (40, 0), (110, 1)):
        assert rel_l2(out[0, row, head], ref[0, row, head]) <= 1e-2, (row, head)
    assert (lse.cpu() - lse_ref).abs().max() <= 2e-3


def test_flash_attention_lazy_max_on_growing_scores(device):
    """Scores that climb steadily along the key axis: the running reference max lags behind
    the true max between rescales (probabilities above 1 in the meantime) and is moved
    several times; both P.V precisions must still match the oracle."""
    B, S, H, D = 1, 1024, 2, 128
    q = torch.randn(B, S, H, D, generator=g(81)).bfloat16()
    k = torch.randn(B, S, H, D, generator=g(82)).bfloat16() * 0.3
    v = torch.randn(B, S, H, D, generator=g(83)).bfloat16()
    # add a component along q's mean direction that grows with the key index: scaled scores
    # rise by roughly 40 (58 in log2 units) from the first to the last key
    u = q.float().mean(dim=1, keepdim=True)
    u = u / u.norm(dim=-1, keepdim=True)
    ramp = torch.linspace(0, 1, S).view(1, S, 1, 1)
    k = (k.float() + ramp * u * 40.0).bfloat16()
    ref, lse_ref = ops.attention(q, k, v, return_lse=True)
    for fn, tol in ((_C.flash_attn_fwd, 4e-3), (_C.attention, 1.5e-3)):
        out = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=device)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=device)
        fn(q.to(device), k.to(device), v.to(device), out, lse, D ** -0.5)
        assert torch.isfinite(out.float()).all()
        assert rel_l2(out, ref) <= tol
        assert (lse.cpu() - lse_ref).abs().max() <= 2e-3


def test_flash_attention_strided_views(device):
    """q/k/v as slices of a fused [B,S,3,H,D] buffer, output into a wider row."""
    B, S, H, D = 2, 130, 4, 128
    qkv = torch.randn(B, S, 3, H, D, generator=g(47)).bfloat16()
    ref = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    qd = qkv.to(device)
    wide = torch.zeros(B, S, 5 * H * D, dtype=torch.bfloat16, device=device)
    o = wide[:, :, : H * D].unflatten(-1, (H, D))
    _C.flash_attn_fwd(qd[:, :, 0], qd[:, :, 1], qd[:, :, 2], o, None, D ** -0.5)
    assert rel_l2(o, ref) <= 4e-3
    assert wide[:, :, H * D:].abs().sum().item() == 0


def _f16_bits_to_float(t):
    return t.cpu().view(torch.float16).float()


@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D", [(2, 333, 333, 3, 3, 128), (1, 200, 77, 4, 2, 64),
                                             (1, 640, 640, 24, 24, 128)])
def test_flash_attention_half_pv_vs_oracle(device, B, Sq, Sk, H, Hk, D):
    """P and V in IEEE half: about 1/3 of the bf16-P form's distance to the fp32-softmax oracle
    (what is left is mostly the bf16 rounding of the output, which the oracle shares)."""
    q = torch.randn(B, Sq, H, D, generator=g(41)).bfloat16()
    k = torch.randn(B, Sk, Hk, D, generator=g(42)).bfloat16()
    v = torch.randn(B, Sk, Hk, D, generator=g(43)).bfloat16()
    ref, lse_ref = ops.attention(q, k, v, return_lse=True)
    out = torch.empty(B, Sq, H, D, dtype=torch.bfloat16, device=device)
    lse = torch.empty(B, H, Sq, dtype=torch.float32, device=device)
    _C.attention(q.to(device), k.to(device), v.to(device), out, lse, D ** -0.5)
    assert rel_l2(out, ref) <= 1e-3
    assert (lse.cpu() - lse_ref).abs().max() <= 1e-4
    out_bf = torch.empty_like(out)
    _C.flash_attn_fwd(q.to(device), k.to(device), v.to(device), out_bf, None, D ** -0.5)
    assert rel_l2(out, ref) < rel_l2(out_bf, ref)


def test_bf16_to_f16_exact_and_saturating(device):
    x = torch.randn(37, 256, generator=g(48)).bfloat16()
    x[0, :4] = torch.tensor([1e6, -1e6, 65280.0, 0.0]).bfloat16()
    wide = torch.zeros(37, 512, dtype=torch.bfloat16)
    wide[:, 128:384] = x
    y = torch.empty(37, 256, dtype=torch.float16, device=device)
    _C.bf16_to_f16(wide.to(device)[:, 128:384], y)
    exp = x.float().clamp(-65504, 65504).half()
    assert torch.equal(y.cpu(), exp)
    v = torch.randn(2, 9, 3, 128, generator=g(49)).bfloat16()
    big = torch.randn(2, 9, 3, 3, 128, generator=g(50)).bfloat16()
    assert torch.equal(_C.value_to_f16(v.to(device)).cpu(), v.half())
    assert torch.equal(_C.value_to_f16(big.to(device)[:, :, 2]).cpu(), big[:, :, 2].half())


@pytest.mark.parametrize("fp8", [False, True])
def test_linear_ex_half_output_columns(device, fp8):
    """Columns [256, 384) of the result come out in half format holding the SAME bf16 values."""
    M, N, K = 300, 384, 256
    x = torch.randn(M, K, generator=g(51)).bfloat16()
    w = (torch.randn(N, K, generator=g(52)) * K ** -0.5).bfloat16()
    b = (torch.randn(N, generator=g(53)) * 0.1).bfloat16()
    kw = {}
    if fp8:
        xq, xs = ops.per_token_quant_fp8(x)
        wq, ws = ops.quantize_weight_per_channel(w)
        a_d, w_d = xq.to(device), wq.to(device)
        kw = dict(sa=xs.view(-1).to(device), sb=ws.to(device), fp8=True)
    else:
        a_d, w_d = x.to(device), w.to(device)
    plain = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    mixed = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    _C.linear_ex(a_d, w_d, plain, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, bias=b.to(device), **kw)
    _C.linear_ex(a_d, w_d, mixed, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, bias=b.to(device),
                 f16_cols=(256, 384), **kw)
    assert torch.equal(mixed[:, :256].cpu(), plain[:, :256].cpu())
    as_half = _f16_bits_to_float(mixed[:, 256:].contiguous())
    assert torch.equal(as_half, plain[:, 256:].float().cpu())
    with pytest.raises(RuntimeError):
        _C.linear_ex(a_d, w_d, mixed, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, f16_cols=(250, 384), **kw)


@pytest.mark.parametrize("T,H,D", [(37, 3, 128), (512, 24, 128), (5, 2, 64)])
def test_merge_attn_states_vs_oracle(device, T, H, D):
    """merge_state_v2 (sgl-kernel/tests/test_merge_state_v2.py): contiguous [tokens, heads, D],
    with empty sides (+-inf log-sum-exps) mixed in."""
    import sgl_kernel

    va = torch.randn(T, H, D, generator=g(81)).bfloat16()
    vb = torch.randn(T, H, D, generator=g(82)).bfloat16()
    sa = torch.randn(T, H, generator=g(83)) * 3
    sb = torch.randn(T, H, generator=g(84)) * 3
    sa[0, 0] = float("inf")       # treated as -inf: an empty side
    sb[1 % T, 0] = float("-inf")
    ref_v, ref_s = ops.merge_attn_states(va, sa, vb, sb)
    got_v, got_s = sgl_kernel.merge_state_v2(va.to(device), sa.to(device), vb.to(device), sb.to(device))
    assert bf16_ulp_diff(got_v, ref_v).max() <= 1          # fma vs mul+add in fp32
    assert torch.allclose(got_s.cpu(), ref_s, rtol=1e-6, atol=1e-6)
    assert torch.equal(got_v[0, 0].cpu(), vb[0, 0]) and torch.equal(got_v[1 % T, 0].cpu(), va[1 % T, 0])
    with pytest.raises(RuntimeError):
        sgl_kernel.merge_state_v2(va.to(device)[:, :, ::2], sa.to(device), vb.to(device), sb.to(device))


def test_attention_over_key_halves_merges_to_the_whole(device):
    """Size-independent property at the Flux shape of one sample: attention over the first and
    the second half of the keys, merged with their log-sum-exps (strided [B, H, S] layout, in
    place), equals attention over all keys up to the extra bf16 rounding of the two partials."""
    B, S, H, D = 1, 4608, 24, 128
    qkv = (torch.randn(B, S, 3, H, D, generator=g(91)) * 1.5).bfloat16().to(device)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    scale = D ** -0.5
    whole = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=device)
    lse_w = torch.empty(B, H, S, dtype=torch.float32, device=device)
    _C.attention(q, k, v, whole, lse_w, scale)
    parts = []
    for sl in (slice(0, 2000), slice(2000, S)):      # ragged split, second half not tile aligned
        o = torch.empty_like(whole)
        l = torch.empty_like(lse_w)
        _C.attention(q, k[:, sl], v[:, sl], o, l, scale)
        parts.append((o, l))
    (oa, la), (ob, lb) = parts
    _C.merge_attn_states(oa, la, oa, la, ob, lb)      # in place over the first partial
    assert rel_l2(oa, whole) <= 3e-3
    assert (la - lse_w).abs().max().item() <= 2e-5


@pytest.mark.parametrize("D,half_v", [(128, True), (128, False), (64, True)])
def test_attention_workgroup_width_does_not_change_the_bits(device, D, half_v):
    """Launches with a workgroup for every CU run 256 query rows per workgroup, smaller ones
    128: a slice of the queries (a small launch) must give the very rows of the big launch."""
    B, S, H = 4, 1000, 24                      # 4 * 24 * ceil(1000 / 256) = 384 workgroups
    gen = torch.Generator(device=device).manual_seed(23)
    q, k, v = ((torch.randn(B, S, H, D, generator=gen, device=device) * 1.3).bfloat16()
               for _ in range(3))
    vv = v.to(torch.float16) if half_v else v
    scale = D ** -0.5
    big = torch.empty_like(q)
    lse_big = torch.empty(B, H, S, dtype=torch.float32, device=device)
    _C.flash_attn_fwd(q, k, vv, big, lse_big, scale, v_is_f16=half_v)
    for (b, lo, hi) in [(0, 0, 128), (3, 700, 1000), (1, 250, 390)]:
        qs = q[b:b + 1, lo:hi, :2]             # 2 heads, <= 3 query blocks: a small launch
        small = torch.empty_like(qs)
        lse_small = torch.empty(1, 2, hi - lo, dtype=torch.float32, device=device)
        _C.flash_attn_fwd(qs, k[b:b + 1, :, :2], vv[b:b + 1, :, :2], small, lse_small, scale,
                          v_is_f16=half_v)
        assert torch.equal(small, big[b:b + 1, lo:hi, :2])
        assert torch.equal(lse_small, lse_big[b:b + 1, :2, lo:hi])


def test_attention_at_video_length(device):
    """One sequence-parallel rank's attention of the HunyuanVideo configuration (3 of 24 heads
    over 118 800 image + 256 text tokens): the merge property over a ragged key split, and
    sampled query rows against an fp64 softmax(q K^T) V computed with torch on the device."""
    B, S, H, D = 1, 118800 + 256, 3, 128
    gen = torch.Generator(device=device).manual_seed(17)
    q, k, v = ((torch.randn(B, S, H, D, generator=gen, device=device) * 1.2).bfloat16()
               for _ in range(3))
    scale = D ** -0.5
    whole = torch.empty_like(q)
    lse_w = torch.empty(B, H, S, dtype=torch.float32, device=device)
    _C.attention(q, k, v, whole, lse_w, scale)
    rows = torch.tensor([0, 31, 32, 127, 128, 59527, 118799, 118800, S - 1], device=device)
    qs = q[0, rows].double().transpose(0, 1)                      # [H, rows, D]
    sc = torch.einsum("hrd,shd->hrs", qs, k[0].double()) * scale
    ref = torch.einsum("hrs,shd->rhd", torch.softmax(sc, dim=-1), v[0].double())
    assert rel_l2(whole[0, rows], ref) <= 4e-3                    # 1e-3 bar + bf16 output rounding
    ref_lse = torch.logsumexp(sc, dim=-1)                         # [H, rows]
    assert (lse_w[0][:, rows].double() - ref_lse).abs().max().item() <= 1e-4
    cut = 50001
    parts = []
    for sl in (slice(0, cut), slice(cut, S)):
        o, l = torch.empty_like(whole), torch.empty_like(lse_w)
        _C.attention(q, k[:, sl], v[:, sl], o, l, scale)
        parts.append((o, l))
    (oa, la), (ob, lb) = parts
    _C.merge_attn_states(oa, la, oa, la, ob, lb)
    assert rel_l2(oa, whole) <= 3e-3
    assert (la - lse_w).abs().max().item() <= 5e-5


def test_cfg_parallel_partial_product_is_the_eager_bf16_product(device):
    """The per-rank contribution of CFG parallelism, s * pred in bf16 (denoising.py:1438-1444),
    through the guidance-combine kernel with a zero second operand."""
    from sglang_diffusion_amd.runtime.parallel import _hip_scale

    x = (torch.randn(3, 4096, 64, generator=g(95)) * 2).bfloat16().to(device)
    for s in (3.5, 1.0 - 3.5, 0.0, 1.0):
        assert torch.equal(_hip_scale(x, s), s * x)


@pytest.mark.parametrize("B,S,H,D,split", [(2, 200, 3, 128, 37), (1, 130, 2, 64, 0), (1, 96, 2, 128, 96)])
def test_attention_with_query_transform_equals_the_separate_pass(device, B, S, H, D, split):
    """RMSNorm + interleaved RoPE applied to Q inside the attention kernel == the standalone
    qk_norm_rope pass on q followed by plain attention, bit for bit; qk_norm_rope with q = None
    touches only k."""
    qkv = (torch.randn(B, S, 3, H, D, generator=g(101)) * 1.3).bfloat16().to(device)
    w = [(torch.rand(D, generator=g(102 + i)) + 0.5).to(device) for i in range(4)]
    ang = torch.rand(S, D // 2, generator=g(110)) * 6.28
    cos, sin = torch.cos(ang).to(device), torch.sin(ang).to(device)
    scale = D ** -0.5
    # reference: both sides in place, then attention
    ref = qkv.clone()
    _C.qk_norm_rope(ref[:, :, 0], ref[:, :, 1], w[0], w[1], w[2], w[3], split, 1e-6, cos, sin, True)
    v16 = _C.value_to_f16(ref[:, :, 2])
    o_ref = torch.empty(B, S, H, D, dtype=torch.bfloat16, device=device)
    l_ref = torch.empty(B, H, S, dtype=torch.float32, device=device)
    _C.flash_attn_fwd(ref[:, :, 0], ref[:, :, 1], v16, o_ref, l_ref, scale, False, True)
    # fused: k only in place, q transformed on load
    got = qkv.clone()
    _C.qk_norm_rope(None, got[:, :, 1], w[0], w[1], w[2], w[3], split, 1e-6, cos, sin, True)
    assert torch.equal(got[:, :, 1], ref[:, :, 1]) and torch.equal(got[:, :, 0], qkv[:, :, 0])
    o = torch.empty_like(o_ref)
    l = torch.empty_like(l_ref)
    _C.flash_attn_fwd(got[:, :, 0], got[:, :, 1], v16, o, l, scale, False, True,
                      q_norm=(w[0], w[2], split, 1e-6), q_rope=(cos, sin))
    assert torch.equal(o, o_ref) and torch.equal(l, l_ref)
    assert torch.equal(got[:, :, 0], qkv[:, :, 0]), "q in memory must stay untouched"
    # each half alone
    only_norm = qkv.clone()
    _C.qk_norm_rope(only_norm[:, :, 0], only_norm[:, :, 1], w[0], w[1], w[2], w[3], split, 1e-6,
                    None, None, True)
    o1 = torch.empty_like(o_ref)
    _C.flash_attn_fwd(only_norm[:, :, 0], got[:, :, 1], v16, o1, None, scale, False, True)
    o2 = torch.empty_like(o_ref)
    _C.flash_attn_fwd(qkv[:, :, 0], got[:, :, 1], v16, o2, None, scale, False, True,
                      q_norm=(w[0], w[2], split, 1e-6))
    assert torch.equal(o1, o2)
    with pytest.raises(RuntimeError):
        _C.flash_attn_fwd(qkv[:, :, 0], got[:, :, 1], v16, o2, None, scale, False, True,
                          q_rope=(cos[:-1], sin[:-1]))


# ------------------------------------------------------------------------- norm / rope --
@pytest.mark.parametrize("rows,dim,wf32", [(37, 128, True), (5, 3072, False), (1000, 128, True)])
def test_rmsnorm_vs_oracle(device, rows, dim, wf32):
    x = torch.randn(rows, dim, generator=g(51)).bfloat16()
    w = torch.rand(dim, generator=g(52)) + 0.5
    w = w if wf32 else w.bfloat16()
    ref = ops.rmsnorm(x, w, 1e-6)
    out = torch.empty_like(x, device=device)
    _C.rmsnorm(out, x.to(device), w.to(device), 1e-6)
    assert bf16_ulp_diff(out, ref).max() <= 1
    assert mismatch_frac(out, ref) < 5e-3
    r = torch.randn(rows, dim, generator=g(53)).bfloat16()
    fo, fr = ops.fused_add_rmsnorm(x, r, w, 1e-6)
    xd, rd = x.to(device).clone(), r.to(device).clone()
    _C.fused_add_rmsnorm(xd, rd, w.to(device), 1e-6)
    assert torch.equal(rd.cpu(), fr)
    assert bf16_ulp_diff(xd, fo).max() <= 1


@pytest.mark.parametrize("D,interleaved", [(128, True), (64, True), (128, False)])
def test_qk_norm_rope_vs_oracle(device, D, interleaved):
    B, S, H, Tt = 2, 45, 3, 13
    q = torch.randn(B, S, H, D, generator=g(61)).bfloat16()
    k = torch.randn(B, S, H, D, generator=g(62)).bfloat16()
    w = [torch.rand(D, generator=g(63 + i)) + 0.5 for i in range(4)]  # q_txt,k_txt,q_img,k_img
    ang = torch.rand(S, D // 2, generator=g(70)) * 6.28
    cos, sin = ang.cos(), ang.sin()
    qr, kr = q.clone(), k.clone()
    qt, kt = ops.qk_norm_rope(q[:, :Tt], k[:, :Tt], w[0], w[1], 1e-6, None, None)
    qi, ki = ops.qk_norm_rope(q[:, Tt:], k[:, Tt:], w[2], w[3], 1e-6, None, None)
    qr, kr = torch.cat([qt, qi], 1), torch.cat([kt, ki], 1)
    fn = ops.rope_interleaved if interleaved else ops.rope_neox
    qr, kr = fn(qr, cos, sin), fn(kr, cos, sin)
    qd, kd = q.to(device), k.to(device)
    _C.qk_norm_rope(qd, kd, *[t.to(device) for t in w], Tt, 1e-6, cos.to(device), sin.to(device),
                    interleaved)
    for got, ref in ((qd, qr), (kd, kr)):
        assert_close_bf16(got, ref, rel=1e-3, max_ulps=1.0, what="qk_norm_rope")
        assert mismatch_frac(got, ref) < 5e-3


def test_rotary_embedding_cache_op_vs_oracle(device):
    T, Hq, Hk, D, rot = 50, 4, 2, 128, 64
    q = torch.randn(T, Hq * D, generator=g(71)).bfloat16()
    k = torch.randn(T, Hk * D, generator=g(72)).bfloat16()
    pos = torch.randint(0, 100, (T,), generator=g(73))
    ang = torch.rand(100, rot // 2, generator=g(74)) * 6.28
    cache = torch.cat([ang.cos(), ang.sin()], -1)
    for neox in (False, True):
        qr, kr = ops.rotary_embedding_cache(pos, q, k, D, cache, neox)
        qd, kd = q.to(device).clone(), k.to(device).clone()
        _C.rotary_embedding(pos.to(device), qd, kd, D, cache.to(device), neox)
        assert_close_bf16(qd, qr, max_ulps=1.0)
        assert_close_bf16(kd, kr, max_ulps=1.0)


@pytest.mark.parametrize("B,T,D", [(2, 33, 3072), (1, 7, 256), (3, 5, 1536)])
def test_layernorm_modulate_vs_oracle(device, B, T, D):
    x = torch.randn(B, T, D, generator=g(81)).bfloat16() * 2 + 0.3
    shift = torch.randn(B, D, generator=g(82)).bfloat16()
    scale = torch.randn(B, D, generator=g(83)).bfloat16()
    ref = ops.layernorm_modulate(x, shift, scale)
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=device)
    _C.layernorm_modulate(out, x.view(B * T, D).to(device), shift.to(device), scale.to(device), B,
                          T, 1e-6)
    assert_close_bf16(out.view(B, T, D), ref, max_ulps=1.0, what="layernorm_modulate")
    assert mismatch_frac(out.view(B, T, D), ref) < 5e-3
    # plain LayerNorm (no modulation)
    ref0 = ops.layernorm_modulate(x, None, None)
    _C.layernorm_modulate(out, x.view(B * T, D).to(device), None, None, B, T, 1e-6)
    assert_close_bf16(out.view(B, T, D), ref0, max_ulps=1.0)
    # fused quantising variant == modulate then per-token quant of ITS bf16 output
    q = torch.empty(B * T, D, dtype=FP8, device=device)
    s = torch.empty(B * T, dtype=torch.float32, device=device)
    _C.layernorm_modulate(out, x.view(B * T, D).to(device), shift.to(device), scale.to(device), B,
                          T, 1e-6)
    _C.layernorm_modulate_quant(q, s, x.view(B * T, D).to(device), shift.to(device),
                                scale.to(device), B, T, 1e-6)
    qr, sr = ops.per_token_quant_fp8(out.cpu())
    assert torch.equal(s.cpu(), sr) and mismatch_frac(q, qr) == 0.0


def test_gate_residual_activation_exact(device):
    B, T, D = 2, 19, 512
    y = torch.randn(B, T, D, generator=g(91)).bfloat16()
    r = torch.randn(B, T, D, generator=g(92)).bfloat16()
    gate = torch.randn(B, D, generator=g(93)).bfloat16()
    ref = ops.gate_residual(r, gate, y)
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=device)
    _C.gate_residual(out, r.view(B * T, D).to(device), gate.to(device), y.view(B * T, D).to(device),
                     B, T)
    assert torch.equal(out.cpu().view(B, T, D), ref)
    x = (torch.randn(40, 256, generator=g(94)) * 3).bfloat16()
    o = torch.empty_like(x, device=device)
    _C.activation(o, x.to(device), 0)
    assert_close_bf16(o, torch.nn.functional.silu(x), max_ulps=1.0)
    assert mismatch_frac(o, torch.nn.functional.silu(x)) < 5e-3
    _C.activation(o, x.to(device), 1)
    assert_close_bf16(o, torch.nn.functional.gelu(x, approximate="tanh"), max_ulps=1.0)
    # eager tanh implementations differ by ~1 fp32 ulp among themselves, which moves a few
    # bf16 roundings in the cancelling tail (x << 0)
    assert mismatch_frac(o, torch.nn.functional.gelu(x, approximate="tanh")) < 1e-2


def test_timestep_embedding_vs_oracle(device):
    t = torch.tensor([1000.0, 999.0, 500.5, 12.25, 0.0, 3504.0])
    ref = ops.timestep_embedding(t, 256, True, 0.0)
    out = torch.empty(6, 256, dtype=torch.float32, device=device)
    _C.timestep_embedding(t.to(device), out, True, 0.0, 1.0, 10000.0)
    # reference tolerance: atol 1e-3 (jit_kernel/tests/test_timestep_embedding.py:64)
    assert (out.cpu() - ref).abs().max() <= 1e-3
    outb = torch.empty(6, 256, dtype=torch.bfloat16, device=device)
    _C.timestep_embedding(t.to(device), outb, True, 0.0, 1.0, 10000.0)
    assert bf16_ulp_diff(outb, ref.bfloat16()).float().mean() < 0.2


def test_euler_step_and_cfg_bit_exact(device):
    n = 4 * 4096 * 64
    lat = torch.randn(n, generator=g(95)).bfloat16()
    v = torch.randn(n, generator=g(96)).bfloat16()
    for sig, sn in ((1.0, 0.9837), (0.5, 0.4711), (0.0209, 0.0)):
        ref = ops.euler_step(lat, v, sig, sn)
        out = torch.empty_like(lat, device=device)
        _C.euler_step(out, lat.to(device), v.to(device), sig, sn)
        assert torch.equal(out.cpu(), ref)
    ref = ops.cfg_combine(lat, v, 3.5)
    out = torch.empty_like(lat, device=device)
    _C.cfg_combine(out, lat.to(device), v.to(device), 3.5)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("B,T,C", [(4, 4096, 64), (1, 96, 64), (3, 1000, 8)])
def test_cfg_normalize_and_guidance_rescale_vs_oracle(device, B, T, C):
    """The optional steps after the CFG combine.  Norms / stds are sums over up to 1M
    elements: the device's summation order is not eager's, so a factor can land on the other
    side of a bf16 rounding once in a while -- bit-equal in all but a tiny fraction of
    elements, never more than one ulp apart."""
    cond = torch.randn(B, T, C, generator=g(201)).bfloat16()
    uncond = (torch.randn(B, T, C, generator=g(202)) * 0.9 + 0.05).bfloat16()
    noise = ops.cfg_combine(cond, uncond, 4.0)          # norm well above cond's
    # --- norm cap: active, then inactive (factor large enough), then disabled
    for factor, active in ((1.0, True), (0.7, True), (50.0, False), (0.0, False)):
        ref = ops.cfg_normalize(noise, cond, factor)
        got = noise.to(device).clone()
        _C.cfg_normalize(got, cond.to(device), factor)
        if not active:
            assert torch.equal(got.cpu(), noise)
            assert torch.equal(ref, noise)
        else:
            assert not torch.equal(ref, noise)
            assert bf16_ulp_diff(got, ref).max() <= 1 and mismatch_frac(got, ref) < 0.01
            # the cap holds
            n_got = torch.linalg.vector_norm(got.float().cpu())
            assert n_got <= torch.linalg.vector_norm(cond.float()) * factor * 1.01
    # --- std rescale, per sample
    for gr in (0.7, 1.0, 0.25):
        ref = ops.rescale_noise_cfg(noise, cond, gr)
        got = noise.to(device).clone()
        _C.guidance_rescale(got, cond.to(device), gr)
        assert bf16_ulp_diff(got, ref).max() <= 1 and mismatch_frac(got, ref) < 0.01
    got = noise.to(device).clone()
    _C.guidance_rescale(got, cond.to(device), 0.0)
    assert torch.equal(got.cpu(), noise)
    # samples are independent: sample 0 alone gives sample 0 of the batch
    if B > 1:
        full = noise.to(device).clone()
        _C.guidance_rescale(full, cond.to(device), 0.7)
        one = noise[:1].to(device).clone()
        _C.guidance_rescale(one, cond[:1].to(device), 0.7)
        assert torch.equal(full[:1], one)


@pytest.mark.parametrize("rows,dim", [(5, 256), (77, 3072), (3, 8)])
def test_layernorm_affine_vs_eager(device, rows, dim):
    """nn.LayerNorm with affine parameters (text token refiner): fp32 statistics, one rounding."""
    x = (torch.randn(rows, dim, generator=g(301)) * 2 + 0.3).bfloat16()
    w = (torch.randn(dim, generator=g(302)) * 0.2 + 1.0).bfloat16()
    b = (torch.randn(dim, generator=g(303)) * 0.1).bfloat16()
    ref = torch.nn.functional.layer_norm(x, (dim,), w, b, 1e-6)
    out = torch.empty(rows, dim, dtype=torch.bfloat16, device=device)
    _C.layernorm_affine(out, x.to(device), w.to(device), b.to(device), 1e-6)
    assert_close_bf16(out, ref, rel=1e-3, max_ulps=1.0, what="layernorm_affine")
    assert mismatch_frac(out, ref) < 0.01


def test_token_mean_vs_eager(device):
    x = torch.randn(3, 21, 192, generator=g(304)).bfloat16()
    view = x[:, 1:]                                   # the reference averages enc[:, 1:]
    out = torch.empty(3, 192, dtype=torch.bfloat16, device=device)
    _C.token_mean(out, x.to(device)[:, 1:])
    ref = torch.mean(view, dim=1)
    assert_close_bf16(out, ref, rel=1e-3, max_ulps=1.0, what="token_mean")


def test_empty_inputs_are_no_ops(device):
    """Zero rows / zero tokens / zero elements: every entry point returns without launching
    (an empty batch reaches the kernels when a scheduler step has nothing for a rank)."""
    bf = dict(dtype=torch.bfloat16, device=device)
    e2 = torch.empty(0, 256, **bf)
    w = torch.zeros(128, 256, **bf)
    out = torch.empty(0, 128, **bf)
    _C.linear_bf16(e2, w, None, out)
    _C.fp8_scaled_mm(torch.empty(0, 256, dtype=FP8, device=device),
                     torch.zeros(128, 256, dtype=FP8, device=device),
                     torch.empty(0, device=device), torch.ones(128, device=device), None, out)
    _C.layernorm_modulate(torch.empty(0, 256, **bf), e2, torch.zeros(1, 256, **bf),
                          torch.zeros(1, 256, **bf), 0, 7, 1e-6)
    _C.layernorm_affine(torch.empty(0, 256, **bf), e2, torch.ones(256, **bf), torch.zeros(256, **bf), 1e-6)
    q = torch.empty(0, 5, 2, 128, **bf)
    _C.qk_norm_rope(q, q.clone(), None, None, None, None, 0, 1e-6, None, None, True)
    _C.flash_attn_fwd(q, q, q, torch.empty_like(q), None, 0.1)
    z = torch.empty(0, **bf)
    _C.cfg_combine(z, z, z, 3.5)
    _C.euler_step(z, z, z, 1.0, 0.9)
    _C.cfg_normalize(torch.empty(0, 8, **bf), torch.empty(0, 8, **bf), 1.0)
    _C.guidance_rescale(torch.empty(0, 8, **bf), torch.empty(0, 8, **bf), 0.7)
    torch.cuda.synchronize()


# ----------------------------------------------------------------- switches and guards --
@pytest.mark.parametrize("half_v", [True, False])
@pytest.mark.parametrize("B,S,Sk,H", [(1, 200, 333, 2),      # 128-row workgroups, ragged last tile
                                      (4, 1000, 1000, 24),   # 256-row workgroups (a workgroup per CU)
                                      (1, 64, 1, 1), (2, 257, 129, 3)])
def test_attention_asm_reads_and_builtin_reads_give_the_same_bits(device, B, S, Sk, H, half_v,
                                                                 monkeypatch):
    """Head size 128 reads its K and V fragments from LDS with inline asm and hand-counted waits
    by default; MI355_ATTN_ASMV=0 (read when the library is loaded; `reload_env` re-reads it)
    selects the compiler-scheduled builtin reads.  Both forms must produce the same output and
    log-sum-exp bits on both workgroup widths, both P.V precisions and ragged key counts."""
    D = 128
    gen = torch.Generator(device=device).manual_seed(41)
    q = (torch.randn(B, S, H, D, generator=gen, device=device) * 1.3).bfloat16()
    k = (torch.randn(B, Sk, H, D, generator=gen, device=device) * 1.3).bfloat16()
    v = (torch.randn(B, Sk, H, D, generator=gen, device=device) * 1.3).bfloat16()
    vv = v.to(torch.float16) if half_v else v
    outs = []
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("MI355_ATTN_ASMV", mode)
            _C.reload_env()
            o = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device=device)
            lse = torch.empty(B, H, S, dtype=torch.float32, device=device)
            _C.flash_attn_fwd(q, k, vv, o, lse, D ** -0.5, v_is_f16=half_v)
            torch.cuda.synchronize()
            outs.append((o, lse))
    finally:
        monkeypatch.delenv("MI355_ATTN_ASMV", raising=False)
        _C.reload_env()
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1], outs[1][1])


def test_half_output_is_refused_where_it_is_not_built(device):
    """out_f16 belongs to the plain fp8 GEMMs.  A bf16 problem must be refused on EVERY path,
    the small-M (<= 16 rows) weight-streaming one included, instead of silently writing bf16."""
    for M in (4, 64):
        x = torch.randn(M, 128, device=device).bfloat16()
        w = torch.randn(32, 128, device=device).bfloat16()
        out = torch.zeros(M, 32, dtype=torch.float16, device=device)
        with pytest.raises(RuntimeError, match="half output"):
            _C.linear_ex(x, w, out, M=M, N=32, K=128, lda=128, ldw=128, ldo=32, out_f16=True)
