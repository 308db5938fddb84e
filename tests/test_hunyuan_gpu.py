"""HunyuanVideo MM-DiT blocks (sequence-parallel configuration's model) on the HIP kernels
against the CPU oracle's restatement of the reference's eager path.  Parity is UNPINNED for
these blocks (see oracle/hunyuan_oracle.py): both sides restate the same reading of the
reference; the operators underneath are the ones pinned on the Flux path."""
import pytest
import torch

from oracle.hunyuan_oracle import HunyuanBlocksOracle
from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.hunyuanvideo import (MMDoubleStreamBlock, MMSingleStreamBlock,
                                                      process_weights_after_loading)
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu

HID, HEADS = 256, 2      # head size 128, as HunyuanVideo (3072 / 24)


def _inputs(B, Ti, Tt, seed=0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, Ti, HID, generator=g).bfloat16()
    txt = (torch.randn(B, Tt, HID, generator=g) * 0.5).bfloat16()
    vec = torch.randn(B, HID, generator=g).bfloat16()
    ang = torch.rand(Ti, HID // HEADS // 2, generator=g) * 6.28
    return img, txt, vec, torch.cos(ang), torch.sin(ang)


def _build(cls, device, fp8, prefix):
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    blk = cls(HID, HEADS, 4.0, dtype=torch.bfloat16, prefix=prefix, quant_config=qc)
    init_synthetic_weights(blk, seed=3, host_rng=True)
    state = {f"{prefix}.{k}": v.detach().clone() for k, v in blk.state_dict().items()}
    blk = blk.to(device)
    process_weights_after_loading(blk)
    return blk, state


@pytest.mark.parametrize("fp8", [False, True])
def test_double_stream_block_vs_oracle(device, fp8):
    blk, state = _build(MMDoubleStreamBlock, device, fp8, "double_blocks.0")
    orc = HunyuanBlocksOracle(state, num_heads=HEADS, fp8=fp8)
    img, txt, vec, cos, sin = _inputs(2, 96, 24)
    ri, rt = orc.double_block("double_blocks.0", img, txt, vec, cos, sin)
    with torch.no_grad():
        oi, ot = blk(img.to(device), txt.to(device), vec.to(device), (cos.to(device), sin.to(device)))
    ei, et = rel_l2(oi, ri), rel_l2(ot, rt)
    print(f"double block fp8={fp8}: img {ei:.2e} txt {et:.2e}")
    # one attention call per block: the per-operator budget (2e-3 attention, 1e-3 elsewhere).
    # W8A8: five quantised linears in a chain (qkv, proj, fc_in, fc_out); a one-ulp bf16
    # difference upstream moves an e4m3 rounding (3 mantissa bits) downstream, as on the Flux path
    tol = 2e-2 if fp8 else 4e-3
    assert ei <= tol and et <= tol
    if fp8:
        assert blk.img_mlp.fc_in.weight.dtype == torch.float8_e4m3fn
        assert blk.img_mod.linear.weight.dtype == torch.bfloat16


@pytest.mark.parametrize("fp8", [False, True])
def test_single_stream_block_vs_oracle(device, fp8):
    blk, state = _build(MMSingleStreamBlock, device, fp8, "single_blocks.0")
    orc = HunyuanBlocksOracle(state, num_heads=HEADS, fp8=fp8)
    img, txt, vec, cos, sin = _inputs(2, 80, 16, seed=1)
    x = torch.cat([img, txt], dim=1)
    ref = orc.single_block("single_blocks.0", x, vec, 16, cos, sin)
    with torch.no_grad():
        out = blk(x.to(device), vec.to(device), 16, (cos.to(device), sin.to(device)))
    e = rel_l2(out, ref)
    print(f"single block fp8={fp8}: {e:.2e}")
    assert e <= (2e-2 if fp8 else 4e-3)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B", [1, 2])
def test_fused_block_schedule_equals_the_layer_path(device, fp8, B):
    """The blocks' fused schedule (joint q|k|v buffer through row maps, V written in half by the
    projection, GELU and the gated residual in GEMM epilogues, LayerNorm + modulation + fp8
    quantisation in one pass, attention writing linear2's input in place) against the same
    blocks run module by module: the same kernels on the same values, so the same bits."""
    img, txt, vec, cos, sin = _inputs(B, 96, 24, seed=4)
    d = lambda t: t.to(device)  # noqa: E731
    fr = (d(cos), d(sin))
    blk, _ = _build(MMDoubleStreamBlock, device, fp8, "double_blocks.0")
    with torch.no_grad():
        assert blk.fused and blk._fused_kind() == ("fp8" if fp8 else "bf16")
        fi, ft = blk(d(img), d(txt), d(vec), fr)
        blk.fused = False
        li, lt = blk(d(img), d(txt), d(vec), fr)
    assert torch.equal(fi, li) and torch.equal(ft, lt)
    sblk, _ = _build(MMSingleStreamBlock, device, fp8, "single_blocks.0")
    x = torch.cat([img, txt], dim=1)
    with torch.no_grad():
        fo = sblk(d(x), d(vec), 24, fr)
        sblk.fused = False
        lo = sblk(d(x), d(vec), 24, fr)
    assert torch.equal(fo, lo)


def test_text_rows_are_not_rotated(device):
    """The joint rotation table's identity rows leave the text q / k bit for bit at their
    RMSNorm values: rotating the video angles must not change the text output's dependence
    on its own q, and a zero-length video part is plain attention over the text."""
    blk, state = _build(MMSingleStreamBlock, device, False, "single_blocks.0")
    orc = HunyuanBlocksOracle(state, num_heads=HEADS)
    _, txt, vec, _, _ = _inputs(1, 8, 32, seed=2)
    empty = torch.zeros(0, HID // HEADS // 2)
    ref = orc.single_block("single_blocks.0", txt, vec, 32, empty, empty)
    with torch.no_grad():
        out = blk(txt.to(device), vec.to(device), 32, (empty.to(device), empty.to(device)))
    assert rel_l2(out, ref) <= 4e-3


def test_fused_norm_layers_match_their_parts_and_the_oracle(device):
    """LayerNormScaleShift, ScaleResidualLayerNormScaleShift and MulAdd (the reference's layer
    names and call signatures): the one-pass residual + gate * x -> LayerNorm -> scale / shift
    kernel gives the bits of the separate gate-residual and norm-modulate kernels, and both
    agree with the oracle's eager restatement."""
    from sglang_diffusion_amd.layers.elementwise import MulAdd
    from sglang_diffusion_amd.layers.layernorm import (LayerNormScaleShift,
                                                       ScaleResidualLayerNormScaleShift)

    B, T, D = 2, 75, 384
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, D, generator=g).bfloat16()
    res = torch.randn(B, T, D, generator=g).bfloat16()
    gate, shift, scale = ((torch.randn(B, D, generator=g) * 0.5).bfloat16() for _ in range(3))
    O = HunyuanBlocksOracle
    want_res = O.mul_add(x, gate, res)
    want_out = O.norm_scale_shift(want_res, shift, scale)
    d = lambda t: t.to(device)  # noqa: E731
    norm, fused, muladd = LayerNormScaleShift(D), ScaleResidualLayerNormScaleShift(D), MulAdd()
    out, res_out = fused(d(res), d(x), d(gate), d(shift), d(scale))
    # parts: MulAdd then LayerNormScaleShift
    part_res = muladd(d(x), d(gate), d(res))
    part_out = norm(part_res, d(shift), d(scale))
    assert torch.equal(res_out, part_res) and torch.equal(out, part_out)
    assert torch.equal(res_out.cpu(), want_res)
    assert rel_l2(out, want_out) <= 1e-3
    # [B, 1, D] modulation rows and the integer gate of the cross-attention call sites
    out3, res3 = fused(d(res), d(x), d(gate)[:, None], d(shift)[:, None], d(scale)[:, None])
    assert torch.equal(out3, out) and torch.equal(res3, res_out)
    out1, res1 = fused(d(res), d(x), 1, d(shift), d(scale))
    assert torch.equal(res1.cpu(), res + x)
    assert torch.equal(out1, norm(res1, d(shift), d(scale)))
    with pytest.raises(ValueError):
        fused(d(res), d(x), 2, d(shift), d(scale))
    with pytest.raises(NotImplementedError):
        fused(d(res), d(x), d(gate)[:, None, None].expand(B, 3, 1, D), d(shift), d(scale))
    with pytest.raises(NotImplementedError):
        LayerNormScaleShift(D, elementwise_affine=True)


@pytest.mark.parametrize("fp8", [False, True])
def test_whole_video_dit_vs_oracle(device, fp8):
    """HunyuanVideoTransformer3DModel (patch embedding, text token refiner, time / pooled text /
    guidance modulation, double + single stream blocks, final layer, unpatchify) at a small
    size with head size 128 and 3-D rotary axes (16, 56, 56), against the oracle's eager
    restatement.  Parity UNPINNED as for the blocks (no reference fixture)."""
    from oracle.hunyuan_oracle import HunyuanVideoOracle
    from sglang_diffusion_amd.layers.attention import Mi355AttentionMetadataBuilder
    from sglang_diffusion_amd.layers.forward_context import set_forward_context
    from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig,
                                                          HunyuanVideoTransformer3DModel)

    # The bf16 case checks the structure.  W8A8 at a toy width only checks that the same layers
    # are quantised: an input that differs by one bf16 ulp flips an e4m3 code now and then, a
    # flip is a 6-12 % step, few channels average it, and eight quantised GEMMs in a row stack
    # it (a block alone, fed identical inputs, already sits at 1-2e-2 here); a layer
    # quantised on one side only shows as > 1e-1
    heads, nl = 4, (1 if fp8 else 2)
    cfg = HunyuanVideoArchConfig(num_attention_heads=heads, num_layers=nl, num_single_layers=nl,
                                 text_embed_dim=192, pooled_projection_dim=64, guidance_embeds=True)
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    model = HunyuanVideoTransformer3DModel(cfg, quant_config=qc)
    init_synthetic_weights(model, seed=5, host_rng=True)
    g = torch.Generator().manual_seed(8)
    for n, p in model.named_parameters():          # affine LayerNorm parameters away from (1, 0)
        if ".norm1." in n or ".norm2." in n:
            p.data = (torch.randn(p.shape, generator=g) * 0.1 + (1.0 if n.endswith("weight") else 0.0)
                      ).to(p.dtype)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(device)
    process_weights_after_loading(model)
    B, T, Hh, Ww, L = 2, 3, 8, 12, 20
    x = torch.randn(B, 16, T, Hh, Ww, generator=g).bfloat16()
    enc = (torch.randn(B, 1 + L, 192, generator=g) * 0.5).bfloat16()
    t = torch.tensor([900.0, 417.0])
    guid = torch.tensor([6016.0, 6016.0]).bfloat16()
    orc = HunyuanVideoOracle(state, num_heads=heads, num_layers=nl, num_single_layers=nl,
                             pooled_dim=64, guidance_embeds=True, fp8=fp8)
    ref = orc.forward(x, enc, t, guid)
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)
    with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=md):
        out = model(x.to(device), enc.to(device), t.to(device), guid.to(device))
    assert out.shape == ref.shape == (B, 16, T, Hh, Ww)
    e = rel_l2(out, ref)
    print(f"whole video DiT fp8={fp8}: rel {e:.2e}")
    assert e <= (6e-2 if fp8 else 1e-2)


def test_rotary_table_shards_by_frames():
    """The (t, h, w) rotation table of a sequence-parallel rank is the slice of the full table
    that belongs to its frames (rotary_embedding/mrope.py:333-343)."""
    from sglang_diffusion_amd.models.hunyuanvideo import get_rotary_pos_embed

    full = get_rotary_pos_embed((4, 3, 5), (16, 56, 56), 256, "cpu")
    assert full[0].shape == (60, 64)
    for r in range(2):
        part = get_rotary_pos_embed((4, 3, 5), (16, 56, 56), 256, "cpu", sp_rank=r, sp_world_size=2)
        assert torch.equal(part[0], full[0][r * 30:(r + 1) * 30])
        assert torch.equal(part[1], full[1][r * 30:(r + 1) * 30])
    with pytest.raises(ValueError):
        get_rotary_pos_embed((3, 3, 5), (16, 56, 56), 256, "cpu", sp_rank=0, sp_world_size=2)


@pytest.mark.parametrize("fp8", [False, True])
def test_checkpoint_loads_into_hbm_and_runs(device, tmp_path, fp8):
    """A diffusers-named, sharded checkpoint goes file -> HBM through the rename / merge table
    and gives the bits of the model it was written from (W8A8: quantised on the device while
    loading on one side, after .to(device) on the other)."""
    from sglang_diffusion_amd.layers.attention import Mi355AttentionMetadataBuilder
    from sglang_diffusion_amd.layers.forward_context import set_forward_context
    from sglang_diffusion_amd.runtime.hunyuan_loader import load_hunyuan_transformer
    from tests.test_hunyuan_loader import CFG, _write_checkpoint

    src, _, path = _write_checkpoint(tmp_path)
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    loaded, report = load_hunyuan_transformer(path, device=device, quant_config=qc)
    assert report.unexpected_keys == [] and report.bytes_loaded > 0
    assert all(p.device.type == "cuda" for p in loaded.parameters())
    if fp8:
        from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig,
                                                              HunyuanVideoTransformer3DModel)
        twin = HunyuanVideoTransformer3DModel(HunyuanVideoArchConfig(**CFG), quant_config=qc)
        twin.load_state_dict(src.state_dict())
        src = twin
    src = src.to(device)
    process_weights_after_loading(src)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 16, 2, 8, 8, generator=g).bfloat16().to(device)
    enc = (torch.randn(1, 1 + 12, CFG["text_embed_dim"], generator=g) * 0.5).bfloat16().to(device)
    t = torch.tensor([500.0], device=device)
    guid = torch.tensor([6016.0], device=device).bfloat16()
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)
    with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=md):
        a = loaded(x, enc, t, guid)
        b = src(x, enc, t, guid)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def test_adjacent_stream_slices_reach_the_kernel_without_a_copy(device):
    """UlyssesAttention on one rank: when the replicated (text) tokens sit right behind the main
    tokens in one q|k|v buffer the kernel reads that buffer; separate tensors are concatenated.
    Same bits either way."""
    from sglang_diffusion_amd.layers.attention import (Mi355AttentionMetadataBuilder,
                                                       UlyssesAttention, _append_tokens)
    from sglang_diffusion_amd.layers.forward_context import set_forward_context

    B, Ti, Tt, H, D = 2, 200, 40, 2, 128
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, Ti + Tt, 3, H, D, generator=g).bfloat16().to(device)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    joint = _append_tokens(q[:, :Ti], q[:, Ti:])
    assert joint.data_ptr() == q.data_ptr() and joint.shape == q.shape and torch.equal(joint, q)
    apart = _append_tokens(q[:, :Ti], q[:, Ti:].clone())
    assert apart.data_ptr() != q.data_ptr() and torch.equal(apart, q)
    # a gap between the two slices, or a different head subset, is not adjacency
    assert _append_tokens(q[:, :Ti - 1], q[:, Ti:]).shape[1] == Ti + Tt - 1
    assert _append_tokens(q[:, :Ti - 1], q[:, Ti:]).data_ptr() != q.data_ptr()
    attn = UlyssesAttention(num_heads=H, head_size=D)
    md = Mi355AttentionMetadataBuilder().build(current_timestep=0)
    with set_forward_context(current_timestep=0, attn_metadata=md):
        a_img, a_txt = attn(q[:, :Ti], k[:, :Ti], v[:, :Ti], q[:, Ti:], k[:, Ti:], v[:, Ti:])
        b_img, b_txt = attn(q[:, :Ti], k[:, :Ti], v[:, :Ti],
                            q[:, Ti:].clone(), k[:, Ti:].clone(), v[:, Ti:].clone())
    assert torch.equal(a_img, b_img) and torch.equal(a_txt, b_txt)
