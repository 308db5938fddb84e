"""The plugin interfaces (attention backend, quant method, linear, norms, rope) called the way
the reference's layers call them, checked against the oracle."""
import pytest
import torch

from oracle import ops
from tests.helpers import assert_close_bf16, bf16_ulp_diff, rel_l2

pytestmark = pytest.mark.gpu


def test_attention_backend_api(device):
    from sglang_diffusion_amd.layers.attention import (AttentionBackendEnum, LocalAttention,
                                                       Mi355AttentionBackend, USPAttention,
                                                       get_attn_backend)
    from sglang_diffusion_amd.layers.forward_context import set_forward_context

    be = get_attn_backend(128, torch.bfloat16)
    assert be is Mi355AttentionBackend and be.get_enum() == AttentionBackendEnum.AITER
    with pytest.raises(ValueError):
        get_attn_backend(128, torch.bfloat16, {AttentionBackendEnum.SLIDING_TILE_ATTN})
    impl = be.get_impl_cls()(num_heads=4, head_size=128, softmax_scale=128 ** -0.5, causal=False)
    q, k, v = (torch.randn(2, 70, 4, 128).bfloat16() for _ in range(3))
    md = be.get_builder_cls()().build(current_timestep=3)
    out, lse = impl.forward(q.to(device), k.to(device), v.to(device), md, return_softmax_lse=True)
    ref, lref = ops.attention(q, k, v, return_lse=True)
    assert rel_l2(out, ref) <= 4e-3 and (lse.cpu() - lref).abs().max() <= 1e-4
    layer = USPAttention(num_heads=4, head_size=128, causal=False)
    with set_forward_context(current_timestep=0, attn_metadata=md):
        o2 = layer(q.to(device), k.to(device), v.to(device))
    assert torch.equal(o2, out)
    o3 = LocalAttention(num_heads=4, head_size=128)(q.to(device), k.to(device), v.to(device))
    assert torch.equal(o3, out)
    with pytest.raises(NotImplementedError):
        be.get_impl_cls()(num_heads=4, head_size=128, softmax_scale=1.0, causal=True)


def test_fp8_linear_method_per_channel_and_block(device):
    from sglang_diffusion_amd.layers.linear import ColumnParallelLinear
    from sglang_diffusion_amd.layers.quantization import (Fp8Config, Fp8LinearMethod,
                                                          UnquantizedLinearMethod,
                                                          get_quantization_config)

    assert get_quantization_config("fp8") is Fp8Config
    cfg = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "dynamic",
                                 "modules_to_not_convert": ["model.proj_out"]})
    assert cfg.is_checkpoint_fp8_serialized and cfg.ignored_layers == ["proj_out"]
    w = (torch.randn(384, 256) * 256 ** -0.5).bfloat16()
    b = (torch.randn(384) * 0.1).bfloat16()
    x = torch.randn(2, 50, 256).bfloat16()
    # (a) bf16 checkpoint quantised online
    qc = Fp8Config(is_checkpoint_fp8_serialized=False)
    lin = ColumnParallelLinear(256, 384, bias=True, gather_output=True, params_dtype=torch.bfloat16,
                               quant_config=qc, prefix="blk.attn.to_q")
    assert isinstance(lin.quant_method, Fp8LinearMethod)
    lin.weight.data.copy_(w)
    lin.bias.data.copy_(b)
    lin = lin.to(device)
    lin.quant_method.process_weights_after_loading(lin)
    y, ob = lin(x.to(device))
    assert ob is None and y.shape == (2, 50, 384)
    wq, ws = ops.quantize_weight_per_channel(w)
    assert rel_l2(y, ops.fp8_linear(x, wq, ws, b)) <= 1e-3
    # (b) ignored layer stays bf16
    lin2 = ColumnParallelLinear(256, 384, quant_config=cfg, prefix="proj_out",
                                params_dtype=torch.bfloat16)
    assert isinstance(lin2.quant_method, UnquantizedLinearMethod)
    # (c) 128x128 block-quantised fp8 checkpoint
    qcb = Fp8Config(is_checkpoint_fp8_serialized=True, weight_block_size=[128, 128])
    lin3 = ColumnParallelLinear(256, 384, bias=True, params_dtype=torch.bfloat16, quant_config=qcb,
                                prefix="blk.attn.to_k")
    wqb, wsb = ops.quantize_weight_block(w)
    assert lin3.weight.dtype == torch.float8_e4m3fn and lin3.weight_scale_inv.shape == (3, 2)
    lin3.weight_loader(lin3.weight, wqb)
    lin3.weight_loader(lin3.weight_scale_inv, wsb)
    lin3.bias.data.copy_(b)
    lin3 = lin3.to(device)
    lin3.quant_method.process_weights_after_loading(lin3)
    y3, _ = lin3(x.to(device))
    assert rel_l2(y3, ops.fp8_block_linear(x, wqb, wsb, b)) <= 1e-3
    with pytest.raises(ValueError):
        Fp8Config(is_checkpoint_fp8_serialized=False, weight_block_size=[128, 128])


def test_rmsnorm_qknorm_rope_layers(device):
    from sglang_diffusion_amd.layers.layernorm import RMSNorm, apply_qk_norm
    from sglang_diffusion_amd.layers.rotary import apply_flashinfer_rope_qk_inplace

    nq, nk = RMSNorm(128, eps=1e-6).to(device), RMSNorm(128, eps=1e-6).to(device)
    nq.weight.data.copy_(torch.rand(128) + 0.5)
    nk.weight.data.copy_(torch.rand(128) + 0.5)
    q = torch.randn(2, 30, 3, 128).bfloat16()
    k = torch.randn(2, 30, 3, 128).bfloat16()
    qo, ko = apply_qk_norm(q.to(device), k.to(device), nq, nk, 128)
    assert bf16_ulp_diff(qo, ops.rmsnorm(q, nq.weight.cpu(), 1e-6)).max() <= 1
    assert bf16_ulp_diff(ko, ops.rmsnorm(k, nk.weight.cpu(), 1e-6)).max() <= 1
    x = torch.randn(9, 128).bfloat16()
    assert bf16_ulp_diff(nq(x.to(device)), ops.rmsnorm(x, nq.weight.cpu(), 1e-6)).max() <= 1
    ang = torch.rand(30, 64) * 6.28
    cache = torch.cat([ang.cos(), ang.sin()], -1)
    qr, kr = apply_flashinfer_rope_qk_inplace(q.to(device), k.to(device), cache.to(device),
                                              is_neox=False)
    assert_close_bf16(qr, ops.rope_interleaved(q, ang.cos(), ang.sin()), max_ulps=1.0)
    # ROCm-fallback semantics switch: cos/sin rounded to bf16 first (rotary_embedding/utils.py:80-83)
    qb, _ = apply_flashinfer_rope_qk_inplace(q.to(device), k.to(device), cache.to(device),
                                             is_neox=False, round_cos_sin_to_input_dtype=True)
    refb = ops.rope_interleaved(q, ang.cos().bfloat16(), ang.sin().bfloat16())
    assert_close_bf16(qb, refb, max_ulps=1.0)
    with pytest.raises(ValueError):
        apply_flashinfer_rope_qk_inplace(q.to(device)[0], k.to(device)[0], cache.to(device))
