"""Checkpoint loader: diffusers-layout directory -> FluxTransformer2DModel.

The checkpoints are synthetic (written here with safetensors; no Flux weights ship with the
reference and there is no network), laid out as the reference's loader expects them
(loader/component_loaders/transformer_loader.py:122-199): config.json + *.safetensors, names
optionally prefixed with "transformer.", quantisation declared in config.json or in the
safetensors header."""
import json
import os

import pytest
import torch

from sglang_diffusion_amd.layers.quantization import Fp8Config
from sglang_diffusion_amd.models.flux import FluxArchConfig, FluxTransformer2DModel
from sglang_diffusion_amd.runtime import loader as L
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights
from tests.helpers import assert_close_bf16
from tests.flux_small import SMALL, build_small, build_small_block_fp8, small_inputs


def _plain_state(guidance=True, seed=0):
    cfg = FluxArchConfig(**SMALL, guidance_embeds=guidance)
    m = FluxTransformer2DModel(cfg, quant_config=None)
    init_synthetic_weights(m, seed=seed, host_rng=True)
    return cfg, {k: v.detach().clone() for k, v in m.state_dict().items()}


# ------------------------------------------------------------------ host logic (no GPU) --
def test_roundtrip_bf16_single_file_cpu(tmp_path):
    cfg, state = _plain_state()
    files = L.save_flux_checkpoint(str(tmp_path), cfg, state)
    assert [os.path.basename(f) for f in files] == ["diffusion_pytorch_model.safetensors"]
    model, rep = L.load_flux_transformer(str(tmp_path), device="cpu")
    assert rep.missing_keys == [] and rep.unexpected_keys == []
    assert rep.bytes_loaded == sum(v.numel() * v.element_size() for v in state.values())
    got = model.state_dict()
    assert set(got) == set(state)
    for k, v in state.items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    assert all(not p.requires_grad for p in model.parameters())
    assert model.config.num_layers == SMALL["num_layers"]
    assert model.config.axes_dims_rope == cfg.axes_dims_rope


def test_sharded_prefixed_and_fp32_checkpoint_cpu(tmp_path):
    """Several shards, names carrying the whole-pipeline prefix, tensors stored in fp32: the
    loader strips the prefix (configs/models/dits/flux.py:28-33) and casts to the parameter's
    dtype (fsdp_load.py:262-268)."""
    cfg, state = _plain_state(guidance=False)
    wide = {"transformer." + k: v.float() for k, v in state.items()}
    files = L.save_flux_checkpoint(str(tmp_path), cfg, wide, shard_bytes=1 << 20)
    assert len(files) > 2
    assert os.path.isfile(tmp_path / "diffusion_pytorch_model.safetensors.index.json")
    model, rep = L.load_flux_transformer(str(tmp_path), device="cpu")
    assert not model.config.guidance_embeds
    got = model.state_dict()
    for k, v in state.items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k


def test_missing_unexpected_and_shape_errors_cpu(tmp_path):
    cfg, state = _plain_state()
    # a parameter the checkpoint lacks
    short = dict(state)
    gone = next(k for k in short if k.endswith("to_q.weight"))
    del short[gone]
    d1 = tmp_path / "missing"
    L.save_flux_checkpoint(str(d1), cfg, short)
    with pytest.raises(ValueError, match="missing from the checkpoint"):
        L.load_flux_transformer(str(d1), device="cpu")
    # a tensor the model does not have: reported, or an error under strict
    extra = dict(state)
    extra["transformer_blocks.0.attn.lora_down.weight"] = torch.zeros(4, 4)
    d2 = tmp_path / "extra"
    L.save_flux_checkpoint(str(d2), cfg, extra)
    _, rep = L.load_flux_transformer(str(d2), device="cpu")
    assert rep.unexpected_keys == ["transformer_blocks.0.attn.lora_down.weight"]
    with pytest.raises(ValueError, match="not found in custom model state dict"):
        L.load_flux_transformer(str(d2), device="cpu", strict=True)
    # a wrong shape
    bad = dict(state)
    bad[gone] = state[gone][:, :-1].contiguous()
    d3 = tmp_path / "shape"
    L.save_flux_checkpoint(str(d3), cfg, bad)
    with pytest.raises((ValueError, AssertionError)):
        L.load_flux_transformer(str(d3), device="cpu")
    # directory problems
    with pytest.raises(FileNotFoundError):
        L.load_flux_transformer(str(tmp_path / "nowhere"), device="cpu")
    d4 = tmp_path / "noweights"
    os.makedirs(d4)
    (d4 / "config.json").write_text(json.dumps({"_class_name": "FluxTransformer2DModel"}))
    with pytest.raises(ValueError, match="no safetensors files"):
        L.load_flux_transformer(str(d4), device="cpu")
    (d4 / "config.json").write_text(json.dumps({"_class_name": "WanTransformer3DModel"}))
    with pytest.raises(ValueError, match="WanTransformer3DModel"):
        L.load_flux_transformer(str(d4), device="cpu")


def test_arch_config_from_diffusers_keys():
    """The published Flux.1-dev transformer config.json (keys as diffusers writes them)."""
    hf = {"_class_name": "FluxTransformer2DModel", "_diffusers_version": "0.30.0.dev0",
          "_name_or_path": "../checkpoints/flux-dev/transformer", "attention_head_dim": 128,
          "guidance_embeds": True, "in_channels": 64, "joint_attention_dim": 4096,
          "num_attention_heads": 24, "num_layers": 19, "num_single_layers": 38, "patch_size": 1,
          "pooled_projection_dim": 768, "axes_dims_rope": [16, 56, 56]}
    cfg = L.arch_config_from_hf(hf)
    assert cfg == FluxArchConfig()
    assert cfg.hidden_size == 3072 and cfg.axes_dims_rope == (16, 56, 56)
    schnell = L.arch_config_from_hf({**hf, "guidance_embeds": False})
    assert not schnell.guidance_embeds
    with pytest.raises(ValueError, match="axes_dims_rope"):
        L.arch_config_from_hf({**hf, "axes_dims_rope": [16, 56, 48]})


def test_quant_config_sources(tmp_path):
    """config.json first (hf_diffusers_utils.py:352-378), then the safetensors header
    (hf_diffusers_utils.py:911-945), including the diffusers per-layer format."""
    assert L.get_quant_config({"num_layers": 2}) is None
    qc = L.get_quant_config({"quantization_config": {
        "quant_method": "fp8", "activation_scheme": "dynamic", "weight_block_size": [128, 128],
        "modules_to_not_convert": ["model.proj_out"]}})
    assert isinstance(qc, Fp8Config) and qc.is_checkpoint_fp8_serialized
    assert qc.weight_block_size == [128, 128] and qc.ignored_layers == ["proj_out"]
    with pytest.raises(ValueError):
        L.get_quant_config({"quantization_config": {"quant_method": "awq-marlin"}})

    from safetensors.torch import save_file
    t = {"x": torch.zeros(2)}
    p0 = str(tmp_path / "plain.safetensors")
    save_file(t, p0)
    assert L.read_safetensors_metadata(p0) == {}
    assert L.get_quant_config_from_safetensors_metadata(p0) is None
    p1 = str(tmp_path / "explicit.safetensors")
    save_file(t, p1, metadata={"_quantization_metadata": json.dumps(
        {"quant_method": "fp8", "activation_scheme": "dynamic"})})
    q1 = L.get_quant_config_from_safetensors_metadata(p1)
    assert isinstance(q1, Fp8Config) and q1.weight_block_size is None
    p2 = str(tmp_path / "perlayer.safetensors")
    save_file(t, p2, metadata={"_quantization_metadata": json.dumps(
        {"format_version": "1.0", "layers": {"a.b": {"format": "float8_e4m3fn"}}})})
    q2 = L.get_quant_config_from_safetensors_metadata(p2)
    assert isinstance(q2, Fp8Config) and q2.activation_scheme == "dynamic"
    p3 = str(tmp_path / "garbage.safetensors")
    save_file(t, p3, metadata={"_quantization_metadata": "{not json"})
    assert L.get_quant_config_from_safetensors_metadata(p3) is None
    p4 = tmp_path / "truncated.safetensors"
    p4.write_bytes(b"\xff" * 8 + b"{}")
    with pytest.raises(ValueError, match="header length"):
        L.read_safetensors_metadata(str(p4))


def test_param_name_mapping():
    assert L.map_param_name("transformer.x_embedder.weight") == "x_embedder.weight"
    assert (L.map_param_name("transformer.transformer_blocks.3.attn.to_q.weight")
            == "transformer_blocks.3.attn.to_q.weight")
    assert (L.map_param_name("single_transformer_blocks.0.proj_out.bias")
            == "single_transformer_blocks.0.proj_out.bias")


def test_converter_matches_oracle_and_model_layout_cpu(tmp_path):
    """The offline fp8 converter (convert_hf_to_fp8.py): which tensors it quantises, the codes and
    scales it writes (against the oracle's restatement of the block rule), the config it leaves,
    and that a model built from that config expects exactly the tensors the directory holds."""
    from oracle import ops as O
    from safetensors import safe_open
    from sglang_diffusion_amd.tools import convert_hf_to_fp8 as C

    cfg, state = _plain_state()
    src, dst = tmp_path / "bf16", tmp_path / "fp8"
    L.save_flux_checkpoint(str(src), cfg, state, shard_bytes=2 << 20)
    qd = C.convert_fp8(str(src), str(dst), "block", [128, 128], device=torch.device("cpu"))
    assert qd["quant_method"] == "fp8" and qd["activation_scheme"] == "dynamic"
    assert qd["weight_block_size"] == [128, 128]
    assert "proj_out" in qd["modules_to_not_convert"]
    assert "transformer_blocks.0.ff.net.0.proj" in qd["modules_to_not_convert"]
    assert json.loads((dst / "config.json").read_text())["quantization_config"] == qd
    got = {}
    for f in L.list_safetensors_files(str(dst)):
        with safe_open(f, framework="pt", device="cpu") as h:
            got.update({k: h.get_tensor(k) for k in h.keys()})
    index = json.loads((dst / "model.safetensors.index.json").read_text())
    assert set(index["weight_map"]) == set(got)
    quantised = [k for k in state if C.is_quantised_key(k)]
    assert quantised and all(k.endswith(".weight") for k in quantised)
    for k, v in state.items():
        if k in quantised:
            q, s = O.quantize_weight_block(v)
            assert got[k].dtype == torch.float8_e4m3fn
            assert torch.equal(got[k].view(torch.uint8), q.view(torch.uint8)), k
            assert torch.equal(got[k.replace(".weight", ".weight_scale_inv")], s), k
        else:
            assert torch.equal(got[k], v), k
    # the model built from the written config holds an fp8 weight exactly where the converter
    # wrote one: the ignore list ("proj_out", "...ff.net.0.proj", ...) is matched by substring
    # against the prefixes the layers were built with (srt/layers/quantization/utils.py:80)
    model, rep = L.load_flux_transformer(str(dst), device="cpu", strict=True) \
        if False else (None, None)
    shell = FluxTransformer2DModel(cfg, quant_config=L.get_quant_config(
        json.loads((dst / "config.json").read_text())))
    params = dict(shell.named_parameters())
    assert set(params) == set(got)
    for k, p in params.items():
        assert p.dtype == got[k].dtype and p.shape == got[k].shape, k

    # tensor strategy: one scale of shape [1] per layer
    dst2 = tmp_path / "fp8_tensor"
    qd2 = C.convert_fp8(str(src), str(dst2), "tensor", None, device=torch.device("cpu"))
    assert "weight_block_size" not in qd2
    with safe_open(L.list_safetensors_files(str(dst2))[0], framework="pt", device="cpu") as h:
        names = set(h.keys())
    some = next(k for k in quantised if k in names)
    with safe_open(L.list_safetensors_files(str(dst2))[0], framework="pt", device="cpu") as h:
        s = h.get_tensor(some.replace(".weight", ".weight_scale"))
        q = h.get_tensor(some)
    assert s.shape == (1,) and torch.equal(s, (state[some].abs().max().float() / 448.0).view(1))
    # the reference divides the bf16 weight by a 0-d fp32 scale: type promotion keeps bf16, so
    # the quotient is rounded to bf16 before it is rounded to e4m3 (convert_hf_to_fp8.py:90-95)
    want = (state[some] / s.reshape(())).clamp(-448, 448)
    assert want.dtype == torch.bfloat16
    assert torch.equal(q.view(torch.uint8), want.to(torch.float8_e4m3fn).view(torch.uint8))

    # an all-zero block: scale 0, zero codes (dequantises to the zeros it was), ragged edges
    w = torch.randn(200, 300).bfloat16()
    w[:128, 128:256] = 0
    q, s = C.block_fp8(w, [128, 128])
    assert s.shape == (2, 3) and s[0, 1] == 0 and not torch.isnan(q.float()).any()
    assert (q[:128, 128:256].float() == 0).all()
    rq, rs = O.quantize_weight_block(w)
    keep = torch.ones(200, 300, dtype=torch.bool)
    keep[:128, 128:256] = False
    assert torch.equal(q.view(torch.uint8)[keep], rq.view(torch.uint8)[keep]) and torch.equal(s, rs)

    with pytest.raises(ValueError, match="block-size"):
        C.convert_fp8(str(src), str(tmp_path / "x"), "block", None)
    with pytest.raises(ValueError, match="differ"):
        C.convert_fp8(str(src), str(src), "tensor", None)


# ------------------------------------------------------------------------- on the GPU --
def _run(model, cfg, device, seed=3, fused=None):
    """(module output, fused output); fused=False / True runs only that path."""
    from sglang_diffusion_amd.runtime import pipeline_config as pc

    B, Hpx, Wpx, Tt = 2, 96, 64, 16
    Ti = (Hpx // 16) * (Wpx // 16)
    hid, enc, pooled, t, guid = small_inputs(B, Ti, Tt, cfg, seed=seed)
    fr = pc.get_freqs_cis(Tt, Wpx, Hpx, device, model.rotary_emb)
    args = (hid.to(device), enc.to(device), pooled.to(device), t.to(device), guid.to(device), fr)
    with torch.no_grad():
        if fused is not None:
            return ((model.forward_fused if fused else model)(*args).clone(),)
        return model(*args).clone(), model.forward_fused(*args).clone()


@pytest.mark.gpu
@pytest.mark.parametrize("fp8", [False, True])
def test_loaded_bf16_checkpoint_equals_directly_built_model(device, tmp_path, fp8):
    """A bf16 checkpoint, loaded plain or quantised to W8A8 on the device while loading, gives
    the very bits of the model whose parameters were set in place."""
    ref_model, state, cfg = build_small(device, fp8=fp8)
    L.save_flux_checkpoint(str(tmp_path), cfg, state, shard_bytes=4 << 20)
    qc = Fp8Config(is_checkpoint_fp8_serialized=False, activation_scheme="dynamic") if fp8 else None
    model, rep = L.load_flux_transformer(str(tmp_path), device=device, quant_config=qc)
    assert rep.unexpected_keys == []
    assert all(p.device.type == "cuda" for p in model.parameters())
    if fp8:
        lin = model.transformer_blocks[0].attn.to_q
        assert lin.weight.dtype == torch.float8_e4m3fn and lin.weight_scale.dtype == torch.float32
    for got, want in zip(_run(model, cfg, device), _run(ref_model, cfg, device)):
        assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["config", "header"])
def test_loaded_block_fp8_checkpoint_equals_directly_built_model(device, tmp_path, where):
    """An fp8-serialised checkpoint (e4m3 weights + weight_scale_inv per 128x128 block), its
    quantisation declared in config.json or only in the safetensors header of a separate
    weights file (the reference's --transformer-weights-path route)."""
    ref_model, _, cfg = build_small_block_fp8(device)
    state = {k: v.detach().cpu() for k, v in ref_model.state_dict().items()}
    assert any(k.endswith("weight_scale_inv") for k in state)
    qd = {"quant_method": "fp8", "activation_scheme": "dynamic", "weight_block_size": [128, 128]}
    if where == "config":
        L.save_flux_checkpoint(str(tmp_path), cfg, state, quantization_config=qd)
        model, rep = L.load_flux_transformer(str(tmp_path), device=device)
    else:
        base, wdir = tmp_path / "base", tmp_path / "quantised"
        _, plain = _plain_state()
        L.save_flux_checkpoint(str(base), cfg, plain)
        L.save_flux_checkpoint(str(wdir), cfg, state,
                               metadata={"_quantization_metadata": json.dumps(qd)})
        model, rep = L.load_flux_transformer(str(base), device=device,
                                             transformer_weights_path=str(wdir))
    assert rep.unexpected_keys == []
    lin = model.single_transformer_blocks[0].proj_mlp
    assert lin.weight.dtype == torch.float8_e4m3fn and hasattr(lin, "weight_scale_inv")
    for got, want in zip(_run(model, cfg, device), _run(ref_model, cfg, device)):
        assert torch.equal(got, want)


@pytest.mark.gpu
def test_loaded_per_tensor_fp8_checkpoint(device, tmp_path):
    """An fp8-serialised checkpoint with ONE scale per layer (stored as a 0-d tensor, as
    offline converters write it): the scale is widened to per-channel after loading
    (fp8.py:392-396) and the layer computes x @ (q * s)^T with dynamic per-token activations."""
    from oracle import ops as O

    cfg, plain = _plain_state()
    qc = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="dynamic")
    shell = FluxTransformer2DModel(cfg, quant_config=qc)
    state = {}
    for name, p in shell.named_parameters():
        if name.endswith(".weight_scale"):
            continue
        if p.dtype == torch.float8_e4m3fn:
            w = plain[name].float()
            s = w.abs().max().clamp_min(1e-12) / 448.0
            state[name] = (w / s).clamp(-448, 448).to(torch.float8_e4m3fn)
            state[name[:-len("weight")] + "weight_scale"] = s.reshape(())
        else:
            state[name] = plain[name]
    L.save_flux_checkpoint(str(tmp_path), cfg, state, quantization_config={
        "quant_method": "fp8", "activation_scheme": "dynamic"})
    model, rep = L.load_flux_transformer(str(tmp_path), device=device)
    lin = model.transformer_blocks[1].attn.to_q
    name = next(n for n, m in model.named_modules() if m is lin)
    n_out = lin.weight.shape[0]
    assert lin.weight_scale.shape == (n_out,)
    assert torch.equal(lin.weight_scale.cpu(),
                       state[name + ".weight_scale"].reshape(1).expand(n_out).float())
    assert torch.equal(lin.weight.cpu().view(torch.uint8),
                       state[name + ".weight"].view(torch.uint8))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(48, lin.weight.shape[1], generator=g).bfloat16()
    y, _ = lin(x.to(device))
    bias = lin.bias.detach().cpu() if lin.bias is not None else None
    want = O.fp8_linear(x, state[name + ".weight"], lin.weight_scale.cpu(), bias)
    # same rounding points, different fp32 summation order inside the GEMM
    assert_close_bf16(y.cpu(), want, rel=1e-3, max_ulps=1.0, what="per-tensor fp8 linear")


@pytest.mark.gpu
def test_loaded_static_activation_scheme_checkpoint(device, tmp_path):
    """activation_scheme = "static": every quantised layer carries a calibrated per-tensor
    `input_scale` beside its per-tensor `weight_scale`; activations are quantised with that
    fixed scale (values beyond 448 * scale saturate) instead of per token."""
    from oracle import ops as O

    cfg, plain = _plain_state()
    qc = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
    shell = FluxTransformer2DModel(cfg, quant_config=qc)
    state = {}
    for name, p in shell.named_parameters():
        if name.endswith(".weight_scale") or name.endswith(".input_scale"):
            continue
        if p.dtype == torch.float8_e4m3fn:
            w = plain[name].float()
            s = w.abs().max().clamp_min(1e-12) / 448.0
            state[name] = (w / s).clamp(-448, 448).to(torch.float8_e4m3fn)
            state[name[:-len("weight")] + "weight_scale"] = s.reshape(())
            # a "calibrated" range of 3: most of a unit-variance activation, the tail saturates
            state[name[:-len("weight")] + "input_scale"] = torch.tensor(3.0 / 448.0)
        else:
            state[name] = plain[name]
    assert any(k.endswith(".input_scale") for k in state)
    L.save_flux_checkpoint(str(tmp_path), cfg, state, quantization_config={
        "quant_method": "fp8", "activation_scheme": "static"})
    model, rep = L.load_flux_transformer(str(tmp_path), device=device, strict=True)
    assert rep.unexpected_keys == []
    lin = model.single_transformer_blocks[0].proj_mlp
    name = next(n for n, m in model.named_modules() if m is lin)
    assert lin.input_scale.shape == (1,) and lin.input_scale.dtype == torch.float32
    assert torch.equal(lin.input_scale.cpu(), state[name + ".input_scale"].reshape(1))
    g = torch.Generator().manual_seed(6)
    x = torch.randn(40, lin.weight.shape[1], generator=g).bfloat16()
    x[3, 5] = 7.5   # beyond the calibrated range: must saturate at 448 * scale = 3
    y, _ = lin(x.to(device))
    bias = lin.bias.detach().cpu() if lin.bias is not None else None
    want = O.fp8_linear_static(x, state[name + ".weight"], lin.weight_scale.cpu(),
                               lin.input_scale.cpu(), bias)
    assert_close_bf16(y.cpu(), want, rel=1e-3, max_ulps=1.0, what="static-scale fp8 linear")
    # and it is NOT what the dynamic scheme would give
    dyn = O.fp8_linear(x, state[name + ".weight"], lin.weight_scale.cpu(), bias)
    assert (y.cpu().float() - dyn.float()).abs().max() > 1e-2
    # the whole forward runs on the layer path; the fused schedule says why it does not
    outs = _run(model, cfg, device, fused=False)
    assert all(torch.isfinite(o.float()).all() for o in outs)
    with pytest.raises(NotImplementedError, match="static activation"):
        _run(model, cfg, device, fused=True)


@pytest.mark.gpu
def test_converted_checkpoint_loads_and_matches(device, tmp_path):
    """bf16 directory -> converter (run on the device) -> loader: the very bits of the model
    whose block-quantised parameters were set in place from the oracle's quantiser."""
    from sglang_diffusion_amd.tools import convert_hf_to_fp8 as C

    ref_model, plain, cfg = build_small_block_fp8(device)
    src, dst = tmp_path / "bf16", tmp_path / "fp8"
    L.save_flux_checkpoint(str(src), cfg, plain)
    C.main(["--model-dir", str(src), "--save-dir", str(dst), "--strategy", "block",
            "--block-size", "128", "128"])
    model, rep = L.load_flux_transformer(str(dst), device=device, strict=True)
    assert rep.unexpected_keys == []
    for (ka, a), (kb, b) in zip(sorted(model.state_dict().items()),
                                sorted(ref_model.state_dict().items())):
        assert ka == kb and a.dtype == b.dtype
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.float8_e4m3fn else a,
                           b.view(torch.uint8) if b.dtype == torch.float8_e4m3fn else b), ka
    for got, want in zip(_run(model, cfg, device), _run(ref_model, cfg, device)):
        assert torch.equal(got, want)
