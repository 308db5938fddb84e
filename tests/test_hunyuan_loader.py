"""HunyuanVideo checkpoint naming: the rename / merge table (configs/models/dits/hunyuanvideo.py:
37-149, loader/utils.py:33-139 of the reference) and the loader on top of it, on the CPU."""
import json
import os

import pytest
import torch

from sglang_diffusion_amd.models.hunyuanvideo import (HunyuanVideoArchConfig,
                                                      HunyuanVideoTransformer3DModel)
from sglang_diffusion_amd.runtime.hunyuan_loader import (HUNYUAN_PARAM_NAMES_MAPPING,
                                                         load_hunyuan_transformer,
                                                         map_hunyuan_name, merged_weights,
                                                         to_diffusers_state_dict)
from sglang_diffusion_amd.runtime.synthetic import init_synthetic_weights

CFG = dict(num_attention_heads=2, num_layers=2, num_single_layers=3, num_refiner_layers=2,
           text_embed_dim=64, pooled_projection_dim=32, guidance_embeds=True)


def test_names_of_the_reference_table():
    """Spot checks written out from the reference's table, one per group of rows."""
    cases = {
        "context_embedder.time_text_embed.timestep_embedder.linear_1.weight":
            ("txt_in.t_embedder.mlp.fc_in.weight", None, None),
        "context_embedder.proj_in.bias": ("txt_in.input_embedder.bias", None, None),
        "context_embedder.token_refiner.refiner_blocks.1.attn.to_k.weight":
            ("txt_in.refiner_blocks.1.self_attn_qkv.weight", 1, 3),
        "context_embedder.token_refiner.refiner_blocks.0.ff.net.0.proj.weight":
            ("txt_in.refiner_blocks.0.mlp.fc_in.weight", None, None),
        "context_embedder.token_refiner.refiner_blocks.0.ff.net.2.bias":
            ("txt_in.refiner_blocks.0.mlp.fc_out.bias", None, None),
        "context_embedder.token_refiner.refiner_blocks.0.norm_out.linear.weight":
            ("txt_in.refiner_blocks.0.adaLN_modulation.linear.weight", None, None),
        "x_embedder.proj.weight": ("img_in.proj.weight", None, None),
        "time_text_embed.guidance_embedder.linear_2.bias": ("guidance_in.mlp.fc_out.bias", None, None),
        "time_text_embed.text_embedder.linear_1.weight": ("vector_in.fc_in.weight", None, None),
        "transformer_blocks.17.norm1_context.linear.weight":
            ("double_blocks.17.txt_mod.linear.weight", None, None),
        "transformer_blocks.3.attn.add_v_proj.bias": ("double_blocks.3.txt_attn_qkv.bias", 2, 3),
        "transformer_blocks.3.attn.to_q.weight_scale":
            ("double_blocks.3.img_attn_qkv.weight_scale", 0, 3),
        "transformer_blocks.3.attn.to_out.0.weight": ("double_blocks.3.img_attn_proj.weight", None, None),
        "transformer_blocks.3.attn.to_add_out.weight": ("double_blocks.3.txt_attn_proj.weight", None, None),
        "transformer_blocks.3.attn.norm_added_k.weight":
            ("double_blocks.3.txt_attn_k_norm.weight", None, None),
        "transformer_blocks.0.ff_context.net.0.proj.bias": ("double_blocks.0.txt_mlp.fc_in.bias", None, None),
        "transformer_blocks.0.ff.net.0.weight": ("double_blocks.0.img_mlp.fc_in.weight", None, None),
        "single_transformer_blocks.39.proj_mlp.weight": ("single_blocks.39.linear1.weight", 3, 4),
        "single_transformer_blocks.39.attn.to_v.bias": ("single_blocks.39.linear1.bias", 2, 4),
        "single_transformer_blocks.39.proj_out.weight": ("single_blocks.39.linear2.weight", None, None),
        "single_transformer_blocks.39.norm.linear.bias": ("single_blocks.39.modulation.linear.bias", None, None),
        "single_transformer_blocks.39.attn.norm_q.weight": ("single_blocks.39.q_norm.weight", None, None),
        "norm_out.linear.weight": ("final_layer.adaLN_modulation.linear.weight", None, None),
        "proj_out.bias": ("final_layer.linear.bias", None, None),
        "transformer.proj_out.bias": ("final_layer.linear.bias", None, None),
        "something.else.weight": ("something.else.weight", None, None),
    }
    for src, want in cases.items():
        assert map_hunyuan_name(src) == want, src
    assert len(HUNYUAN_PARAM_NAMES_MAPPING) == 49
    # one pass is a fixed point: no output of the table matches a pattern of the table
    for src in cases:
        name = map_hunyuan_name(src)[0]
        assert map_hunyuan_name(name)[0] == name


def test_pieces_merge_in_table_order_whatever_order_they_arrive_in():
    q, k, v, m = (torch.full((2, 3), float(i)) for i in range(4))
    stream = [("single_transformer_blocks.0.proj_mlp.weight", m),
              ("single_transformer_blocks.0.attn.to_v.weight", v),
              ("proj_out.weight", torch.ones(1)),
              ("single_transformer_blocks.0.attn.to_q.weight", q),
              ("single_transformer_blocks.0.attn.to_k.weight", k)]
    got = dict(merged_weights(iter(stream)))
    assert set(got) == {"final_layer.linear.weight", "single_blocks.0.linear1.weight"}
    assert torch.equal(got["single_blocks.0.linear1.weight"], torch.cat([q, k, v, m], 0))
    with pytest.raises(ValueError, match="pieces missing"):
        list(merged_weights(iter(stream[:2])))
    with pytest.raises(ValueError, match="appears twice"):
        list(merged_weights(iter([stream[1], stream[1]])))


def _write_checkpoint(tmp_path, shards=2):
    from safetensors.torch import save_file

    cfg = HunyuanVideoArchConfig(**CFG)
    model = HunyuanVideoTransformer3DModel(cfg)
    init_synthetic_weights(model, seed=11, host_rng=True)
    sd = to_diffusers_state_dict(model, cfg)
    d = tmp_path / "transformer"
    os.makedirs(d)
    names = sorted(sd)
    for s in range(shards):        # q, k and v of a block land in different files
        part = {n: sd[n].contiguous() for n in names[s::shards]}
        save_file(part, str(d / f"diffusion_pytorch_model-{s + 1:05d}-of-{shards:05d}.safetensors"))
    with open(d / "config.json", "w") as f:
        json.dump(dict(CFG, _class_name="HunyuanVideoTransformer3DModel",
                       _diffusers_version="0.36.0", rope_axes_dim=[16, 56, 56]), f)
    return model, sd, str(d)


def test_checkpoint_round_trip_on_the_cpu(tmp_path):
    model, sd, path = _write_checkpoint(tmp_path)
    assert "transformer_blocks.1.attn.add_k_proj.weight" in sd
    assert "single_transformer_blocks.2.proj_mlp.bias" in sd
    assert "transformer_blocks.0.ff.net.0.proj.weight" in sd and "transformer_blocks.0.ff.net.2.weight" in sd
    assert not any(k.startswith(("double_blocks", "single_blocks", "txt_in")) for k in sd)
    loaded, report = load_hunyuan_transformer(path, device="cpu")
    assert report.unexpected_keys == [] and report.missing_keys == []
    want = dict(model.named_parameters())
    got = dict(loaded.named_parameters())
    assert set(want) == set(got)
    for k in want:
        assert want[k].dtype == got[k].dtype and torch.equal(want[k], got[k]), k


def test_loader_errors(tmp_path):
    from safetensors.torch import load_file, save_file

    _, _, path = _write_checkpoint(tmp_path, shards=1)
    f = os.path.join(path, os.listdir(path)[0] if os.listdir(path)[0].endswith(".safetensors")
                     else [x for x in os.listdir(path) if x.endswith(".safetensors")][0])
    sd = load_file(f)
    del sd["transformer_blocks.0.attn.to_k.weight"]
    save_file(sd, f)
    with pytest.raises(ValueError, match="pieces missing"):
        load_hunyuan_transformer(path, device="cpu")
    with open(os.path.join(path, "config.json"), "w") as fh:
        json.dump({"_class_name": "FluxTransformer2DModel"}, fh)
    with pytest.raises(ValueError, match="holds a FluxTransformer2DModel"):
        load_hunyuan_transformer(path, device="cpu")
