"""Diffusers-format HunyuanVideo transformer checkpoint -> HunyuanVideoTransformer3DModel.

The reference renames every checkpoint tensor through the regular-expression table of
configs/models/dits/hunyuanvideo.py:37-149 and, where the model holds one fused projection for
several checkpoint tensors (q|k|v of the double blocks and the token refiner, q|k|v|proj_mlp of
the single blocks), concatenates the pieces along the output dimension in the order the table
gives them (loader/utils.py:33-139).  This module holds the same table, built from a compact
description, a streaming version of the merge (a fused tensor is assembled the moment its last
piece arrives, nothing else is kept) and the loader on top of runtime/loader.py.
"""
from __future__ import annotations

import re
from dataclasses import fields
from typing import Any, Dict, Iterator, List, Optional, Tuple, Union

import torch

from ..layers.quantization import QuantizationConfig
from ..models.hunyuanvideo import (HunyuanVideoArchConfig, HunyuanVideoTransformer3DModel,
                                   process_weights_after_loading)
from .loader import (LoadReport, get_component_config, get_quant_config,
                     get_quant_config_from_safetensors_metadata, list_safetensors_files,
                     load_weights, safetensors_weights_iterator)

Rule = Union[str, Tuple[str, int, int]]

_CLASS_NAMES = ("HunyuanVideoTransformer3DModel",)
_ROWS: List[Tuple[str, str, Optional[Tuple[int, int]], bool]] = []


def _rule(table: Dict[str, Rule], src: str, dst: str, part: Optional[Tuple[int, int]] = None,
          proj: bool = False) -> None:
    """One row: `{i}` in src/dst is a block index; the trailing `.weight` / `.bias` /
    `.weight_scale` … is carried over.  proj: diffusers' GELU wrapper puts the first FF linear
    under `.proj`, plain activations do not."""
    _ROWS.append((src, dst, part, proj))
    indexed = "{i}" in src
    pat = "^" + re.escape(src).replace(re.escape("{i}"), r"(\d+)")
    pat += (r"(?:\.proj)?" if proj else "") + r"\.(.*)$"
    repl = dst.replace("{i}", r"\1") + (r".\2" if indexed else r".\1")
    table[pat] = repl if part is None else (repl, part[0], part[1])


def build_param_names_mapping() -> Dict[str, Rule]:
    """configs/models/dits/hunyuanvideo.py:37-149 (same patterns, same order)."""
    t: Dict[str, Rule] = {}
    ce, rb = "context_embedder", "context_embedder.token_refiner.refiner_blocks.{i}"
    # text token refiner
    _rule(t, f"{ce}.time_text_embed.timestep_embedder.linear_1", "txt_in.t_embedder.mlp.fc_in")
    _rule(t, f"{ce}.time_text_embed.timestep_embedder.linear_2", "txt_in.t_embedder.mlp.fc_out")
    _rule(t, f"{ce}.proj_in", "txt_in.input_embedder")
    _rule(t, f"{ce}.time_text_embed.text_embedder.linear_1", "txt_in.c_embedder.fc_in")
    _rule(t, f"{ce}.time_text_embed.text_embedder.linear_2", "txt_in.c_embedder.fc_out")
    for n in ("norm1", "norm2"):
        _rule(t, f"{rb}.{n}", f"txt_in.refiner_blocks.{{i}}.{n}")
    for k, n in enumerate(("to_q", "to_k", "to_v")):
        _rule(t, f"{rb}.attn.{n}", "txt_in.refiner_blocks.{i}.self_attn_qkv", (k, 3))
    _rule(t, f"{rb}.attn.to_out.0", "txt_in.refiner_blocks.{i}.self_attn_proj")
    _rule(t, f"{rb}.ff.net.0", "txt_in.refiner_blocks.{i}.mlp.fc_in", proj=True)
    _rule(t, f"{rb}.ff.net.2", "txt_in.refiner_blocks.{i}.mlp.fc_out", proj=True)
    _rule(t, f"{rb}.norm_out.linear", "txt_in.refiner_blocks.{i}.adaLN_modulation.linear")
    # patch embedding and the conditioning embedders
    _rule(t, "x_embedder.proj", "img_in.proj")
    for src, dst in (("timestep_embedder", "time_in.mlp"), ("guidance_embedder", "guidance_in.mlp"),
                     ("text_embedder", "vector_in")):
        _rule(t, f"time_text_embed.{src}.linear_1", f"{dst}.fc_in")
        _rule(t, f"time_text_embed.{src}.linear_2", f"{dst}.fc_out")
    # double stream blocks
    tb, db = "transformer_blocks.{i}", "double_blocks.{i}"
    _rule(t, f"{tb}.norm1.linear", f"{db}.img_mod.linear")
    _rule(t, f"{tb}.norm1_context.linear", f"{db}.txt_mod.linear")
    _rule(t, f"{tb}.attn.norm_q", f"{db}.img_attn_q_norm")
    _rule(t, f"{tb}.attn.norm_k", f"{db}.img_attn_k_norm")
    for k, n in enumerate(("to_q", "to_k", "to_v")):
        _rule(t, f"{tb}.attn.{n}", f"{db}.img_attn_qkv", (k, 3))
    for k, n in enumerate(("add_q_proj", "add_k_proj", "add_v_proj")):
        _rule(t, f"{tb}.attn.{n}", f"{db}.txt_attn_qkv", (k, 3))
    _rule(t, f"{tb}.attn.to_out.0", f"{db}.img_attn_proj")
    _rule(t, f"{tb}.attn.to_add_out", f"{db}.txt_attn_proj")
    _rule(t, f"{tb}.attn.norm_added_q", f"{db}.txt_attn_q_norm")
    _rule(t, f"{tb}.attn.norm_added_k", f"{db}.txt_attn_k_norm")
    _rule(t, f"{tb}.ff.net.0", f"{db}.img_mlp.fc_in", proj=True)
    _rule(t, f"{tb}.ff.net.2", f"{db}.img_mlp.fc_out", proj=True)
    _rule(t, f"{tb}.ff_context.net.0", f"{db}.txt_mlp.fc_in", proj=True)
    _rule(t, f"{tb}.ff_context.net.2", f"{db}.txt_mlp.fc_out", proj=True)
    # single stream blocks: linear1 = q | k | v | proj_mlp
    st, sb = "single_transformer_blocks.{i}", "single_blocks.{i}"
    _rule(t, f"{st}.attn.norm_q", f"{sb}.q_norm")
    _rule(t, f"{st}.attn.norm_k", f"{sb}.k_norm")
    for k, n in enumerate(("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp")):
        _rule(t, f"{st}.{n}", f"{sb}.linear1", (k, 4))
    _rule(t, f"{st}.proj_out", f"{sb}.linear2")
    _rule(t, f"{st}.norm.linear", f"{sb}.modulation.linear")
    # output head
    _rule(t, "norm_out.linear", "final_layer.adaLN_modulation.linear")
    _rule(t, "proj_out", "final_layer.linear")
    return t


HUNYUAN_PARAM_NAMES_MAPPING: Dict[str, Rule] = build_param_names_mapping()
_COMPILED = [(re.compile(p), r) for p, r in HUNYUAN_PARAM_NAMES_MAPPING.items()]


def map_hunyuan_name(name: str) -> Tuple[str, Optional[int], Optional[int]]:
    """(target name, piece index, piece count); names no row matches are kept
    (loader/utils.py:46-90; the table's outputs match none of its own patterns, so one pass
    is the fixed point the reference iterates to)."""
    if name.startswith("transformer."):
        name = name[len("transformer."):]
    for pat, rule in _COMPILED:
        if pat.match(name):
            if isinstance(rule, tuple):
                return pat.sub(rule[0], name), rule[1], rule[2]
            return pat.sub(rule, name), None, None
    return name, None, None


def merged_weights(weights: Iterator[Tuple[str, torch.Tensor]]
                   ) -> Iterator[Tuple[str, torch.Tensor]]:
    """loader/utils.py:95-139 as a stream: renamed tensors pass through; the pieces of a fused
    parameter wait until all have arrived and leave as one tensor, concatenated along
    dimension 0 in piece order."""
    waiting: Dict[str, Dict[int, torch.Tensor]] = {}
    for raw, tensor in weights:
        name, idx, count = map_hunyuan_name(raw)
        if idx is None:
            yield name, tensor
            continue
        pieces = waiting.setdefault(name, {})
        if idx in pieces:
            raise ValueError(f"{raw}: piece {idx} of {name} appears twice in the checkpoint")
        pieces[idx] = tensor
        if len(pieces) == count:
            del waiting[name]
            yield name, torch.cat([pieces[i] for i in range(count)], dim=0)
    if waiting:
        some = ", ".join(f"{k} (has pieces {sorted(v)})" for k, v in list(waiting.items())[:4])
        raise ValueError(f"{len(waiting)} fused parameter(s) with pieces missing from the "
                         f"checkpoint: {some}")


def hunyuan_arch_config_from_hf(hf_config: Dict[str, Any]) -> HunyuanVideoArchConfig:
    """Keys of the diffusers config overwrite the defaults (configs/models/base.py:71-81)."""
    known = {f.name for f in fields(HunyuanVideoArchConfig)}
    kw = {k: v for k, v in hf_config.items() if k in known}
    if "rope_axes_dim" in kw:
        kw["rope_axes_dim"] = tuple(int(a) for a in kw["rope_axes_dim"])
    cfg = HunyuanVideoArchConfig(**kw)
    if sum(cfg.rope_axes_dim) != cfg.attention_head_dim:
        raise ValueError(f"rope_axes_dim {cfg.rope_axes_dim} does not add up to "
                         f"attention_head_dim {cfg.attention_head_dim}")
    return cfg


def load_hunyuan_transformer(component_path: str, device: Union[torch.device, str] = "cuda",
                             transformer_weights_path: Optional[str] = None,
                             quant_config: Optional[QuantizationConfig] = None,
                             strict: bool = False
                             ) -> Tuple[HunyuanVideoTransformer3DModel, LoadReport]:
    """TransformerLoader.load_customized (loader/component_loaders/transformer_loader.py:122-199)
    for HunyuanVideo on one GPU; arguments as load_flux_transformer."""
    device = torch.device(device)
    hf_config = get_component_config(component_path)
    cls_name = hf_config.get("_class_name", _CLASS_NAMES[0])
    if cls_name not in _CLASS_NAMES:
        raise ValueError(f"{component_path} holds a {cls_name}; this loader builds "
                         f"{_CLASS_NAMES[0]}")
    files = list_safetensors_files(transformer_weights_path or component_path)
    if not files:
        raise ValueError(f"no safetensors files found in "
                         f"{transformer_weights_path or component_path}")
    if quant_config is None:
        quant_config = get_quant_config(hf_config)
    if quant_config is None and transformer_weights_path is not None:
        quant_config = get_quant_config_from_safetensors_metadata(files[0])
    cfg = hunyuan_arch_config_from_hf(hf_config)
    with torch.device(device):
        model = HunyuanVideoTransformer3DModel(cfg, quant_config=quant_config)
    model.eval()
    report = load_weights(model, merged_weights(safetensors_weights_iterator(files, device)),
                          strict=strict, mapping={})
    process_weights_after_loading(model)
    for p in model.parameters():
        p.requires_grad_(False)
    return model, report


def to_diffusers_state_dict(model: HunyuanVideoTransformer3DModel,
                            cfg: HunyuanVideoArchConfig) -> Dict[str, torch.Tensor]:
    """The inverse of the table for a model's own parameters: fused projections are cut back
    into the checkpoint's tensors (equal thirds for q|k|v; D, D, D, mlp for linear1).  Used to
    write synthetic checkpoints in the on-disk naming."""
    D = cfg.hidden_size
    inverse: Dict[str, List[Tuple[str, int]]] = {}
    n_idx = max(cfg.num_layers, cfg.num_single_layers, cfg.num_refiner_layers)
    for src, dst, part, proj in _ROWS:
        # diffusers keeps the first FF linear inside its activation module (`net.0.proj`)
        tail = ".proj" if proj and src.endswith("net.0") else ""
        for i in (range(n_idx) if "{i}" in src else [0]):
            for leaf in ("weight", "bias"):
                inverse.setdefault(f"{dst.replace('{i}', str(i))}.{leaf}", []).append(
                    (f"{src.replace('{i}', str(i))}{tail}.{leaf}", part[0] if part else 0))
    out: Dict[str, torch.Tensor] = {}
    for name, p in model.named_parameters():
        srcs = sorted(inverse.get(name, []), key=lambda s: s[1])
        if not srcs:
            raise KeyError(f"no checkpoint name maps to {name}")
        if len(srcs) == 1:
            out[srcs[0][0]] = p.detach().clone()
            continue
        sizes = [D] * 3 + ([p.shape[0] - 3 * D] if len(srcs) == 4 else [])
        for (src, _), piece in zip(srcs, torch.split(p.detach(), sizes, dim=0)):
            out[src] = piece.clone()
    return out
