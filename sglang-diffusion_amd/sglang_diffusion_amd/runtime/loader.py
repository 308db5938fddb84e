"""Diffusers-format Flux transformer checkpoint -> FluxTransformer2DModel on one MI355X.

Mirrors the reference's transformer loader for the single-GPU, replicated-weights case
(loader/component_loaders/transformer_loader.py:42-199, loader/fsdp_load.py:56-153,221-400):

  1. `config.json` of the component directory is the architecture
     (utils/hf_diffusers_utils.py:315-342; its keys overwrite the arch defaults,
     configs/models/base.py:71-81);
  2. the quantisation config comes from `config.json["quantization_config"]`
     (utils/hf_diffusers_utils.py:352-378) or, failing that, from the `_quantization_metadata`
     entry of a safetensors header (utils/hf_diffusers_utils.py:911-945);
  3. every `*.safetensors` file of the directory is read (loader/utils.py:175-177), names go
     through the model's `param_names_mapping` (configs/models/dits/flux.py:28-33), each tensor
     is cast to the dtype of the parameter it lands in and handed to the parameter's
     `weight_loader` when it has one (fsdp_load.py:268-300);
  4. every quantised layer post-processes its weights (fsdp_load.py:141-146).

The model is built directly in HBM (288 GB holds the 12 B-parameter bf16 network ten times over),
so a tensor travels file -> device parameter once, with no host-side copy of the model.  Unlike the
reference this loader never deletes a file it cannot read: it raises.
"""
from __future__ import annotations

import glob
import json
import os
import re
import struct
from dataclasses import fields
from typing import Any, Dict, Iterator, List, NamedTuple, Optional, Tuple

import torch

from ..layers.quantization import QuantizationConfig, get_quantization_config
from ..models.flux import FluxArchConfig, FluxTransformer2DModel

# configs/models/dits/flux.py:28-33: diffusers checkpoints already use the module names of the
# model; pipelines that save the whole pipeline in one file prefix them with "transformer.".
FLUX_PARAM_NAMES_MAPPING: Dict[str, str] = {
    r"^transformer\.(\w*)\.(.*)$": r"\1.\2",
}

_CLASS_NAMES = ("FluxTransformer2DModel",)


class LoadReport(NamedTuple):
    """What load_model_from_full_model_state_dict returns in the reference
    (fsdp_load.py:236-239), plus the byte count that was read."""

    missing_keys: List[str]
    unexpected_keys: List[str]
    bytes_loaded: int


def list_safetensors_files(model_path: str) -> List[str]:
    """loader/utils.py:175-177; a single file is accepted as is
    (transformer_loader.py:55-60)."""
    if os.path.isfile(model_path) and model_path.endswith(".safetensors"):
        return [model_path]
    return sorted(glob.glob(os.path.join(str(model_path), "*.safetensors")))


def read_safetensors_metadata(path: str) -> Dict[str, str]:
    """The `__metadata__` object of a safetensors header: 8-byte little-endian header length,
    then that many bytes of JSON.  Nothing of the payload is touched."""
    with open(path, "rb") as f:
        raw = f.read(8)
        if len(raw) != 8:
            raise ValueError(f"{path}: shorter than a safetensors header")
        (n,) = struct.unpack("<Q", raw)
        size = os.fstat(f.fileno()).st_size
        if n == 0 or n > size - 8:
            raise ValueError(f"{path}: header length {n} does not fit the file ({size} bytes)")
        header = json.loads(f.read(n).decode("utf-8"))
    meta = header.get("__metadata__") or {}
    return {str(k): str(v) for k, v in meta.items()}


def get_component_config(component_path: str) -> Dict[str, Any]:
    """utils/hf_diffusers_utils.py:315-342 for a transformer directory: `config.json`."""
    path = os.path.join(component_path, "config.json")
    if not os.path.isfile(path):
        raise FileNotFoundError(f"no config.json under {component_path}")
    with open(path, "r", encoding="utf-8") as f:
        cfg = json.load(f)
    if not isinstance(cfg, dict):
        raise ValueError(f"{path}: a JSON object is expected")
    return cfg


def get_quant_config(model_config: Dict[str, Any]) -> Optional[QuantizationConfig]:
    """utils/hf_diffusers_utils.py:352-378."""
    qc = model_config.get("quantization_config")
    if qc is None:
        return None
    if "quant_method" not in qc:
        raise ValueError("quantization_config has no quant_method")
    quant_cls = get_quantization_config(qc["quant_method"])
    return quant_cls.from_config(dict(qc))


def get_quant_config_from_safetensors_metadata(path: str) -> Optional[QuantizationConfig]:
    """utils/hf_diffusers_utils.py:911-945: `_quantization_metadata` of the header; the
    diffusers per-layer format ({"format_version", "layers": {name: {"format": "float8…"}}})
    means dynamic-activation fp8."""
    meta = read_safetensors_metadata(path)
    text = meta.get("_quantization_metadata")
    if not text:
        return None
    try:
        d = json.loads(text)
    except ValueError:
        return None
    if not isinstance(d, dict):
        return None
    if "quant_method" not in d and "format_version" in d and "layers" in d:
        layers = d.get("layers") or {}
        if any(isinstance(v, dict) and "float8" in v.get("format", "") for v in layers.values()):
            d["quant_method"] = "fp8"
            d["activation_scheme"] = "dynamic"
    if "quant_method" not in d:
        return None
    return get_quantization_config(d["quant_method"]).from_config(d)


def arch_config_from_hf(hf_config: Dict[str, Any]) -> FluxArchConfig:
    """Keys of the diffusers config overwrite the defaults (configs/models/base.py:71-81);
    bookkeeping keys (`_class_name`, `_diffusers_version`, `quantization_config` …) are not
    architecture and are dropped here instead of being set as attributes."""
    known = {f.name for f in fields(FluxArchConfig)}
    kw = {k: v for k, v in hf_config.items() if k in known}
    if "axes_dims_rope" in kw:
        kw["axes_dims_rope"] = tuple(int(a) for a in kw["axes_dims_rope"])
    cfg = FluxArchConfig(**kw)
    if sum(cfg.axes_dims_rope) != cfg.attention_head_dim:
        raise ValueError(f"axes_dims_rope {cfg.axes_dims_rope} does not add up to "
                         f"attention_head_dim {cfg.attention_head_dim}")
    return cfg


def map_param_name(name: str, mapping: Dict[str, str] = FLUX_PARAM_NAMES_MAPPING) -> str:
    """First matching pattern rewrites the name (loader/utils.py get_param_names_mapping);
    names no pattern matches are kept."""
    for pat, repl in mapping.items():
        if re.match(pat, name):
            return re.sub(pat, repl, name)
    return name


def safetensors_weights_iterator(files: List[str], device: torch.device
                                 ) -> Iterator[Tuple[str, torch.Tensor]]:
    """loader/weight_utils.py:151-220: (name, tensor) over every file, each tensor read from
    the memory-mapped file straight onto `device`."""
    from safetensors import safe_open

    for path in files:
        with safe_open(path, framework="pt", device=str(device)) as f:
            for name in f.keys():
                yield name, f.get_tensor(name)


def load_weights(model: torch.nn.Module, weights: Iterator[Tuple[str, torch.Tensor]],
                 strict: bool = False,
                 mapping: Dict[str, str] = FLUX_PARAM_NAMES_MAPPING) -> LoadReport:
    """fsdp_load.py:221-400 without the sharded branch: each checkpoint tensor is cast to the
    dtype of the parameter (so fp8 parameters stay fp8) and goes through the parameter's
    `weight_loader` if it has one.  Checkpoint names with no parameter raise under `strict`
    and are reported otherwise; parameters the checkpoint does not hold are reported as
    missing and raise here in both modes, because an uninitialised DiT weight cannot produce
    a meaningful image (the reference only accepts that for training-time extras,
    fsdp_load.py:369-400)."""
    params = dict(model.named_parameters())
    params.update({k: v for k, v in model.named_buffers() if k not in params})
    seen = set()
    unexpected: List[str] = []
    nbytes = 0
    with torch.no_grad():
        for raw_name, tensor in weights:
            name = map_param_name(raw_name, mapping)
            p = params.get(name)
            if p is None:
                if strict:
                    raise ValueError(f"Parameter {name} not found in custom model state dict. "
                                     f"The hf to custom mapping may be incorrect.")
                unexpected.append(name)
                continue
            if name in seen:
                raise ValueError(f"{name} appears twice in the checkpoint")
            seen.add(name)
            nbytes += tensor.numel() * tensor.element_size()
            t = tensor.to(device=p.device, dtype=p.dtype)
            loader = getattr(p, "weight_loader", None)
            if loader is not None:
                loader(p, t)
            else:
                if t.dim() == 0 and p.dim() == 1 and p.numel() == 1:
                    t = t.reshape(1)
                if p.shape != t.shape:
                    raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != parameter "
                                     f"shape {tuple(p.shape)}")
                p.data.copy_(t)
    missing = sorted(k for k, v in model.named_parameters() if k not in seen)
    if missing:
        head = ", ".join(missing[:8]) + (" …" if len(missing) > 8 else "")
        raise ValueError(f"{len(missing)} parameter(s) missing from the checkpoint: {head}")
    return LoadReport(missing, sorted(unexpected), nbytes)


def load_flux_transformer(component_path: str, device: torch.device | str = "cuda",
                          transformer_weights_path: Optional[str] = None,
                          quant_config: Optional[QuantizationConfig] = None,
                          strict: bool = False
                          ) -> Tuple[FluxTransformer2DModel, LoadReport]:
    """TransformerLoader.load_customized (transformer_loader.py:122-199) for Flux.

    component_path            the `transformer/` directory of a diffusers checkpoint
    transformer_weights_path  optional file or directory whose safetensors replace the
                              directory's own (the reference's --transformer-weights-path,
                              transformer_loader.py:42-71); its header metadata is then also
                              a source of the quantisation config (transformer_loader.py:73-89)
    quant_config              overrides what the checkpoint declares, e.g.
                              Fp8Config(is_checkpoint_fp8_serialized=False, ...) to quantise a
                              bf16 checkpoint on the device while loading
    """
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
    if quant_config is None and transformer_weights_path:
        for path in files:
            quant_config = get_quant_config_from_safetensors_metadata(path)
            if quant_config is not None:
                break

    cfg = arch_config_from_hf(hf_config)
    with torch.device(device):
        model = FluxTransformer2DModel(cfg, quant_config=quant_config)
    model.eval()
    report = load_weights(model, safetensors_weights_iterator(files, device), strict=strict)
    model.process_weights_after_loading()
    for p in model.parameters():
        p.requires_grad_(False)
    return model, report


def save_flux_checkpoint(component_path: str, cfg: FluxArchConfig,
                         state: Dict[str, torch.Tensor],
                         quantization_config: Optional[Dict[str, Any]] = None,
                         shard_bytes: Optional[int] = None,
                         metadata: Optional[Dict[str, str]] = None) -> List[str]:
    """Write `state` as a diffusers-layout transformer directory (config.json +
    diffusion_pytorch_model[-0000i-of-0000n].safetensors [+ index]).  The counterpart of the
    loader for tests, synthetic benchmarks and offline-quantised checkpoints made here."""
    from safetensors.torch import save_file

    os.makedirs(component_path, exist_ok=True)
    hf = {"_class_name": _CLASS_NAMES[0]}
    for f in fields(FluxArchConfig):
        v = getattr(cfg, f.name)
        hf[f.name] = list(v) if isinstance(v, tuple) else v
    if quantization_config is not None:
        hf["quantization_config"] = quantization_config
    with open(os.path.join(component_path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(hf, f, indent=2)

    names = sorted(state)
    shards: List[List[str]] = [[]]
    used = 0
    for n in names:
        b = state[n].numel() * state[n].element_size()
        if shard_bytes and shards[-1] and used + b > shard_bytes:
            shards.append([])
            used = 0
        shards[-1].append(n)
        used += b
    out = []
    index = {}
    for i, part in enumerate(shards):
        fname = ("diffusion_pytorch_model.safetensors" if len(shards) == 1 else
                 f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors")
        path = os.path.join(component_path, fname)
        save_file({n: state[n].detach().cpu().contiguous() for n in part}, path,
                  metadata=metadata)
        out.append(path)
        index.update({n: fname for n in part})
    if len(shards) > 1:
        with open(os.path.join(component_path,
                               "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": index}, f)
    return out
