"""Offline conversion of a bf16 diffusers transformer directory to an fp8-serialised one.

Same command line, same selection of tensors, same arithmetic and same output layout as the
reference's tool (python/sglang/multimodal_gen/tools/convert_hf_to_fp8.py), so a directory
written by either tool loads in either runtime:

    python -m sglang_diffusion_amd.tools.convert_hf_to_fp8 \
        --model-dir  /path/to/FLUX.1-dev/transformer \
        --save-dir   /path/to/FLUX.1-dev/transformer-FP8 \
        --strategy block --block-size 128 128

  block    e4m3 weight + `weight_scale_inv` [ceil(N/bn), ceil(K/bk)]  (convert_hf_to_fp8.py:48-79)
  tensor   e4m3 weight + `weight_scale` [1]                          (convert_hf_to_fp8.py:90-95)
  channel  e4m3 weight + `weight_scale` [N, 1]                       (convert_hf_to_fp8.py:82-87);
           its config.json declares "compressed-tensors", which the diffusion runtime's
           registry does not know (runtime/layers/quantization/__init__.py:10-17), so such a
           directory is for other consumers.

Differences, all deliberate: a weight block that is entirely zero gets scale 0 and zero
codes (the reference divides 0 by 0 there and stores NaN codes), and the block scale keeps
its two dimensions when a weight has a single row or column of blocks (the reference's
squeeze() drops them and the loader then rejects the shape); and the division by 448 is a true
IEEE division on every device (on a GPU, torch turns `tensor / python_scalar` into a multiplication
by the rounded reciprocal, so the reference's tool run on a GPU can store a scale one ulp away
from the one it stores on the host; codes and scale stay consistent with each other either way).

Tensors are converted one at a time on the HIP device when there is one (the host otherwise;
the arithmetic is IEEE fp32 division and comparison, identical on both), so the peak footprint
is one weight, not one file.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

FP8 = torch.float8_e4m3fn
FP8_MAX = torch.finfo(FP8).max
FP8_MIN = torch.finfo(FP8).min

# convert_hf_to_fp8.py:140-160: a key is quantised iff it contains "weight" and none of these.
_SKIP_SUBSTRINGS = ("layernorm", "embed", "router", "mlp.gate.", "norm", "lm_head", "eh_proj",
                    "net", "txt_mod", "img_mod", "img_in", "txt_in", "time_in", "vector_in",
                    "adaLN_modulation", "all_final_layer", "feed_forward")


def is_quantised_key(key: str) -> bool:
    return ("weight" in key and not any(s in key for s in _SKIP_SUBSTRINGS)
            and key != "proj_out.weight")


def _fp8_max_like(t: torch.Tensor) -> torch.Tensor:
    """448 as a one-element fp32 tensor on t's device: dividing by a TENSOR is a true division."""
    return torch.full((1,), FP8_MAX, dtype=torch.float32, device=t.device)


def block_fp8(weight: torch.Tensor, block_size: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    bn, bk = int(block_size[0]), int(block_size[1])
    n, k = weight.shape
    nt, kt = -(-n // bn), -(-k // bk)
    t = F.pad(weight, (0, kt * bk - k, 0, nt * bn - n), mode="constant", value=0.0)
    t = t.reshape(nt, bn, kt, bk)
    bmax = t.abs().amax(dim=(1, 3), keepdim=True)
    scale = bmax.to(torch.float32) / _fp8_max_like(bmax)
    safe = torch.where(scale == 0, torch.ones_like(scale), scale)
    q = (t / safe).clamp(min=FP8_MIN, max=FP8_MAX).reshape(nt * bn, kt * bk).to(FP8)
    return q[:n, :k].contiguous(), scale.reshape(nt, kt)


def channel_fp8(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    cmax = weight.abs().amax(dim=-1, keepdim=True)
    scale = cmax.clamp(min=1e-12).to(torch.float32) / _fp8_max_like(cmax)
    return (weight / scale).clamp(min=FP8_MIN, max=FP8_MAX).to(FP8), scale


def tensor_fp8(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = weight.abs().max().clamp(min=1e-12).to(torch.float32) / _fp8_max_like(weight).reshape(())
    return (weight / scale).clamp(min=FP8_MIN, max=FP8_MAX).to(FP8), scale.view(1)


def quant_fp8(weight, strategy: str, block_size=None):
    if strategy == "tensor":
        return tensor_fp8(weight)
    if strategy == "channel":
        return channel_fp8(weight)
    return block_fp8(weight, block_size)


def quantization_config_for(strategy: str, block_size: Optional[Sequence[int]],
                            not_converted: List[str]) -> Dict:
    """convert_hf_to_fp8.py:214-263."""
    if strategy in ("block", "tensor"):
        qc: Dict = {"activation_scheme": "dynamic", "fmt": "e4m3", "quant_method": "fp8"}
        if block_size:
            qc["weight_block_size"] = [int(b) for b in block_size]
        if not_converted:
            qc["modules_to_not_convert"] = sorted(set(not_converted))
        return qc
    act = {"actorder": None, "block_structure": None, "dynamic": True, "group_size": None,
           "num_bits": 8, "observer": None, "observer_kwargs": {}, "strategy": "token",
           "symmetric": True, "type": "float"}
    wts = {"actorder": None, "block_structure": None, "dynamic": False, "group_size": None,
           "num_bits": 8, "observer": "minmax", "observer_kwargs": {}, "strategy": strategy,
           "symmetric": True, "type": "float"}
    return {"config_groups": {"group_0": {"input_activations": act, "output_activations": None,
                                          "targets": ["Linear"], "weights": wts}},
            "format": "float-quantized", "ignore": sorted(set(not_converted)),
            "quant_method": "compressed-tensors", "quantization_status": "compressed"}


def convert_file(src: str, dst: str, strategy: str, block_size, device: torch.device
                 ) -> Tuple[Dict[str, int], List[str]]:
    """One safetensors file -> its converted twin.  Returns ({output key: numel}, module names
    left unconverted)."""
    from safetensors import safe_open
    from safetensors.torch import save_file

    out: Dict[str, torch.Tensor] = {}
    skipped: List[str] = []
    with safe_open(src, framework="pt", device="cpu") as f:
        for key in f.keys():
            t = f.get_tensor(key)
            if is_quantised_key(key):
                if t.dim() != 2:
                    raise ValueError(f"{src}: {key} is selected for fp8 but has shape "
                                     f"{tuple(t.shape)}")
                q, s = quant_fp8(t.to(device), strategy, block_size)
                out[key] = q.cpu()
                suffix = ".weight_scale_inv" if block_size else ".weight_scale"
                out[key.replace(".weight", suffix)] = s.cpu()
            else:
                skipped.append(key.replace(".weight", ""))
                out[key] = t
    save_file(out, dst, metadata={"format": "pt"})
    return {k: v.numel() for k, v in out.items()}, skipped


def convert_fp8(input_path: str, output_path: str, strategy: str = "block",
                block_size: Optional[Sequence[int]] = None,
                device: Optional[torch.device] = None) -> Dict:
    """convert_hf_to_fp8.py:179-283.  Returns the quantization_config that was written."""
    if strategy == "block" and not block_size:
        raise ValueError("--strategy block needs --block-size BN BK")
    if strategy != "block" and block_size:
        raise ValueError("--block-size only goes with --strategy block")
    if block_size and len(block_size) != 2:
        raise ValueError("--block-size takes two integers")
    input_path = os.path.abspath(input_path)
    if os.path.abspath(output_path) == input_path:
        raise ValueError("save-dir must differ from model-dir")
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    os.makedirs(output_path, exist_ok=True)

    names = sorted(os.listdir(input_path))
    for name in names:
        p = os.path.join(input_path, name)
        # the old weight index names the old files' contents; a fresh one is written below
        if (not name.endswith(".safetensors") and not os.path.isdir(p)
                and not name.endswith(".safetensors.index.json")):
            shutil.copyfile(p, os.path.join(output_path, name))
    files = [n for n in names if n.endswith(".safetensors")]
    if not files:
        raise ValueError(f"no safetensors files found in {input_path}")

    weight_map: Dict[str, str] = {}
    total = 0
    not_converted: List[str] = []
    for name in files:
        sizes, skipped = convert_file(os.path.join(input_path, name),
                                      os.path.join(output_path, name), strategy, block_size,
                                      device)
        for k, n in sizes.items():
            weight_map[k] = name
            total += n
        not_converted.extend(skipped)

    qc = quantization_config_for(strategy, block_size, not_converted)
    cfg_path = os.path.join(input_path, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path, "r", encoding="utf-8") as f:
            cfg = json.load(f)
        cfg["quantization_config"] = qc
        with open(os.path.join(output_path, "config.json"), "w", encoding="utf-8") as f:
            json.dump(cfg, f, indent=2)
    with open(os.path.join(output_path, "model.safetensors.index.json"), "w",
              encoding="utf-8") as f:
        json.dump({"weight_map": weight_map, "metadata": {"total_size": total}}, f, indent=2)
    return qc


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model-dir", type=str, required=True,
                    help="directory of the bf16 safetensors model (e.g. the transformer subfolder)")
    ap.add_argument("--save-dir", type=str, required=True,
                    help="directory to write the converted model to")
    ap.add_argument("--strategy", type=str, default="block", choices=["block", "channel", "tensor"])
    ap.add_argument("--block-size", type=int, nargs="*", default=None, help="e.g. --block-size 128 128")
    ap.add_argument("--max-workers", type=int, default=1,
                    help="accepted for command-line compatibility; files are converted in turn")
    args = ap.parse_args(argv)
    if os.path.exists(args.save_dir) and not os.path.isdir(args.save_dir):
        raise ValueError("The save_dir should be a directory.")
    convert_fp8(args.model_dir, args.save_dir, args.strategy, args.block_size or None)


if __name__ == "__main__":
    main()
