"""Quantization plugin API and the fp8 W8A8 linear method on MI355X.

Mirrors python/sglang/multimodal_gen/runtime/layers/quantization/configs/base_config.py:19-155
(QuantizeMethodBase / QuantizationConfig), layers/linear.py:88-159 (LinearMethodBase /
UnquantizedLinearMethod), layers/quantization/__init__.py:10-56 (registry) and
layers/quantization/fp8.py:77-498 (Fp8Config / Fp8LinearMethod).  `apply` goes to the
hand-written gfx950 kernels through the C ABI; there is no other GEMM backend to dispatch to.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn
from torch.nn import Parameter

from .. import _C

FP8_DTYPE = torch.float8_e4m3fn  # gfx950 is OCP fp8, not fnuz (srt/.../fp8_kernel.py:95-109)
FP8_MAX = 448.0


class QuantizeMethodBase(ABC):
    @abstractmethod
    def create_weights(self, layer: nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def embedding(self, layer: nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        return


class QuantizationConfig(ABC):
    quantized_model_path: Optional[str] = None

    def __init__(self):
        super().__init__()
        self.packed_modules_mapping: Dict[str, List[str]] = dict()

    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int:
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    @abstractmethod
    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError

    def get_cache_scale(self, name: str) -> Optional[str]:
        return None


class LinearMethodBase(QuantizeMethodBase):
    @abstractmethod
    def create_weights(self, layer: nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs) -> None:
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


def set_weight_attrs(weight: torch.Tensor, attrs: Optional[Dict[str, Any]]) -> None:
    if attrs is None:
        return
    for k, v in attrs.items():
        setattr(weight, k, v)


def _flat2d(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.stride(0) % 8) != 0:
        x2 = x2.contiguous()
    return x2


class UnquantizedLinearMethod(LinearMethodBase):
    """linear.py:126-159; F.linear is replaced by the bf16 MFMA GEMM."""

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs) -> None:
        weight = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition,
                                       dtype=params_dtype), requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", weight)
        set_weight_attrs(weight, extra_weight_attrs)

    def apply(self, layer, x, bias=None, act: int = 0) -> torch.Tensor:
        if x.dtype != torch.bfloat16:
            raise TypeError(f"bf16 activations expected, got {x.dtype}")
        x2 = _flat2d(x)
        out = torch.empty((x2.shape[0], layer.weight.shape[0]), dtype=x.dtype, device=x.device)
        _C.linear_bf16(x2, layer.weight, bias, out, act)
        return out.view(*x.shape[:-1], layer.weight.shape[0])


def is_layer_skipped(prefix: str, ignored_layers: List[str]) -> bool:
    """python/sglang/srt/layers/quantization/utils.py is_layer_skipped, exact / prefix match."""
    return any(prefix == ig or (ig and ig in prefix) for ig in ignored_layers)


class Fp8Config(QuantizationConfig):
    """fp8.py:77-156."""

    ACTIVATION_SCHEMES = ["static", "dynamic"]

    def __init__(self, is_checkpoint_fp8_serialized: bool = False,
                 activation_scheme: str = "dynamic", ignored_layers: Optional[List[str]] = None,
                 weight_block_size: Optional[List[int]] = None) -> None:
        super().__init__()
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        if activation_scheme not in self.ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []
        if weight_block_size is not None:
            if not is_checkpoint_fp8_serialized:
                raise ValueError("The block-wise quantization only supports fp8-serialized "
                                 "checkpoint for now.")
            if len(weight_block_size) != 2:
                raise ValueError("The quantization block size of weight must have 2 dimensions, "
                                 f"but got {len(weight_block_size)} dimensions.")
            if activation_scheme != "dynamic":
                raise ValueError("The block-wise quantization only supports dynamic activation "
                                 f"scheme for now, but got {activation_scheme} activation scheme.")
        self.weight_block_size = weight_block_size

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        is_serialized = "fp8" in quant_method
        activation_scheme = cls.get_from_keys(config, ["activation_scheme"])
        ignored = cls.get_from_keys_or(config, ["ignored_layers", "modules_to_not_convert"], None)
        if ignored:
            ignored = [layer.replace("model.", "") for layer in ignored]
        wbs = cls.get_from_keys_or(config, ["weight_block_size"], None)
        return cls(is_checkpoint_fp8_serialized=is_serialized, activation_scheme=activation_scheme,
                   ignored_layers=ignored, weight_block_size=wbs)

    def get_quant_method(self, layer: nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        from .linear import LinearBase

        if isinstance(layer, LinearBase):
            if is_layer_skipped(prefix, self.ignored_layers):
                return UnquantizedLinearMethod()
            return Fp8LinearMethod(self)
        return None

    def get_scaled_act_names(self) -> List[str]:
        return []


class Fp8LinearMethod(LinearMethodBase):
    """fp8.py:159-498 on the per-token x per-channel path (the reference's sm89+ default,
    srt/layers/quantization/fp8_utils.py:1242-1276) and the 128x128 block path
    (fp8_utils.py:518-548).  Weights stay [N, K] row-major: that IS the column-major
    [K, N] operand fp8_scaled_mm wants, so the reference's `weight.t()` view is a no-op here.
    """

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.cutlass_fp8_supported = True  # per-token x per-channel GEMM is available
        self.block_quant = quant_config.weight_block_size is not None
        self.static_act = quant_config.activation_scheme == "static"

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                       output_size, params_dtype, **extra_weight_attrs):
        n = sum(output_partition_sizes)
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = n
        layer.orig_dtype = params_dtype
        serialized = self.quant_config.is_checkpoint_fp8_serialized
        wdtype = FP8_DTYPE if serialized else params_dtype
        weight = Parameter(torch.empty(n, input_size_per_partition, dtype=wdtype),
                           requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        set_weight_attrs(weight, extra_weight_attrs)
        layer.register_parameter("weight", weight)
        if serialized:
            if self.block_quant:
                bn, bk = self.quant_config.weight_block_size
                scale = Parameter(torch.full(((n + bn - 1) // bn,
                                              (input_size_per_partition + bk - 1) // bk),
                                             torch.finfo(torch.float32).min, dtype=torch.float32),
                                  requires_grad=False)
                layer.register_parameter("weight_scale_inv", scale)
            else:
                scale = Parameter(torch.full((len(output_partition_sizes),),
                                             torch.finfo(torch.float32).min, dtype=torch.float32),
                                  requires_grad=False)
                layer.register_parameter("weight_scale", scale)
            set_weight_attrs(scale, extra_weight_attrs)
        if self.static_act:
            # one calibrated activation scale per logical shard (fp8.py:284-298)
            if not serialized:
                raise ValueError("static activation scales come from an fp8-serialised checkpoint")
            iscale = Parameter(torch.full((len(output_partition_sizes),),
                                          torch.finfo(torch.float32).min, dtype=torch.float32),
                               requires_grad=False)
            set_weight_attrs(iscale, extra_weight_attrs)
            layer.register_parameter("input_scale", iscale)
        else:
            layer.register_parameter("input_scale", None)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        if self.block_quant:
            layer.weight = Parameter(layer.weight.data.contiguous(), requires_grad=False)
            layer.weight_scale_inv = Parameter(layer.weight_scale_inv.data.contiguous(),
                                               requires_grad=False)
            return
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # bf16 checkpoint: per-channel online quantisation = per_token_group_quant_fp8(weight,
            # K) (fp8.py:357-364): eps 1e-10, scale = absmax/448, q = clamp(w / scale).
            w = layer.weight.data
            n, k = w.shape
            if not w.is_cuda:
                raise RuntimeError("fp8 weight quantisation runs on the HIP device: move the "
                                   "layer to the GPU before process_weights_after_loading")
            q = torch.empty((n, k), dtype=FP8_DTYPE, device=w.device)
            s = torch.empty((n, 1), dtype=torch.float32, device=w.device)
            _C.per_token_group_quant_fp8(w.contiguous(), q, s, k, 1e-10, -FP8_MAX, FP8_MAX, False)
            layer.weight = Parameter(q.contiguous(), requires_grad=False)
            layer.weight_scale = Parameter(s.reshape(-1).contiguous(), requires_grad=False)
        else:
            # fp8 checkpoint with one scale per logical shard -> per-channel
            # (convert_to_channelwise, fp8.py:392-396)
            ws = layer.weight_scale.data
            if ws.numel() != layer.weight.shape[0]:
                ws = torch.cat([ws[i].expand(wd) for i, wd in enumerate(layer.logical_widths)])
            layer.weight = Parameter(layer.weight.data.contiguous(), requires_grad=False)
            layer.weight_scale = Parameter(ws.float().contiguous(), requires_grad=False)
        if self.static_act:
            # the shards share one activation: keep the largest scale (fp8.py:423-433)
            layer.input_scale = Parameter(layer.input_scale.data.float().max().reshape(1),
                                          requires_grad=False)
        else:
            layer.input_scale = None

    def apply(self, layer, x, bias=None) -> torch.Tensor:
        if x.dtype != torch.bfloat16:
            raise TypeError(f"bf16 activations expected, got {x.dtype}")
        x2 = _flat2d(x)
        M, K = x2.shape
        N = layer.weight.shape[0]
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        q = torch.empty((M, K), dtype=FP8_DTYPE, device=x.device)
        if self.block_quant:
            bn, bk = self.quant_config.weight_block_size
            if (bn, bk) != (128, 128):
                raise NotImplementedError("only 128x128 weight blocks are built")
            xs = torch.empty((M, K // bk), dtype=torch.float32, device=x.device)
            _C.per_token_group_quant_fp8(x2, q, xs, bk, 1e-10, -FP8_MAX, FP8_MAX, False)
            ws = layer.weight_scale_inv
            _C.fp8_blockwise_scaled_mm(q, layer.weight, xs, ws, out, (xs.stride(0), xs.stride(1)),
                                       (ws.stride(1), ws.stride(0)))
            if bias is not None:
                out += bias  # `output += bias` after the bf16 cast (fp8_utils.py:546-548)
        elif layer.input_scale is not None:
            # static per-tensor scale: q = clamp(x * (1/s)), the scale repeated per token for
            # the per-token x per-channel GEMM (static_quant_fp8(..., repeat_scale=True),
            # srt/layers/quantization/fp8_kernel.py:607-700, fp8_utils.py:1234-1240)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            _C.per_tensor_quant_fp8(x2, q, layer.input_scale, True)
            xs = layer.input_scale.expand(M).contiguous()
            _C.fp8_scaled_mm(q, layer.weight, xs, layer.weight_scale, bias, out)
        else:
            xs = torch.empty((M,), dtype=torch.float32, device=x.device)
            _C.per_token_quant_fp8(x2, q, xs)
            _C.fp8_scaled_mm(q, layer.weight, xs, layer.weight_scale, bias, out)
        return out.view(*x.shape[:-1], N)


# registry, quantization/__init__.py:10-56
_CUSTOMIZED_METHOD_TO_QUANT_CONFIG: Dict[str, type] = {"fp8": Fp8Config}
QUANTIZATION_METHODS = ["fp8"]


def register_quantization_config(quantization: str):
    def _wrapper(cls):
        if not issubclass(cls, QuantizationConfig):
            raise ValueError("The quantization config must be a subclass of `QuantizationConfig`.")
        _CUSTOMIZED_METHOD_TO_QUANT_CONFIG[quantization] = cls
        if quantization not in QUANTIZATION_METHODS:
            QUANTIZATION_METHODS.append(quantization)
        return cls

    return _wrapper


def get_quantization_config(quantization: str) -> type:
    if quantization not in _CUSTOMIZED_METHOD_TO_QUANT_CONFIG:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return _CUSTOMIZED_METHOD_TO_QUANT_CONFIG[quantization]
