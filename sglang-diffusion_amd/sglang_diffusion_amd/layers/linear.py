"""Linear layers, mirroring python/sglang/multimodal_gen/runtime/layers/linear.py:162-431.

The Flux DiT only instantiates ColumnParallelLinear(gather_output=True) with tp=1
(models/dits/flux.py:256-333,473-487,788-831), so tensor-parallel sharding is not built:
tp_size is fixed at 1 and `forward` returns `(output, None)` exactly like the reference.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch.nn import Parameter

from .quantization import (QuantizationConfig, QuantizeMethodBase, UnquantizedLinearMethod,
                           set_weight_attrs)


class LinearBase(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = ""):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        self.params_dtype = params_dtype or torch.get_default_dtype()
        self.quant_config = quant_config
        self.prefix = prefix
        if quant_config is None:
            self.quant_method: Optional[QuantizeMethodBase] = UnquantizedLinearMethod()
        else:
            self.quant_method = quant_config.get_quant_method(self, prefix=prefix)

    def forward(self, x: torch.Tensor):
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = True,
                 skip_bias_add: bool = False, params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = "",
                 output_sizes: Optional[List[int]] = None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        assert self.quant_method is not None
        self.quant_method.create_weights(self, self.input_size, output_sizes or [self.output_size],
                                         self.input_size, self.output_size, self.params_dtype,
                                         weight_loader=self.weight_loader)
        if bias:
            self.bias = Parameter(torch.empty(self.output_size, dtype=self.params_dtype),
                                  requires_grad=False)
            set_weight_attrs(self.bias, {"output_dim": 0, "weight_loader": self.weight_loader})
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param: Parameter, loaded_weight: torch.Tensor) -> None:
        if len(loaded_weight.shape) == 0:
            loaded_weight = loaded_weight.reshape(1)
        assert param.size() == loaded_weight.size(), (
            f"Tried to load weights of size {loaded_weight.size()} to a parameter of size "
            f"{param.size()}")
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor):
        bias = self.bias if not self.skip_bias_add else None
        assert self.quant_method is not None
        output = self.quant_method.apply(self, x, bias)
        return output, (self.bias if self.skip_bias_add else None)

    def extra_repr(self) -> str:
        return (f"in_features={self.input_size}, output_features={self.output_size}, "
                f"bias={self.bias is not None}")


class ColumnParallelLinear(ReplicatedLinear):
    """tp_size == 1: identical to the replicated layer; `gather_output` is accepted and is a
    no-op, as tensor_model_parallel_all_gather is for a single rank (linear.py:415-419)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True,
                 gather_output: bool = False, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None,
                 output_sizes: Optional[List[int]] = None, prefix: str = "", tp_group=None):
        if tp_group is not None and tp_group.size() != 1:
            raise NotImplementedError("tensor parallel linears are outside the DiT hot path")
        self.tp_size, self.tp_rank = 1, 0
        self.gather_output = gather_output
        super().__init__(input_size, output_size, bias, skip_bias_add, params_dtype, quant_config,
                         prefix, output_sizes)
        self.input_size_per_partition = input_size
        self.output_size_per_partition = output_size
        self.output_partition_sizes = output_sizes or [output_size]


class MergedColumnParallelLinear(ColumnParallelLinear):
    """linear.py:434-: several logical outputs packed along N (used for fused QKV)."""

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = True,
                 gather_output: bool = False, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None,
                 quant_config: Optional[QuantizationConfig] = None, prefix: str = ""):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias, gather_output, skip_bias_add,
                         params_dtype, quant_config, list(output_sizes), prefix)
