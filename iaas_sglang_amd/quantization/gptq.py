"""GPTQ int4 linear method (AutoGPTQ v1 checkpoint convention, optional act-order).

Config mirrors GPTQConfig (python/sglang/srt/layers/quantization/gptq.py:57-158).  The
reference takes its GPTQLinearMethod and gptq_gemm from vllm (gptq.py:18-19,
quantization/__init__.py:25), which is not part of the reference tree: PARITY UNPINNED.
Arithmetic implemented: W[k,n] = (w[k,n] - (z[g_idx[k],n] + 1)) * s[g_idx[k],n], x @ W.
"""
from __future__ import annotations

from fractions import Fraction
from typing import Any, Dict, List, Optional

import torch
from torch.nn import Parameter

from .. import ops
from .._compat import (GroupQuantScaleParameter, LinearBase, LinearMethodBase, PackedvLLMParameter,
                       QuantizationConfig)
from .._lib import MI_W4_GPTQ
from .awq import W4FusedDecodeMixin


class GPTQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, lm_head_quantized: bool = False,
                 dynamic: Optional[Dict[str, Dict[str, Any]]] = None) -> None:
        super().__init__()
        self.dynamic = dynamic or {}
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        self.lm_head_quantized = lm_head_quantized
        self.pack_factor = Fraction(32, self.weight_bits)
        if self.weight_bits != 4:
            raise ValueError(f"Only 4-bit GPTQ is on this hot path (SURVEY 2.1), got {self.weight_bits} bits.")

    def __repr__(self) -> str:
        return (f"GPTQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}), lm_head_quantized={self.lm_head_quantized}), dynamic={self.dynamic}")

    def get_scaled_act_names(self) -> List[str]:
        raise NotImplementedError  # as the reference (gptq.py:123-128)

    @classmethod
    def get_name(cls) -> str:
        return "gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        dynamic = cls.get_from_keys_or(config, ["dynamic"], default={}) or {}
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]), cls.get_from_keys_or(config, ["lm_head"], default=False),
                   dynamic)

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if isinstance(layer, LinearBase) or hasattr(layer, "output_partition_sizes") or prefix == "":
            return GPTQLinearMethod(self)
        return None


class GPTQLinearMethod(W4FusedDecodeMixin, LinearMethodBase):
    def __init__(self, quant_config: GPTQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        group = cfg.group_size if cfg.group_size != -1 else input_size
        if input_size_per_partition % group != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        N = sum(output_partition_sizes)
        weight_loader = extra_weight_attrs.get("weight_loader")
        qweight = PackedvLLMParameter(data=torch.empty(input_size_per_partition // 8, N, dtype=torch.int32),
                                      input_dim=0, output_dim=1, packed_dim=0, packed_factor=8,
                                      weight_loader=weight_loader)
        g_idx = torch.nn.Parameter(torch.arange(input_size_per_partition, dtype=torch.int32) // group,
                                   requires_grad=False)
        qzeros = PackedvLLMParameter(data=torch.empty(input_size_per_partition // group, N // 8, dtype=torch.int32),
                                     input_dim=0, output_dim=1, packed_dim=1, packed_factor=8,
                                     weight_loader=weight_loader)
        scales = GroupQuantScaleParameter(data=torch.empty(input_size_per_partition // group, N, dtype=params_dtype),
                                          input_dim=0, output_dim=1, weight_loader=weight_loader)
        layer.register_parameter("qweight", qweight)
        layer.register_parameter("g_idx", g_idx)
        layer.register_parameter("qzeros", qzeros)
        layer.register_parameter("scales", scales)
        layer.mi_group_size = group
        layer.mi_out_features = N

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        g_idx = layer.g_idx.data if self.quant_config.desc_act else None
        qw, zs, perm = ops.w4_repack(layer.qweight.data.contiguous(), layer.qzeros.data.contiguous(),
                                     layer.scales.data.contiguous(), layer.mi_group_size, MI_W4_GPTQ, g_idx)
        layer.qweight = Parameter(qw, requires_grad=False)
        layer.qzeros = Parameter(zs, requires_grad=False)
        layer.scales = Parameter(layer.scales.data, requires_grad=False)
        layer.mi_perm = perm
        layer.mi_w4_native = True

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not getattr(layer, "mi_w4_native", False):
            raise RuntimeError("GPTQLinearMethod.apply before process_weights_after_loading")
        N = layer.mi_out_features
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = ops.w4a16_gemm(x2, layer.qweight, layer.qzeros, N, layer.mi_group_size, layer.mi_perm, bias)
        return out.reshape(x.shape[:-1] + (N,))
