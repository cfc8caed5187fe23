"""AWQ int4 (group-quantised, zero-point) linear method.

Mirrors AWQConfig / AWQLinearMethod (python/sglang/srt/layers/quantization/awq.py:27-101,
104-204).  The reference dequantises the whole weight to fp16 on every call and then runs a
dense matmul (:199-203); here the checkpoint tensors are repacked once in
process_weights_after_loading and `apply` is one fused dequant+MFMA GEMM (mi_w4a16_gemm).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
from torch.nn import Parameter

from .. import ops
from .._compat import (GroupQuantScaleParameter, LinearBase, LinearMethodBase, PackedvLLMParameter,
                       QuantizationConfig, UnquantizedLinearMethod)
from .._lib import MI_W4_AWQ


def is_layer_skipped_awq(prefix: str, modules_to_not_convert: List[str]) -> bool:
    return any(module_name in prefix for module_name in modules_to_not_convert)  # awq.py:23-24


class AWQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, zero_point: bool,
                 modules_to_not_convert: Optional[List[str]] = None) -> None:
        super().__init__()
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.modules_to_not_convert = modules_to_not_convert or []
        if self.weight_bits != 4:
            raise ValueError("Currently, only 4-bit weight quantization is supported for AWQ, "
                             f"but got {self.weight_bits} bits.")
        self.pack_factor = 32 // self.weight_bits

    def __repr__(self) -> str:
        return (f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, "
                f"zero_point={self.zero_point}, modules_to_not_convert={self.modules_to_not_convert})")

    def get_scaled_act_names(self) -> List[str]:
        return []

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]  # the reference allows half only (awq.py:67-68)

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]),
                   cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]),
                   cls.get_from_keys_or(config, ["modules_to_not_convert"], None))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if isinstance(layer, LinearBase) or hasattr(layer, "output_partition_sizes") or prefix == "":
            if is_layer_skipped_awq(prefix, self.modules_to_not_convert):
                return UnquantizedLinearMethod()
            return AWQLinearMethod(self)
        return None


class W4FusedDecodeMixin:
    """Decode-batch fused forms of an int4 linear whose weights are in the native layout (after
    process_weights_after_loading): the GEMM's split-K slabs are consumed by ONE kernel that also runs the decoder
    layer's next op(s) -- what harness.LlamaStack.forward_decode_fused16 calls.  16-bit activations in and out;
    bit-identical to apply() followed by the unfused kernels."""

    @staticmethod
    def fused_decode16_ok(layer: torch.nn.Module, M: int) -> bool:
        if not getattr(layer, "mi_w4_native", False) or getattr(layer, "mi_perm", None) is not None:
            return False
        if getattr(layer, "bias", None) is not None:
            return False
        K = layer.qweight.numel() * 8 // layer.mi_out_features
        return ops.w4a16_fused_ok(M, layer.mi_out_features, K, layer.mi_group_size)

    def apply_add_rmsnorm16(self, layer, x, residual, norm_weight, eps):
        return ops.w4a16_gemm_add_rmsnorm(x, layer.qweight, layer.qzeros, layer.mi_out_features, layer.mi_group_size,
                                          residual, norm_weight, eps)

    def apply_rope_kvwrite16(self, layer, x, positions, cos_sin_cache, k_cache, v_cache, loc, num_q_heads, num_kv_heads,
                             head_dim):
        return ops.w4a16_gemm_rope_kvwrite(x, layer.qweight, layer.qzeros, layer.mi_group_size, positions, cos_sin_cache,
                                           k_cache, v_cache, loc, num_q_heads, num_kv_heads, head_dim)

    def apply_silu_mul16(self, layer, x):
        return ops.w4a16_gemm_silu_mul(x, layer.qweight, layer.qzeros, layer.mi_out_features, layer.mi_group_size)


class AWQLinearMethod(W4FusedDecodeMixin, LinearMethodBase):
    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        group = cfg.group_size if cfg.group_size != -1 else input_size_per_partition
        if input_size_per_partition % group != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        output_size_per_partition = sum(output_partition_sizes)
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        weight_loader = extra_weight_attrs.get("weight_loader")
        qweight = PackedvLLMParameter(
            data=torch.empty(input_size_per_partition, output_size_per_partition // cfg.pack_factor, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=cfg.pack_factor, weight_loader=weight_loader)
        qzeros = PackedvLLMParameter(
            data=torch.empty(input_size_per_partition // group, output_size_per_partition // cfg.pack_factor,
                             dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=cfg.pack_factor, weight_loader=weight_loader)
        scales = GroupQuantScaleParameter(
            data=torch.empty(input_size_per_partition // group, output_size_per_partition, dtype=params_dtype),
            input_dim=0, output_dim=1, weight_loader=weight_loader)
        layer.register_parameter("qweight", qweight)
        layer.register_parameter("qzeros", qzeros)
        layer.register_parameter("scales", scales)
        layer.mi_group_size = group
        layer.mi_out_features = output_size_per_partition

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        qw, zs, _ = ops.w4_repack(layer.qweight.data.contiguous(), layer.qzeros.data.contiguous(),
                                  layer.scales.data.contiguous(), layer.mi_group_size, MI_W4_AWQ)
        # keep the reference's attribute names alive (awq.py:183-186) but free the checkpoint layout
        layer.qweight = Parameter(qw, requires_grad=False)
        layer.qzeros = Parameter(zs, requires_grad=False)
        layer.scales = Parameter(layer.scales.data, requires_grad=False)
        layer.mi_w4_native = True

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not getattr(layer, "mi_w4_native", False):
            raise RuntimeError("AWQLinearMethod.apply before process_weights_after_loading")
        N = layer.mi_out_features
        out_shape = x.shape[:-1] + (N,)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = ops.w4a16_gemm(x2, layer.qweight, layer.qzeros, N, layer.mi_group_size, None, bias)
        return out.reshape(out_shape)
