"""Per-tensor FP8 (OCP e4m3fn) linear method.

Mirrors Fp8Config / Fp8LinearMethod (python/sglang/srt/layers/quantization/fp8.py:106-182,
185-465, non-block, non-marlin branch) and apply_fp8_linear's HIP branch
(quantization/fp8_utils.py:654-749): static or dynamic per-tensor activation scale,
per-tensor weight scale, `torch._scaled_mm` semantics -- computed by our HIP kernels
(mi_fp8_quant_per_tensor / _per_token + mi_fp8_gemm).  gfx950 is OCP fp8, so none of the
reference's fnuz renormalisation applies (fp8_kernel.py:51-63 keys it on gfx94).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
from torch.nn import Parameter

from .. import ops
from .._compat import (LinearBase, LinearMethodBase, ModelWeightParameter, PerTensorScaleParameter,
                       QuantizationConfig, UnquantizedLinearMethod)

ACTIVATION_SCHEMES = ["static", "dynamic"]
FP8 = torch.float8_e4m3fn


def _is_layer_skipped(prefix: str, ignored_layers: List[str]) -> bool:
    # quantization/utils.py:20-55 (unfused names only; fused shards are resolved by the caller's mapping)
    return prefix in ignored_layers


class Fp8Config(QuantizationConfig):
    """fp8.py:106-182."""

    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic",
                 ignored_layers: Optional[List[str]] = None, weight_block_size: Optional[List[int]] = None) -> None:
        super().__init__()
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []
        if weight_block_size is not None:
            raise NotImplementedError("block-wise FP8 is outside this hot path (SURVEY 2.1: block-fp8 OUT)")
        self.weight_block_size = None

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def get_min_capability(cls) -> int:
        return 80  # compared with major*10+minor; gfx950 reports (9,5) -> 95 (model_loader/loader.py:130-141)

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized="fp8" in quant_method,
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]),
                   ignored_layers=cls.get_from_keys_or(config, ["ignored_layers"], None),
                   weight_block_size=cls.get_from_keys_or(config, ["weight_block_size"], None))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if isinstance(layer, LinearBase) or hasattr(layer, "output_partition_sizes") or prefix == "":
            if _is_layer_skipped(prefix, self.ignored_layers):
                return UnquantizedLinearMethod()
            return Fp8LinearMethod(self)
        if hasattr(layer, "tp_k_head_num") and hasattr(layer, "k_scale"):     # RadixAttention (fp8.py:210-211)
            from .kv_cache import Fp8KVCacheMethod
            return Fp8KVCacheMethod(self)
        return None  # MoE hooks: not on this path

    def get_scaled_act_names(self) -> List[str]:
        return []


def apply_fp8_linear(input: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                     use_per_token_if_dynamic: bool = False) -> torch.Tensor:
    """fp8_utils.py:510-749, HIP branch: quantise the activation, then one fused GEMM with the
    scale epilogue.  `weight` is the [K, N] view of [N, K] storage (fp8.py:406)."""
    input_2d = input.reshape(-1, input.shape[-1])
    if not input_2d.is_contiguous():
        input_2d = input_2d.contiguous()
    output_shape = [*input.shape[:-1], weight.shape[1]]
    if input_scale is not None:                      # static (:658)
        qx, xs = ops.fp8_quant_per_tensor(input_2d, input_scale.reshape(1))
    elif use_per_token_if_dynamic:                   # compressed-tensors dynamic (:582-588)
        qx, xs = ops.fp8_quant_per_token(input_2d)
        xs = xs.reshape(-1)
    else:                                            # dynamic per tensor (:669-674)
        qx, xs = ops.fp8_quant_per_tensor(input_2d)
    out = ops.fp8_gemm(qx, weight, xs, weight_scale.reshape(-1), input.dtype, bias)
    return out.view(*output_shape)


class Fp8FusedDecodeMixin:
    """Decode-shaped fused forms shared by the per-tensor FP8 linear methods (Fp8LinearMethod and the
    compressed-tensors W8A8-FP8 scheme with tensor scales)."""

    # this linear + the op(s) that consume its output in
    # a Llama decoder layer, one GEMM launch + one consumer launch, bit-identical to the unfused sequence.
    # All take an fp8 activation already quantised with layer.input_scale (static scheme).
    @staticmethod
    def fused_decode_ok(layer: torch.nn.Module, M: int) -> bool:
        s = getattr(layer, "input_scale", None)
        K = layer.weight.shape[0]
        return (s is not None and s.numel() == 1 and layer.weight_scale.numel() == 1 and 0 < M <= 512
                and K % 128 == 0 and getattr(layer, "bias", None) is None)

    @staticmethod
    def fused_silu_ok(layer: torch.nn.Module, M: int) -> bool:
        """apply_silu_mul at this row count: decode batches as fused_decode_ok; prefill (M > 512) through the tile
        kernel's epilogue when the halves are 128-column aligned."""
        s = getattr(layer, "input_scale", None)
        K, N = layer.weight.shape
        if s is None or s.numel() != 1 or layer.weight_scale.numel() != 1 or getattr(layer, "bias", None) is not None:
            return False
        return K % 128 == 0 and (M <= 512 or (N // 2) % 128 == 0)

    def apply_add_rmsnorm(self, layer, qx, residual, norm_weight, eps, next_scale=None, want_out=False):
        """(out | None, fp8 | None) = rmsnorm(linear(qx) + residual) * norm_weight [-> fp8 with next_scale]."""
        return ops.fp8_gemm_add_rmsnorm(qx, layer.weight, layer.input_scale.reshape(1), layer.weight_scale.reshape(1),
                                        residual, norm_weight, eps, next_scale, want_out)

    def apply_rope_kvwrite(self, layer, qx, positions, cos_sin_cache, k_cache, v_cache, loc, num_q_heads,
                           num_kv_heads, head_dim):
        """q = rope(linear(qx)[:, :q]); k (rope) and v rows go straight into the KV pool at `loc`."""
        return ops.fp8_gemm_rope_kvwrite(qx, layer.weight, layer.input_scale.reshape(1), layer.weight_scale.reshape(1),
                                         positions, cos_sin_cache, k_cache, v_cache, loc, num_q_heads, num_kv_heads,
                                         head_dim)

    def apply_silu_mul(self, layer, qx, next_scale, act_dtype):
        """fp8(silu(gate) * up) of linear(qx) = [gate | up], quantised with next_scale."""
        return ops.fp8_gemm_silu_mul(qx, layer.weight, layer.input_scale.reshape(1), layer.weight_scale.reshape(1),
                                     next_scale, act_dtype)


class Fp8LinearMethod(Fp8FusedDecodeMixin, LinearMethodBase):
    """fp8.py:185-465 restricted to per-tensor scales."""

    @staticmethod
    def static_input_scale(layer: torch.nn.Module) -> Optional[torch.Tensor]:
        """The calibrated per-tensor activation scale, if this layer has one: a producer kernel
        (ops.rmsnorm_fp8 / silu_and_mul_fp8) may then emit the fp8 activation directly."""
        s = getattr(layer, "input_scale", None)
        return s if s is not None and s.numel() == 1 else None

    def apply_prequantized(self, layer: torch.nn.Module, qx: torch.Tensor, out_dtype: torch.dtype,
                           bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`apply` minus the activation quantisation: qx is already fp8 with layer.input_scale.  `out`: write there
        (e.g. the all-reduce staging buffer of a row-parallel linear)."""
        return ops.fp8_gemm(qx, layer.weight, layer.input_scale.reshape(1), layer.weight_scale.reshape(-1),
                            out_dtype, bias, out)

    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                       output_partition_sizes: List[int], input_size: int, output_size: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        output_size_per_partition = sum(output_partition_sizes)
        weight_loader = extra_weight_attrs.get("weight_loader")
        layer.logical_widths = output_partition_sizes
        layer.input_size_per_partition = input_size_per_partition
        layer.output_size_per_partition = output_size_per_partition
        layer.orig_dtype = params_dtype
        weight_dtype = FP8 if self.quant_config.is_checkpoint_fp8_serialized else params_dtype
        weight = ModelWeightParameter(
            data=torch.empty(output_size_per_partition, input_size_per_partition, dtype=weight_dtype),
            input_dim=1, output_dim=0, weight_loader=weight_loader)
        layer.register_parameter("weight", weight)
        if self.quant_config.is_checkpoint_fp8_serialized:
            scale = PerTensorScaleParameter(data=torch.empty(len(output_partition_sizes), dtype=torch.float32),
                                            weight_loader=weight_loader)
            scale[:] = torch.finfo(torch.float32).min
            layer.register_parameter("weight_scale", scale)
            if self.quant_config.activation_scheme == "static":
                scale = PerTensorScaleParameter(data=torch.empty(len(output_partition_sizes), dtype=torch.float32),
                                                weight_loader=weight_loader)
                scale[:] = torch.finfo(torch.float32).min
                layer.register_parameter("input_scale", scale)
            else:
                layer.register_parameter("input_scale", None)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        w = layer.weight.data
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # bf16/fp16 checkpoint: per-tensor dynamic weight quant (fp8.py:359-366, input_to_float8:
            # scale = 448/amax, returns 1/scale) -- on the device through our quant kernel
            w2 = w.reshape(w.shape[0], -1).contiguous()
            qweight, weight_scale = ops.fp8_quant_per_tensor(w2, weight_mode=True)
            layer.weight = Parameter(qweight.t(), requires_grad=False)
            layer.weight_scale = Parameter(weight_scale, requires_grad=False)
            layer.input_scale = None
            return
        weight_scale = layer.weight_scale.data
        max_w_scale = weight_scale.max()
        # N shard scales -> one (requantize_with_max_scale, quantization/utils.py:94-119)
        unfused = bool(weight_scale[-1] > torch.finfo(FP8).min)
        if unfused and len(layer.logical_widths) > 1:
            start = 0
            for idx, width in enumerate(layer.logical_widths):
                end = start + width
                dq = (w[start:end, :].to(torch.float16) * weight_scale[idx]).contiguous()
                q, _ = ops.fp8_quant_per_tensor(dq, max_w_scale.reshape(1).to(torch.float32))
                w[start:end, :] = q
                start = end
        layer.weight = Parameter(w.t(), requires_grad=False)
        layer.weight_scale = Parameter(max_w_scale.reshape(1), requires_grad=False)
        if self.quant_config.activation_scheme == "static":
            layer.input_scale = Parameter(layer.input_scale.data.max().reshape(1), requires_grad=False)
        else:
            layer.input_scale = None

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return apply_fp8_linear(input=x, weight=layer.weight, weight_scale=layer.weight_scale,
                                input_scale=layer.input_scale, bias=bias, use_per_token_if_dynamic=False)
