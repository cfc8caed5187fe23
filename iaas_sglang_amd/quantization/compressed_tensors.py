"""`compressed-tensors` checkpoints, FP8 W8A8 scheme only -- the format of the reference's own FP8 test model
(neuralmagic/Meta-Llama-3.1-8B-Instruct-FP8, python/sglang/test/test_utils.py:54).

Mirrors the surface of CompressedTensorsConfig / CompressedTensorsLinearMethod
(quantization/compressed_tensors/compressed_tensors.py:76-160,286-330,389-399,614-660) and of the scheme
CompressedTensorsW8A8Fp8 (schemes/compressed_tensors_w8a8_fp8.py:28-160): per-tensor or per-channel weight
scales, static per-tensor or dynamic per-token activation scales; `apply_weights` is
apply_fp8_linear(..., use_per_token_if_dynamic=True) -> our mi_fp8_quant_* + mi_fp8_gemm (row scales are an
epilogue variant of the same GEMM).  The config is parsed from the plain `quantization_config` dict; neither the
`compressed_tensors` package nor pydantic is needed.  Every other scheme of the format (int8, wNa16, sparse-24,
MoE) is outside this hot path and raises NotImplementedError.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional

import torch
from torch.nn import Parameter

from .. import ops
from .._compat import (ChannelQuantScaleParameter, LinearBase, LinearMethodBase, ModelWeightParameter,
                       PerTensorScaleParameter, QuantizationConfig, UnquantizedLinearMethod)
from .fp8 import FP8, Fp8FusedDecodeMixin, apply_fp8_linear

ACTIVATION_QUANT_FORMATS = ("float-quantized", "naive-quantized", "int-quantized")   # formats that carry input_activations


class CompressedTensorsW8A8Fp8:
    """One layer's scheme: fp8 e4m3 weights with `strategy` in {"tensor", "channel"} scales; activations static
    per tensor (`is_static_input_scheme`) or dynamic per token."""

    def __init__(self, strategy: str, is_static_input_scheme: bool):
        if strategy not in ("tensor", "channel"):
            raise ValueError(f"Unknown quantization strategy {strategy}")
        self.strategy = strategy
        self.is_static_input_scheme = is_static_input_scheme

    @classmethod
    def get_min_capability(cls) -> int:
        return 89

    def create_weights(self, layer: torch.nn.Module, output_partition_sizes: List[int], input_size_per_partition: int,
                       params_dtype: torch.dtype, weight_loader, **kwargs):
        n = sum(output_partition_sizes)
        layer.logical_widths = output_partition_sizes
        layer.register_parameter("weight", ModelWeightParameter(
            data=torch.empty(n, input_size_per_partition, dtype=FP8), input_dim=1, output_dim=0,
            weight_loader=weight_loader))
        if self.strategy == "channel":
            scale = ChannelQuantScaleParameter(data=torch.empty((n, 1), dtype=torch.float32), output_dim=0,
                                               weight_loader=weight_loader)
        else:
            scale = PerTensorScaleParameter(data=torch.empty(len(output_partition_sizes), dtype=torch.float32),
                                            weight_loader=weight_loader)
        scale[:] = torch.finfo(torch.float32).min
        layer.register_parameter("weight_scale", scale)
        if self.is_static_input_scheme:
            inp = PerTensorScaleParameter(data=torch.empty(len(output_partition_sizes), dtype=torch.float32),
                                          weight_loader=weight_loader)
            inp[:] = torch.finfo(torch.float32).min
            layer.register_parameter("input_scale", inp)

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        w = layer.weight.data
        if self.strategy == "tensor":
            # fused modules (qkv, gate_up) arrive with one scale per shard: requantise every shard with the
            # largest one so that a single per-tensor scale serves the GEMM (quantization/utils.py:94-119)
            ws = layer.weight_scale.data
            max_scale = ws.max()
            if len(layer.logical_widths) > 1 and bool(ws[-1] > torch.finfo(FP8).min):
                start = 0
                for idx, width in enumerate(layer.logical_widths):
                    dq = (w[start:start + width].to(torch.float16) * ws[idx]).contiguous()
                    q, _ = ops.fp8_quant_per_tensor(dq, max_scale.reshape(1).to(torch.float32))
                    w[start:start + width] = q
                    start += width
            layer.weight = Parameter(w.t(), requires_grad=False)
            layer.weight_scale = Parameter(max_scale.reshape(1), requires_grad=False)
        else:   # channel scales already line up with the rows: just take the [K, N] view
            layer.weight = Parameter(w.t(), requires_grad=False)
            layer.weight_scale = Parameter(layer.weight_scale.data, requires_grad=False)
        if self.is_static_input_scheme and getattr(layer, "input_scale", None) is not None:
            layer.input_scale = Parameter(layer.input_scale.data.max().reshape(1), requires_grad=False)
        else:
            layer.input_scale = None

    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return apply_fp8_linear(input=x, weight=layer.weight, weight_scale=layer.weight_scale,
                                input_scale=layer.input_scale, bias=bias, use_per_token_if_dynamic=True)


class CompressedTensorsLinearMethod(Fp8FusedDecodeMixin, LinearMethodBase):
    """compressed_tensors.py:614-660: delegates to the scheme attached to the layer by get_quant_method."""

    def __init__(self, quantization_config: "CompressedTensorsConfig"):
        self.quantization_config = quantization_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        layer.scheme.create_weights(layer=layer, input_size=input_size, input_size_per_partition=input_size_per_partition,
                                    output_partition_sizes=output_partition_sizes, output_size=output_size,
                                    params_dtype=params_dtype, weight_loader=extra_weight_attrs.get("weight_loader"))

    def process_weights_after_loading(self, layer) -> None:
        layer.scheme.process_weights_after_loading(layer)

    def apply(self, layer, x, bias=None):
        if getattr(layer, "scheme", None) is None:
            raise ValueError("A scheme must be defined for each layer")
        return layer.scheme.apply_weights(layer, x, bias=bias)

    # the fused decode forms (Fp8FusedDecodeMixin) apply unchanged when the scheme is per-tensor / static
    @staticmethod
    def static_input_scale(layer):
        s = getattr(layer, "input_scale", None)
        return s if s is not None and s.numel() == 1 and layer.weight_scale.numel() == 1 else None

    def apply_prequantized(self, layer, qx, out_dtype, bias=None, out=None):
        return ops.fp8_gemm(qx, layer.weight, layer.input_scale.reshape(1), layer.weight_scale.reshape(-1), out_dtype,
                            bias, out)


class CompressedTensorsConfig(QuantizationConfig):
    """compressed_tensors.py:76-160: target -> {weights, input_activations} from `config_groups`."""

    def __init__(self, target_scheme_map: Dict[str, Any], ignore: List[str], quant_format: Optional[str],
                 config: Optional[Dict[str, Any]] = None, packed_modules_mapping: Optional[Dict[str, List[str]]] = None):
        super().__init__()
        self.target_scheme_map = target_scheme_map
        self.ignore = ignore
        self.quant_format = quant_format
        self.config = config
        self.packed_modules_mapping = packed_modules_mapping or {}

    def get_name(self) -> str:
        return "compressed_tensors"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 70

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    def get_scaled_act_names(self) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CompressedTensorsConfig":
        fmt = config.get("format")
        scheme_map: Dict[str, Any] = {}
        for group in (config.get("config_groups") or {}).values():
            for target in group.get("targets") or []:
                acts = group.get("input_activations") if fmt in ACTIVATION_QUANT_FORMATS else None
                scheme_map[target] = {"weights": dict(group.get("weights") or {}),
                                      "input_activations": dict(acts) if acts else None}
        return cls(scheme_map, list(config.get("ignore") or []), fmt, config=config,
                   packed_modules_mapping=config.get("packed_modules_mapping") or {})

    # ---- scheme selection
    @staticmethod
    def _is_fp8_w8a8(wq: Optional[Dict[str, Any]], aq: Optional[Dict[str, Any]]) -> bool:
        """compressed_tensors.py:286-315: float weights, symmetric, static, per tensor/channel; activations
        dynamic (any) or static symmetric per tensor."""
        if not wq or not aq:
            return False
        if wq.get("type") != "float" or aq.get("type") != "float" or wq.get("num_bits") != 8:
            return False
        if not wq.get("symmetric", True) or wq.get("dynamic", False) or wq.get("strategy") not in ("tensor", "channel"):
            return False
        if aq.get("dynamic", False):
            return True
        return bool(aq.get("symmetric", True)) and aq.get("strategy") == "tensor"

    def _match_target(self, layer: torch.nn.Module, layer_name: Optional[str]) -> Optional[str]:
        """A target is a layer name, a `re:` pattern on it, or the module's class name
        (compressed_tensors.py:436-470 / utils.find_matched_target)."""
        names = [layer_name] if layer_name else []
        # a fused module (qkv_proj) matches through any of its shards (q_proj, ...)
        for fused, shards in self.packed_modules_mapping.items():
            if layer_name and layer_name.endswith(fused):
                names += [layer_name[: -len(fused)] + sh for sh in shards]
        cls_names = [type(layer).__name__]
        if any("Linear" in c.__name__ for c in type(layer).__mro__):
            cls_names.append("Linear")
        for target in self.target_scheme_map:
            if target.startswith("re:"):
                if any(re.match(target[3:], n) for n in names):
                    return target
            elif target in names or any(n.endswith("." + target) for n in names) or target in cls_names:
                return target
        return None

    def get_scheme(self, layer: torch.nn.Module, layer_name: Optional[str] = None) -> Optional[CompressedTensorsW8A8Fp8]:
        target = self._match_target(layer, layer_name)
        if target is None:
            raise ValueError(f"Unable to find matching target for {layer_name} in the compressed-tensors config")
        wq, aq = self.target_scheme_map[target]["weights"], self.target_scheme_map[target]["input_activations"]
        if self._is_fp8_w8a8(wq, aq):
            return CompressedTensorsW8A8Fp8(strategy=wq["strategy"], is_static_input_scheme=not aq.get("dynamic", False))
        raise NotImplementedError("compressed-tensors: only the FP8 W8A8 scheme (float 8-bit weights per tensor/channel, "
                                  f"fp8 activations) is on this hot path; got weights={wq}, input_activations={aq}")

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if any(prefix == ig or prefix.endswith("." + ig) or (ig.startswith("re:") and re.match(ig[3:], prefix))
               for ig in self.ignore):
            return UnquantizedLinearMethod()
        if isinstance(layer, LinearBase) or hasattr(layer, "output_partition_sizes"):
            layer.scheme = self.get_scheme(layer=layer, layer_name=prefix)
            return CompressedTensorsLinearMethod(self)
        return None   # MoE / attention-layer hooks: not on this path
