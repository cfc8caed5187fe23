"""Interface mirrors of the reference types our plugins sit behind.

When SGLang is importable the real classes are used (so the plugins ARE subclasses of the
reference's ABCs and parameter classes and the unmodified TP weight loaders work).  When it is
not (this repo's tests, the GPU box) minimal stand-ins with the same names, attributes and
method signatures are used -- they carry no arithmetic.  Citations: /root/reference paths.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from enum import IntEnum, auto
from typing import Any, Dict, List, Optional

import torch
from torch.nn import Parameter

try:  # pragma: no cover - exercised only inside an SGLang install
    from sglang.srt.layers.attention.base_attn_backend import AttentionBackend
    from sglang.srt.layers.linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
    from sglang.srt.layers.parameter import (ChannelQuantScaleParameter, GroupQuantScaleParameter,
                                             ModelWeightParameter, PackedvLLMParameter, PerTensorScaleParameter)
    from sglang.srt.layers.quantization.base_config import QuantizationConfig, QuantizeMethodBase
    from sglang.srt.model_executor.forward_batch_info import ForwardMode

    HAVE_SGLANG = True
except Exception:  # ImportError or any of SGLang's own import-time failures
    HAVE_SGLANG = False

    class ForwardMode(IntEnum):
        """python/sglang/srt/model_executor/forward_batch_info.py:60-123."""
        EXTEND = auto()
        DECODE = auto()
        MIXED = auto()
        IDLE = auto()
        TARGET_VERIFY = auto()
        DRAFT_EXTEND = auto()
        DUMMY_FIRST = auto()

        def is_extend(self):
            return self in (ForwardMode.EXTEND, ForwardMode.MIXED, ForwardMode.DRAFT_EXTEND,
                            ForwardMode.TARGET_VERIFY)

        def is_prefill(self):
            return self.is_extend()

        def is_decode(self):
            return self == ForwardMode.DECODE

        def is_mixed(self):
            return self == ForwardMode.MIXED

        def is_idle(self):
            return self == ForwardMode.IDLE

        def is_target_verify(self):
            return self == ForwardMode.TARGET_VERIFY

        def is_draft_extend(self):
            return self == ForwardMode.DRAFT_EXTEND

        def is_decode_or_idle(self):
            return self in (ForwardMode.DECODE, ForwardMode.IDLE)

        def is_cuda_graph(self):
            return self in (ForwardMode.DECODE, ForwardMode.TARGET_VERIFY, ForwardMode.IDLE)

    class AttentionBackend(ABC):
        """python/sglang/srt/layers/attention/base_attn_backend.py:14-115."""

        @abstractmethod
        def init_forward_metadata(self, forward_batch):
            raise NotImplementedError()

        def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
            raise NotImplementedError()

        def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                     encoder_lens, forward_mode, spec_info):
            raise NotImplementedError()

        def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                    encoder_lens, forward_mode, spec_info, seq_lens_cpu):
            raise NotImplementedError()

        def get_cuda_graph_seq_len_fill_value(self):
            raise NotImplementedError()

        def forward(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True, **kwargs):
            if forward_batch.forward_mode.is_decode():
                return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)
            return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)

        def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
            raise NotImplementedError()

        def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
            raise NotImplementedError()

        def support_triton(self):
            return True

    class QuantizeMethodBase(ABC):
        """python/sglang/srt/layers/quantization/base_config.py:11-35."""

        @abstractmethod
        def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
            raise NotImplementedError

        @abstractmethod
        def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
            raise NotImplementedError

        def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
            return

    class LinearMethodBase(QuantizeMethodBase):
        """python/sglang/srt/layers/linear.py:111-149."""

    class UnquantizedLinearMethod(LinearMethodBase):
        """linear.py:152-193.  In a real install skipped / unquantised layers are handled by the reference's own class;
        this stand-in does the same thing -- a plain library GEMM (F.linear -> hipBLASLt), not a hot-path kernel."""

        def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size,
                           output_size, params_dtype, **extra):
            w = Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
                          requires_grad=False)
            layer.register_parameter("weight", w)

        def apply(self, layer, x, bias=None):
            return torch.nn.functional.linear(x, layer.weight, bias)

    class LinearBase(torch.nn.Module):
        """Marker base (python/sglang/srt/layers/linear.py:153)."""

    class QuantizationConfig(ABC):
        """python/sglang/srt/layers/quantization/base_config.py:38-128."""

        def __init__(self):
            super().__init__()
            self.packed_modules_mapping: Dict[str, List[str]] = dict()

        @abstractmethod
        def get_name(self) -> str: ...

        @abstractmethod
        def get_supported_act_dtypes(self) -> List[torch.dtype]: ...

        @classmethod
        @abstractmethod
        def get_min_capability(cls) -> int: ...

        @staticmethod
        @abstractmethod
        def get_config_filenames() -> List[str]: ...

        @classmethod
        @abstractmethod
        def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig": ...

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
        def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]: ...

        @abstractmethod
        def get_scaled_act_names(self) -> List[str]: ...

    class _Param(Parameter):
        """Stand-in for layers/parameter.py classes: a Parameter that remembers the loader attributes
        (input_dim/output_dim/packed_dim/packed_factor/weight_loader) the TP loaders read."""

        def __new__(cls, data: torch.Tensor, **kwargs):
            return super().__new__(cls, data=data, requires_grad=False)

        def __init__(self, data: torch.Tensor, weight_loader=None, input_dim=None, output_dim=None,
                     packed_dim=None, packed_factor=None, **kwargs):
            self.weight_loader = weight_loader
            self.input_dim, self.output_dim = input_dim, output_dim
            self.packed_dim, self.packed_factor = packed_dim, packed_factor

    class ModelWeightParameter(_Param):
        pass

    class PerTensorScaleParameter(_Param):
        pass

    class PackedvLLMParameter(_Param):
        pass

    class GroupQuantScaleParameter(_Param):
        pass

    class ChannelQuantScaleParameter(_Param):
        pass
