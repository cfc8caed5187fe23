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

        @abstractmethod
        def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int,
                           output_partition_sizes: List[int], input_size: int, output_size: int,
                           params_dtype: torch.dtype, **extra_weight_attrs):
            raise NotImplementedError

        @abstractmethod
        def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
            raise NotImplementedError

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

    # ---- layers/parameter.py:29-441: the parameter classes a linear method registers and the TP weight loaders of
    # linear.py drive (`param.load_*_weight(...)`, isinstance checks on these very names).  Same names, inheritance,
    # properties and loader signatures; the loaders are slice-and-copy only.  tests/test_interfaces_cpu.py checks them
    # against the reference's classes, signature by signature and on reference-run vectors.
    def _copy_exact(dst: torch.Tensor, src: torch.Tensor) -> None:
        assert dst.shape == src.shape, f"{tuple(dst.shape)} != {tuple(src.shape)}"
        dst.copy_(src)

    class BasevLLMParameter(Parameter):
        def __new__(cls, data: torch.Tensor, **kwargs):
            return super().__new__(cls, data=data, requires_grad=False)

        def __init__(self, data: torch.Tensor, weight_loader):
            self._weight_loader = weight_loader

        @property
        def weight_loader(self):
            return self._weight_loader

        def _assert_and_load(self, loaded_weight: torch.Tensor):
            _copy_exact(self.data, loaded_weight)

        def load_column_parallel_weight(self, loaded_weight: torch.Tensor):
            _copy_exact(self.data, loaded_weight)

        def load_row_parallel_weight(self, loaded_weight: torch.Tensor):
            _copy_exact(self.data, loaded_weight)

        def load_merged_column_weight(self, loaded_weight: torch.Tensor, **kwargs):
            _copy_exact(self.data, loaded_weight)

        def load_qkv_weight(self, loaded_weight: torch.Tensor, **kwargs):
            _copy_exact(self.data, loaded_weight)

    class _ColumnvLLMParameter(BasevLLMParameter):
        """Sharded along `output_dim` (column parallelism; fused qkv / gate_up shards)."""

        def __init__(self, output_dim: int, **kwargs):
            self._output_dim = output_dim
            super().__init__(**kwargs)

        @property
        def output_dim(self):
            return self._output_dim

        def _packed_along_output(self) -> bool:
            return isinstance(self, PackedvLLMParameter) and self.packed_dim == self.output_dim

        def load_column_parallel_weight(self, loaded_weight: torch.Tensor, tp_rank: int,
                                        use_presharded_weights: bool = False):
            if not use_presharded_weights:
                n = self.data.shape[self.output_dim]
                loaded_weight = loaded_weight.narrow(self.output_dim, tp_rank * n, n)
            _copy_exact(self.data, loaded_weight)

        def load_merged_column_weight(self, loaded_weight: torch.Tensor, **kwargs):
            offset, size = kwargs.get("shard_offset"), kwargs.get("shard_size")
            if self._packed_along_output():
                size, offset = self.adjust_shard_indexes_for_packing(shard_offset=offset, shard_size=size)
            if not kwargs.get("use_presharded_weights"):
                loaded_weight = loaded_weight.narrow(self.output_dim, kwargs.get("tp_rank") * size, size)
            _copy_exact(self.data.narrow(self.output_dim, offset, size), loaded_weight)

        def load_qkv_weight(self, loaded_weight: torch.Tensor, tp_rank: int, use_presharded_weights: bool = False,
                            **kwargs):
            offset, size = kwargs.get("shard_offset"), kwargs.get("shard_size")
            if self._packed_along_output():
                size, offset = self.adjust_shard_indexes_for_packing(shard_offset=offset, shard_size=size)
            # q heads are split over the ranks; a kv head may be shared by `num_heads` ranks
            shard = tp_rank if kwargs.get("shard_id") == "q" else tp_rank // kwargs.get("num_heads")
            if not use_presharded_weights:
                loaded_weight = loaded_weight.narrow(self.output_dim, shard * size, size)
            _copy_exact(self.data.narrow(self.output_dim, offset, size), loaded_weight)

    class RowvLLMParameter(BasevLLMParameter):
        """Sharded along `input_dim` (row parallelism)."""

        def __init__(self, input_dim: int, **kwargs):
            self._input_dim = input_dim
            super().__init__(**kwargs)

        @property
        def input_dim(self):
            return self._input_dim

        def load_row_parallel_weight(self, loaded_weight: torch.Tensor, tp_rank: int,
                                     use_presharded_weights: bool = False):
            if not use_presharded_weights:
                n = self.data.shape[self.input_dim]
                loaded_weight = loaded_weight.narrow(self.input_dim, tp_rank * n, n)
            if loaded_weight.dim() == 0:
                loaded_weight = loaded_weight.reshape(1)
            _copy_exact(self.data, loaded_weight)

    class ModelWeightParameter(_ColumnvLLMParameter, RowvLLMParameter):
        pass

    class GroupQuantScaleParameter(_ColumnvLLMParameter, RowvLLMParameter):
        pass

    class ChannelQuantScaleParameter(_ColumnvLLMParameter):
        pass

    class PerTensorScaleParameter(BasevLLMParameter):
        """One scale per logical matrix of a fused module; loaders address the shard id, never the TP rank."""

        def __init__(self, **kwargs):
            self.qkv_idxs = {"q": 0, "k": 1, "v": 2}
            super().__init__(**kwargs)

        def _shard_id_as_int(self, shard_id) -> int:
            if isinstance(shard_id, int):
                return shard_id
            assert isinstance(shard_id, str) and shard_id in self.qkv_idxs
            return self.qkv_idxs[shard_id]

        def _load_whole(self, *args, **kwargs):
            kwargs.pop("tp_rank", None)
            kwargs.pop("use_presharded_weights", None)
            BasevLLMParameter.load_row_parallel_weight(self, *args, **kwargs)

        def load_row_parallel_weight(self, *args, **kwargs):
            self._load_whole(*args, **kwargs)

        def load_column_parallel_weight(self, *args, **kwargs):
            self._load_whole(*args, **kwargs)

        def load_merged_column_weight(self, *args, **kwargs):
            self._load_into_shard_id(*args, **kwargs)

        def load_qkv_weight(self, *args, **kwargs):
            self._load_into_shard_id(*args, **kwargs)

        def _load_into_shard_id(self, loaded_weight: torch.Tensor, shard_id, **kwargs):
            if loaded_weight.dim() != 0:          # compressed-tensors scales carry a shape of [1], AutoFP8's none
                assert loaded_weight.shape[0] == 1
                loaded_weight = loaded_weight[0]
            _copy_exact(self.data[self._shard_id_as_int(shard_id)], loaded_weight)

    class PackedvLLMParameter(ModelWeightParameter):
        """int4/int8 values packed `packed_factor` to a storage element along `packed_dim`."""

        def __init__(self, packed_factor, packed_dim: int, marlin_tile_size: Optional[int] = None, **kwargs):
            self._packed_factor, self._packed_dim, self._marlin_tile_size = packed_factor, packed_dim, marlin_tile_size
            super().__init__(**kwargs)

        @property
        def packed_dim(self):
            return self._packed_dim

        @property
        def packed_factor(self):
            return self._packed_factor

        @property
        def marlin_tile_size(self):
            return self._marlin_tile_size

        def adjust_shard_indexes_for_packing(self, shard_size, shard_offset):
            size, offset = shard_size // self.packed_factor, shard_offset // self.packed_factor
            if self.marlin_tile_size is not None:
                return size * self.marlin_tile_size, offset * self.marlin_tile_size
            return size, offset
