"""fp8 KV-cache scaling factors: the quant method attached to attention layers and the scales-file loader.

Mirrors (no arithmetic beyond the is-it-set logic):
  * `BaseKVCacheMethod`   python/sglang/srt/layers/quantization/kv_cache.py:17-82
  * `Fp8KVCacheMethod`    python/sglang/srt/layers/quantization/fp8.py:1151-1158
  * `kv_cache_scales_loader`  python/sglang/srt/model_loader/weight_utils.py:923-966 with the checks of its pydantic
    schema (`KVCacheQuantSchema` / `QuantParamSchema`, weight_utils.py:849-920) done on the plain dict -- file format
    of test/srt/kv_cache_scales_llama3_8b.json, selected with `--quantization-param-path`
  * `load_kv_cache_scales`    python/sglang/srt/models/llama.py:359-378
The scales end up where MiAttnBackend reads them: `layer.k_scale_float` / `layer.v_scale_float` (checkpoint scales)
or `layer.k_scale` / `layer.v_scale` (scales file), consumed by `mi_kv_write_fp8` and the fp8-KV attention kernels.
gfx950 stores OCP e4m3fn, so the reference's fnuz doubling (kv_cache.py:51-53) does not apply.
"""
from __future__ import annotations

import json
import logging
from typing import Iterable, List, Optional, Tuple

import torch

from .._compat import QuantizationConfig, QuantizeMethodBase

logger = logging.getLogger(__name__)


class BaseKVCacheMethod(QuantizeMethodBase):
    """Adds `k_scale` / `v_scale` to an attention layer so a checkpoint can load them."""

    def __init__(self, quant_config: QuantizationConfig):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module):
        # -1.0 = "not loaded"; a checkpoint that carries k_scale / v_scale overwrites it
        layer.k_scale = torch.nn.Parameter(torch.tensor(-1.0, dtype=torch.float32), requires_grad=False)
        layer.v_scale = torch.nn.Parameter(torch.tensor(-1.0, dtype=torch.float32), requires_grad=False)

    def apply(self, layer: torch.nn.Module) -> torch.Tensor:
        raise RuntimeError(f"{self.__class__.__name__}.apply should not be called.")

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        if layer.k_scale > 0.0 and layer.v_scale > 0.0:          # separate scales present
            k_scale = layer.k_scale.to("cpu").tolist()
            v_scale = layer.v_scale.to("cpu").tolist()
        elif layer.k_scale < 0.0 and layer.v_scale < 0.0:        # nothing loaded
            k_scale = v_scale = 1.0
        else:                                                     # a single kv_scale was remapped onto k_scale
            assert layer.k_scale > 0.0
            dup = max(layer.k_scale, layer.v_scale)
            k_scale = v_scale = dup.to("cpu").tolist()
        if not isinstance(k_scale, float) or not isinstance(v_scale, float):
            raise ValueError("Only support per-tensor scaling factor for fp8 KV cache")
        layer.k_scale.copy_(k_scale)
        layer.v_scale.copy_(v_scale)
        layer.k_scale_float = k_scale
        layer.v_scale_float = v_scale


class Fp8KVCacheMethod(BaseKVCacheMethod):
    """What `Fp8Config.get_quant_method` returns for an attention layer (fp8.py:1151-1158)."""


def kv_cache_scales_loader(filename: str, tp_rank: int, tp_size: int, num_hidden_layers: int,
                           model_type: Optional[str]) -> Iterable[Tuple[int, float]]:
    """(layer index, scaling factor) pairs of `tp_rank` from a KV-cache scales JSON.  Any problem with the file is
    logged and yields no scales (every layer then keeps 1.0), exactly like the reference."""
    try:
        with open(filename) as f:
            doc = json.load(f)
        if model_type is not None and doc.get("model_type") != model_type:
            raise ValueError(f"Model type is {model_type} but loaded scaling factors belonging to different model "
                             f"type {doc.get('model_type')}!")
        kv = doc["kv_cache"]
        if kv["dtype"] != "float8_e4m3fn":
            raise ValueError(f"Loaded scaling factors intended for KV cache dtype = {kv['dtype']} rather than "
                             "float8_e4m3fn!")
        factors = {int(r): {int(layer): float(s) for layer, s in m.items()} for r, m in kv["scaling_factor"].items()}
        if len(factors) != tp_size:
            raise ValueError(f"Loaded dictionary has TP size {len(factors)} but LLM engine is currently running with "
                             f"TP size {tp_size}.")
        for r in range(tp_size):
            if r not in factors:
                raise ValueError(f"KV cache scales map for TP rank {r} not found.")
            if len(factors[r]) != num_hidden_layers:
                raise ValueError(f"KV cache scales map for TP rank {r} is malformed. Expected {num_hidden_layers} "
                                 f"layers, got {len(factors[r])}.")
        for i in range(num_hidden_layers):
            if i not in factors[tp_rank]:
                raise ValueError(f"Could not find KV cache scales for layer {i} in TP rank {tp_rank}.")
        return factors[tp_rank].items()
    except FileNotFoundError:
        logger.error("File or directory '%s' not found.", filename)
    except json.JSONDecodeError:
        logger.error("Error decoding JSON in file '%s'.", filename)
    except Exception as e:  # noqa: BLE001  the reference swallows every validation error the same way
        logger.error("An error occurred while reading '%s': %s", filename, e)
    logger.warning("Defaulting to KV cache scaling factors = 1.0 for all layers in TP rank %d as an error occurred "
                   "during loading.", tp_rank)
    return []


def load_kv_cache_scales(attn_layers: List[torch.nn.Module], quantization_param_path: str, tp_rank: int = 0,
                         tp_size: int = 1, model_type: Optional[str] = None) -> int:
    """`LlamaModel.load_kv_cache_scales` (llama.py:359-378) over a list of attention layers (index = layer id): the
    per-layer factor becomes both k_scale and v_scale.  Returns how many layers were set."""
    n = 0
    for layer_idx, scaling_factor in kv_cache_scales_loader(quantization_param_path, tp_rank, tp_size, len(attn_layers),
                                                            model_type):
        attn = attn_layers[layer_idx]
        if not hasattr(attn, "k_scale"):
            raise RuntimeError("Self attention has no KV cache scaling factor attribute!")
        # llama.py:371-376 assigns the python float over the attribute; on a layer that went through
        # Fp8KVCacheMethod.create_weights the attribute is a registered nn.Parameter (a plain assignment raises), and
        # process_weights_after_loading has already frozen k_scale_float / v_scale_float (1.0 for a checkpoint without
        # kv scales) -- the file's factor has to replace those too, or the backend would keep reading 1.0
        for name in ("k_scale", "v_scale"):
            cur = getattr(attn, name)
            if isinstance(cur, torch.nn.Parameter) or isinstance(cur, torch.Tensor):
                cur.data.fill_(scaling_factor)
            else:
                setattr(attn, name, scaling_factor)
            setattr(attn, name + "_float", float(scaling_factor))
        n += 1
    return n
