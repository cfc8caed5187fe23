"""QuantizationConfig / LinearMethodBase plugins on gfx950 HIP kernels: per-tensor FP8,
AWQ int4, GPTQ int4.  Class names match the reference's (`Fp8LinearMethod`, `AWQLinearMethod`,
`GPTQLinearMethod`) so they are picked up by WEIGHT_LOADER_V2_SUPPORTED
(python/sglang/srt/layers/linear.py:42-60)."""
from .awq import AWQConfig, AWQLinearMethod
from .compressed_tensors import CompressedTensorsConfig, CompressedTensorsLinearMethod, CompressedTensorsW8A8Fp8
from .fp8 import Fp8Config, Fp8LinearMethod, apply_fp8_linear
from .gptq import GPTQConfig, GPTQLinearMethod
from .kv_cache import BaseKVCacheMethod, Fp8KVCacheMethod, kv_cache_scales_loader, load_kv_cache_scales

MI_QUANTIZATION_METHODS = {"fp8": Fp8Config, "awq": AWQConfig, "gptq": GPTQConfig,
                           "compressed-tensors": CompressedTensorsConfig}

__all__ = ["Fp8Config", "Fp8LinearMethod", "apply_fp8_linear", "AWQConfig", "AWQLinearMethod",
           "GPTQConfig", "GPTQLinearMethod", "CompressedTensorsConfig", "CompressedTensorsLinearMethod",
           "CompressedTensorsW8A8Fp8", "BaseKVCacheMethod", "Fp8KVCacheMethod", "kv_cache_scales_loader",
           "load_kv_cache_scales", "MI_QUANTIZATION_METHODS"]
