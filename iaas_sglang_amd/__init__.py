"""MI355X-native (gfx950) hot path for SGLang's batched prefill/decode:
paged-KV extend/decode attention behind ``AttentionBackend`` and FP8 / AWQ / GPTQ
quantized linears behind ``QuantizationConfig`` / ``LinearMethodBase``, as hand-written
HIP kernels reached through the C ABI in ``include/mi_hotpath.h``.

Importing this package loads ``libmi_hotpath.so``; it fails loudly if that is missing.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)

__all__ = ["_lib", "ops"]
__version__ = "0.1.0"
