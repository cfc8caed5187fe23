"""CPU oracle: FP8 / AWQ / GPTQ quantized-linear arithmetic of the reference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  File:line citations are
relative to /root/reference.  gfx950 FP8 is OCP e4m3fn (max 448); the
reference's fnuz handling is keyed on gfx94 only
(python/sglang/srt/layers/quantization/fp8_kernel.py:51-63).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0

AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]  # nibble j of an int32 -> logical col 8c+AWQ_ORDER... see below


# ---------------------------------------------------------------------------
# FP8 activation / weight quantisation
# ---------------------------------------------------------------------------
def input_to_float8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-tensor dynamic quant used for bf16 checkpoints' weights.
    Restates quantization/fp8_utils.py:310-326: scale = fp_max/amax (amax
    clamped at 1e-12), saturate, cast; returns (q, 1/scale)."""
    mn, mx = x.aminmax()
    amax = torch.maximum(mn.abs(), mx.abs()).float().clamp(min=1e-12)
    scale = FP8_MAX / amax
    q = (x.float() * scale).clamp(min=-FP8_MAX, max=FP8_MAX).to(FP8)
    return q.contiguous(), scale.float().reciprocal()


def per_tensor_quant_fp8(x: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """Per-tensor activation quant, dynamic (scale None) or static.
    Restates sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-97: dynamic scale =
    absmax/448 (:42); q = sat(x * (1/scale)) (:54,:71); and the torch model the
    reference tests it with (sgl-kernel/tests/test_per_tensor_quant_fp8.py:29-36).
    Returns (q [M,K] fp8, scale fp32 [1])."""
    if scale is None:
        scale = (x.float().abs().max() / FP8_MAX).reshape(1)
    scale = scale.float().reshape(1)
    inv = 1.0 / scale
    q = (x.float() * inv).clamp(min=-FP8_MAX, max=FP8_MAX).to(FP8)
    return q, scale


def per_token_quant_fp8(x: torch.Tensor):
    """Per-row dynamic quant.  Restates per_token_quant_fp8.cu:9-75 and
    sgl-kernel/tests/test_per_token_quant_fp8.py:14-22.  Returns (q, scale [M,1])."""
    scale = (x.float().abs().amax(dim=-1, keepdim=True) / FP8_MAX)
    inv = 1.0 / scale
    q = (x.float() * inv).clamp(min=-FP8_MAX, max=FP8_MAX).to(FP8)
    return q, scale


def scaled_mm(a_q, b_q, scale_a, scale_b, out_dtype, bias=None):
    """out = ((A @ B) * sa[:,None]) * sb[None,:] -> out_dtype (+ bias).
    Restates sgl-kernel/tests/test_fp8_gemm.py:6-14, the truth for
    fp8_scaled_mm (csrc/gemm/fp8_gemm_kernel.cu:1071-1146).  a_q [M,K] fp8,
    b_q [K,N] fp8 (column-major view), scale_a [M] or [1], scale_b [N] or [1]."""
    o = torch.matmul(a_q.to(torch.float32), b_q.to(torch.float32))
    o = o * scale_a.reshape(-1, 1).float()
    o = o * scale_b.reshape(1, -1).float()
    o = o.to(out_dtype)
    if bias is not None:
        o = o + bias.view(1, -1)
    return o


def per_tensor_dequantize(tensor, inv_scale):
    """quantization/utils.py:58-63: through fp16, whatever the model dtype."""
    return tensor.to(torch.float16) * inv_scale


def convert_to_channelwise(weight_scale, logical_widths: List[int]):
    """quantization/utils.py:71-91: one fp32 scale per output channel, [sum(widths), 1]."""
    out = torch.empty((sum(logical_widths), 1), dtype=torch.float32)
    if weight_scale.dim() == 0:
        out.fill_(weight_scale.item())
        return out
    start = 0
    for idx, w in enumerate(logical_widths):
        out[start:start + w, :] = weight_scale[idx]
        start += w
    return out


def requantize_with_max_scale(weight, weight_scale, logical_widths: List[int]):
    """Fuse N per-shard scales into one.  Restates quantization/utils.py:94-119
    (+ per_tensor_dequantize :58-63 which goes through fp16).  Pinned by running the reference function
    (tests/golden/make_golden_elementwise.py -> elementwise.pt: unfused and fused-checkpoint cases)."""
    max_w = weight_scale.max()
    unfused = bool(weight_scale[-1] > torch.finfo(FP8).min)
    weight = weight.clone()
    if unfused:
        start = 0
        for idx, w in enumerate(logical_widths):
            end = start + w
            dq = weight[start:end, :].to(torch.float16) * weight_scale[idx]
            weight[start:end, :], _ = per_tensor_quant_fp8(dq, max_w.reshape(1))
            start = end
    return max_w, weight


def fp8_linear(x, weight_kn, weight_scale, input_scale=None, bias=None,
               per_token: bool = False):
    """y = fp8(x) @ W_fp8 with scale epilogue, in x.dtype.

    Restates apply_fp8_linear's non-CUTLASS HIP branch, quantization/
    fp8_utils.py:654-749: static -> static quant (:658); dynamic + per-tensor
    weight -> per-tensor dynamic quant (:669-674); per-tensor both ->
    torch._scaled_mm semantics (:715-723), otherwise the unfused fallback
    `(A@B in fp32) * x_scale * w_scale.t() + bias` (:479-507).
    weight_kn is the [K,N] view stored by Fp8LinearMethod (fp8.py:364,406).
    Arithmetic is spelled in fp32 (`scaled_mm` above) rather than calling
    torch._scaled_mm so it runs on any host."""
    x2 = x.reshape(-1, x.shape[-1])
    if input_scale is not None:
        qx, xs = per_tensor_quant_fp8(x2, input_scale)
    elif per_token:
        qx, xs = per_token_quant_fp8(x2)
    else:
        qx, xs = per_tensor_quant_fp8(x2)
    o = torch.matmul(qx.float(), weight_kn.float())
    o = o * xs.reshape(-1, 1) * weight_scale.reshape(1, -1).float()
    if bias is not None:
        o = o + bias.float()
    return o.to(x.dtype).reshape(*x.shape[:-1], weight_kn.shape[1])


# ---------------------------------------------------------------------------
# AWQ int4
# ---------------------------------------------------------------------------
def _awq_unpack(packed: torch.Tensor) -> torch.Tensor:
    """[R, C/8] int32 -> [R, C] int 0..15 in logical column order.
    Restates sgl-kernel/tests/test_awq_dequant.py:9-22,36-50: nibble j (bits
    4j..4j+3) is extracted in order, then columns are permuted by
    AWQ_REVERSE_ORDER [0,4,1,5,2,6,3,7] within every group of 8."""
    shifts = torch.arange(0, 32, 4)
    nib = (packed[:, :, None] >> shifts[None, None, :]) & 0xF       # [R, C/8, 8]
    nib = nib[:, :, AWQ_ORDER]
    return nib.reshape(packed.shape[0], -1).to(torch.int32)


def awq_dequantize(qweight, scales, qzeros, group_size: Optional[int] = None):
    """W[K,N] = (w - z) * s.  Restates awq_dequantize_torch,
    sgl-kernel/tests/test_awq_dequant.py:28-57 (the reference test's truth for
    csrc/gemm/awq_kernel.cu:126-221).  qweight [K,N/8] i32, qzeros [K/g,N/8]
    i32, scales [K/g,N] fp16/bf16."""
    K = qweight.shape[0]
    g = K // scales.shape[0] if group_size is None or group_size == -1 else group_size
    w = _awq_unpack(qweight)
    z = _awq_unpack(qzeros).repeat_interleave(g, dim=0)
    s = scales.repeat_interleave(g, dim=0)
    return (w - z) * s


def awq_linear(x, qweight, scales, qzeros, bias=None):
    """Restates AWQLinearMethod.apply, quantization/awq.py:188-204:
    out = x @ awq_dequantize(...) (+ bias), in x.dtype."""
    W = awq_dequantize(qweight, scales, qzeros).to(x.dtype)
    out = torch.matmul(x.reshape(-1, x.shape[-1]), W)
    if bias is not None:
        out = out + bias
    return out.reshape(*x.shape[:-1], W.shape[1])


# ---------------------------------------------------------------------------
# GPTQ int4  -- PARITY UNPINNED: the arithmetic lives in vllm
# (vllm.model_executor.layers.quantization.gptq.GPTQLinearMethod +
# _custom_ops.gptq_gemm; pins vllm==0.9.0.1 / 0.6.7.dev2, quantization/
# __init__.py:117, python/pyproject.toml:80), which is not under
# /root/reference.  The reference's own tests at this boundary check only the
# quant_method class (test/srt/test_gptqmodel_dynamic.py:130-145).  This is
# the published AutoGPTQ v1 convention: qweight packed along K, qzeros packed
# along N sequentially and stored minus one, optional act-order g_idx.
# ---------------------------------------------------------------------------
def gptq_dequantize(qweight, scales, qzeros, g_idx=None, group_size: int = 128):
    """qweight [K/8,N] i32, qzeros [K/g,N/8] i32, scales [K/g,N] -> W [K,N]."""
    shifts = torch.arange(0, 32, 4)
    Kp, N = qweight.shape
    w = ((qweight[:, None, :] >> shifts[None, :, None]) & 0xF).reshape(Kp * 8, N).to(torch.int32)
    z = ((qzeros[:, :, None] >> shifts[None, None, :]) & 0xF).reshape(qzeros.shape[0], -1).to(torch.int32) + 1
    K = Kp * 8
    if g_idx is None:
        g_idx = torch.arange(K) // group_size
    g_idx = g_idx.to(torch.int64)
    return (w - z[g_idx]) * scales[g_idx]


def gptq_linear(x, qweight, scales, qzeros, g_idx=None, group_size=128, bias=None):
    W = gptq_dequantize(qweight, scales, qzeros, g_idx, group_size).to(x.dtype)
    out = torch.matmul(x.reshape(-1, x.shape[-1]), W)
    if bias is not None:
        out = out + bias
    return out.reshape(*x.shape[:-1], W.shape[1])
