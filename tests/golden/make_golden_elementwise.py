#!/usr/bin/env python3
"""Generate tests/golden/elementwise.pt by RUNNING THE REFERENCE'S OWN torch forms on CPU.

Build container only (needs /root/reference); the fixture holds plain tensors (inputs + expected outputs).
Same import recipe as make_golden.py (namespace stubs, no reference `__init__.py` runs).
Reference code executed:

  * python/sglang/srt/layers/layernorm.py:128-146      RMSNorm.forward_native (with / without residual)
  * python/sglang/srt/layers/activation.py:56-58       SiluAndMul.forward_native
  * python/sglang/srt/layers/rotary_embedding.py:49-74,138-166  RotaryEmbedding.forward_native (neox; the cache is
        cast to the model dtype first, as the class does on non-CUDA devices, :104-105)
  * sgl-kernel/tests/test_norm.py:8-15,40-50           llama_rms_norm, fused_add_rms_norm
  * sgl-kernel/tests/test_rotary_embedding.py:9-120    RotaryEmbedding.forward_native (fp32 arithmetic, one rounding)
  * sgl-kernel/tests/test_merge_state_v2.py:101-135    merge_state_torch
  * python/sglang/srt/layers/quantization/utils.py:58-119  per_tensor_dequantize, convert_to_channelwise,
        requantize_with_max_scale.  On ROCm that module binds `scaled_fp8_quant` from vllm._custom_ops (absent here);
        the name is bound to the reference's own torch form of that op, `torch_scaled_fp8_quant`
        (sgl-kernel/tests/test_per_tensor_quant_fp8.py:29-36), returning (q, scale) as the op does.

Usage:  python tests/golden/make_golden_elementwise.py
"""
import importlib
import os
import sys
import types

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF


def load_reference():
    m = mg.load_reference()
    su = sys.modules["sglang.srt.utils"]
    su.is_npu = lambda: False
    su.is_cpu = lambda: False
    su.cpu_has_amx_support = lambda: False
    su.set_weight_attrs = lambda *a, **k: None
    sk = sys.modules["sgl_kernel"]
    for n in ["apply_rope_with_cos_sin_cache_inplace", "merge_state", "merge_state_v2"]:
        setattr(sk, n, None)
    vllm = types.ModuleType("vllm")
    vllm.__path__ = []
    vco = types.ModuleType("vllm._custom_ops")
    for n in ["fused_add_rms_norm", "rms_norm", "rotary_embedding"]:   # only forward_native is called here
        setattr(vco, n, None)
    ptq = m["ptq"]
    vco.scaled_fp8_quant = lambda x, scale: (ptq.torch_scaled_fp8_quant(x, scale), scale)
    sys.modules["vllm"], sys.modules["vllm._custom_ops"] = vllm, vco
    vllm._custom_ops = vco
    sys.modules["sglang.srt.layers.quantization.fp8_kernel"].scaled_fp8_quant = vco.scaled_fp8_quant
    dist = types.ModuleType("sglang.srt.distributed")
    dist.divide = lambda a, b: a // b
    dist.get_tensor_model_parallel_rank = lambda: 0
    dist.get_tensor_model_parallel_world_size = lambda: 1
    sys.modules[dist.__name__] = dist
    sgu = types.ModuleType("sglang.utils")
    sgu.resolve_obj_by_qualname = lambda name: None
    sys.modules[sgu.__name__] = sgu

    def load_file(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    m["layernorm"] = importlib.import_module("sglang.srt.layers.layernorm")
    # activation.py:211-216 REPLACES its own SiluAndMul with vllm's class on non-NVIDIA platforms (vllm is absent
    # here).  The class this fixture pins is the one the file defines (:55-58), so the module body is run with the
    # platform probe answering "cuda" (its sgl_kernel imports resolve to the unused stub names); nothing else is
    # imported under that setting.
    for n in ["gelu_and_mul", "gelu_tanh_and_mul", "silu_and_mul"]:
        setattr(sk, n, None)
    su.is_cuda = lambda: True
    try:
        m["activation"] = importlib.import_module("sglang.srt.layers.activation")
    finally:
        su.is_cuda = lambda: False
    m["rotary"] = importlib.import_module("sglang.srt.layers.rotary_embedding")
    m["qutils"] = importlib.import_module("sglang.srt.layers.quantization.utils")
    m["t_norm"] = load_file("_ref_test_norm", f"{REF}/sgl-kernel/tests/test_norm.py")
    m["t_rope"] = load_file("_ref_test_rope", f"{REF}/sgl-kernel/tests/test_rotary_embedding.py")
    m["t_merge"] = load_file("_ref_test_merge", f"{REF}/sgl-kernel/tests/test_merge_state_v2.py")
    return m


def cases(m):
    out = {}
    gen = torch.Generator().manual_seed(11)
    # ---- RMSNorm: the srt module's native form and the sgl-kernel test's form on the same inputs
    for name, rows, H, dtype, eps in [("rmsnorm_bf16_4096", 19, 4096, torch.bfloat16, 1e-5),
                                      ("rmsnorm_fp16_1024", 7, 1024, torch.float16, 1e-6),
                                      ("rmsnorm_bf16_8192", 3, 8192, torch.bfloat16, 1e-5),
                                      ("rmsnorm_fp16_128", 33, 128, torch.float16, 1e-6)]:
        x = torch.randn(rows, H, generator=gen).to(dtype)
        res = torch.randn(rows, H, generator=gen).to(dtype)
        w = (torch.randn(H, generator=gen) * 0.5 + 1.0).to(dtype)
        norm = m["layernorm"].RMSNorm(H, eps=eps)
        norm.weight.data = w.clone()
        with torch.no_grad():
            y = norm.forward_native(x.clone())
            y_add, res_out = norm.forward_native(x.clone(), res.clone())
        y_k = m["t_norm"].llama_rms_norm(x.clone(), w, eps)
        y_add_k, res_out_k = m["t_norm"].fused_add_rms_norm(x.clone(), res.clone(), w, eps)
        out[name] = dict(x=x, residual=res, weight=w, eps=torch.tensor(eps), y=y, y_add=y_add, residual_out=res_out,
                         y_kernel_test=y_k, y_add_kernel_test=y_add_k, residual_out_kernel_test=res_out_k)
    # ---- SiLU-and-mul
    act = m["activation"].SiluAndMul()
    for name, rows, inter, dtype in [("silu_mul_bf16", 9, 1792, torch.bfloat16), ("silu_mul_fp16", 5, 688, torch.float16)]:
        x = (torch.randn(rows, 2 * inter, generator=gen) * 2).to(dtype)
        out[name] = dict(x=x, y=act.forward_native(x))
    # ---- NeoX RoPE
    for name, T, Hq, Hkv, D, dtype, base, maxpos in [("rope_bf16_d128", 13, 8, 2, 128, torch.bfloat16, 500000, 1024),
                                                     ("rope_fp16_d128", 6, 4, 4, 128, torch.float16, 10000, 512),
                                                     ("rope_fp16_d64", 10, 12, 12, 64, torch.float16, 10000, 256)]:
        rope = m["rotary"].RotaryEmbedding(D, D, maxpos, base, True, dtype)
        rope_k = m["t_rope"].RotaryEmbedding(D, D, maxpos, base, True, dtype)
        pos = torch.randint(0, maxpos, (T,), generator=gen)
        pos[0] = 0
        pos[-1] = maxpos - 1
        q = torch.randn(T, Hq * D, generator=gen).to(dtype)
        k = torch.randn(T, Hkv * D, generator=gen).to(dtype)
        qo, ko = rope.forward_native(pos, q.clone(), k.clone())
        qk, kk = rope_k.forward_native(pos, q.clone(), k.clone())
        out[name] = dict(positions=pos, q=q, k=k, head_dim=torch.tensor(D), base=torch.tensor(float(base)),
                         max_pos=torch.tensor(maxpos), cos_sin_cache_f32=rope_k.cos_sin_cache.clone(),
                         q_out=qo, k_out=ko, q_out_kernel_test=qk,
                         k_out_kernel_test=kk)
    # ---- merge_state
    for name, n, h, d, dtype in [("merge_bf16", 11, 8, 128, torch.bfloat16), ("merge_fp16", 5, 4, 64, torch.float16)]:
        oa = torch.randn(n, h, d, generator=gen).to(dtype)
        ob = torch.randn(n, h, d, generator=gen).to(dtype)
        la = torch.randn(n, h, generator=gen) * 3
        lb = torch.randn(n, h, generator=gen) * 3
        la[0, 0] = float("inf")            # the inf -> -inf guard
        lb[1, 1] = float("inf")
        o, lse = m["t_merge"].merge_state_torch(oa.clone(), la.clone(), ob.clone(), lb.clone())
        out[name] = dict(o_a=oa, lse_a=la, o_b=ob, lse_b=lb, o=o, lse=lse)
    # ---- weight-scale utilities of the FP8 loaders
    qu = m["qutils"]
    FP8 = torch.float8_e4m3fn
    widths = [64, 16, 16]
    w = ((torch.rand(sum(widths), 96, generator=gen) - 0.5) * 2 * 300).to(FP8)
    ws = torch.tensor([0.011, 0.027, 0.0041])
    max_s, w_re = qu.requantize_with_max_scale(w.clone(), ws.clone(), widths)
    out["requantize_unfused"] = dict(weight=w.view(torch.uint8), weight_scale=ws, widths=torch.tensor(widths),
                                     max_scale=max_s, weight_out=w_re.view(torch.uint8))
    ws_f = torch.tensor([0.02, torch.finfo(FP8).min, torch.finfo(FP8).min])     # fused checkpoint: one scale loaded
    max_f, w_f = qu.requantize_with_max_scale(w.clone(), ws_f.clone(), widths)
    out["requantize_fused"] = dict(weight=w.view(torch.uint8), weight_scale=ws_f, widths=torch.tensor(widths),
                                   max_scale=max_f, weight_out=w_f.view(torch.uint8))
    out["convert_to_channelwise"] = dict(weight_scale=ws, widths=torch.tensor(widths),
                                         out=qu.convert_to_channelwise(ws.clone(), widths))
    out["per_tensor_dequantize"] = dict(weight=w.view(torch.uint8), scale=torch.tensor(0.013),
                                        out=qu.per_tensor_dequantize(w, torch.tensor(0.013)))
    return out


def main():
    m = load_reference()
    c = cases(m)
    torch.save(c, os.path.join(HERE, "elementwise.pt"))
    for name, d in c.items():
        print(name, {k: tuple(v.shape) for k, v in d.items()})
    print("elementwise.pt", os.path.getsize(os.path.join(HERE, "elementwise.pt")), "bytes")


if __name__ == "__main__":
    main()
