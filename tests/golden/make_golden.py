#!/usr/bin/env python3
"""Generate tests/golden/*.pt by RUNNING THE REFERENCE'S OWN FILES on CPU.

Runs only in the build container (needs /root/reference); the resulting .pt
fixtures (inputs + expected outputs, plain tensors) are committed and travel to
the GPU box, the reference does not.  Recipe: SURVEY.md section 8c --
pre-seed `sys.modules` with empty namespace packages so no reference
`__init__.py` executes, then import single files:

  * python/sglang/srt/layers/attention/torch_native_backend.py
      -> TorchNativeAttnBackend._run_sdpa_forward_decode / _extend
  * sgl-kernel/tests/test_awq_dequant.py       -> awq_dequantize_torch
  * sgl-kernel/tests/test_fp8_gemm.py          -> torch_scaled_mm
  * sgl-kernel/tests/test_per_tensor_quant_fp8.py -> torch_scaled_fp8_quant
  * sgl-kernel/tests/test_per_token_quant_fp8.py  -> torch_per_token_quant_fp8
  * python/sglang/srt/layers/quantization/fp8_utils.py
      -> input_to_float8, _apply_fallback_scaled_mm  (+ torch._scaled_mm itself,
         the op the reference calls at fp8_utils.py:715)

Usage:  python tests/golden/make_golden.py   (writes next to this file)
"""
import importlib
import importlib.util
import os
import sys
import types

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
ROOT = f"{REF}/python"


def _ns(pkg):
    m = types.ModuleType(pkg)
    m.__path__ = [f"{ROOT}/{pkg.replace('.', '/')}"]
    sys.modules[pkg] = m
    return m


def load_reference():
    for pkg in ["sglang", "sglang.srt", "sglang.srt.layers", "sglang.srt.layers.attention",
                "sglang.srt.layers.quantization", "sglang.srt.model_executor", "sglang.srt.mem_cache"]:
        _ns(pkg)
    fbi = types.ModuleType("sglang.srt.model_executor.forward_batch_info")
    fbi.ForwardBatch = type("ForwardBatch", (), {})
    sys.modules[fbi.__name__] = fbi
    tnb = importlib.import_module("sglang.srt.layers.attention.torch_native_backend")

    # names the sgl-kernel test files import at module level (never called here)
    sk = types.ModuleType("sgl_kernel")
    for n in ["awq_dequantize", "fp8_scaled_mm", "sgl_per_tensor_quant_fp8", "sgl_per_token_quant_fp8"]:
        setattr(sk, n, None)
    sys.modules["sgl_kernel"] = sk
    su = types.ModuleType("sglang.srt.utils")
    su.is_hip = lambda: True
    su.is_cuda = lambda: False
    su.align = lambda x, a: (x + a - 1) // a * a
    su.get_bool_env_var = lambda name, default="false": os.getenv(name, default).lower() in ("1", "true")
    su.get_cuda_version = lambda: (0, 0)
    su.get_device_capability = lambda *a, **k: (9, 5)
    su.is_flashinfer_available = lambda: False
    sys.modules["sglang.srt.utils"] = su
    lu = types.ModuleType("sglang.srt.layers.utils")
    lu.is_sm100_supported = lambda *a, **k: False
    sys.modules["sglang.srt.layers.utils"] = lu
    dg = types.ModuleType("sglang.srt.layers.quantization.deep_gemm_wrapper")
    sys.modules[dg.__name__] = dg
    sys.modules["sglang.srt.layers.quantization"].deep_gemm_wrapper = dg
    fk = types.ModuleType("sglang.srt.layers.quantization.fp8_kernel")
    fk.fp8_dtype = torch.float8_e4m3fn       # gfx950 is OCP e4m3fn (fp8_kernel.py:51-63)
    fk.fp8_max = 448.0
    fk.is_fp8_fnuz = lambda: False
    for n in ["sglang_per_token_group_quant_fp8", "per_token_group_quant_fp8", "scaled_fp8_quant",
              "sglang_per_token_quant_fp8", "static_quant_fp8", "w8a8_block_fp8_matmul_deepgemm",
              "w8a8_block_fp8_matmul_triton"]:
        setattr(fk, n, None)
    sys.modules[fk.__name__] = fk

    def load_file(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    mods = {
        "tnb": tnb,
        "awq": load_file("_ref_test_awq", f"{REF}/sgl-kernel/tests/test_awq_dequant.py"),
        "gemm": load_file("_ref_test_fp8_gemm", f"{REF}/sgl-kernel/tests/test_fp8_gemm.py"),
        "ptq": load_file("_ref_test_ptq", f"{REF}/sgl-kernel/tests/test_per_tensor_quant_fp8.py"),
        "tokq": load_file("_ref_test_tokq", f"{REF}/sgl-kernel/tests/test_per_token_quant_fp8.py"),
        "fp8u": importlib.import_module("sglang.srt.layers.quantization.fp8_utils"),
    }
    return mods


# ---------------------------------------------------------------------------
def make_pool(gen, slots, Hkv, D, dtype, mean=0.0, std=1.0):
    k = (torch.randn(slots, Hkv, D, generator=gen) * std + mean).to(dtype)
    v = (torch.randn(slots, Hkv, D, generator=gen) * std + mean).to(dtype)
    return k, v


def scattered_table(gen, lens, slots, max_reqs, ctx, shared_prefix=0):
    """req_to_token rows with scattered, non-contiguous slot ids (slot 0 unused).
    shared_prefix>0: rows 0 and 1 share their first `shared_prefix` slots
    (radix-cache prefix sharing)."""
    perm = torch.randperm(slots - 1, generator=gen) + 1
    req_to_token = torch.zeros(max_reqs, ctx, dtype=torch.int32)
    rows = torch.randperm(max_reqs, generator=gen)[: len(lens)]
    off = 0
    for i, L in enumerate(lens):
        req_to_token[rows[i], :L] = perm[off : off + L].to(torch.int32)
        off += L
    if shared_prefix and len(lens) > 1:
        req_to_token[rows[1], :shared_prefix] = req_to_token[rows[0], :shared_prefix]
    return req_to_token, rows.to(torch.int64)


def attention_cases(tnb):
    be = tnb.TorchNativeAttnBackend.__new__(tnb.TorchNativeAttnBackend)
    out = {}
    gen = torch.Generator().manual_seed(0)

    def decode_case(name, Hq, Hkv, D, lens, dtype, mean=0.0, std=1.0):
        B = len(lens)
        slots = sum(lens) + 8
        k_cache, v_cache = make_pool(gen, slots, Hkv, D, dtype, mean, std)
        req_to_token, rpi = scattered_table(gen, lens, slots, max_reqs=B + 3, ctx=max(lens) + 4)
        seq_lens = torch.tensor(lens, dtype=torch.int64)
        q = (torch.randn(B, Hq, D, generator=gen) * std + mean).to(dtype)
        k_new = torch.randn(B, Hkv, D, generator=gen).to(dtype)
        v_new = torch.randn(B, Hkv, D, generator=gen).to(dtype)
        # set_kv_buffer (memory_pool.py:454-455): slot of the LAST token of each request
        loc = torch.stack([req_to_token[rpi[i], lens[i] - 1] for i in range(B)]).to(torch.int64)
        k_after, v_after = k_cache.clone(), v_cache.clone()
        k_after[loc] = k_new
        v_after[loc] = v_new
        scaling = 1.0 / D ** 0.5
        o = torch.empty_like(q)
        be._run_sdpa_forward_decode(q, o, k_after, v_after, req_to_token, rpi, seq_lens,
                                    scaling=scaling, enable_gqa=(Hq != Hkv), causal=False)
        out[name] = dict(q=q, k_new=k_new, v_new=v_new, k_cache=k_cache, v_cache=v_cache, req_to_token=req_to_token,
                         req_pool_indices=rpi, seq_lens=seq_lens, out_cache_loc=loc,
                         scaling=torch.tensor(scaling), o=o)

    def extend_case(name, Hq, Hkv, D, pre, ext, dtype, causal=True, shared_prefix=0):
        B = len(pre)
        lens = [p + e for p, e in zip(pre, ext)]
        slots = sum(lens) + 8
        k_cache, v_cache = make_pool(gen, slots, Hkv, D, dtype)
        req_to_token, rpi = scattered_table(gen, lens, slots, max_reqs=B + 2, ctx=max(lens) + 4,
                                            shared_prefix=shared_prefix)
        E = sum(ext)
        q = torch.randn(E, Hq, D, generator=gen).to(dtype)
        k_new = torch.randn(E, Hkv, D, generator=gen).to(dtype)
        v_new = torch.randn(E, Hkv, D, generator=gen).to(dtype)
        loc = torch.cat([req_to_token[rpi[i], pre[i] : lens[i]] for i in range(B)]).to(torch.int64)
        k_after, v_after = k_cache.clone(), v_cache.clone()
        k_after[loc] = k_new
        v_after[loc] = v_new
        scaling = 1.0 / D ** 0.5
        o = torch.empty_like(q)
        be._run_sdpa_forward_extend(q, o, k_after, v_after, req_to_token, rpi,
                                    torch.tensor(lens, dtype=torch.int64),
                                    torch.tensor(pre, dtype=torch.int32),
                                    torch.tensor(ext, dtype=torch.int32),
                                    scaling=scaling, enable_gqa=(Hq != Hkv), causal=causal)
        out[name] = dict(q=q, k_new=k_new, v_new=v_new, k_cache=k_cache, v_cache=v_cache, req_to_token=req_to_token,
                         req_pool_indices=rpi, seq_lens=torch.tensor(lens, dtype=torch.int64),
                         extend_prefix_lens=torch.tensor(pre, dtype=torch.int32),
                         extend_seq_lens=torch.tensor(ext, dtype=torch.int32),
                         out_cache_loc=loc, scaling=torch.tensor(scaling),
                         causal=torch.tensor(causal), o=o)

    # GQA group 4 (Llama-3 shape ratio), ragged incl. S=1 and a non-multiple-of-tile length
    decode_case("decode_gqa4_d128_bf16", 8, 2, 128, [1, 7, 64, 129, 200, 33], torch.bfloat16)
    decode_case("decode_gqa8_d128_bf16", 8, 1, 128, [5, 257, 31], torch.bfloat16)         # 70B TP8 rank shape
    decode_case("decode_mha_d128_fp16", 4, 4, 128, [3, 90, 150], torch.float16)           # Llama-2 MHA
    decode_case("decode_mha_d64_fp16", 4, 4, 64, [2, 77, 128, 19], torch.float16)         # OPT-125m head_dim
    decode_case("decode_gqa4_d128_bf16_shifted", 8, 2, 128, [40, 100], torch.bfloat16,
                mean=0.1, std=0.2)  # test_triton_attention_kernels.py:67-72 input distribution
    extend_case("extend_gqa4_d128_bf16_noprefix", 8, 2, 128, [0, 0, 0], [1, 17, 70], torch.bfloat16)
    extend_case("extend_gqa4_d128_bf16_prefix", 8, 2, 128, [64, 0, 33, 100], [17, 5, 64, 1], torch.bfloat16,
                shared_prefix=32)
    extend_case("extend_mha_d64_fp16_prefix", 4, 4, 64, [10, 0], [30, 45], torch.float16)
    extend_case("extend_gqa4_d128_bf16_noncausal", 8, 2, 128, [0, 9], [20, 11], torch.bfloat16, causal=False)
    return out


def quant_cases(m):
    out = {}
    gen = torch.Generator().manual_seed(1)
    # --- AWQ dequant (test_awq_dequant.py:28-57), group 128 and group == K
    for name, K, N8, g, sdt in [("awq_g128_fp16", 256, 16, 128, torch.float16),
                                ("awq_gK_bf16", 128, 9, 128, torch.bfloat16)]:
        qweight = torch.randint(0, torch.iinfo(torch.int32).max, (K, N8), dtype=torch.int32, generator=gen)
        qweight[::3] *= -1  # exercise the sign bit / top nibble
        qzeros = torch.randint(0, torch.iinfo(torch.int32).max, (K // g, N8), dtype=torch.int32, generator=gen)
        scales = torch.rand(K // g, N8 * 8, generator=gen).to(sdt)
        W = m["awq"].awq_dequantize_torch(qweight, scales, qzeros, g)
        out[name] = dict(qweight=qweight, qzeros=qzeros, scales=scales, group=torch.tensor(g), W=W)
    # --- fp8 scaled mm truth (test_fp8_gemm.py:6-35)
    for name, M, N, K, with_bias, odt in [("scaled_mm_bf16_bias", 5, 48, 256, True, torch.bfloat16),
                                          ("scaled_mm_fp16", 33, 16, 512, False, torch.float16)]:
        a = ((torch.rand(M, K, generator=gen) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)
        b = ((torch.rand(N, K, generator=gen) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.randn(M, generator=gen) * 0.001
        sb = torch.randn(N, generator=gen) * 0.001
        bias = torch.randn(N, generator=gen).to(odt) if with_bias else None
        o = m["gemm"].torch_scaled_mm(a, b.t(), sa, sb, odt, bias)
        d = dict(a=a.view(torch.uint8), b_nk=b.view(torch.uint8), scale_a=sa, scale_b=sb, o=o)
        if bias is not None:
            d["bias"] = bias
        out[name] = d
    # --- per-tensor / per-token quant truth (test_per_tensor_quant_fp8.py:29-36, test_per_token...:14-22)
    x = torch.rand(7, 512, generator=gen).to(torch.float16)
    x[2] *= 30
    x[4] = -x[4]
    dyn_scale = (x.float().abs().max() / 448.0).reshape(1)        # per_tensor_quant_fp8.cu:42
    out["per_tensor_dynamic"] = dict(x=x, scale=dyn_scale,
                                     q=m["ptq"].torch_scaled_fp8_quant(x, dyn_scale).view(torch.uint8))
    st = torch.tensor([0.0123])
    out["per_tensor_static"] = dict(x=x, scale=st, q=m["ptq"].torch_scaled_fp8_quant(x, st).view(torch.uint8))
    tok_scale = x.float().abs().amax(dim=-1) / 448.0               # per_token_quant_fp8.cu:49
    out["per_token"] = dict(x=x, scale=tok_scale,
                            q=m["tokq"].torch_per_token_quant_fp8(x, tok_scale).view(torch.uint8))
    # --- input_to_float8 (fp8_utils.py:310-326)
    w = (torch.rand(24, 64, generator=gen) * 2e-3 - 1e-3).to(torch.bfloat16)  # dummy-weight range
    qw, inv = m["fp8u"].input_to_float8(w, torch.float8_e4m3fn)
    out["input_to_float8"] = dict(x=w, q=qw.view(torch.uint8), inv_scale=inv)
    # --- the per-tensor linear the reference runs on HIP: torch._scaled_mm (fp8_utils.py:715-723)
    xa = torch.randn(19, 64, generator=gen).to(torch.bfloat16)
    xs = (xa.float().abs().max() / 448.0).reshape(1)
    qx = m["ptq"].torch_scaled_fp8_quant(xa, xs)
    bias = torch.randn(24, generator=gen).to(torch.bfloat16)
    y = torch._scaled_mm(qx, qw.t(), out_dtype=torch.bfloat16, scale_a=xs, scale_b=inv.reshape(1), bias=bias)
    out["fp8_linear_per_tensor"] = dict(x=xa, w_nk=qw.view(torch.uint8), w_scale=inv.reshape(1),
                                        bias=bias, x_scale=xs, y=y)
    # --- per-token activations x per-tensor weight: unfused fallback (fp8_utils.py:479-507)
    ts = (xa.float().abs().amax(dim=-1, keepdim=True) / 448.0)
    qxt = m["tokq"].torch_per_token_quant_fp8(xa, ts)
    y2 = m["fp8u"]._apply_fallback_scaled_mm(qxt, qw.t(), ts, inv.reshape(1, 1), xa.shape, [19, 24], bias,
                                             torch.bfloat16)
    out["fp8_linear_per_token_fallback"] = dict(x=xa, w_nk=qw.view(torch.uint8), w_scale=inv.reshape(1),
                                                bias=bias, y=y2)
    return out


def main():
    torch.manual_seed(0)
    m = load_reference()
    att = attention_cases(m["tnb"])
    torch.save(att, os.path.join(HERE, "attention.pt"))
    q = quant_cases(m)
    torch.save(q, os.path.join(HERE, "quant.pt"))
    for name, d in {**att, **q}.items():
        print(name, {k: tuple(v.shape) for k, v in d.items()})
    for f in ["attention.pt", "quant.pt"]:
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
