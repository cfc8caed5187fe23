#!/usr/bin/env python3
"""Micro-benchmark of the prefill-side kernels at Llama-3-8B shapes: FP8 GEMM at M = tokens and
ragged extend attention (no prefix / with prefix).  Not a test."""
import math
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402

FP8 = torch.float8_e4m3fn
dev = "cuda"
M = int(os.environ.get("M", "16384"))
if os.environ.get("GEMM", "1") == "1":
    for N, K in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
        w = torch.randn(N, K, device=dev).to(FP8)
        x = torch.randn(M, K, device=dev).to(FP8)
        sa = torch.ones(1, device=dev); sb = torch.ones(1, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"fp8_gemm M={M} N={N} K={K}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
if os.environ.get("ATTN", "1") == "1":
    Hq, Hkv, D = 32, 8, 128
    cases = [(8, 2048, 0), (2, 8192, 0), (8, 512, 1536), (64, 1, 2047)]
    if os.environ.get("CASES"):      # e.g. CASES="1x512x32000,4x8x8192" (requests x new tokens x cached prefix)
        cases = [tuple(int(v) for v in c.split("x")) for c in os.environ["CASES"].split(",")]
    for B, S, P in cases:
        E = B * S
        q = torch.randn(E, Hq, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
        k = torch.randn(E, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
        v = torch.randn(E, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
        slots = B * P + 1
        kb = torch.randn(slots, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
        vb = torch.randn(slots, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
        o = torch.empty_like(q)
        ext = torch.full((B,), S, dtype=torch.int32, device=dev)
        pre = torch.full((B,), P, dtype=torch.int32, device=dev)
        qo = ops.kv_indptr(ext); kvp = ops.kv_indptr(pre).clone()
        idx = (torch.randperm(max(B * P, 1), device=dev).to(torch.int32) + 1)
        xs = int(os.environ.get("XSPLITS", "1"))
        wsx = torch.empty(max(1, ops.decode_workspace_numel(E, Hq, D, xs)), dtype=torch.float32, device=dev)
        def run():
            if xs > 1:
                ops.extend_attention_splitkv(q, k, v, o, kb, vb, qo, kvp, idx, S, 1 / math.sqrt(D), xs, wsx)
            else:
                ops.extend_attention(q, k, v, o, kb, vb, qo, kvp, idx, S, 1 / math.sqrt(D), 0.0, True, -1)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flops = 4 * Hq * D * B * (S * P + S * (S + 1) / 2)
        print(f"extend_attn B={B} ext={S} prefix={P}: {ms:8.3f} ms  {flops/ms/1e9:8.1f} TFLOP/s", flush=True)
