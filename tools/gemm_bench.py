#!/usr/bin/env python3
"""Micro-benchmark of mi_fp8_gemm at the Llama-3-8B decode shapes (M=128).  Not a test."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402

FP8 = torch.float8_e4m3fn
dev = "cuda"
M = int(os.environ.get("M", "128"))
pad = int(os.environ.get("XPAD", "0"))
shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
if os.environ.get("SHAPES"):   # e.g. SHAPES="768x4096,4096x512,3584x4096,4096x1792" (N x K; TP=8 per-rank shapes)
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
nbuf = 6
for N, K in shapes:
    ws = [torch.randn(N, K, device=dev).to(FP8) for _ in range(nbuf)]      # rotate buffers: defeat the L3
    xfull = torch.randn(M, K + pad, device=dev).to(FP8)
    x = xfull[:, :K]
    sa = torch.ones(1, device=dev); sb = torch.ones(1, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for w in ws:
        ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        for w in ws:
            ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
    print(f"M={M} N={N} K={K} xpad={pad} rot={os.environ.get('MI_GEMM_ROTATE','1')}: {us:8.1f} us  {N*K/us/1e3:7.1f} GB/s weights", flush=True)
