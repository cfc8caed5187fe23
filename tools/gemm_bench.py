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
wpad = int(os.environ.get("WPAD", "0"))     # row pitch of the weights = K + WPAD bytes
shapes = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]
if os.environ.get("SHAPES"):   # e.g. SHAPES="768x4096,4096x512,3584x4096,4096x1792" (N x K; TP=8 per-rank shapes)
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
nbuf = int(os.environ.get("NBUF", "6"))
for N, K in shapes:
    const = os.environ.get("CONST", "0") == "1"     # constant operands: how much of the time is data-dependent (power)
    mk = (lambda *sh: torch.full(sh, 0.001953125, device=dev)) if const else (lambda *sh: torch.randn(*sh, device=dev))
    ws = [mk(N, K + wpad).to(FP8)[:, :K] for _ in range(nbuf)]      # rotate buffers: defeat the L3
    xfull = mk(M, K + pad).to(FP8)
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
    if os.environ.get("SILU", "0") == "1" and N % 256 == 0 and M > 512:     # the fused gate_up form (fp8 [M, N/2] out)
        qs = torch.full((1,), 0.05, device=dev)
        ops.fp8_gemm_silu_mul(x, ws[0].t(), sa, sb, qs, torch.bfloat16)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            for w in ws:
                ops.fp8_gemm_silu_mul(x, w.t(), sa, sb, qs, torch.bfloat16)
        e1.record(); e1.synchronize()
        us2 = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
        print(f"M={M} N={N} K={K} fused SiLU*up fp8 out: {us2:8.1f} us  {2*M*N*K/us2/1e6:7.1f} TFLOP/s", flush=True)
    print(f"M={M} N={N} K={K} {2*M*N*K/us/1e6:7.1f} TFLOP/s xpad={pad} wpad={wpad} rot={os.environ.get('MI_GEMM_ROTATE','1')}: {us:8.1f} us  {N*K/us/1e3:7.1f} GB/s weights", flush=True)
