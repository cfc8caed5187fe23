#!/usr/bin/env python3
"""Micro-benchmark of mi_w4a16_gemm at the Llama-2-7B AWQ decode shapes (config C4: M=64, g=128)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402
from iaas_sglang_amd._lib import MI_W4_AWQ  # noqa: E402

dev = "cuda"
M = int(os.environ.get("M", "64"))
g = 128
for N, K in [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]:
    gen = torch.Generator(device=dev).manual_seed(0)
    sets = []
    for _ in range(4):
        qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, device=dev, generator=gen)
        qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // g, N // 8), dtype=torch.int32, device=dev, generator=gen)
        scales = (torch.rand(K // g, N, device=dev, generator=gen) * 1e-2).to(torch.float16)
        sets.append(ops.w4_repack(qweight, qzeros, scales, g, MI_W4_AWQ)[:2])
    x = torch.randn(M, K, device=dev).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    for qw, zs in sets:
        ops.w4a16_gemm(x, qw, zs, N, g, None, None, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        for qw, zs in sets:
            ops.w4a16_gemm(x, qw, zs, N, g, None, None, out)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(sets))
    print(f"w4a16 M={M} N={N} K={K}: {us:8.1f} us  {N*K/2/us/1e3:7.1f} GB/s int4 weights", flush=True)
