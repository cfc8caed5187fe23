#!/usr/bin/env python3
"""Diagnostic (library built with `make TUNING=1`): k-loop / epilogue / end-barrier ticks per tile and wave of
fp8_gemm_tile_kernel at prefill shapes.  usage: M=16384 python tools/tile_stamps.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops, _lib  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (512 * 12 * 8))()
FP8 = torch.float8_e4m3fn
M = int(os.environ.get("M", "16384"))
for N, K in [(6144, 4096), (4096, 4096), (4096, 14336)]:
    w = torch.randn(N, K, device="cuda").to(FP8)
    x = torch.randn(M, K, device="cuda").to(FP8)
    sa = torch.ones(1, device="cuda"); sb = torch.ones(1, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    torch.cuda.synchronize()
    assert lib.mi_debug_xd_stamps(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    e1.record(); e1.synchronize()
    assert lib.mi_debug_xd_stamps(buf, 0) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(512, 12, 8)
    tiles = t[:, :8, 3].sum()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={M} N={N} K={K}: {ms:.3f} ms ({2 * M * N * K / ms / 1e9:.0f} TFLOP/s); per tile and wave: k-loop {float(t[:, :8, 0].sum() / tiles):.0f} "
          f"({float(t[:, :8, 0].sum() / tiles) / (K // 128):.0f} per k-step), epilogue {float(t[:, :8, 1].sum() / tiles):.0f}, "
          f"end barrier {float(t[:, :8, 2].sum() / tiles):.0f} ticks; by wave epilogue: "
          + " ".join(f"{float(t[:, wv, 1].sum() / t[:, wv, 3].sum().clamp(min=1)):.0f}" for wv in range(8)), flush=True)
    ks = t[:, :8, 7].sum()
    print("   per k-step and wave: reads + MFMA issue " + " ".join(f"{float(t[:, wv, 4].sum() / t[:, wv, 7].sum().clamp(min=1)):.0f}" for wv in range(8))
          + f" | DMA wait {float(t[:, :8, 5].sum() / ks):.0f} | barrier " + " ".join(f"{float(t[:, wv, 6].sum() / t[:, wv, 7].sum().clamp(min=1)):.0f}" for wv in range(8)), flush=True)
