#!/usr/bin/env python3
"""Diagnostic (library built with `make TUNING=1`): where a work item of extend_attn32_kernel spends its cycles (wave 0,
s_memtime at 100 MHz ticks x ... see mi_x32_stamps in extend_attn.hip).  usage: CASES=16x2048x0,4x8192x0 python tools/x32_stamps.py"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops, _lib  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
dev = "cuda"
Hq, Hkv, D = 32, 8, 128
for case in os.environ.get("CASES", "16x2048x0,8x2048x0,4x8192x0").split(","):
    B, S, P = (int(v) for v in case.split("x"))
    E = B * S
    q = torch.randn(E, Hq, D, device=dev).to(torch.bfloat16)
    k = torch.randn(E, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(E, Hkv, D, device=dev).to(torch.bfloat16)
    kb = torch.randn(B * P + 1, Hkv, D, device=dev).to(torch.bfloat16)
    vb = torch.randn(B * P + 1, Hkv, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    qo = ops.kv_indptr(torch.full((B,), S, dtype=torch.int32, device=dev))
    kvp = ops.kv_indptr(torch.full((B,), P, dtype=torch.int32, device=dev)).clone()
    idx = torch.randperm(max(B * P, 1), device=dev).to(torch.int32) + 1
    run = lambda: ops.extend_attention(q, k, v, o, kb, vb, qo, kvp, idx, S, 1 / math.sqrt(D), 0.0, True, -1)  # noqa: E731
    run(); torch.cuda.synchronize()
    assert lib.mi_debug_x32_stamps(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); e1.synchronize()
    assert lib.mi_debug_x32_stamps(buf, 0) == 0
    t = [float(x) for x in buf]
    n = t[0]
    print(f"{case}: {e0.elapsed_time(e1) / reps:.3f} ms, items/launch {n / reps:.0f}, key tiles/item {t[5] / n:.2f}; s_memtime ticks per item: "
          f"prologue issue {t[1] / n:.0f}, first tile wait {t[2] / n:.0f}, tile loop {t[3] / n:.0f} ({t[3] / t[5]:.0f} per tile), "
          f"epilogue {t[4] / n:.0f}, end barrier {t[6] / n:.0f}", flush=True)
