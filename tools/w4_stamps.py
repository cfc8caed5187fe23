#!/usr/bin/env python3
"""Diagnostic (library built with `make TUNING=1`): where a phase of w4a16_xw_kernel spends its cycles at the C4 shapes
(Llama-2-7B AWQ, M = 64, fp16).  Waves 0-7 consumers (compute | - | barrier), waves 8-11 loaders (issue | vmcnt | barrier)."""
import ctypes as C
import os
import sys

os.environ["MI_W4_STAMPS"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops, _lib  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
M = int(os.environ.get("M", "64"))
dt = torch.float16
buf = (C.c_ulonglong * (512 * 12 * 8))()
g = torch.Generator().manual_seed(0)
for N, K in [(22016, 4096), (12288, 4096), (4096, 4096), (4096, 11008)]:
    ws = []
    for _ in range(3):
        qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
        qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 128, N // 8), dtype=torch.int32, generator=g)
        scales = (torch.rand(K // 128, N, generator=g) * 0.01 + 0.001).to(dt)
        ws.append(ops.w4_repack(qweight.cuda(), qzeros.cuda(), scales.cuda(), 128, ops.MI_W4_AWQ)[:2])
    x = torch.randn(M, K, generator=g).to(dt).cuda()
    for qw, zs in ws:
        ops.w4a16_gemm(x, qw, zs, N, 128)
    torch.cuda.synchronize()
    assert lib.mi_debug_w4_stamps(None, 1) == 0
    reps = 8
    for _ in range(reps):
        for qw, zs in ws:
            ops.w4a16_gemm(x, qw, zs, N, 128)
    torch.cuda.synchronize()
    assert lib.mi_debug_w4_stamps(buf, 0) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(512, 12, 8)
    used = t[:, :, 3] > 0
    nwg = int(used[:, 0].sum())
    ph = t[:nwg, :, 3]
    print(f"N={N} K={K}: workgroups(y=0) {nwg}, phases/wg {float(ph[:, 0].mean()) / (reps * len(ws)):.1f}")
    tot = t[:nwg, :8, 4].sum() / max(t[:nwg, :8, 3].sum(), 1) * float(ph[:, 0].mean()) / (reps * len(ws))
    rt = t[:nwg, :8, 5].sum() / max(t[:nwg, :8, 3].sum(), 1) * float(ph[:, 0].mean()) / (reps * len(ws))
    print(f"   consumers: entry -> end of main loop {float(tot):9.0f} cycles = {float(rt) / 100:7.2f} us")
    for name, i in (("compute/issue", 0), ("vmcnt wait", 1), ("barrier", 2)):
        pw = (t[:nwg, :, i] / t[:nwg, :, 3].clamp(min=1)).mean(0)
        print(f"   {name:13s} cyc/phase per wave: " + " ".join(f"{float(v):6.0f}" for v in pw))
