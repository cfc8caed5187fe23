#!/usr/bin/env python3
"""Diagnostic (library built with `make TUNING=1`): where a phase of fp8_gemm_xd_kernel spends its cycles.
Runs the decode GEMM shapes at M=128 with MI_XD_VAR bit 16 set and prints, per shape, the mean cycles per phase that a
wave spends (a) between the barrier and its next counted wait (DMA issue + fragment reads + MFMAs), (b) in the counted
vmcnt wait, (c) in the barrier."""
import ctypes as C
import os
import sys

os.environ["MI_XD_VAR"] = str(int(os.environ.get("MI_XD_VAR", "0")) | 16)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops, _lib  # noqa: E402

lib = C.CDLL(_lib.LIB_PATH)
FP8 = torch.float8_e4m3fn
M = int(os.environ.get("M", "128"))
buf = (C.c_ulonglong * (512 * 12 * 8))()
for N, K in [(28672, 4096), (6144, 4096), (4096, 4096), (4096, 14336)]:
    ws = [torch.randn(N, K, device="cuda").to(FP8) for _ in range(4)]
    x = torch.randn(M, K, device="cuda").to(FP8)
    sa = torch.ones(1, device="cuda"); sb = torch.ones(1, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for w in ws:
        ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    torch.cuda.synchronize()
    assert lib.mi_debug_xd_stamps(None, 1) == 0
    reps = 8
    for _ in range(reps):
        for w in ws:
            ops.fp8_gemm(x, w.t(), sa, sb, torch.bfloat16, None, out)
    torch.cuda.synchronize()
    assert lib.mi_debug_xd_stamps(buf, 0) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).view(512, 12, 8)
    used = t[:, :, 3] > 0
    ph = t[:, :, 3][used]
    nwg = int(used[:, 0].sum())
    print(f"N={N} K={K}: workgroups {int(used[:, 0].sum())}, phases/wg {float(ph.mean()) / (reps * len(ws)):.1f}")
    nwg = int(used[:, 0].sum())
    # role kernel (MI_GEMM_XW != 0): waves 0-3 consumers (compute | - | barrier), waves 4-7 loaders (issue | vmcnt | barrier)
    roles = os.environ.get("MI_GEMM_XW", "1") != "0"
    if roles and nwg:
        norm = float(ph.mean()) / (reps * len(ws)) / max(float(t[:nwg, :4, 3].sum()), 1.0)
        tot, rt = float(t[:nwg, :4, 4].sum()) * norm, float(t[:nwg, :4, 5].sum()) * norm
        print(f"   consumers: entry -> end of main loop {tot:9.0f} cycles = {rt / 100:7.2f} us  (clock {tot / max(rt, 1e-9) * 100:6.0f} MHz)")
        a0, a1 = t[:nwg, :4, 6], t[:nwg, :4, 7]
        base = float(a0.min())
        print(f"   last launch, 100 MHz ticks from the first entry: entries {float(a0.mean() - base) / 100:6.2f} us mean, "
              f"{float(a0.max() - base) / 100:6.2f} max; end of main loop {float(a1.mean() - base) / 100:6.2f} mean, "
              f"{float(a1.min() - base) / 100:6.2f} min, {float(a1.max() - base) / 100:6.2f} max")
    rows = (("compute/issue", 0), ("vmcnt wait", 1), ("barrier", 2)) if roles else \
           (("compute", 0), ("vmcnt wait", 1), ("barrier", 2), ("  dma issue", 4), ("  frag reads", 5), ("  mfma issue", 6))
    for name, i in rows:
        per = (t[:, :, i][used] / ph)
        pw = (t[:nwg, :, i] / t[:nwg, :, 3].clamp(min=1)).mean(0)
        print(f"   {name:13s} mean {float(per.mean()):7.0f} cyc/phase   per wave: " + " ".join(f"{float(v):6.0f}" for v in pw))
