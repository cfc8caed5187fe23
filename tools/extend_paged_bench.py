#!/usr/bin/env python3
"""Micro-benchmark: extend attention over a cached prefix on a page-aligned pool, token-granular kv_indices against one
index per page (mi_extend_attn_paged).  Default: 16 x 2048 new tokens over a 2048-key prefix, Llama-3-8B heads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402

dev = "cuda"
B, EXT, PRE = int(os.environ.get("B", "16")), int(os.environ.get("EXT", "2048")), int(os.environ.get("PRE", "2048"))
Hq, Hkv, D, dtype = 32, 8, 128, torch.bfloat16
for P in (16, 64):
    g = torch.Generator().manual_seed(P)
    L = PRE + EXT
    need = -(-L // P)
    order = torch.randperm(B * need, generator=g) + 1
    slots = (B * need + 1) * P
    r2t = torch.zeros(B, L, dtype=torch.int32)
    for i in range(B):
        pg = order[i * need: (i + 1) * need]
        r2t[i] = (pg[:, None] * P + torch.arange(P)[None, :]).reshape(-1)[:L].to(torch.int32)
    kc = torch.randn(slots, Hkv, D, device=dev).to(dtype)
    vc = torch.randn(slots, Hkv, D, device=dev).to(dtype)
    q = torch.randn(B * EXT, Hq, D, device=dev).to(dtype)
    kn = torch.randn(B * EXT, Hkv, D, device=dev).to(dtype)
    vn = torch.randn(B * EXT, Hkv, D, device=dev).to(dtype)
    rpi = torch.arange(B, dtype=torch.int64, device=dev)
    pre_d = torch.full((B,), PRE, dtype=torch.int32, device=dev)
    ext_d = torch.full((B,), EXT, dtype=torch.int32, device=dev)
    kvp, qop = ops.kv_indptr(pre_d).clone(), ops.kv_indptr(ext_d).clone()
    idx = torch.empty(B * PRE, dtype=torch.int32, device=dev)
    ops.kv_indices(r2t.to(dev), rpi, pre_d, kvp, idx)
    pi, px = ops.kv_page_tables(r2t.to(dev), rpi, pre_d, P)
    pi = pi.clone()
    o = torch.empty_like(q)

    def tok():
        ops.extend_attention(q, kn, vn, o, kc, vc, qop, kvp, idx, EXT, D ** -0.5, 0.0, True, -1)

    def paged():
        ops.extend_attention_paged(q, kn, vn, o, kc, vc, qop, kvp, idx, pi, px, P, EXT, D ** -0.5, 0.0, True, -1)

    res = {}
    for name, fn in (("token-granular", tok), ("page-granular", paged)):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        res[name] = e0.elapsed_time(e1) / 10
    flops = 4.0 * Hq * D * B * EXT * (PRE + (EXT + 1) / 2)
    print(f"page {P}: {B} x {EXT} over {PRE} cached keys: " + ", ".join(f"{k} {v:.3f} ms ({flops / v / 1e9:.0f} TFLOP/s)" for k, v in res.items()), flush=True)
