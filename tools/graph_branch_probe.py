#!/usr/bin/env python3
"""Do two parallel branches of a captured hipGraph run side by side on this runtime?  Two decode-attention launches that
each occupy 16 of the 256 CUs (B = 2, S = 16384, 8 splits), (a) back to back on one stream, (b) on two streams forked
and joined inside one captured graph, (c) the same two streams eagerly.  If branches overlap, (b) takes about as long as
ONE launch; if the runtime serialises them, as long as (a)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402

dev = "cuda"
B, S, Hq, Hkv, D, splits = 2, 16384, 32, 8, 128, 8
dt = torch.bfloat16


def make():
    k = torch.randn(B * S + 1, Hkv, D, device=dev).to(dt)
    v = torch.randn(B * S + 1, Hkv, D, device=dev).to(dt)
    q = torch.randn(B, Hq, D, device=dev).to(dt)
    o = torch.empty_like(q)
    lens = torch.full((B,), S, dtype=torch.int32, device=dev)
    indptr = ops.kv_indptr(lens).clone()
    idx = (torch.randperm(B * S, device=dev) + 1).to(torch.int32)
    ws = torch.empty(ops.decode_workspace_numel(B, Hq, D, splits), dtype=torch.float32, device=dev)
    return lambda: ops.decode_attention(q, k, v, o, indptr, idx, D ** -0.5, 0.0, splits, ws)


fa, fb = make(), make()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    fa(); fb()


def forked():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        fa()
    with torch.cuda.stream(s2):
        fb()
    cur.wait_stream(s1); cur.wait_stream(s2)


def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def graph_of(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):          # ten pairs per replay: the replay's fixed cost does not hide the answer
            fn()
    return g


one = graph_of(fa)
g_ser, g_fork = graph_of(serial), graph_of(forked)
print(f"one launch x10 per replay       : {timeit(one.replay) / 10:7.1f} us per launch")
print(f"graph, one stream (a, b) x10    : {timeit(g_ser.replay) / 10:7.1f} us per pair")
print(f"graph, two branches (a | b) x10 : {timeit(g_fork.replay) / 10:7.1f} us per pair")
print(f"eager, two streams              : {timeit(forked):7.1f} us per pair (host-bound if small)")
