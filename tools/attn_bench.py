#!/usr/bin/env python3
"""Micro-benchmark of mi_decode_attn at the BASELINE shape (B=128, S=2048, Hq=32, Hkv=8, D=128).
Cycles over several layer-sized pools so the 256 MB Infinity Cache cannot help.  Not a test."""
import math
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import ops  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "128")); S = int(os.environ.get("S", "2048"))
Hq, Hkv, D = int(os.environ.get("HQ", "32")), int(os.environ.get("HKV", "8")), 128
npool = int(os.environ.get("NPOOL", "6"))
contig = os.environ.get("CONTIG", "0") == "1"
slots = B * S + 1
pools = [(torch.randn(slots, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16),
          torch.randn(slots, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)) for _ in range(npool)]
if os.environ.get("VOFF") is not None:      # K and V carved out of ONE allocation, V shifted by VOFF bytes past the end of K:
    voff = int(os.environ["VOFF"]) // 2      # does the relative placement of the two buffers (memory channels) matter?
    n = slots * Hkv * D
    pools = []
    for _ in range(npool):
        big = torch.randn(2 * n + voff + 64, device=dev, dtype=torch.float32).to(torch.bfloat16)
        pools.append((big[:n].view(slots, Hkv, D), big[n + voff: 2 * n + voff].view(slots, Hkv, D)))
q = torch.randn(B, Hq, D, device=dev, dtype=torch.float32).to(torch.bfloat16)
o = torch.empty_like(q)
g = torch.Generator(device=dev).manual_seed(0)
PAGE = int(os.environ.get("PAGE", "1"))          # PAGE >= 16: page-aligned layout (scattered pages), page-granular indices
if PAGE > 1:
    assert S % PAGE == 0 and os.environ.get("RAGGED", "0") != "1"
    npg = B * S // PAGE
    page_ids = (torch.arange(npg, device=dev) if contig else torch.randperm(npg, device=dev, generator=g)).to(torch.int32)
    idx = (page_ids.view(-1, 1) * PAGE + torch.arange(PAGE, device=dev, dtype=torch.int32).view(1, -1)).reshape(-1) + 1
    slots = B * S + 1
    # page p covers slots p*PAGE+1 .. : shift by one page so that ids line up with (id << shift): use a pool with one extra page
    idx = idx - 1 + PAGE
    pools = [(torch.randn(slots + PAGE, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16),
              torch.randn(slots + PAGE, Hkv, D, device=dev, dtype=torch.float32).to(torch.bfloat16)) for _ in range(npool)]
    page_indices = page_ids + 1
    page_indptr = (torch.arange(B + 1, device=dev) * (S // PAGE)).to(torch.int32)
else:
    idx = (torch.arange(B * S, device=dev) if contig else torch.randperm(B * S, device=dev, generator=g)).to(torch.int32) + 1
if os.environ.get("RAGGED", "0") == "1":      # S_i ~ U[1, 2S], mean S (SURVEY 8d ragged variant), same pool size
    gl = torch.Generator().manual_seed(0)
    lens = torch.randint(1, 2 * S + 1, (B,), generator=gl)
    while int(lens.sum()) > B * S:
        lens = (lens.float() * 0.98).long().clamp(min=1)
    sl = lens.to(torch.int64).to(dev)
else:
    sl = torch.full((B,), S, dtype=torch.int64, device=dev)
indptr = ops.kv_indptr(sl)
tot = int(sl.sum())
abytes = 2 * tot * Hkv * D * 2 + 2 * B * Hq * D * 2 + 4 * tot
print(f"tokens {tot} max {int(sl.max())} min {int(sl.min())}", flush=True)
CHUNK = int(os.environ.get('CHUNK', '0'))
WORK = None
if os.environ.get('WORKLIST', '0') == '1':      # the backend's ragged plan: fixed chunks + longest-first launch list
    from types import SimpleNamespace
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    be = MiAttnBackend.__new__(MiAttnBackend)
    be.num_kv_head, be.cu_count, be.max_kv_splits, be.device = Hkv, ops.cu_count(), int(os.environ.get('MAXS', '8')), dev
    if os.environ.get('FLOOR'):                 # pin the chunk (otherwise the backend's simulated choice)
        be.min_split_chunk = int(os.environ['FLOOR'])
    ns_, CHUNK, WORK = be._choose_split_plan(B, tot, sl.cpu())
    os.environ['SPLITS'] = str(ns_)
    print(f'work list: {0 if WORK is None else WORK.shape[0]} entries, chunk {CHUNK}, splits {ns_}', flush=True)
    if os.environ.get('PLAN', '0') == '1':      # the graph-replay form: capacity-sized launch, plan read on the device
        ns_, CHUNK, wl = be._plan_on_host(B, tot, sl.cpu(), force_list=True)
        cap = B * be.max_kv_splits
        buf = torch.zeros(4 + 2 * cap, dtype=torch.int32)
        buf[0], buf[1], buf[2] = wl.shape[0], ns_, CHUNK
        buf[4: 4 + 2 * wl.shape[0]] = wl.reshape(-1)
        buf = buf.to(dev)
        WORK, CHUNK = (buf[4:].view(cap, 2), buf[:4]), 0
        os.environ['SPLITS'] = str(be.max_kv_splits)
        print(f'device plan: {wl.shape[0]} live of {cap} entries, splits {ns_}', flush=True)
KV8 = os.environ.get("KV8", "0") == "1"       # fp8 e4m3 pool (unit scales); token-granular or paged
if KV8:
    pools = [(k.to(torch.float8_e4m3fn), v.to(torch.float8_e4m3fn)) for k, v in pools]
    abytes = 2 * tot * Hkv * D + 2 * B * Hq * D * 2 + 4 * tot


def run(k, v, ns, ws):
    if KV8 and PAGE == 1:
        ops.decode_attention_fp8kv(q, k, v, indptr, idx, 1 / math.sqrt(D), 1.0, 1.0, 0.0, ns, ws, o=o, split_chunk=CHUNK, work=WORK)
        return
    if PAGE > 1 and os.environ.get("PAGED_KERNEL", "1") == "1":
        ops.decode_attention_paged(q, k, v, indptr, page_indptr, page_indices, PAGE, 1 / math.sqrt(D), 0.0, ns, ws, o=o,
                                   split_chunk=CHUNK, work=WORK)
    else:
        ops.decode_attention(q, k, v, o, indptr, idx, 1 / math.sqrt(D), 0.0, ns, ws, split_chunk=CHUNK, work=WORK)


for ns in [int(x) for x in os.environ.get("SPLITS", "1,2,4,8").split(",")]:
    ws = torch.empty(max(1, ops.decode_workspace_numel(B, Hq, D, ns)), dtype=torch.float32, device=dev)
    for k, v in pools:
        run(k, v, ns, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    e0.record()
    for _ in range(reps):
        for k, v in pools:
            run(k, v, ns, ws)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * npool)
    print(f"B={B} S={S} Hq={Hq} Hkv={Hkv} contig={int(contig)} page={PAGE} paged_kernel={os.environ.get('PAGED_KERNEL', '1') if PAGE > 1 else '-'} splits={ns} W={os.environ.get('MI_DECODE_W','auto')}: "
          f"{us:8.1f} us  {abytes/us/1e3:7.1f} GB/s", flush=True)
