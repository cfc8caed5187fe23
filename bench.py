#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s of the hot path at BASELINE.json's configuration --
Llama-3-8B, per-tensor FP8 linears, TP = --gpus, batch 128, KV length 2048, bf16 paged KV
(page_size 1, slots scattered over the pool), synthetic data, dummy weights.

One "step" = one decode step of the whole layer stack through the plugin surfaces
(MiAttnBackend.init_forward_metadata + forward, Fp8LinearMethod.apply, plus the glue kernels and
the bf16 lm_head), captured once in a hipGraph and replayed.  Inputs (weights, KV pool,
req_to_token) are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          (TP=N over RCCL)
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", choices=["llama3-8b", "llama3-70b"], default="llama3-8b",
                    help="llama3-8b = BASELINE.json's headline (C3); llama3-70b = its TP=8 configuration (C5, batch 256)")
    ap.add_argument("--batch", type=int, default=0, help="decode batch (default: 128 for llama3-8b, 256 for llama3-70b)")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; invalidates the metric)")
    ap.add_argument("--contiguous", action="store_true", help="contiguous KV slots instead of scattered")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--act-scheme", choices=["static", "dynamic"], default="static",
                    help="FP8 activation scale: static = serialized FP8 checkpoint with calibrated per-tensor "
                         "input scales (the reference's own FP8 test model format), dynamic = bf16 checkpoint "
                         "quantised at load, per-tensor absmax every call")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "single-GPU rehearsals of the TP path with MI_BENCH_SAME_GPU=1)")
    ap.add_argument("--no-custom-ar", action="store_true", help="TP>1: use RCCL only (skip the native xGMI all-reduce)")
    ap.add_argument("--splits", type=int, default=0, help="force the split-KV count (0 = backend heuristic)")
    ap.add_argument("--prefill-batch", type=int, default=-1,
                    help="sequences of --seq tokens in the prefill leg (-1 = the decode batch: the metric's 128 x 2048; 0 = skip)")
    ap.add_argument("--prefill-chunk", type=int, default=8,
                    help="sequences per EXTEND batch of the prefill leg (8 x 2048 tokens: the 134-MB activations between the "
                         "GEMMs and the norm / rope / KV-write kernels stay in the 256-MB Infinity Cache; measured on one box: "
                         "2 -> 1942, 4 -> 2021, 8 -> 2044, 16 -> 2015, 32 -> 2012 TFLOP/s)")
    ap.add_argument("--no-plugin-surface", action="store_true", help="skip the unfused plugin-surface-only step timing")
    ap.add_argument("--tbo", choices=["auto", "on", "off"], default="auto",
                    help="two-micro-batch overlap of the decode step (each half of the batch on its own stream, own "
                         "all-reduce communicator): auto = on for TP > 1 (C5: all-reduce overlapped with attention)")
    ap.add_argument("--kv-dtype", choices=["bf16", "fp8"], default="bf16",
                    help="KV cache dtype: bf16 is BASELINE.json's configuration (the headline); fp8 = e4m3fn pool "
                         "(SURVEY 8f row 1), reported as a variant -- a different workload, not the headline")
    ap.add_argument("--kernel-reps", type=int, default=3, help="passes over all layers for the roofline timing")
    return ap.parse_args()


def dist_setup(n, backend="nccl"):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MI_BENCH_SAME_GPU") == "1":   # rehearsal: every rank on GPU 0 (needs --dist-backend gloo)
        local = 0
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit(f"--gpus {n} needs torch.distributed.run with --nproc-per-node {n}")
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    group = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
        group = torch.distributed.group.WORLD
    return world, rank, local, group


def make_custom_ar(world, rank, dev, msg_bytes):
    """Native xGMI all-reduce for TP>1, self-checked against RCCL once; any problem -> RCCL only."""
    if world == 1 or os.environ.get("MI_DISABLE_CUSTOM_AR") == "1":
        return None
    from iaas_sglang_amd.custom_all_reduce import CustomAllreduce
    cpu_group = torch.distributed.new_group(backend="gloo")
    ca = CustomAllreduce(cpu_group, dev, max_size=max(8 * 1024 * 1024, msg_bytes))   # ranks agree on .disabled
    if ca.disabled:
        if rank == 0:
            print(f"[bench] native all-reduce unavailable ({ca.init_error}); using RCCL", file=sys.stderr)
        return None
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    x = torch.randint(-4, 5, (msg_bytes // 2,), device=dev, generator=g).to(torch.bfloat16)
    ref = x.clone()
    torch.distributed.all_reduce(ref)
    y = ca.all_reduce(x)
    torch.cuda.synchronize()
    ok = bool(torch.equal(y, ref)) and not ca.timed_out()
    flag = torch.tensor([1 if ok else 0], device=dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)       # every rank takes the same branch below
    if int(flag.item()) != 1:
        if rank == 0:
            print("[bench] native all-reduce self-check failed; using RCCL", file=sys.stderr)
        return None
    # the fused all-reduce + add + RMSNorm + fp8 form, fed from the registered staging buffer as the step feeds it, must
    # give the bits of the unfused sequence; any local problem (an exception included) switches it off on every rank
    fused = 1
    try:
        from iaas_sglang_amd import ops
        rows, Hh = max(1, msg_bytes // 2 // 4096), 4096
        gc = torch.Generator(device=dev).manual_seed(7)
        xs = torch.randn(rows, Hh, device=dev, generator=torch.Generator(device=dev).manual_seed(200 + rank)).to(torch.bfloat16)
        res = torch.randn(rows, Hh, device=dev, generator=gc).to(torch.bfloat16)
        w = torch.randn(Hh, device=dev, generator=gc).to(torch.bfloat16)
        qs = torch.tensor([0.02], device=dev)
        r1, r2 = res.clone(), res.clone()
        want = ops.rmsnorm_fp8(ca.all_reduce(xs), w, 1e-5, qs, residual=r1)
        buf = ca.staging((rows, Hh), torch.bfloat16)
        buf.copy_(xs)
        _, got = ca.all_reduce_add_rmsnorm(buf, r2, w, 1e-5, q_scale=qs, want_out=False)
        torch.cuda.synchronize()
        fused = int(torch.equal(got.view(torch.uint8), want.view(torch.uint8)) and torch.equal(r1, r2) and not ca.timed_out())
    except Exception as e:  # noqa: BLE001
        fused = 0
        print(f"[bench] rank {rank}: fused all-reduce+norm self-check raised {type(e).__name__}: {e}", file=sys.stderr)
    fused_ok = torch.tensor([fused], device=dev)
    torch.distributed.all_reduce(fused_ok, op=torch.distributed.ReduceOp.MIN)
    if int(fused_ok.item()) != 1:
        ca.fuse_norm = False
        if rank == 0:
            print("[bench] fused all-reduce+norm self-check failed; using the unfused sequence", file=sys.stderr)
    return ca


def barrier_sync(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def attention_bytes(B, S, Hkv, D, Hq, esize=2, kv_esize=None):
    """Algorithmic bytes of ONE decode-attention launch: K+V rows once, q, o, kv_indices."""
    return 2 * B * S * Hkv * D * (kv_esize or esize) + 2 * B * Hq * D * esize + 4 * B * S


def time_attention_kernel(stack, runner, backend, fb, reps):
    """Average duration of one decode-attention launch, HIP events on the launch stream, cycling
    through every layer's pool so nothing is served from the 256 MB Infinity Cache."""
    from iaas_sglang_amd import ops
    s = stack.shape
    B = fb.batch_size
    q = torch.randn(B, stack.Hq, s.head_dim, device=runner.device, dtype=torch.float32).to(stack.dtype)
    o = torch.empty_like(q)
    md = backend.forward_metadata
    pool = runner.token_to_kv_pool

    fp8_pool = pool.get_key_buffer(0).element_size() == 1

    def one_pass():
        for li in range(s.layers):
            if fp8_pool:
                ops.decode_attention_fp8kv(q, pool.get_key_buffer(li), pool.get_value_buffer(li), md.kv_indptr,
                                           md.kv_indices, s.head_dim ** -0.5, 1.0, 1.0, 0.0, md.num_kv_splits,
                                           md.workspace, o=o)
            else:
                ops.decode_attention(q, pool.get_key_buffer(li), pool.get_value_buffer(li), o, md.kv_indptr,
                                     md.kv_indices, s.head_dim ** -0.5, 0.0, md.num_kv_splits, md.workspace)

    one_pass()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        one_pass()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * s.layers)


def ragged_attention_leg(H, stack, runner, backend, B, S, dev, reps=2):
    """Decode attention on the RAGGED variant of the workload (S_i ~ U[1, 2S], same total pool): the
    backend's ragged plan (fixed 512-key splits + a longest-first launch list) against the same layer pools."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, 2 * S + 1, (B,), generator=g)
    while int(lens.sum()) > B * S:
        lens = (lens.float() * 0.98).long().clamp(min=1)
    fb = H.make_decode_batch(runner, backend, B, 0, dev, seed=1, ragged=lens)
    backend.init_forward_metadata(fb)
    md = backend.forward_metadata
    # host time of the plan itself (what init_forward_metadata / graph replay spend on it before the launches)
    import time
    backend._plan_on_host(B, int(lens.sum()), fb.seq_lens_cpu)
    t0 = time.perf_counter()
    for _ in range(5):
        backend._plan_on_host(B, int(lens.sum()), fb.seq_lens_cpu)
    plan_ms = (time.perf_counter() - t0) / 5 * 1e3
    s = stack.shape
    q = torch.randn(B, stack.Hq, s.head_dim, device=dev, dtype=torch.float32).to(stack.dtype)
    o = torch.empty_like(q)
    pool = runner.token_to_kv_pool
    if pool.get_key_buffer(0).element_size() == 1:
        return None

    def one_pass():
        for li in range(s.layers):
            ops.decode_attention(q, pool.get_key_buffer(li), pool.get_value_buffer(li), o, md.kv_indptr, md.kv_indices,
                                 s.head_dim ** -0.5, 0.0, md.num_kv_splits, md.workspace, split_chunk=md.split_chunk,
                                 work=md.work)
    one_pass()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        one_pass()
    e1.record(stream)
    e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (reps * s.layers)
    tot = int(lens.sum())
    nbytes = 2 * tot * stack.Hkv * s.head_dim * 2 + 2 * B * stack.Hq * s.head_dim * 2 + 4 * tot
    return {"keys": tot, "max_len": int(lens.max()), "min_len": int(lens.min()), "avg_launch_us": round(t * 1e6, 2),
            "achieved_GBps": round(nbytes / t / 1e9, 1), "kv_splits": md.num_kv_splits, "split_chunk": md.split_chunk,
            "work_items": None if md.work is None else int(md.work.shape[0]), "plan_host_ms": round(plan_ms, 3)}


MFMA_PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak


def prefill_leg(H, stack, runner, backend, shape, nseq, S, dev, world, chunk=16):
    """Prefill TFLOP/s, the other half of BASELINE.json's metric, at the metric's size: `nseq` sequences x S new
    tokens (no cached prefix) through the whole layer stack, as ceil(nseq / chunk) EXTEND batches of `chunk`
    sequences each (what a scheduler with a 32k-token prefill budget would issue), eager launches, every chunk
    writing its K/V rows to scattered pool slots.  One untimed chunk first (lazy inits, scratch growth); the timed region covers ALL
    chunks between two barrier + synchronize pairs."""
    chunks = []
    done = 0
    while done < nseq:
        n = min(chunk, nseq - done)
        chunks.append(n)
        done += n

    def run(n, seed, first_req):
        # chunk i: request rows first_req .. first_req + n - 1, its own slot range of the pool, its own activations
        fb = H.make_extend_batch(runner, backend, [0] * n, [S] * n, dev, seed=seed, req_offset=first_req,
                                 slot_offset=first_req * S)
        hidden = torch.randn(n * S, shape.hidden, device=dev, dtype=torch.float32).to(stack.dtype)
        return fb, hidden

    def forward(fb, hidden):
        backend.init_forward_metadata(fb)
        return stack.forward(hidden, fb.positions, fb, backend, last_token_logits=fb.extend_seq_lens)

    warm = run(chunks[0], 99, 0)             # one untimed chunk (lazy inits, scratch growth)
    forward(*warm)
    del warm
    batches, first = [], 0
    for i, n in enumerate(chunks):          # DISTINCT chunks: the stack updates `hidden` in place (residual stream)
        batches.append(run(n, 3 + i, first))
        first += n
    barrier_sync(world)
    t0 = time.perf_counter()
    for b in batches:
        forward(*b)
    barrier_sync(world)
    sec = time.perf_counter() - t0
    T = nseq * S
    lin = 2.0 * T * shape.layers * ((shape.num_heads + 2 * shape.num_kv_heads) * shape.head_dim * shape.hidden
                                    + shape.num_heads * shape.head_dim * shape.hidden + 3 * shape.intermediate * shape.hidden)
    attn = 4.0 * shape.num_heads * shape.head_dim * nseq * (S * (S + 1) / 2) * shape.layers     # causal QK^T + PV
    head = 2.0 * nseq * shape.vocab * shape.hidden
    flops = lin + attn + head
    # MFMA roofline of the leg: fp8 linears at the dense fp8 peak, attention and the bf16 lm_head at the bf16 peak
    ideal = (lin / MFMA_PEAK_FP8_TFLOPS + (attn + head) / MFMA_PEAK_BF16_TFLOPS) / 1e12 / world
    return {"tflops": round(flops / sec / 1e12, 2), "ms": round(sec * 1e3, 2), "tokens": T,
            "tokens_per_s": round(T / sec, 1),
            "roofline": {"bound": "mfma", "achieved": round(flops / sec / 1e12, 2),
                         "peak": round(flops / ideal / 1e12, 1), "unit": "TFLOP/s",
                         "frac": round(ideal / sec, 4),
                         "note": "peak = flops / (fp8 linear flops at 5 PF + bf16 attention and lm_head flops at 2.5 PF per GPU)"},
            "sample": f"{nseq} sequences x {S} tokens{' = the metric' + chr(39) + 's batch' if nseq == 128 and S == 2048 else ' (a SAMPLE of the metric' + chr(39) + 's 128 x 2048 batch)'}, no prefix, {shape.layers} layers, eager, "
                      f"{len(chunks)} DISTINCT EXTEND batches of {chunk} sequences (own request rows, pool slots and activations); "
                      f"flops = linears {lin:.3e} + causal attention {attn:.3e} + lm_head {head:.3e}"}


def cpu_baseline(shape, B, S, sample_requests=64, timed_layers=12):
    """The reference's torch-native arithmetic (our CPU restatement, oracle/) for ONE decoder layer,
    timed on the host cores on a bounded sample: linears / norm / rope / activation at the full batch
    B, attention on `sample_requests` of the B requests at the full KV length S (its per-request loop
    is linear in the number of requests, torch_native_backend.py:147-178) and scaled by B/sample;
    extrapolated to `shape.layers` layers + lm_head."""
    from oracle import attention as oa
    from oracle import elementwise as oe
    from oracle import quant as oq
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))        # the GPU box gives one GPU's job a 16-core share
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    D, Hq, Hkv, H, I = shape.head_dim, shape.num_heads, shape.num_kv_heads, shape.hidden, shape.intermediate
    dt = torch.bfloat16
    Bs = min(B, sample_requests)

    def w(n, k):
        t = (torch.rand(n, k, generator=g) * 2e-3 - 1e-3).to(dt)
        q, inv = oq.input_to_float8(t)
        return q.t(), inv.reshape(1)

    qkv_w, qkv_s = w((Hq + 2 * Hkv) * D, H)
    o_w, o_s = w(H, Hq * D)
    gu_w, gu_s = w(2 * I, H)
    dn_w, dn_s = w(H, I)
    norm_w = torch.ones(H, dtype=dt)
    slots = Bs * S + 1
    kc = torch.randn(slots, Hkv, D, generator=g, dtype=torch.float32).to(dt)
    vc = torch.randn(slots, Hkv, D, generator=g, dtype=torch.float32).to(dt)
    r2t = (torch.randperm(Bs * S, generator=g) + 1).to(torch.int32).view(Bs, S)
    rpi = torch.arange(Bs)
    sl = torch.full((Bs,), S, dtype=torch.int64)
    loc = r2t[:, S - 1].to(torch.int64)
    pos = torch.full((B,), S - 1, dtype=torch.int64)
    cache = oe.rope_cos_sin_cache(D, S + 8, shape.rope_theta)
    x0 = torch.randn(B, H, generator=g).to(dt)
    t_attn = [0.0]

    def layer(hidden, residual):
        x, residual = oe.rmsnorm(hidden, norm_w, shape.rms_eps, residual)
        qkv = oq.fp8_linear(x, qkv_w, qkv_s)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        q, k = oe.rope_neox(pos, q.contiguous(), k.contiguous(), cache, D)
        ta = time.perf_counter()
        a_s = oa.forward_decode(q[:Bs], k[:Bs].reshape(-1, Hkv, D), v[:Bs].reshape(-1, Hkv, D), kc, vc, r2t, rpi,
                                sl, loc, Hq, Hkv, D ** -0.5)
        t_attn[0] = time.perf_counter() - ta
        a = a_s.repeat(B // Bs + 1, 1)[:B]
        hidden = oq.fp8_linear(a, o_w, o_s)
        x, residual = oe.rmsnorm(hidden, norm_w, shape.rms_eps, residual)
        hidden = oq.fp8_linear(oe.silu_and_mul(oq.fp8_linear(x, gu_w, gu_s)), dn_w, dn_s)
        return hidden, residual

    h, r = layer(x0, x0.clone())          # warm-up
    times = []
    for _ in range(timed_layers):
        t0 = time.perf_counter()
        h, r = layer(h, r)
        total = time.perf_counter() - t0
        times.append(total - t_attn[0] + t_attn[0] * (B / Bs))
    t_layer = sorted(times)[len(times) // 2]
    lm = (torch.rand(8192, H, generator=g) * 2e-3 - 1e-3).to(dt)     # 8192 of the vocab rows
    t0 = time.perf_counter()
    _ = h @ lm.t()
    t_lm = (time.perf_counter() - t0) * (shape.vocab / 8192)
    step = t_layer * shape.layers + t_lm
    return {"value": round(B / step, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ (torch CPU restatement of the reference torch-native path): {timed_layers} timed "
                      f"decoder layers, linears/glue at B={B}, attention on {Bs} of {B} requests at S={S} scaled "
                      f"x{B / Bs:g} (median {t_layer:.2f} s/layer) x {shape.layers} layers + lm_head "
                      f"extrapolated from 8192 of {shape.vocab} rows"}


def main():
    a = parse()
    if os.environ.get("MI_BENCH_WATCHDOG"):     # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MI_BENCH_WATCHDOG"]), exit=True)
    world, rank, local, group = dist_setup(a.gpus, a.dist_backend)
    dev = torch.device("cuda", local)
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config
    import dataclasses

    shape = {"llama3-8b": H.LLAMA3_8B, "llama3-70b": H.LLAMA3_70B}[a.model]
    if not a.batch:
        a.batch = 128 if a.model == "llama3-8b" else 256
    if a.prefill_batch < 0:
        a.prefill_batch = a.batch
    if a.layers:
        shape = dataclasses.replace(shape, layers=a.layers)
    tp, B, S, dtype = world, a.batch, a.seq, torch.bfloat16
    if a.splits:
        os.environ["MI_ATTN_MAX_KV_SPLITS"] = str(a.splits)

    kv8 = a.kv_dtype == "fp8"
    runner = H.make_runner(shape, max_reqs=B, ctx=2 * S + 8, pool_tokens=B * S, dtype=dtype, device=dev, tp=tp,
                           fill_kv=True, seed=rank, max_kv_splits=a.splits or 8,
                           kv_dtype=torch.float8_e4m3fn if kv8 else None)
    backend = MiAttnBackend(runner)
    if a.splits:
        backend._choose_splits = lambda bs, tot, cap=None: a.splits
    custom_ar = make_custom_ar(world, rank, dev, B * shape.hidden * 2) if not a.no_custom_ar else None
    tbo = a.tbo == "on" or (a.tbo == "auto" and world > 1)
    # the second micro-batch needs its own staging buffer and barrier flags: a second communicator
    custom_ar_b = make_custom_ar(world, rank, dev, B * shape.hidden * 2) if (tbo and custom_ar is not None) else None
    static = a.act_scheme == "static"
    cfg = Fp8Config(is_checkpoint_fp8_serialized=static, activation_scheme=a.act_scheme)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, dev, tp=tp, rank=rank, group=group,
                         custom_ar=custom_ar)
    fb = H.make_decode_batch(runner, backend, B, S, dev, scattered=not a.contiguous, seed=0)
    ids = torch.randint(0, shape.vocab, (B,), device=dev)
    out_ids = torch.empty_like(ids)
    if static:
        backend.init_forward_metadata(fb)
        stack.calibrate_static_input_scales(torch.index_select(stack.embed, 0, ids), fb.positions, fb, backend)

    def step():
        hidden = torch.index_select(stack.embed, 0, ids)
        backend.init_forward_metadata(fb)
        logits = stack.forward(hidden, fb.positions, fb, backend)
        torch.argmax(logits, dim=-1, out=out_ids)

    serial_step = step
    if tbo:
        backend_b = MiAttnBackend(runner)
        halves = H.split_decode_batch(fb, backend_b, B // 2)
        tbo_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        tbo_ars = [custom_ar, custom_ar_b] if custom_ar is not None and custom_ar_b is not None else None
        if world > 1 and tbo_ars is None:
            raise SystemExit("--tbo with TP > 1 needs the native all-reduce (two communicators); RCCL calls of two "
                             "streams would have to be ordered identically on every rank")

        def step():   # noqa: F811  the timed step: both halves of the batch, each on its own stream
            hidden = torch.index_select(stack.embed, 0, ids)
            logits = stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], tbo_streams, tbo_ars)
            torch.argmax(logits, dim=-1, out=out_ids)

    def capture(fn):
        """hipGraph of one step (None when capture is impossible, e.g. gloo collectives): eager once on a side stream
        (allocator warm-up, lazy inits), then the capture."""
        fn()
        barrier_sync(world)
        if a.no_graph or not (world == 1 or a.dist_backend == "nccl"):
            return None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with contextlib.ExitStack() as es:
                for ca in (custom_ar, custom_ar_b):
                    if ca is not None:
                        es.enter_context(ca.capture())
                with torch.cuda.graph(g):
                    fn()
            return g
        except Exception as e:  # e.g. a collective that cannot be captured: fall back to eager launches
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            torch.cuda.synchronize()
            return None

    graph = capture(step)
    run = graph.replay if graph is not None else step
    timed_is_tbo = tbo
    tbo_auto_note = None
    if tbo and a.tbo == "auto":
        # auto: the two-micro-batch step is the timed one only if it is the faster one on this machine -- a short synced
        # comparison (max over ranks, so every rank takes the same branch) before the timed region
        g_ser = capture(serial_step)
        run_ser = g_ser.replay if g_ser is not None else serial_step

        def quick(fn, n=3):
            fn()
            barrier_sync(world)
            t_ = time.perf_counter()
            for _ in range(n):
                fn()
            barrier_sync(world)
            from iaas_sglang_amd.parallel import max_over_ranks as _mx
            return _mx(time.perf_counter() - t_, world, dev) / n * 1e3

        q_tbo, q_ser = quick(run), quick(run_ser)
        tbo_auto_note = f"auto: two-batch {q_tbo:.3f} ms vs serial {q_ser:.3f} ms in a 3-step comparison"
        if q_ser < q_tbo:
            run, graph, timed_is_tbo = run_ser, g_ser, False

    for _ in range(a.warmup):
        run()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier_sync(world)
    elapsed = time.perf_counter() - t0
    from iaas_sglang_amd.parallel import max_over_ranks
    elapsed = max_over_ranks(elapsed, world, dev)
    ms_per_step = elapsed / a.steps * 1e3
    tokens_per_s = B / (ms_per_step * 1e-3)          # TP: one batch of B tokens per step for the whole job

    def synced_median(fn, n):
        """bench_one_batch.latency_test_run_once (bench_one_batch.py:332-430): every decode step bracketed by a device
        synchronisation, the median latency reported."""
        lat = []
        for _ in range(n):
            barrier_sync(world)
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t)
        return max_over_ranks(sorted(lat)[len(lat) // 2], world, dev) * 1e3

    median_ms = synced_median(run, max(5, a.steps))

    def time_variant(fn):
        g_ = capture(fn)
        r_ = g_.replay if g_ is not None else fn
        for _ in range(a.warmup):
            r_()
        barrier_sync(world)
        t_ = time.perf_counter()
        for _ in range(a.steps):
            r_()
        barrier_sync(world)
        return max_over_ranks(time.perf_counter() - t_, world, dev) / a.steps * 1e3

    # ---- communication overlap: the serial step beside the two-micro-batch step, and each of them
    # with the all-reduces dropped (numerically meaningless, timing only): exposed communication = step - that
    overlap = None
    if tbo or world > 1:
        try:
            overlap = {"mode": ("two micro-batches of %d requests, one HIP stream and one native all-reduce communicator "
                                "each: the all-reduce of one half runs beside the attention / GEMMs of the other" % (B // 2))
                               if tbo else "none (serial step)"}
            if tbo:
                overlap["timed_step"] = "two micro-batches" if timed_is_tbo else "serial"
                if tbo_auto_note:
                    overlap["selection"] = tbo_auto_note
                overlap["ms_per_step_two_batch"] = round(ms_per_step if timed_is_tbo else time_variant(step), 4)
                overlap["ms_per_step_serial"] = round(time_variant(serial_step) if timed_is_tbo else ms_per_step, 4)
            if world > 1:
                H.LlamaStack.comm_disabled = True
                try:
                    if tbo:
                        overlap["exposed_comm_us_two_batch"] = round((overlap["ms_per_step_two_batch"] - time_variant(step)) * 1e3, 1)
                    ser = overlap.get("ms_per_step_serial", ms_per_step)
                    overlap["exposed_comm_us_serial"] = round((ser - time_variant(serial_step)) * 1e3, 1)
                finally:
                    H.LlamaStack.comm_disabled = False
        except Exception as e:  # informational
            overlap = {"error": f"{type(e).__name__}: {e}"}

    # ---- the same step through the plugin surfaces ONLY, as an unmodified model file would drive them
    # (models/llama.py:245-268: norm -> qkv_proj.apply -> rope -> attn.forward -> o_proj.apply -> norm -> gate_up.apply
    # -> act -> down.apply): no fused linear+consumer entry points, no producer-side fp8 quantisation
    plugin_only = None
    if not a.no_plugin_surface:
        try:
            H.LlamaStack.fuse_decode_layer, H.Linear.fuse_producer_quant = False, False
            g2 = capture(serial_step)
            run2 = g2.replay if g2 is not None else serial_step
            for _ in range(a.warmup):
                run2()
            barrier_sync(world)
            t1 = time.perf_counter()
            for _ in range(a.steps):
                run2()
            barrier_sync(world)
            ms2 = max_over_ranks(time.perf_counter() - t1, world, dev) / a.steps * 1e3
            plugin_only = {"ms_per_step": round(ms2, 4), "tokens_per_s": round(B / (ms2 * 1e-3), 1),
                           "median_synced_ms": round(synced_median(run2, max(5, a.steps)), 4), "hipgraph": g2 is not None,
                           "what": "apply()/forward() per layer exactly as models/llama.py:245-268 calls them; the "
                                   "headline additionally uses the fused linear+consumer entry points of DESIGN 3.6"}
            del g2
        except Exception as e:  # informational: never lose the headline to it
            plugin_only = {"error": f"{type(e).__name__}: {e}"}
        finally:
            H.LlamaStack.fuse_decode_layer, H.Linear.fuse_producer_quant = True, True

    # ---- roofline of the dominant kernel (decode attention), measured live with HIP events
    Hq, Hkv, D = stack.Hq, stack.Hkv, shape.head_dim
    t_attn = time_attention_kernel(stack, runner, backend, fb, a.kernel_reps)
    abytes = attention_bytes(B, S, Hkv, D, Hq, kv_esize=1 if kv8 else 2)
    achieved = abytes / t_attn / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "decode_attn_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("batch") == B and tj.get("seq") == S and tj.get("tp") == tp and not kv8:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    step_bytes = (shape.layers * (abytes + (stack.q_size + 2 * stack.kv_size) * shape.hidden + stack.q_size * shape.hidden
                                  + 3 * stack.inter * shape.hidden) + stack.vocab_shard * shape.hidden * 2)
    kv_splits = backend.forward_metadata.num_kv_splits
    prefill = None
    if a.prefill_batch > 0:
        try:
            prefill = prefill_leg(H, stack, runner, backend, shape, a.prefill_batch, S, dev, world, chunk=a.prefill_chunk)
        except Exception as e:   # the decode number must not be lost to a prefill-side problem
            prefill = {"error": f"{type(e).__name__}: {e}"}
    ragged = None
    if world == 1:
        try:
            ragged = ragged_attention_leg(H, stack, runner, backend, B, S, dev)
        except Exception as e:   # informational only
            ragged = {"error": f"{type(e).__name__}: {e}"}
    result = {
        "metric": "decode tokens/s (Llama-3-8B FP8, batch 128, KV seq 2048)" + (" [variant: fp8 KV cache]" if kv8 else "")
                  + ("" if (a.model, B, S, a.layers) == ("llama3-8b", 128, 2048, 0) else
                     f" [variant: {shape.name}, batch {B}, KV seq {S}" + (f", {a.layers} layers" if a.layers else "") + "]"),
        "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "median_synced_ms": round(median_ms, 4), "tokens_per_s_median_synced": round(B / (median_ms * 1e-3), 1),
        "plugin_surface_only": plugin_only,
        "overlap": overlap,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "fp8_e4m3 x fp8_e4m3 -> f32 (linears), " + ("fp8 e4m3" if kv8 else "bf16") + " KV/f32 softmax (attention)",
        "data": "synthetic",
        "config": {"workload": f"{shape.name} decode step, per-tensor FP8 linears ({a.act_scheme} activation scale), "
                               f"{'fp8 e4m3' if kv8 else 'bf16'} paged KV page_size=1 {'contiguous' if a.contiguous else 'scattered'} slots, "
                               f"batch {B}, KV seq {S}, {shape.layers} layers, TP={tp}",
                   "global_batch": B, "seq_len": S, "parallelism": f"tp{tp}", "hipgraph": graph is not None,
                   "two_batch_overlap": bool(tbo and timed_is_tbo),
                   "all_reduce": None if tp == 1 else (
                       ("native-xgmi fused with add+rmsnorm+fp8 quant" if custom_ar.should_fuse_norm(B, shape.hidden, dtype)
                        else "native-xgmi") if custom_ar is not None else "rccl"),
                   "kv_splits": kv_splits},
        "prefill": prefill,
        "ragged_decode_attention": ragged,
        "step_hbm_roofline_frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": {"kernel": "decode_attn_kernel (+ split merge)", "bound": "hbm", "achieved": round(achieved, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": abytes, "avg_launch_us": round(t_attn * 1e6, 2)},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline({"llama3-8b": H.LLAMA3_8B, "llama3-70b": H.LLAMA3_70B}[a.model], B, S)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
