#!/usr/bin/env python3
"""Decode-step timing of the secondary BASELINE configurations through the same harness as bench.py (not a test, not
the headline): C2 = Llama-3-8B bf16 linears (library GEMM) + our attention, ragged batch 32; C4 = Llama-2-7B AWQ int4
g128, batch 64, fp16.  usage: python tools/config_bench.py c2|c4 [seq]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import harness as H  # noqa: E402
from iaas_sglang_amd._compat import UnquantizedLinearMethod  # noqa: E402
from iaas_sglang_amd.attention_backend import MiAttnBackend  # noqa: E402
from iaas_sglang_amd.quantization import AWQConfig  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda", 0)
if cfg == "c4":
    shape, dtype, B = H.LLAMA2_7B, torch.float16, 64
    awq = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    make = lambda: awq.get_quant_method(None, "")     # noqa: E731
    lens = torch.full((B,), S, dtype=torch.int64)
else:
    shape, dtype, B = H.LLAMA3_8B, torch.bfloat16, 32
    make = UnquantizedLinearMethod
    lens = torch.randint(1, 2 * S + 1, (B,), generator=torch.Generator().manual_seed(0)).to(torch.int64)   # ragged
tot = int(lens.sum())
runner = H.make_runner(shape, max_reqs=B, ctx=2 * S + 8, pool_tokens=tot + 8, dtype=dtype, device=dev, fill_kv=True)
backend = MiAttnBackend(runner)
stack = H.LlamaStack(shape, make, dtype, dev)
if os.environ.get("MI_NO_FUSE") == "1":     # A/B: the unfused plugin-surface sequence
    H.LlamaStack.fuse_decode_layer = False
fb = H.make_decode_batch(runner, backend, B, 0, dev, seed=0, ragged=lens)
ids = torch.randint(0, shape.vocab, (B,), device=dev)
out_ids = torch.empty_like(ids)


def step():
    hidden = torch.index_select(stack.embed, 0, ids)
    backend.init_forward_metadata(fb)
    logits = stack.forward(hidden, fb.positions, fb, backend)
    torch.argmax(logits, dim=-1, out=out_ids)


step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    graph.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
wbytes = sum(p.numel() * p.element_size() for L in stack.layers for lin in (L.qkv, L.o, L.gate_up, L.down)
             for p in lin.parameters()) + stack.lm_head.numel() * 2
kvbytes = 2 * tot * (shape.num_kv_heads * shape.head_dim) * 2 * shape.layers
print(f"{cfg}: {shape.name} B={B} keys={tot}: {ms:.3f} ms/step = {B / ms * 1e3:.0f} tok/s; bytes/step "
      f"{(wbytes + kvbytes) / 1e9:.2f} GB -> {(wbytes + kvbytes) / ms / 1e6:.0f} GB/s "
      f"({(wbytes + kvbytes) / ms / 1e6 / 8000:.1%} of 8 TB/s)", flush=True)
