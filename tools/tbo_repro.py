#!/usr/bin/env python3
"""Repro harness for the two-micro-batch decode step (debugging aid): N steps, optional sync per step."""
import faulthandler
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "60")), exit=True)
from iaas_sglang_amd import harness as H  # noqa: E402
from iaas_sglang_amd.attention_backend import MiAttnBackend  # noqa: E402
from iaas_sglang_amd.quantization import Fp8Config  # noqa: E402
import dataclasses  # noqa: E402

dev = torch.device("cuda", 0)
L = int(os.environ.get("LAYERS", "2"))
B, S = int(os.environ.get("B", "128")), int(os.environ.get("S", "2048"))
shape = dataclasses.replace(H.LLAMA3_8B, layers=L)
runner = H.make_runner(shape, max_reqs=B, ctx=2 * S + 8, pool_tokens=B * S, dtype=torch.bfloat16, device=dev, fill_kv=True)
backend, backend_b = MiAttnBackend(runner), MiAttnBackend(runner)
cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), torch.bfloat16, dev)
fb = H.make_decode_batch(runner, backend, B, S, dev, seed=0)
ids = torch.randint(0, shape.vocab, (B,), device=dev)
backend.init_forward_metadata(fb)
stack.calibrate_static_input_scales(torch.index_select(stack.embed, 0, ids), fb.positions, fb, backend)
halves = H.split_decode_batch(fb, backend_b, B // 2)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
mode = os.environ.get("MODE", "sync")
print("setup done", flush=True)
for it in range(int(os.environ.get("N", "6"))):
    hidden = torch.index_select(stack.embed, 0, ids)
    logits = stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams)
    if mode == "sync" or it == int(os.environ.get("N", "6")) - 1:
        torch.cuda.synchronize()
    print("step", it, "enqueued" if mode != "sync" else "done", float(logits[0, 0]) if mode == "sync" else "", flush=True)
print("all done", flush=True)
