#!/usr/bin/env python3
"""Which kernels (aten ops included, by name) one EXTEND chunk of the bench's prefill leg runs, and for how long:
a torch.profiler table of one 16 x 2048-token chunk through LAYERS layers of the Llama-3-8B stack.  Not a test."""
import dataclasses
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iaas_sglang_amd import harness as H  # noqa: E402
from iaas_sglang_amd.attention_backend import MiAttnBackend  # noqa: E402
from iaas_sglang_amd.quantization import Fp8Config  # noqa: E402

dev = "cuda"
layers = int(os.environ.get("LAYERS", "4"))
nseq, S = int(os.environ.get("NSEQ", "16")), int(os.environ.get("S", "2048"))
shape = dataclasses.replace(H.LLAMA3_8B, layers=layers)
runner = H.make_runner(shape, max_reqs=nseq, ctx=2 * S + 8, pool_tokens=nseq * S, dtype=torch.bfloat16, device=dev, tp=1,
                       fill_kv=False, seed=0, max_kv_splits=8)
backend = MiAttnBackend(runner)
cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), torch.bfloat16, dev)
fb = H.make_extend_batch(runner, backend, [0] * nseq, [S] * nseq, dev, seed=3)
hidden = torch.randn(nseq * S, shape.hidden, device=dev, dtype=torch.float32).to(torch.bfloat16)


def forward():
    backend.init_forward_metadata(fb)
    return stack.forward(hidden, fb.positions, fb, backend, last_token_logits=fb.extend_seq_lens)


forward(); forward()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    forward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
