#!/usr/bin/env python3
"""Generate tests/golden/sched.pt: reference-run vectors for the scheduler-side integer helpers.

Build container only (needs /root/reference).  managers/schedule_batch.py and model_executor/forward_batch_info.py
import most of srt at module level, so the two pure-torch functions are taken out of their files by name (`ast`,
at run time; nothing of the reference is stored here) and executed as they stand:
  * get_last_loc_torch        python/sglang/srt/managers/schedule_batch.py:1900-1909
  * compute_position_torch    python/sglang/srt/model_executor/forward_batch_info.py:734-750
The request-table write has no function of its own in the reference: the non-Triton branch is the inline loop of
ScheduleBatch.prepare_for_extend (schedule_batch.py:1303-1309) over ReqToTokenPool.write (memory_pool.py:77-78,
`self.req_to_token[indices] = values`); the generator replays exactly that loop on a plain tensor.

Usage:  python tests/golden/make_golden_sched.py
"""
import ast
import os
import sys

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/sglang/srt"


def reference_function(path, name):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            node.decorator_list = []
            ns = {"torch": torch}
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def main():
    get_last_loc_torch = reference_function(f"{REF}/managers/schedule_batch.py", "get_last_loc_torch")
    compute_position_torch = reference_function(f"{REF}/model_executor/forward_batch_info.py", "compute_position_torch")
    g = torch.Generator().manual_seed(21)
    out = {}
    for name, bs, ctx, max_reqs, max_ext in [("small", 5, 64, 9, 20), ("one", 1, 700, 3, 600), ("wide", 300, 96, 512, 40),
                                             ("long", 17, 4100, 32, 2500)]:
        rpi = torch.randperm(max_reqs, generator=g)[:bs].to(torch.int64)
        ext = torch.randint(1, max_ext + 1, (bs,), generator=g, dtype=torch.int64)
        pre = torch.minimum(torch.randint(0, ctx, (bs,), generator=g, dtype=torch.int64), ctx - ext)
        pre[0] = 0
        if bs > 3:
            ext[3] = 1
        seq = pre + ext
        r2t = torch.randint(1, 1 << 20, (max_reqs, ctx), generator=g, dtype=torch.int32)
        loc = torch.randperm(1 << 21, generator=g)[: int(ext.sum())].to(torch.int64) + 1
        # --- last_loc of the cached prefixes (before the write)
        last = get_last_loc_torch(r2t, rpi, pre)
        # --- request-table write: schedule_batch.py:1303-1309 over memory_pool.py:77-78
        r2t_after = r2t.clone()
        pt = 0
        for i in range(bs):
            r2t_after[(int(rpi[i]), slice(int(pre[i]), int(seq[i])))] = loc[pt: pt + int(ext[i])].to(torch.int32)
            pt += int(ext[i])
        # --- positions / start locs as ForwardBatch.init_new builds its inputs (int32 lens, :381-397)
        positions, start_loc = compute_position_torch(pre.to(torch.int32), ext.to(torch.int32))
        out[name] = dict(req_to_token=r2t, req_pool_indices=rpi, prefix_lens=pre, seq_lens=seq, extend_lens=ext,
                         out_cache_loc=loc, last_loc=last, req_to_token_after=r2t_after, positions=positions,
                         extend_start_loc=start_loc)
    torch.save(out, os.path.join(HERE, "sched.pt"))
    for k, d in out.items():
        print(k, {n: (tuple(v.shape), str(v.dtype)) for n, v in d.items()})
    print("sched.pt", os.path.getsize(os.path.join(HERE, "sched.pt")), "bytes")


if __name__ == "__main__":
    main()
