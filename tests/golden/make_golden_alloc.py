#!/usr/bin/env python3
"""Generate tests/golden/allocator.pt by RUNNING THE REFERENCE'S OWN allocator file on CPU.

python/sglang/srt/mem_cache/allocator.py is imported with the SURVEY 8c recipe (namespace stubs, no reference
__init__.py executes; its two module-level imports from sglang are stubbed).  Executed reference code:
  * TokenToKVPoolAllocator (allocator.py:113-153): a scripted alloc / free / free-group / clear sequence;
  * PagedTokenToKVPoolAllocator.alloc / free / clear (allocator.py:431-449,524-543): torch code;
  * alloc_extend_kernel_ascend / alloc_decode_kernel_ascend (allocator.py:545-616): the reference's own TORCH
    forms of its Triton kernels alloc_extend_kernel / alloc_decode_kernel (:278-404), which need a GPU.
Runs only in the build container (needs /root/reference).  Usage: python tests/golden/make_golden_alloc.py
"""
import importlib
import os
import sys
import types

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = "/root/reference/python"


def load_allocator():
    for pkg in ["sglang", "sglang.srt", "sglang.srt.mem_cache"]:
        m = types.ModuleType(pkg)
        m.__path__ = [f"{ROOT}/{pkg.replace('.', '/')}"]
        sys.modules[pkg] = m
    mp = types.ModuleType("sglang.srt.mem_cache.memory_pool")
    mp.SWAKVPool = type("SWAKVPool", (), {})
    mp.KVCache = type("KVCache", (), {})
    sys.modules[mp.__name__] = mp
    su = types.ModuleType("sglang.srt.utils")
    su.get_bool_env_var = lambda name, default="false": os.getenv(name, default).lower() in ("1", "true")
    su.next_power_of_2 = lambda n: 1 << (n - 1).bit_length() if n > 0 else 1
    sys.modules[su.__name__] = su
    return importlib.import_module("sglang.srt.mem_cache.allocator")


def token_allocator_script(A):
    """The same scripted sequence tests/test_mem_cache_cpu.py replays on our mirror."""
    a = A.TokenToKVPoolAllocator(12, torch.bfloat16, "cpu", None)
    log = []
    x = a.alloc(5); log.append(("alloc5", x.clone()))
    log.append(("alloc9_none", torch.tensor([-1 if a.alloc(9) is None else 0])))
    y = a.alloc(7); log.append(("alloc7", y.clone()))
    a.free(x[1:4]); log.append(("avail", torch.tensor([a.available_size()])))
    log.append(("alloc2", a.alloc(2).clone()))
    a.free_group_begin(); a.free(y[:3]); a.free(x[:1]); a.free_group_end()
    log.append(("alloc4", a.alloc(4).clone()))
    a.clear(); log.append(("alloc12", a.alloc(12).clone()))
    return log


def paged_cases(A):
    g = torch.Generator().manual_seed(0)
    cases = []
    for page_size, bs in [(4, 5), (16, 9), (64, 33), (1, 7), (8, 1), (32, 128)]:
        num_pages = 4096
        a = A.PagedTokenToKVPoolAllocator(num_pages * page_size, page_size, torch.bfloat16, "cpu", None)
        # scramble the free list the way frees do in a live server
        perm = torch.randperm(num_pages, generator=g)
        a.free_pages = a.free_pages[perm]
        free0 = a.free_pages.clone()
        prefix = torch.randint(0, 5 * page_size, (bs,), generator=g, dtype=torch.int64)
        prefix[0] = 0
        ext = torch.randint(1, 7 * page_size, (bs,), generator=g, dtype=torch.int64)
        if bs > 2:
            prefix[2] = 3 * page_size                             # page-aligned prefix
        # the reference's torch form mishandles an extension that stays inside the old partial page (num2 < 0 is
        # truthy at allocator.py:583; the Triton kernel returns early at :331-332): keep those out of the golden
        # set -- they are covered by oracle/alloc.py (a restatement of the Triton kernel) against the HIP kernel
        bad = (prefix + ext) // page_size < (prefix + page_size - 1) // page_size
        prefix = torch.where(bad, prefix // page_size * page_size, prefix)
        seq = prefix + ext
        # last_loc: the slot of the last prefix token; consistent with the page layout (debug assert :459-461)
        last_page = torch.randint(num_pages + 10, num_pages + 5000, (bs,), generator=g, dtype=torch.int64)
        last_loc = torch.where(prefix > 0, last_page * page_size + (prefix - 1) % page_size, torch.full_like(prefix, -1))
        out = torch.full((int(ext.sum()),), -7, dtype=torch.int64)
        nnp = A.alloc_extend_kernel_ascend(prefix, seq, last_loc, a.free_pages, out, page_size, "cpu")
        cases.append({"kind": "extend", "page_size": page_size, "prefix_lens": prefix, "seq_lens": seq, "last_loc": last_loc,
                      "free_pages": free0, "out_indices": out, "num_new_pages": int(nnp.sum())})
        # decode on top: every request grows by one token
        seq_d = seq + 1
        last_d = out[torch.cumsum(ext, 0) - 1]
        out_d = torch.full((bs,), -7, dtype=torch.int64)
        fp = free0[int(nnp.sum()):]
        nnp_d = A.alloc_decode_kernel_ascend(seq_d, last_d, fp, out_d, page_size)
        cases.append({"kind": "decode", "page_size": page_size, "seq_lens": seq_d, "last_loc": last_d, "free_pages": fp.clone(),
                      "out_indices": out_d, "num_new_pages": int(nnp_d.sum())})
    # class-level torch parts: alloc (page aligned), free (unique pages to the FRONT), clear
    a = A.PagedTokenToKVPoolAllocator(64, 4, torch.bfloat16, "cpu", None)
    log = []
    x = a.alloc(12); log.append(("alloc12", x.clone()))
    log.append(("alloc_too_many", torch.tensor([-1 if a.alloc(64) is None else 0])))
    a.free(x[2:9]); log.append(("free_pages_after_free", a.free_pages.clone()))
    log.append(("alloc8", a.alloc(8).clone()))
    a.clear(); log.append(("free_pages_after_clear", a.free_pages.clone()))
    return cases, log


def main():
    A = load_allocator()
    cases, plog = paged_cases(A)
    torch.save({"token_allocator_log": token_allocator_script(A), "paged_cases": cases, "paged_log": plog},
               os.path.join(HERE, "allocator.pt"))
    print("wrote allocator.pt:", len(cases), "kernel cases")


if __name__ == "__main__":
    main()
