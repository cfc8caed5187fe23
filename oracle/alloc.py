"""CPU oracle: the paged KV-slot allocator kernels (integer, bit-exact).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the reference's Triton kernels
alloc_extend_kernel / alloc_decode_kernel (python/sglang/srt/mem_cache/allocator.py:278-404) request by
request in plain Python.  Pinned by tests/golden/allocator.pt, produced by running the reference's own torch
forms of the same kernels (alloc_extend_kernel_ascend / alloc_decode_kernel_ascend, allocator.py:545-616);
the one case those torch forms mishandle (an extension that stays inside the old partial page) follows
the Triton kernel's early return (:331-332).
"""
from __future__ import annotations

from typing import Tuple

import torch


def alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, page_size: int) -> Tuple[torch.Tensor, int, int]:
    """-> (out_indices int64 [sum extend], num_new_pages, sum_extend_lens); allocator.py:278-364."""
    ps = int(page_size)
    pre = [int(x) for x in prefix_lens]
    seq = [int(x) for x in seq_lens]
    out = torch.empty(sum(s - p for s, p in zip(seq, pre)), dtype=torch.int64)
    pos = 0      # output_start_loc of request pid
    page = 0     # new_page_start_loc of request pid
    for i, (p, s) in enumerate(zip(pre, seq)):
        ceil_pre = (p + ps - 1) // ps
        new_pages = (s + ps - 1) // ps - ceil_pre
        n1 = min(s, ceil_pre * ps) - p                                   # part 1: fill the old partial page
        for j in range(n1):
            out[pos + j] = int(last_loc[i]) + 1 + j
        if p + n1 != s:
            n2 = s // ps * ps - ceil_pre * ps                            # part 2: new full pages
            for j in range(n2):
                out[pos + n1 + j] = int(free_pages[page + j // ps]) * ps + j % ps
            if p + n1 + n2 != s:
                n3 = s - s // ps * ps                                    # part 3: the new partial page
                start = int(free_pages[page + new_pages - 1]) * ps
                for j in range(n3):
                    out[pos + n1 + n2 + j] = start + j
        pos += s - p
        page += new_pages
    return out, page, pos


def alloc_decode(seq_lens, last_loc, free_pages, page_size: int) -> Tuple[torch.Tensor, int]:
    """-> (out_indices int64 [bs], num_new_pages); allocator.py:367-404."""
    ps = int(page_size)
    out = torch.empty(len(seq_lens), dtype=torch.int64)
    page = 0
    for i, s in enumerate(int(x) for x in seq_lens):
        need = (s + ps - 1) // ps - (s - 1 + ps - 1) // ps
        if need == 0:
            out[i] = int(last_loc[i]) + 1
        else:
            out[i] = int(free_pages[page]) * ps
        page += need
    return out, page
