"""KV-cache pools and the slot allocator of the hot path (SURVEY section 8a rows `ReqToTokenPool`,
`MHATokenToKVPool`, `TokenToKVPoolAllocator`).

Mirrors python/sglang/srt/mem_cache/memory_pool.py:49-96 (ReqToTokenPool), :176-456
(MHATokenToKVPool) and mem_cache/allocator.py:36-153 (page_size = 1 allocator): same constructor
arguments, attribute names, return conventions (`alloc` returns None when it cannot be served) and the
same integer results -- the allocator hands out slots 1..size in order, slot 0 of every buffer is the
padding sink (memory_pool.py:235, allocator.py:120-124).  These are host-side bookkeeping classes: the
only device work is `set_kv_buffer`, which is our scatter kernel (mi_kv_write / mi_kv_write_fp8) instead
of `index_put`.  Layer-wise transfer counters, host offload, memory-saver regions and the mooncake
allocator of the reference are outside the hot path and not mirrored.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch

from . import ops

FP8_DTYPES = (torch.float8_e4m3fn,)   # gfx950 is OCP fp8; e5m2 KV is not wired (no kernel reads it)


class ReqToTokenPool:
    """memory_pool.py:49-96 -- request row -> token slots, int32 [size, max_context_len]."""

    def __init__(self, size: int, max_context_len: int, device: str, enable_memory_saver: bool = False):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots = list(range(size))

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        select_index = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select_index

    def free(self, free_index: Union[int, List[int]]):
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = list(range(self.size))


class MHATokenToKVPool:
    """memory_pool.py:176-456 -- per layer k_buffer / v_buffer [size + page_size, head_num, head_dim];
    fp8 pools are stored as uint8 and viewed as fp8 by the getters (memory_pool.py:113-117, 389-405)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int, layer_num: int,
                 device: str, enable_memory_saver: bool = False, start_layer: Optional[int] = None,
                 end_layer: Optional[int] = None):
        self.size, self.page_size, self.dtype, self.device = size, page_size, dtype, device
        if dtype in FP8_DTYPES:
            self.store_dtype = torch.uint8
        elif dtype in (torch.bfloat16, torch.float16):
            self.store_dtype = dtype
        else:
            raise NotImplementedError(f"MHATokenToKVPool: KV dtype {dtype} has no kernel (bf16 / fp16 / fp8_e4m3fn)")
        self.layer_num = layer_num
        self.start_layer = start_layer or 0
        self.end_layer = end_layer or layer_num - 1
        self.head_num, self.head_dim = head_num, head_dim
        # slot 0 (the first page) is the sink padded tokens write to
        self.k_buffer = [torch.zeros((size + page_size, head_num, head_dim), dtype=self.store_dtype, device=device)
                         for _ in range(layer_num)]
        self.v_buffer = [torch.zeros((size + page_size, head_num, head_dim), dtype=self.store_dtype, device=device)
                         for _ in range(layer_num)]
        self.token_stride = head_num * head_dim
        self.layer_transfer_counter = None
        k_size, v_size = self.get_kv_size_bytes()
        self.mem_usage = (k_size + v_size) / (1 << 30)

    def get_kv_size_bytes(self):
        k = sum(b.numel() * b.element_size() for b in self.k_buffer)
        v = sum(b.numel() * b.element_size() for b in self.v_buffer)
        return k, v

    def _view(self, buf: torch.Tensor) -> torch.Tensor:
        return buf.view(self.dtype) if self.store_dtype != self.dtype else buf

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self._view(self.k_buffer[layer_id - self.start_layer])

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self._view(self.v_buffer[layer_id - self.start_layer])

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale: Optional[float] = None, v_scale: Optional[float] = None,
                      layer_id_override: Optional[int] = None):
        """k_buffer[layer][loc] = cache_k; v_buffer[layer][loc] = cache_v (memory_pool.py:418-456).  For an
        fp8 pool the rows are divided by k_scale / v_scale (when given) and cast, in one scatter kernel."""
        layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
        kb, vb = self.k_buffer[layer_id - self.start_layer], self.v_buffer[layer_id - self.start_layer]
        if self.dtype in FP8_DTYPES:
            if cache_k.dtype in FP8_DTYPES:
                raise NotImplementedError("MHATokenToKVPool.set_kv_buffer: rows already in fp8 (hand over T-typed rows)")
            ops.kv_write_fp8(kb, vb, loc, cache_k, cache_v, k_scale, v_scale)
        else:
            if cache_k.dtype != self.dtype:
                cache_k, cache_v = cache_k.to(self.dtype), cache_v.to(self.dtype)
            ops.kv_write(kb, vb, loc, cache_k, cache_v)


class TokenToKVPoolAllocator:
    """allocator.py:113-153 (page_size = 1): a free list of int64 slot indices, served from the front."""

    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache: MHATokenToKVPool):
        self.size, self.page_size, self.dtype, self.device = size, 1, dtype, device
        self._kvcache = kvcache
        self.clear()

    def clear(self):
        # the padded slot 0 is used for writing dummy outputs from padded tokens
        self.free_pages = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group: List[torch.Tensor] = []

    def available_size(self) -> int:
        return len(self.free_pages)

    def get_kvcache(self):
        return self._kvcache

    def restore_state(self, free_pages):
        self.free_pages = free_pages

    def backup_state(self):
        return self.free_pages

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.cat(self.free_group))

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > len(self.free_pages):
            return None
        select_index = self.free_pages[:need_size]
        self.free_pages = self.free_pages[need_size:]
        return select_index

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            self.free_pages = torch.cat((self.free_pages, free_index))
        else:
            self.free_group.append(free_index)

    def alloc_extend(self, *args, **kwargs):
        raise NotImplementedError("alloc_extend is only for paged allocator")

    def alloc_decode(self, *args, **kwargs):
        raise NotImplementedError("alloc_decode is only for paged allocator")


class PagedTokenToKVPoolAllocator(TokenToKVPoolAllocator):
    """allocator.py:407-543: page-aligned allocation; `alloc_extend` / `alloc_decode` are our integer kernels
    (mi_alloc_extend / mi_alloc_decode) instead of the reference's Triton kernels."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, device: str, kvcache: MHATokenToKVPool):
        self.num_pages = size // page_size
        super().__init__(size, dtype, device, kvcache)
        self.page_size = page_size
        self.ret_values = torch.empty((1,), dtype=torch.int64, device=self.device)

    def clear(self):
        # the padded slot 0 (page 0) is used for writing dummy outputs from padded tokens
        self.free_pages = torch.arange(1, self.num_pages + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []

    def available_size(self) -> int:
        return len(self.free_pages) * self.page_size

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        num_pages = need_size // self.page_size
        if num_pages > len(self.free_pages):
            return None
        out_pages = self.free_pages[:num_pages]
        self.free_pages = self.free_pages[num_pages:]
        return (out_pages[:, None] * self.page_size + torch.arange(self.page_size, device=self.device)).reshape(-1)

    def alloc_extend(self, prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor,
                     extend_num_tokens: int) -> Optional[torch.Tensor]:
        out_indices = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        ops.alloc_extend(prefix_lens, seq_lens, last_loc, self.free_pages.contiguous(), out_indices, self.ret_values,
                         self.page_size)
        num_new_pages = int(self.ret_values.item()) >> 32           # the reference syncs here too (allocator.py:482)
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def alloc_decode(self, seq_lens: torch.Tensor, last_loc: torch.Tensor) -> Optional[torch.Tensor]:
        bs = len(seq_lens)
        out_indices = torch.empty((bs,), dtype=torch.int64, device=self.device)
        ops.alloc_decode(seq_lens, last_loc, self.free_pages.contiguous(), out_indices, self.ret_values, self.page_size)
        num_new_pages = int(self.ret_values.item())
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            free_page_indices = torch.unique(free_index // self.page_size)
            self.free_pages = torch.cat((free_page_indices, self.free_pages))     # freed pages go to the FRONT (:531-532)
        else:
            self.free_group.append(free_index)
