"""Thin torch-tensor -> C-ABI wrappers.  PyTorch is only plumbing here (device memory,
the current HIP stream); every computation happens in libmi_hotpath.so.

All wrappers enqueue on ``torch.cuda.current_stream()`` and never synchronise, so they
can be captured into a hipGraph (torch.cuda.graph) as long as the caller preallocates.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from ._lib import (MI_BF16, MI_FP16, MI_SCALE_ROW, MI_SCALE_TENSOR, MI_W4_AWQ, MI_W4_GPTQ,
                   MiHotpathError, check, lib)

FP8_DTYPE = torch.float8_e4m3fn  # gfx950 = OCP e4m3fn
_DT = {torch.bfloat16: MI_BF16, torch.float16: MI_FP16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MiHotpathError(f"unsupported dtype {t.dtype} (bf16 / fp16 only)") from None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise MiHotpathError("hot-path ops need device tensors (no CPU path exists)")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def cu_count() -> int:
    return lib.mi_device_cu_count()


# ------------------------------------------------------------------ integer path
def kv_indptr(lens: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 [B+1] exclusive-scan of lens (int32 or int64)."""
    B = lens.shape[0]
    assert lens.dtype in (torch.int32, torch.int64) and lens.is_contiguous()
    if out is None:
        out = torch.empty(B + 1, dtype=torch.int32, device=lens.device)
    assert out.dtype == torch.int32 and out.numel() >= B + 1
    check(lib.mi_kv_indptr(_ptr(lens), int(lens.dtype == torch.int64), _ptr(out), B, _stream()), "mi_kv_indptr")
    return out[: B + 1]


def kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, lens: torch.Tensor,
               kv_indptr_t: torch.Tensor, out: torch.Tensor,
               kv_start_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gather req_to_token row slices into the flat kv_indices (int32), bit-exact."""
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and req_pool_indices.is_contiguous()
    assert lens.dtype in (torch.int32, torch.int64) and lens.is_contiguous()
    assert kv_indptr_t.dtype == torch.int32 and out.dtype == torch.int32
    if kv_start_idx is not None:
        assert kv_start_idx.dtype == torch.int32
    B = req_pool_indices.shape[0]
    check(lib.mi_kv_indices(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _ptr(lens),
                            int(lens.dtype == torch.int64), _ptr(kv_indptr_t), _ptr(kv_start_idx), _ptr(out),
                            B, _stream()), "mi_kv_indices")
    return out


def kv_page_tables(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, lens: torch.Tensor, page_size: int,
                   page_indptr: Optional[torch.Tensor] = None, page_indices: Optional[torch.Tensor] = None):
    """(page_indptr int32 [B+1], page_indices int32): one page id per page of every request, for decode_attention's
    page-granular form (page-aligned allocation, page_size a power of two).  `page_indices` needs room for
    sum(ceil(len / page_size)) entries (<= B * ceil(max_len / page_size))."""
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and req_pool_indices.is_contiguous()
    assert lens.dtype in (torch.int32, torch.int64) and lens.is_contiguous()
    assert page_size >= 1 and page_size & (page_size - 1) == 0
    B = req_pool_indices.shape[0]
    if page_indptr is None:
        page_indptr = torch.empty(B + 1, dtype=torch.int32, device=lens.device)
    check(lib.mi_kv_page_indptr(_ptr(lens), int(lens.dtype == torch.int64), int(page_size), _ptr(page_indptr), B,
                                _stream()), "mi_kv_page_indptr")
    if page_indices is None:
        page_indices = torch.empty(max(B * (-(-req_to_token.shape[1] // page_size)), 1), dtype=torch.int32,
                                   device=lens.device)
    assert page_indices.dtype == torch.int32 and page_indptr.dtype == torch.int32
    check(lib.mi_kv_page_indices(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _ptr(lens),
                                 int(lens.dtype == torch.int64), _ptr(page_indptr), _ptr(page_indices), B,
                                 int(page_size), _stream()), "mi_kv_page_indices")
    return page_indptr[: B + 1], page_indices


def kv_write(k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor, k: torch.Tensor,
             v: torch.Tensor) -> None:
    """k_cache[loc] = k ; v_cache[loc] = v   (set_kv_buffer)."""
    assert loc.dtype == torch.int64 and loc.is_contiguous()
    T = loc.shape[0]
    k = k.reshape(T, -1)
    v = v.reshape(T, -1)
    assert k.dtype == k_cache.dtype and v.dtype == v_cache.dtype and k.stride(1) == 1 and v.stride(1) == 1
    row_k, row_v = k.shape[1], v.shape[1]
    assert k_cache[0].numel() == row_k and v_cache[0].numel() == row_v
    assert k_cache[0].is_contiguous() and v_cache[0].is_contiguous()
    check(lib.mi_kv_write(_ptr(k_cache), _ptr(v_cache), _ptr(loc), _ptr(k), _ptr(v), T, row_k, row_v,
                          k_cache.stride(0), v_cache.stride(0), k.stride(0), v.stride(0), _dt(k_cache),
                          _stream()), "mi_kv_write")


def kv_write_fp8(k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                 k_scale: Optional[float] = None, v_scale: Optional[float] = None) -> None:
    """fp8 pool (uint8 or float8_e4m3fn storage): cache[loc] = fp8(k / k_scale), fp8(v / v_scale)."""
    assert loc.dtype == torch.int64 and loc.is_contiguous()
    T = loc.shape[0]
    k = k.reshape(T, -1)
    v = v.reshape(T, -1)
    assert k.dtype == v.dtype and k.stride(1) == 1 and v.stride(1) == 1
    assert k_cache.element_size() == 1 and v_cache.element_size() == 1
    assert k_cache[0].numel() == k.shape[1] and v_cache[0].numel() == v.shape[1]
    assert k_cache[0].is_contiguous() and v_cache[0].is_contiguous()
    check(lib.mi_kv_write_fp8(_ptr(k_cache), _ptr(v_cache), _ptr(loc), _ptr(k), _ptr(v), T, k.shape[1], v.shape[1],
                              k_cache.stride(0), v_cache.stride(0), k.stride(0), v.stride(0),
                              float(1.0 if k_scale is None else k_scale), float(1.0 if v_scale is None else v_scale),
                              _dt(k), _stream()), "mi_kv_write_fp8")


def alloc_extend(prefix_lens: torch.Tensor, seq_lens: torch.Tensor, last_loc: torch.Tensor, free_pages: torch.Tensor,
                 out_indices: torch.Tensor, ret_value: torch.Tensor, page_size: int) -> None:
    """Paged slot allocation for an extend batch (all int64 device tensors); ret_value [1] = pages<<32 | tokens."""
    for t in (prefix_lens, seq_lens, last_loc, free_pages, out_indices, ret_value):
        assert t.dtype == torch.int64 and t.is_contiguous()
    bs = prefix_lens.shape[0]
    scratch = torch.empty(2 * max(bs, 1), dtype=torch.int64, device=prefix_lens.device)
    check(lib.mi_alloc_extend(_ptr(prefix_lens), _ptr(seq_lens), _ptr(last_loc), _ptr(free_pages), _ptr(out_indices),
                              _ptr(ret_value), _ptr(scratch), bs, int(page_size), _stream()), "mi_alloc_extend")


def alloc_decode(seq_lens: torch.Tensor, last_loc: torch.Tensor, free_pages: torch.Tensor, out_indices: torch.Tensor,
                 ret_value: torch.Tensor, page_size: int) -> None:
    for t in (seq_lens, last_loc, free_pages, out_indices, ret_value):
        assert t.dtype == torch.int64 and t.is_contiguous()
    bs = seq_lens.shape[0]
    scratch = torch.empty(2 * max(bs, 1), dtype=torch.int64, device=seq_lens.device)
    check(lib.mi_alloc_decode(_ptr(seq_lens), _ptr(last_loc), _ptr(free_pages), _ptr(out_indices), _ptr(ret_value),
                              _ptr(scratch), bs, int(page_size), _stream()), "mi_alloc_decode")


def write_req_to_token(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, pre_lens: torch.Tensor,
                       seq_lens: torch.Tensor, extend_lens: torch.Tensor, out_cache_loc: torch.Tensor) -> None:
    """req_to_token[req_pool_indices[i], pre_lens[i]:seq_lens[i]] = the request's slice of out_cache_loc
    (write_req_to_token_pool_triton, schedule_batch.py:1848-1882)."""
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    for t in (req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc):
        assert t.dtype == torch.int64 and t.is_contiguous() and t.is_cuda
    bs = req_pool_indices.shape[0]
    assert pre_lens.numel() == bs and seq_lens.numel() == bs and extend_lens.numel() == bs
    check(lib.mi_write_req_to_token(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _ptr(pre_lens),
                                    _ptr(seq_lens), _ptr(extend_lens), _ptr(out_cache_loc), bs, _stream()),
          "mi_write_req_to_token")


def get_last_loc(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, prefix_lens: torch.Tensor) -> torch.Tensor:
    """Slot of the last cached token of every request, -1 without a prefix (get_last_loc, schedule_batch.py:1885-1956)."""
    assert req_to_token.dtype == torch.int32 and req_to_token.stride(1) == 1
    assert req_pool_indices.dtype == torch.int64 and prefix_lens.dtype == torch.int64
    assert req_pool_indices.is_contiguous() and prefix_lens.is_contiguous()
    out = torch.empty_like(prefix_lens)
    check(lib.mi_get_last_loc(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _ptr(prefix_lens),
                              _ptr(out), prefix_lens.shape[0], _stream()), "mi_get_last_loc")
    return out


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor, extend_seq_lens_sum: int):
    """(positions int64 [sum], extend_start_loc int32 [bs]) of an extend batch
    (compute_position_triton, forward_batch_info.py:678-732; an empty extend_prefix_lens means no prefixes)."""
    bs = extend_seq_lens.shape[0]
    assert extend_seq_lens.dtype in (torch.int32, torch.int64) and extend_seq_lens.is_contiguous()
    has_prefix = extend_prefix_lens.shape[0] == bs
    if has_prefix:
        assert extend_prefix_lens.dtype == extend_seq_lens.dtype and extend_prefix_lens.is_contiguous()
    positions = torch.empty(int(extend_seq_lens_sum), dtype=torch.int64, device=extend_seq_lens.device)
    start_loc = torch.empty(bs, dtype=torch.int32, device=extend_seq_lens.device)
    check(lib.mi_compute_position(_ptr(extend_prefix_lens) if has_prefix else None, _ptr(extend_seq_lens),
                                  int(extend_seq_lens.dtype == torch.int64), _ptr(positions), _ptr(start_loc), bs,
                                  _stream()), "mi_compute_position")
    return positions, start_loc


# --------------------------------------------------------------------- attention
def decode_workspace_numel(batch: int, num_q_heads: int, v_head_dim: int, num_splits: int) -> int:
    return lib.mi_decode_attn_workspace_bytes(batch, num_q_heads, v_head_dim, num_splits) // 4


def _work_args(work, plan: Optional[torch.Tensor] = None):
    """(pointer, count, plan pointer) of an optional decode work list: int32 [n, 2] = (request, split), contiguous, on
    the device; `work` may be a (list, plan) pair: plan = device int32 {num_work, num_splits, split_chunk}."""
    if isinstance(work, tuple):
        work, plan = work
    if work is None:
        return None, 0, None
    assert work.dtype == torch.int32 and work.dim() == 2 and work.shape[1] == 2 and work.is_contiguous() and work.is_cuda
    if plan is not None:
        assert plan.dtype == torch.int32 and plan.numel() >= 3 and plan.is_cuda and plan.is_contiguous()
    return _ptr(work), work.shape[0], _ptr(plan)


def decode_attention(q: torch.Tensor, k_buf: torch.Tensor, v_buf: torch.Tensor, o: torch.Tensor,
                     kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, sm_scale: float,
                     logit_cap: float = 0.0, num_splits: int = 1,
                     workspace: Optional[torch.Tensor] = None, split_chunk: int = 0,
                     work: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,o [B,Hq,D]; k_buf,v_buf [slots,Hkv,D]; split-KV token attention (split_chunk: keys per split, 0 = S/splits)."""
    B, Hq, D = q.shape
    Hkv = k_buf.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == D and o.stride(2) == 1 and o.stride(1) == D
    assert k_buf.stride(2) == 1 and k_buf.stride(1) == D and v_buf.stride(2) == 1 and v_buf.stride(1) == D
    assert k_buf.dtype == q.dtype and v_buf.dtype == q.dtype and o.dtype == q.dtype
    assert kv_indptr_t.dtype == torch.int32 and kv_indices_t.dtype == torch.int32
    if num_splits > 1:
        need = decode_workspace_numel(B, Hq, D, num_splits)
        assert workspace is not None and workspace.dtype == torch.float32 and workspace.numel() >= need
    check(lib.mi_decode_attn(_ptr(q), _ptr(k_buf), _ptr(v_buf), _ptr(o), _ptr(kv_indptr_t), _ptr(kv_indices_t),
                             _ptr(workspace) if num_splits > 1 else None, B, Hq, Hkv, D, q.stride(0), o.stride(0),
                             k_buf.stride(0), v_buf.stride(0), float(sm_scale), float(logit_cap),
                             int(num_splits), int(split_chunk), *_work_args(work), _dt(q), _stream()), "mi_decode_attn")
    return o


def decode_attention_paged(q: torch.Tensor, k_buf: torch.Tensor, v_buf: torch.Tensor, kv_indptr_t: torch.Tensor,
                           page_indptr: torch.Tensor, page_indices: torch.Tensor, page_size: int, sm_scale: float,
                           logit_cap: float = 0.0, num_splits: int = 1, workspace: Optional[torch.Tensor] = None,
                           o: Optional[torch.Tensor] = None, o_fp8: Optional[torch.Tensor] = None,
                           o_scale: Optional[torch.Tensor] = None, k_scale: float = 1.0, v_scale: float = 1.0,
                           split_chunk: int = 0, work=None):
    """Decode attention with one index per PAGE (page-aligned pool, page_size = 2^k): bf16 / fp16 or fp8 pool, optional
    fp8 copy of the output -- the same kernel, arithmetic and bits as decode_attention / _fp8out / _fp8kv."""
    B, Hq, D = q.shape
    Hkv = k_buf.shape[1]
    kv8 = k_buf.element_size() == 1
    assert q.stride(2) == 1 and q.stride(1) == D
    assert k_buf.stride(2) == 1 and k_buf.stride(1) == D and v_buf.stride(2) == 1 and v_buf.stride(1) == D
    assert kv8 or (k_buf.dtype == q.dtype and v_buf.dtype == q.dtype)
    assert kv_indptr_t.dtype == torch.int32 and page_indptr.dtype == torch.int32 and page_indices.dtype == torch.int32
    assert o is not None or o_fp8 is not None
    if o is not None:
        assert o.dtype == q.dtype and o.stride(2) == 1 and o.stride(1) == D
    if o_fp8 is not None:
        assert o_fp8.dtype == FP8_DTYPE and o_fp8.is_contiguous() and o_scale is not None and o_scale.dtype == torch.float32
    if num_splits > 1:
        need = decode_workspace_numel(B, Hq, D, num_splits)
        assert workspace is not None and workspace.dtype == torch.float32 and workspace.numel() >= need
    check(lib.mi_decode_attn_paged(_ptr(q), _ptr(k_buf), _ptr(v_buf), _ptr(o), _ptr(o_fp8), _ptr(o_scale), int(kv8),
                                   float(k_scale), float(v_scale), _ptr(kv_indptr_t), _ptr(page_indptr),
                                   _ptr(page_indices), int(page_size), _ptr(workspace) if num_splits > 1 else None, B,
                                   Hq, Hkv, D, q.stride(0), o.stride(0) if o is not None else Hq * D, k_buf.stride(0),
                                   v_buf.stride(0), float(sm_scale), float(logit_cap), int(num_splits),
                                   int(split_chunk), *_work_args(work), _dt(q), _stream()), "mi_decode_attn_paged")
    return o if o is not None else o_fp8


def extend_attention(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o: torch.Tensor,
                     k_buf: torch.Tensor, v_buf: torch.Tensor, qo_indptr: torch.Tensor,
                     kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, max_extend_len: int,
                     sm_scale: float, logit_cap: float = 0.0, causal: bool = True,
                     sliding_window: int = -1) -> torch.Tensor:
    """q,o [E,Hq,D]; k_ext,v_ext [E,Hkv,D]; prefix read from k_buf/v_buf through kv_indices."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, o, k_ext, v_ext, k_buf, v_buf):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    assert qo_indptr.dtype == torch.int32 and kv_indptr_t.dtype == torch.int32
    assert kv_indices_t.dtype == torch.int32
    check(lib.mi_extend_attn(_ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o), _ptr(k_buf), _ptr(v_buf),
                             _ptr(qo_indptr), _ptr(kv_indptr_t), _ptr(kv_indices_t), B, int(max_extend_len),
                             Hq, Hkv, D, q.stride(0), o.stride(0), k_ext.stride(0), v_ext.stride(0),
                             k_buf.stride(0), v_buf.stride(0), float(sm_scale), float(logit_cap), int(causal),
                             int(sliding_window), _dt(q), _stream()), "mi_extend_attn")
    return o


def extend_attention_paged(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o: torch.Tensor,
                           k_buf: torch.Tensor, v_buf: torch.Tensor, qo_indptr: torch.Tensor,
                           kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, page_indptr: torch.Tensor,
                           page_indices: torch.Tensor, page_size: int, max_extend_len: int, sm_scale: float,
                           logit_cap: float = 0.0, causal: bool = True, sliding_window: int = -1) -> torch.Tensor:
    """extend_attention on a page-aligned pool: one page id per page of the cached prefix (kv_page_tables on the
    prefix lengths) for the long-extend kernel, kv_indices for every shape that kernel does not take; same bits."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, o, k_ext, v_ext, k_buf, v_buf):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    for t in (qo_indptr, kv_indptr_t, kv_indices_t, page_indptr, page_indices):
        assert t.dtype == torch.int32
    check(lib.mi_extend_attn_paged(_ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o), _ptr(k_buf), _ptr(v_buf), _ptr(qo_indptr),
                                   _ptr(kv_indptr_t), _ptr(kv_indices_t), _ptr(page_indptr), _ptr(page_indices),
                                   int(page_size), B, int(max_extend_len), Hq, Hkv, D, q.stride(0), o.stride(0),
                                   k_ext.stride(0), v_ext.stride(0), k_buf.stride(0), v_buf.stride(0), float(sm_scale),
                                   float(logit_cap), int(causal), int(sliding_window), _dt(q), _stream()),
          "mi_extend_attn_paged")
    return o


def extend_attention_fp8out(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o_fp8: torch.Tensor,
                            o_scale: torch.Tensor, k_buf: torch.Tensor, v_buf: torch.Tensor, qo_indptr: torch.Tensor,
                            kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, max_extend_len: int, sm_scale: float,
                            logit_cap: float = 0.0, causal: bool = True, sliding_window: int = -1,
                            page_indptr: Optional[torch.Tensor] = None, page_indices: Optional[torch.Tensor] = None,
                            page_size: int = 1, o: Optional[torch.Tensor] = None,
                            q_positions: Optional[torch.Tensor] = None,
                            cos_sin_cache_t: Optional[torch.Tensor] = None) -> torch.Tensor:
    """extend_attention (token- or page-granular prefix) returning the output quantised for the following FP8 linear:
    o_fp8 [E, Hq * D] e4m3fn = fp8_quant_per_tensor(static, o_scale) of the T-typed result, bit for bit.  `o`
    (optional [E, Hq, D]) also receives the T-typed result; it is REQUIRED for the shapes the long-extend kernel does
    not take (extend_fp8_out_is_fused says which), where the quantisation is a second launch.  q_positions [E] int64 +
    cos_sin_cache_t (rotary cache cast to q.dtype): q is UNROTATED and the kernel applies NeoX RoPE to Q as it loads it
    (fused shapes only); the caller rotates only k (rope_neox_k_)."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, k_ext, v_ext, k_buf, v_buf) + ((o,) if o is not None else ()):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    assert o_fp8.element_size() == 1 and o_fp8.is_contiguous() and o_fp8.numel() == E * Hq * D
    assert o_scale.dtype == torch.float32 and o_scale.numel() >= 1
    for t in (qo_indptr, kv_indptr_t, kv_indices_t):
        assert t.dtype == torch.int32
    assert (page_indptr is None) == (page_indices is None)
    assert (q_positions is None) == (cos_sin_cache_t is None)
    if q_positions is not None:
        assert q_positions.dtype == torch.int64 and q_positions.numel() == E
        assert cos_sin_cache_t.dtype == q.dtype and cos_sin_cache_t.shape[1] == D and cos_sin_cache_t.is_contiguous()
    check(lib.mi_extend_attn_fp8out(_ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o) if o is not None else None, _ptr(o_fp8),
                                    _ptr(o_scale), _ptr(k_buf), _ptr(v_buf), _ptr(qo_indptr), _ptr(kv_indptr_t),
                                    _ptr(kv_indices_t), _ptr(page_indptr) if page_indptr is not None else None,
                                    _ptr(page_indices) if page_indices is not None else None, int(page_size), B, E,
                                    int(max_extend_len), Hq, Hkv, D, q.stride(0), o.stride(0) if o is not None else Hq * D,
                                    k_ext.stride(0), v_ext.stride(0), k_buf.stride(0), v_buf.stride(0), float(sm_scale),
                                    float(logit_cap), int(causal), int(sliding_window), _ptr(q_positions),
                                    _ptr(cos_sin_cache_t), _dt(q), _stream()),
          "mi_extend_attn_fp8out")
    return o_fp8


def extend_fp8_out_is_fused(head_dim: int, max_extend_len: int, logit_cap: float, sliding_window: int) -> bool:
    """True when mi_extend_attn_fp8out writes the fp8 output from the attention epilogue (no T-typed buffer needed)."""
    return head_dim == 128 and max_extend_len >= 64 and not (logit_cap > 0.0) and sliding_window <= 0


def extend_attention_fp8kv(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o: torch.Tensor,
                           k_buf8: torch.Tensor, v_buf8: torch.Tensor, k_scale: float, v_scale: float,
                           qo_indptr: torch.Tensor, kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor,
                           max_extend_len: int, sm_scale: float, logit_cap: float = 0.0, causal: bool = True,
                           sliding_window: int = -1) -> torch.Tensor:
    """extend_attention with the cached prefix in an fp8 (e4m3fn / uint8 storage) pool [slots, Hkv, 128]."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, o, k_ext, v_ext):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    for t in (k_buf8, v_buf8):
        assert t.element_size() == 1 and t.stride(2) == 1 and t.stride(1) == D and t.shape[2] == D
    assert qo_indptr.dtype == torch.int32 and kv_indptr_t.dtype == torch.int32 and kv_indices_t.dtype == torch.int32
    check(lib.mi_extend_attn_fp8kv(_ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o), _ptr(k_buf8), _ptr(v_buf8), float(k_scale),
                                   float(v_scale), _ptr(qo_indptr), _ptr(kv_indptr_t), _ptr(kv_indices_t), B,
                                   int(max_extend_len), Hq, Hkv, D, q.stride(0), o.stride(0), k_ext.stride(0),
                                   v_ext.stride(0), k_buf8.stride(0), v_buf8.stride(0), float(sm_scale),
                                   float(logit_cap), int(causal), int(sliding_window), _dt(q), _stream()),
          "mi_extend_attn_fp8kv")
    return o


def extend_attention_masked(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o: torch.Tensor,
                            k_buf: torch.Tensor, v_buf: torch.Tensor, qo_indptr: torch.Tensor,
                            kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, custom_mask: torch.Tensor,
                            mask_indptr: torch.Tensor, max_extend_len: int, sm_scale: float, logit_cap: float = 0.0,
                            skip_prefix_custom_mask: bool = True, sliding_window: int = -1) -> torch.Tensor:
    """extend_attention with a tree mask (bool / uint8, flat; request i at mask_indptr[i], [ext_i, pre_i + ext_i])."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, o, k_ext, v_ext, k_buf, v_buf):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    assert qo_indptr.dtype == torch.int32 and kv_indptr_t.dtype == torch.int32 and kv_indices_t.dtype == torch.int32
    assert custom_mask.dtype in (torch.bool, torch.uint8) and custom_mask.is_contiguous()
    assert mask_indptr.dtype == torch.int64 and mask_indptr.numel() >= B + 1
    check(lib.mi_extend_attn_masked(_ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o), _ptr(k_buf), _ptr(v_buf), _ptr(qo_indptr),
                                    _ptr(kv_indptr_t), _ptr(kv_indices_t), _ptr(custom_mask), _ptr(mask_indptr),
                                    int(skip_prefix_custom_mask), B, int(max_extend_len), Hq, Hkv, D, q.stride(0),
                                    o.stride(0), k_ext.stride(0), v_ext.stride(0), k_buf.stride(0), v_buf.stride(0),
                                    float(sm_scale), float(logit_cap), int(sliding_window), _dt(q), _stream()),
          "mi_extend_attn_masked")
    return o


def extend_attention_splitkv(q: torch.Tensor, k_ext: torch.Tensor, v_ext: torch.Tensor, o: torch.Tensor,
                             k_buf: torch.Tensor, v_buf: torch.Tensor, qo_indptr: torch.Tensor,
                             kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor, max_extend_len: int,
                             sm_scale: float, num_splits: int, workspace: Optional[torch.Tensor] = None,
                             logit_cap: float = 0.0, causal: bool = True, sliding_window: int = -1,
                             k_scale: float = 1.0, v_scale: float = 1.0, custom_mask: Optional[torch.Tensor] = None,
                             mask_indptr: Optional[torch.Tensor] = None,
                             skip_prefix_custom_mask: bool = True) -> torch.Tensor:
    """The general extend call (bf16/fp16 or fp8 pool, optional tree mask) with the key range of every query block
    split `num_splits` ways and merged -- for short extends over long prefixes (speculative verify, chunk tails), where
    the unsplit launch has a handful of workgroups.  workspace: decode_workspace_numel(E, Hq, D, num_splits) floats."""
    E, Hq, D = q.shape
    Hkv = k_ext.shape[1]
    B = qo_indptr.shape[0] - 1
    for t in (q, o, k_ext, v_ext):
        assert t.stride(2) == 1 and t.stride(1) == D and t.dtype == q.dtype
    kv8 = k_buf.element_size() == 1
    for t in (k_buf, v_buf):
        assert t.stride(2) == 1 and t.stride(1) == D and (kv8 or t.dtype == q.dtype)
    assert qo_indptr.dtype == torch.int32 and kv_indptr_t.dtype == torch.int32 and kv_indices_t.dtype == torch.int32
    if custom_mask is not None:
        assert custom_mask.dtype in (torch.bool, torch.uint8) and custom_mask.is_contiguous()
        assert mask_indptr is not None and mask_indptr.dtype == torch.int64 and mask_indptr.numel() >= B + 1
    num_splits = int(num_splits)
    if num_splits > 1:
        need = decode_workspace_numel(E, Hq, D, num_splits)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.float32, device=q.device)
        assert workspace.dtype == torch.float32 and workspace.numel() >= need
    check(lib.mi_extend_attn_splitkv(
        _ptr(q), _ptr(k_ext), _ptr(v_ext), _ptr(o), _ptr(k_buf), _ptr(v_buf), int(kv8), float(k_scale), float(v_scale),
        _ptr(qo_indptr), _ptr(kv_indptr_t), _ptr(kv_indices_t), _ptr(custom_mask), _ptr(mask_indptr),
        int(skip_prefix_custom_mask), B, int(max_extend_len), Hq, Hkv, D, q.stride(0), o.stride(0), k_ext.stride(0),
        v_ext.stride(0), k_buf.stride(0), v_buf.stride(0), float(sm_scale), float(logit_cap), int(causal),
        int(sliding_window), _ptr(workspace) if num_splits > 1 else None, E, num_splits, _dt(q), _stream()),
        "mi_extend_attn_splitkv")
    return o


def merge_state(o_a, lse_a, o_b, lse_b, out=None, out_lse=None) -> Tuple[torch.Tensor, torch.Tensor]:
    n, h, d = o_a.shape
    assert o_a.is_contiguous() and o_b.is_contiguous() and lse_a.is_contiguous() and lse_b.is_contiguous()
    assert lse_a.dtype == torch.float32 and lse_b.dtype == torch.float32
    out = torch.empty_like(o_a) if out is None else out
    out_lse = torch.empty_like(lse_a) if out_lse is None else out_lse
    check(lib.mi_merge_state(_ptr(o_a), _ptr(lse_a), _ptr(o_b), _ptr(lse_b), _ptr(out), _ptr(out_lse), n, h, d,
                             _dt(o_a), _stream()), "mi_merge_state")
    return out, out_lse


# --------------------------------------------------------------------------- FP8
def fp8_quant_per_tensor(x: torch.Tensor, scale: Optional[torch.Tensor] = None,
                         out: Optional[torch.Tensor] = None, weight_mode: bool = False
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """scale None -> dynamic (scale = absmax/448 written to a new fp32 [1]); else static.
    weight_mode: the reference's `input_to_float8` arithmetic (multiplier 448/amax, returns 1/multiplier)."""
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    assert not (weight_mode and scale is not None)
    is_static = 2 if weight_mode else int(scale is not None)
    if scale is None:
        scale = torch.empty(1, dtype=torch.float32, device=x.device)
    assert scale.dtype == torch.float32 and scale.numel() == 1
    out = torch.empty(M, K, dtype=FP8_DTYPE, device=x.device) if out is None else out
    check(lib.mi_fp8_quant_per_tensor(_ptr(x), _ptr(out), _ptr(scale), M, K, x.stride(0), is_static, _dt(x),
                                      _stream()), "mi_fp8_quant_per_tensor")
    return out, scale


def fp8_quant_per_token(x: torch.Tensor, out: Optional[torch.Tensor] = None,
                        scales: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    out = torch.empty(M, K, dtype=FP8_DTYPE, device=x.device) if out is None else out
    scales = torch.empty(M, 1, dtype=torch.float32, device=x.device) if scales is None else scales
    check(lib.mi_fp8_quant_per_token(_ptr(x), _ptr(out), _ptr(scales), M, K, x.stride(0), _dt(x), _stream()),
          "mi_fp8_quant_per_token")
    return out, scales


_GEMM_WS = {}
_GEMM_WS_RETIRED = []      # outgrown buffers stay alive: a captured hipGraph may have their address baked in


def _gemm_workspace(nbytes: int, device) -> torch.Tensor:
    """Split-K scratch, one growing buffer per device and stream; stream-ordered use (calls on one stream are
    serialised, so consecutive GEMMs may share it).  A buffer that is outgrown is retired, never freed: graphs captured at start-up
    (decode GEMMs, the fused forms, w4a16) keep writing their slabs to the address they were captured with, and a
    later eager prefill GEMM that needs more scratch must not hand that memory back to the caching allocator under
    them.  Growth is geometric, so the retired buffers add up to less than the live one."""
    # per (device, stream): two streams running GEMMs side by side (two-micro-batch overlap) must not share slabs
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _GEMM_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _GEMM_WS_RETIRED.append(buf)
            nbytes = max(nbytes, 2 * buf.numel())
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _GEMM_WS[key] = buf
    return buf


def fp8_gemm(a: torch.Tensor, b_kn: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
             out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a fp8 [M,K]; b_kn fp8 [K,N] *column-major* (the `.t()` view of [N,K] storage, as the
    reference stores it); scale_a 1 or M values; scale_b 1 or N values."""
    M, K = a.shape
    K2, N = b_kn.shape
    assert K == K2 and a.dtype == FP8_DTYPE and b_kn.dtype == FP8_DTYPE
    assert a.stride(1) == 1 and b_kn.stride(0) == 1, "b must be the [K,N] view of [N,K] row-major storage"
    sa_mode = MI_SCALE_TENSOR if scale_a.numel() == 1 else MI_SCALE_ROW
    sb_mode = MI_SCALE_TENSOR if scale_b.numel() == 1 else MI_SCALE_ROW
    assert sa_mode == MI_SCALE_TENSOR or scale_a.numel() == M
    assert sb_mode == MI_SCALE_TENSOR or scale_b.numel() == N
    assert scale_a.dtype == torch.float32 and scale_b.dtype == torch.float32
    out = torch.empty(M, N, dtype=out_dtype, device=a.device) if out is None else out
    if bias is not None:
        assert bias.dtype == out_dtype and bias.numel() == N and bias.is_contiguous()
    ws_bytes = lib.mi_fp8_gemm_workspace_bytes(M, N, K)
    ws = _gemm_workspace(ws_bytes, a.device) if ws_bytes else None
    check(lib.mi_fp8_gemm(_ptr(a), _ptr(b_kn), _ptr(scale_a), _ptr(scale_b), _ptr(bias), _ptr(out), M, N, K,
                          a.stride(0), b_kn.stride(1), out.stride(0), sa_mode, sb_mode, _DT[out_dtype],
                          _ptr(ws), ws_bytes if ws is not None else 0, _stream()), "mi_fp8_gemm")
    return out


# -------------------------------------------------------------------------- int4
def w4_repack(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group_size: int,
              layout: int, g_idx: Optional[torch.Tensor] = None):
    """Checkpoint-layout int4 weights -> (qw_native, zs_native, perm) for w4a16_gemm.
    For GPTQ act-order (g_idx not sorted) perm = argsort(g_idx) (host-side, once at load)."""
    N = scales.shape[1]
    K = qweight.shape[0] if layout == MI_W4_AWQ else qweight.shape[0] * 8
    assert qweight.dtype == torch.int32 and qzeros.dtype == torch.int32
    assert qweight.is_contiguous() and qzeros.is_contiguous() and scales.is_contiguous()
    perm = None
    if g_idx is not None:
        gi = g_idx.to(torch.int64)
        if not bool((gi[1:] >= gi[:-1]).all()):
            perm = torch.argsort(gi, stable=True).to(torch.int32).contiguous()
    qw = torch.empty(N * K // 8, dtype=torch.int32, device=qweight.device)
    zs = torch.empty(K // group_size, N, dtype=torch.int32, device=qweight.device)
    check(lib.mi_w4_repack(_ptr(qweight), _ptr(qzeros), _ptr(scales), _ptr(perm), _ptr(qw), _ptr(zs), N, K,
                           int(group_size), int(layout), _dt(scales), _stream()), "mi_w4_repack")
    return qw, zs, perm


def gather_columns(x: torch.Tensor, perm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[:, k] = x[:, perm[k]] (GPTQ act-order: activations in the native k order of w4_repack)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.element_size() == 2
    assert perm.dtype == torch.int32 and perm.is_contiguous() and perm.numel() == x.shape[1]
    M, K = x.shape
    out = torch.empty(M, K, dtype=x.dtype, device=x.device) if out is None else out
    check(lib.mi_gather_columns(_ptr(x), _ptr(perm), _ptr(out), M, K, x.stride(0), out.stride(0), _stream()),
          "mi_gather_columns")
    return out


def w4a16_gemm(x: torch.Tensor, qw: torch.Tensor, zs: torch.Tensor, N: int, group_size: int,
               perm: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M, K] (fp16 / bf16) times the repacked int4 weight: decode batches take the x-stationary fused-dequant
    kernels, prefill batches (M > 512) the 256 x 256 tile kernel with the dequant in its MFMA loop -- every shape is a
    hand-written kernel, no library GEMM."""
    assert x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    if bias is not None:
        assert bias.dtype == x.dtype and bias.numel() == N and bias.is_contiguous()
    if perm is not None and M > W4_DENSE_MIN_ROWS and K % 128 == 0:
        x, perm = gather_columns(x, perm), None        # the tile kernel contracts in native k order
    out = torch.empty(M, N, dtype=x.dtype, device=x.device) if out is None else out
    ws_bytes = lib.mi_w4a16_gemm_workspace_bytes(M, N, K) if perm is None else 0
    ws = _gemm_workspace(ws_bytes, x.device) if ws_bytes else None
    check(lib.mi_w4a16_gemm(_ptr(x), _ptr(qw), _ptr(zs), _ptr(perm), _ptr(bias), _ptr(out), M, N, K,
                            int(group_size), x.stride(0), out.stride(0), _dt(x), _ptr(ws),
                            ws_bytes if ws is not None else 0, _stream()), "mi_w4a16_gemm")
    return out


W4_DENSE_MIN_ROWS = 512      # above: the tile kernel; up to it: the decode kernels (128-row chunks)


def w4_dequantize_native(qw: torch.Tensor, zs: torch.Tensor, N: int, K: int, group_size: int,
                         dtype: torch.dtype) -> torch.Tensor:
    """W^T [N, K] in `dtype` from the load-time native layout (mi_w4_dequantize_native): inspection / tests only."""
    w = torch.empty(N, K, dtype=dtype, device=qw.device)
    check(lib.mi_w4_dequantize_native(_ptr(qw), _ptr(zs), _ptr(w), N, K, int(group_size), _DT[dtype], _stream()),
          "mi_w4_dequantize_native")
    return w


def w4_dequantize(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group_size: int,
                  layout: int, g_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense [K,N] weight from the checkpoint layout (the reference's `awq_dequantize` op)."""
    N = scales.shape[1]
    K = qweight.shape[0] if layout == MI_W4_AWQ else qweight.shape[0] * 8
    if g_idx is not None:
        g_idx = g_idx.to(torch.int32).contiguous()
    out = torch.empty(K, N, dtype=scales.dtype, device=scales.device)
    check(lib.mi_w4_dequantize(_ptr(qweight), _ptr(qzeros), _ptr(scales), _ptr(g_idx), _ptr(out), N, K,
                               int(group_size), int(layout), _dt(scales), _stream()), "mi_w4_dequantize")
    return out


# ------------------------------------------------------------ layer glue ("next" rows)
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = rmsnorm(x (+ residual)) * weight; residual (if given) is updated in place to x + residual."""
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous() and weight.dtype == x.dtype
    M, H = x.shape
    out = torch.empty_like(x) if out is None else out
    if residual is not None:
        assert residual.shape == x.shape and residual.stride(1) == 1 and residual.dtype == x.dtype
    check(lib.mi_rmsnorm(_ptr(x), _ptr(residual), _ptr(weight), _ptr(out), M, H, x.stride(0),
                         residual.stride(0) if residual is not None else 0, out.stride(0), float(eps), _dt(x),
                         _stream()), "mi_rmsnorm")
    return out


def rope_neox_(q: torch.Tensor, k: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor,
               head_dim: int) -> None:
    """In-place NeoX RoPE on q [T, Hq*D] and k [T, Hkv*D] (row-strided views allowed)."""
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and positions.dtype == torch.int64
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim
    T = q.shape[0]
    check(lib.mi_rope_neox(_ptr(q), _ptr(k), _ptr(positions), _ptr(cos_sin_cache), T, q.shape[1] // head_dim,
                           k.shape[1] // head_dim, head_dim, q.stride(0), k.stride(0), _dt(q), _stream()),
          "mi_rope_neox")


def rope_neox_k_(k: torch.Tensor, positions: torch.Tensor, cos_sin_cache: torch.Tensor, head_dim: int) -> None:
    """In-place NeoX RoPE on k [T, Hkv*D] only (mi_rope_neox with no q heads): for batches whose attention kernel
    rotates Q as it loads it (extend_attention_fp8out(q_positions=...))."""
    assert k.stride(-1) == 1 and positions.dtype == torch.int64
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim
    check(lib.mi_rope_neox(_ptr(k), _ptr(k), _ptr(positions), _ptr(cos_sin_cache), k.shape[0], 0, k.shape[1] // head_dim,
                           head_dim, k.stride(0), k.stride(0), _dt(k), _stream()), "mi_rope_neox")


def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1
    M, I2 = x.shape
    out = torch.empty(M, I2 // 2, dtype=x.dtype, device=x.device) if out is None else out
    check(lib.mi_silu_and_mul(_ptr(x), _ptr(out), M, I2 // 2, x.stride(0), out.stride(0), _dt(x), _stream()),
          "mi_silu_and_mul")
    return out


def rmsnorm_fp8(x: torch.Tensor, weight: torch.Tensor, eps: float, q_scale: torch.Tensor,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rmsnorm (+ in-place residual add) with the static per-tensor FP8 quantisation of the following
    linear fused in: returns fp8 [M,H]; bit-identical to rmsnorm() followed by fp8_quant_per_tensor(scale)."""
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous() and q_scale.dtype == torch.float32
    M, H = x.shape
    q = torch.empty(M, H, dtype=FP8_DTYPE, device=x.device)
    check(lib.mi_rmsnorm_fp8(_ptr(x), _ptr(residual), _ptr(weight), None, _ptr(q), _ptr(q_scale), M, H, x.stride(0),
                             residual.stride(0) if residual is not None else 0, 0, float(eps), _dt(x), _stream()),
          "mi_rmsnorm_fp8")
    return q


def silu_and_mul_fp8(x: torch.Tensor, q_scale: torch.Tensor) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1 and q_scale.dtype == torch.float32
    M, I2 = x.shape
    q = torch.empty(M, I2 // 2, dtype=FP8_DTYPE, device=x.device)
    check(lib.mi_silu_and_mul_fp8(_ptr(x), None, _ptr(q), _ptr(q_scale), M, I2 // 2, x.stride(0), 0, _dt(x),
                                  _stream()), "mi_silu_and_mul_fp8")
    return q


# ------------------------------------------- decode-shaped FP8 linears fused with their consumer
def decode_attention_fp8out(q: torch.Tensor, k_buf: torch.Tensor, v_buf: torch.Tensor, o_fp8: torch.Tensor,
                            o_scale: torch.Tensor, kv_indptr_t: torch.Tensor, kv_indices_t: torch.Tensor,
                            sm_scale: float, logit_cap: float = 0.0, num_splits: int = 1,
                            workspace: Optional[torch.Tensor] = None, o: Optional[torch.Tensor] = None,
                            split_chunk: int = 0, work: Optional[torch.Tensor] = None) -> torch.Tensor:
    """decode_attention whose output stage also quantises for the following static-scale FP8 linear:
    o_fp8 [B, Hq*D] = quant(o, o_scale), bit-identical to decode_attention + fp8_quant_per_tensor(scale)."""
    B, Hq, D = q.shape
    Hkv = k_buf.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == D
    assert k_buf.stride(2) == 1 and k_buf.stride(1) == D and v_buf.stride(2) == 1 and v_buf.stride(1) == D
    assert k_buf.dtype == q.dtype and v_buf.dtype == q.dtype
    assert o_fp8.dtype == FP8_DTYPE and o_fp8.is_contiguous() and o_fp8.numel() == B * Hq * D
    assert o_scale.dtype == torch.float32 and o_scale.numel() == 1
    if o is not None:
        assert o.dtype == q.dtype and o.stride(2) == 1 and o.stride(1) == D
    if num_splits > 1:
        need = decode_workspace_numel(B, Hq, D, num_splits)
        assert workspace is not None and workspace.dtype == torch.float32 and workspace.numel() >= need
    check(lib.mi_decode_attn_fp8out(_ptr(q), _ptr(k_buf), _ptr(v_buf), _ptr(o), _ptr(o_fp8), _ptr(o_scale),
                                    _ptr(kv_indptr_t), _ptr(kv_indices_t),
                                    _ptr(workspace) if num_splits > 1 else None, B, Hq, Hkv, D, q.stride(0),
                                    o.stride(0) if o is not None else Hq * D, k_buf.stride(0), v_buf.stride(0),
                                    float(sm_scale), float(logit_cap), int(num_splits), int(split_chunk), *_work_args(work),
                                    _dt(q), _stream()), "mi_decode_attn_fp8out")
    return o_fp8


def decode_attention_fp8kv(q: torch.Tensor, k_buf8: torch.Tensor, v_buf8: torch.Tensor, kv_indptr_t: torch.Tensor,
                           kv_indices_t: torch.Tensor, sm_scale: float, k_scale: float = 1.0, v_scale: float = 1.0,
                           logit_cap: float = 0.0, num_splits: int = 1, workspace: Optional[torch.Tensor] = None,
                           o: Optional[torch.Tensor] = None, o_fp8: Optional[torch.Tensor] = None,
                           o_scale: Optional[torch.Tensor] = None, split_chunk: int = 0,
                           work: Optional[torch.Tensor] = None):
    """Decode attention over an fp8 (e4m3fn / uint8 storage) KV pool [slots, Hkv, 128]."""
    B, Hq, D = q.shape
    Hkv = k_buf8.shape[1]
    assert k_buf8.element_size() == 1 and v_buf8.element_size() == 1 and k_buf8.shape[2] == D and v_buf8.shape[2] == D
    assert q.stride(2) == 1 and q.stride(1) == D
    assert k_buf8.stride(2) == 1 and k_buf8.stride(1) == D and v_buf8.stride(2) == 1 and v_buf8.stride(1) == D
    assert o is not None or o_fp8 is not None
    if o is not None:
        assert o.dtype == q.dtype and o.stride(2) == 1 and o.stride(1) == D
    if o_fp8 is not None:
        assert o_fp8.dtype == FP8_DTYPE and o_fp8.is_contiguous() and o_scale is not None and o_scale.dtype == torch.float32
    if num_splits > 1:
        need = decode_workspace_numel(B, Hq, D, num_splits)
        assert workspace is not None and workspace.dtype == torch.float32 and workspace.numel() >= need
    check(lib.mi_decode_attn_fp8kv(_ptr(q), _ptr(k_buf8), _ptr(v_buf8), _ptr(o), _ptr(o_fp8), _ptr(o_scale),
                                   float(k_scale), float(v_scale), _ptr(kv_indptr_t), _ptr(kv_indices_t),
                                   _ptr(workspace) if num_splits > 1 else None, B, Hq, Hkv, D, q.stride(0),
                                   o.stride(0) if o is not None else Hq * D, k_buf8.stride(0), v_buf8.stride(0),
                                   float(sm_scale), float(logit_cap), int(num_splits), int(split_chunk), *_work_args(work),
                                   _dt(q), _stream()), "mi_decode_attn_fp8kv")
    return o if o is not None else o_fp8


def _fused_ws(M: int, N: int, K: int, device):
    nbytes = lib.mi_fp8_gemm_fused_workspace_bytes(M, N, K)
    if nbytes <= 0:
        raise MiHotpathError(f"fused FP8 linear: decode shapes only (M={M} <= 512, K={K} % 128 == 0)")
    return _gemm_workspace(nbytes, device), nbytes


def _check_fp8_operands(a, b_kn, scale_a, scale_b):
    assert a.dtype == FP8_DTYPE and b_kn.dtype == FP8_DTYPE and a.shape[1] == b_kn.shape[0]
    assert a.stride(1) == 1 and b_kn.stride(0) == 1, "b must be the [K,N] view of [N,K] row-major storage"
    assert scale_a.dtype == torch.float32 and scale_a.numel() == 1
    assert scale_b.dtype == torch.float32 and scale_b.numel() == 1


def fp8_gemm_add_rmsnorm(a: torch.Tensor, b_kn: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                         residual: Optional[torch.Tensor], norm_weight: torch.Tensor, eps: float,
                         q_scale: Optional[torch.Tensor] = None, want_out: bool = False):
    """y = rmsnorm(a.b * sa * sb (+ residual, updated in place)) * w; returns (out | None, fp8(y) | None)."""
    _check_fp8_operands(a, b_kn, scale_a, scale_b)
    M, K = a.shape
    N = b_kn.shape[1]
    dt = norm_weight.dtype
    assert norm_weight.is_contiguous() and norm_weight.numel() == N
    if residual is not None:
        assert residual.dtype == dt and residual.shape == (M, N) and residual.is_contiguous()
    out = torch.empty(M, N, dtype=dt, device=a.device) if (want_out or q_scale is None) else None
    q = torch.empty(M, N, dtype=FP8_DTYPE, device=a.device) if q_scale is not None else None
    ws, nbytes = _fused_ws(M, N, K, a.device)
    check(lib.mi_fp8_gemm_add_rmsnorm_fp8(_ptr(a), _ptr(b_kn), _ptr(scale_a), _ptr(scale_b), _ptr(residual),
                                          _ptr(norm_weight), _ptr(out), _ptr(q), _ptr(q_scale), M, N, K, a.stride(0),
                                          b_kn.stride(1), float(eps), _DT[dt], _ptr(ws), nbytes, _stream()),
          "mi_fp8_gemm_add_rmsnorm_fp8")
    return out, q


def fp8_gemm_rope_kvwrite(a: torch.Tensor, b_kn: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                          positions: torch.Tensor, cos_sin_cache: torch.Tensor, k_cache: torch.Tensor,
                          v_cache: torch.Tensor, loc: torch.Tensor, num_q_heads: int, num_kv_heads: int,
                          head_dim: int) -> torch.Tensor:
    """qkv linear + NeoX RoPE + KV-pool write; returns q [M, Hq*D] in the pool dtype."""
    _check_fp8_operands(a, b_kn, scale_a, scale_b)
    M, K = a.shape
    N = b_kn.shape[1]
    assert N == (num_q_heads + 2 * num_kv_heads) * head_dim
    assert positions.dtype == torch.int64 and loc.dtype == torch.int64 and positions.numel() == M and loc.numel() == M
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim and cos_sin_cache.is_contiguous()
    assert k_cache.dtype == v_cache.dtype and k_cache[0].is_contiguous() and v_cache[0].is_contiguous()
    assert k_cache[0].numel() == num_kv_heads * head_dim and v_cache[0].numel() == num_kv_heads * head_dim
    q = torch.empty(M, num_q_heads * head_dim, dtype=k_cache.dtype, device=a.device)
    ws, nbytes = _fused_ws(M, N, K, a.device)
    check(lib.mi_fp8_gemm_rope_kvwrite(_ptr(a), _ptr(b_kn), _ptr(scale_a), _ptr(scale_b), _ptr(positions),
                                       _ptr(cos_sin_cache), _ptr(q), _ptr(k_cache), _ptr(v_cache), _ptr(loc), M,
                                       num_q_heads, num_kv_heads, head_dim, K, a.stride(0), b_kn.stride(1), q.stride(0),
                                       k_cache.stride(0), v_cache.stride(0), _dt(k_cache), _ptr(ws), nbytes, _stream()),
          "mi_fp8_gemm_rope_kvwrite")
    return q


def fp8_gemm_silu_mul(a: torch.Tensor, b_kn: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                      q_scale: torch.Tensor, act_dtype: torch.dtype) -> torch.Tensor:
    """gate_up linear + SiLU(gate)*up + static FP8 quant; returns fp8 [M, I]."""
    _check_fp8_operands(a, b_kn, scale_a, scale_b)
    M, K = a.shape
    N = b_kn.shape[1]
    assert N % 2 == 0 and q_scale.dtype == torch.float32 and q_scale.numel() == 1
    q = torch.empty(M, N // 2, dtype=FP8_DTYPE, device=a.device)
    if M > 512:        # prefill: the tile kernel's own epilogue, no slabs (needs I % 128 == 0, K % 128 == 0)
        if (N // 2) % 128 != 0 or K % 128 != 0:
            raise MiHotpathError(f"fused gate_up at M={M}: needs I % 128 == 0 and K % 128 == 0 (I={N // 2}, K={K})")
        ws, nbytes = None, 0
    else:
        ws, nbytes = _fused_ws(M, N, K, a.device)
    check(lib.mi_fp8_gemm_silu_mul_fp8(_ptr(a), _ptr(b_kn), _ptr(scale_a), _ptr(scale_b), _ptr(q), _ptr(q_scale), M,
                                       N // 2, K, a.stride(0), b_kn.stride(1), _DT[act_dtype], _ptr(ws), nbytes,
                                       _stream()), "mi_fp8_gemm_silu_mul_fp8")
    return q


# ------------------------------------------------------------------ int4 linears fused with their consumer (decode)
def _w4_fused_ws(M: int, N: int, K: int, group_size: int, device):
    nbytes = lib.mi_w4a16_fused_workspace_bytes(M, N, K, int(group_size))
    if nbytes <= 0:
        raise ValueError(f"no fused int4 form for M={M} N={N} K={K} group={group_size} (decode shapes only)")
    return _gemm_workspace(nbytes, device), nbytes


def w4a16_fused_ok(M: int, N: int, K: int, group_size: int) -> bool:
    return 0 < M <= 128 and lib.mi_w4a16_fused_workspace_bytes(M, N, K, int(group_size)) > 0


def w4a16_gemm_add_rmsnorm(x: torch.Tensor, qw: torch.Tensor, zs: torch.Tensor, N: int, group_size: int,
                           residual: Optional[torch.Tensor], norm_weight: torch.Tensor, eps: float) -> torch.Tensor:
    """out = rmsnorm(x.W (+ residual, updated in place)) * w, the int4 GEMM's slabs consumed by one kernel."""
    assert x.dim() == 2 and x.stride(1) == 1 and norm_weight.dtype == x.dtype and norm_weight.numel() == N
    M, K = x.shape
    if residual is not None:
        assert residual.dtype == x.dtype and residual.shape == (M, N) and residual.is_contiguous()
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    ws, nbytes = _w4_fused_ws(M, N, K, group_size, x.device)
    check(lib.mi_w4a16_gemm_add_rmsnorm(_ptr(x), _ptr(qw), _ptr(zs), _ptr(residual), _ptr(norm_weight), _ptr(out), M, N, K,
                                        int(group_size), x.stride(0), float(eps), _dt(x), _ptr(ws), nbytes, _stream()),
          "mi_w4a16_gemm_add_rmsnorm")
    return out


def w4a16_gemm_rope_kvwrite(x: torch.Tensor, qw: torch.Tensor, zs: torch.Tensor, group_size: int, positions: torch.Tensor,
                            cos_sin_cache: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, loc: torch.Tensor,
                            num_q_heads: int, num_kv_heads: int, head_dim: int) -> torch.Tensor:
    """qkv int4 linear + NeoX RoPE + KV-pool write; returns q [M, Hq*D]."""
    assert x.dim() == 2 and x.stride(1) == 1 and k_cache.dtype == x.dtype == v_cache.dtype
    M, K = x.shape
    N = (num_q_heads + 2 * num_kv_heads) * head_dim
    assert positions.dtype == torch.int64 and loc.dtype == torch.int64 and positions.numel() == M and loc.numel() == M
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim and cos_sin_cache.is_contiguous()
    assert k_cache[0].is_contiguous() and v_cache[0].is_contiguous() and k_cache[0].numel() == num_kv_heads * head_dim
    q = torch.empty(M, num_q_heads * head_dim, dtype=x.dtype, device=x.device)
    ws, nbytes = _w4_fused_ws(M, N, K, group_size, x.device)
    check(lib.mi_w4a16_gemm_rope_kvwrite(_ptr(x), _ptr(qw), _ptr(zs), _ptr(positions), _ptr(cos_sin_cache), _ptr(q),
                                         _ptr(k_cache), _ptr(v_cache), _ptr(loc), M, num_q_heads, num_kv_heads, head_dim, K,
                                         int(group_size), x.stride(0), q.stride(0), k_cache.stride(0), v_cache.stride(0),
                                         _dt(x), _ptr(ws), nbytes, _stream()), "mi_w4a16_gemm_rope_kvwrite")
    return q


def w4a16_gemm_silu_mul(x: torch.Tensor, qw: torch.Tensor, zs: torch.Tensor, N: int, group_size: int) -> torch.Tensor:
    """silu(gate) * up of the int4 gate_up linear ([gate | up] columns), [M, N / 2] in the activation dtype."""
    assert x.dim() == 2 and x.stride(1) == 1 and N % 2 == 0
    M, K = x.shape
    out = torch.empty(M, N // 2, dtype=x.dtype, device=x.device)
    ws, nbytes = _w4_fused_ws(M, N, K, group_size, x.device)
    check(lib.mi_w4a16_gemm_silu_mul(_ptr(x), _ptr(qw), _ptr(zs), _ptr(out), M, N // 2, K, int(group_size), x.stride(0),
                                     _dt(x), _ptr(ws), nbytes, _stream()), "mi_w4a16_gemm_silu_mul")
    return out
