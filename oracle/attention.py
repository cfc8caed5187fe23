"""CPU oracle: paged-KV attention of the reference's torch-native backend.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line (relative to /root/reference) whose arithmetic it restates.
Layouts:
  k_cache, v_cache : [slots, Hkv, D]      slot 0 is a padding sink
  req_to_token     : int32 [max_reqs, max_context_len]
  req_pool_indices : int64 [B]; seq_lens int64 [B]; out_cache_loc int64 [tokens]
  q                : [tokens, Hq, D]
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch.nn.functional import scaled_dot_product_attention


# ---------------------------------------------------------------------------
# integer path (bit-exact)
# ---------------------------------------------------------------------------
def kv_indices(
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    lens: torch.Tensor,
    kv_start_idx: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """kv_indptr (int32 [B+1]) and flat kv_indices (int32 [sum lens]).

    Restates python/sglang/srt/layers/attention/utils.py:5-41 (the gather) and
    triton_backend.py:173-175 (kv_indptr = cumsum(lens)); the same python `cat`
    of req_to_token row slices the reference's own test uses as truth
    (test/srt/test_create_kvindices.py:43-49).
    """
    B = int(req_pool_indices.shape[0])
    lens64 = lens.to(torch.int64)
    kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(lens64, dim=0).to(torch.int32)
    rows = []
    for i in range(B):
        s = 0 if kv_start_idx is None else int(kv_start_idx[i])
        rows.append(req_to_token[int(req_pool_indices[i]), s : s + int(lens64[i])])
    if rows:
        flat = torch.cat(rows, dim=0).to(torch.int32).contiguous()
    else:
        flat = torch.empty(0, dtype=torch.int32)
    return kv_indptr, flat


def set_kv_buffer(k_cache, v_cache, loc, k, v) -> None:
    """In-place KV write.  Restates mem_cache/memory_pool.py:454-455
    (`k_buffer[layer][loc] = cache_k`), bf16/fp16 KV (no scale path)."""
    k_cache[loc] = k.to(k_cache.dtype)
    v_cache[loc] = v.to(v_cache.dtype)


# ---------------------------------------------------------------------------
# torch-native flavour: same per-request gather + SDPA as the reference
# ---------------------------------------------------------------------------
def decode_sdpa(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens,
                scaling=None, enable_gqa=False) -> torch.Tensor:
    """o[tokens,Hq,Dv].  Restates torch_native_backend.py:112-180
    (_run_sdpa_forward_decode): one query token per request, keys/values
    gathered through req_to_token[req_pool_idx, :seq_len], non-causal SDPA."""
    B = int(seq_lens.shape[0])
    Hq, Dv = q.shape[1], v_cache.shape[-1]
    o = torch.empty(q.shape[0], Hq, Dv, dtype=q.dtype)
    qT = q.movedim(0, 1)  # [Hq, tokens, D]
    for i in range(B):
        S = int(seq_lens[i])
        tok = req_to_token[int(req_pool_indices[i]), :S].to(torch.int64)
        key = k_cache[tok].movedim(0, 1)      # [Hkv, S, D]
        val = v_cache[tok].movedim(0, 1)
        out = scaled_dot_product_attention(
            qT[:, i : i + 1, :].unsqueeze(0), key.unsqueeze(0), val.unsqueeze(0),
            enable_gqa=enable_gqa, scale=scaling, is_causal=False,
        ).squeeze(0).movedim(1, 0)
        o[i : i + 1] = out
    return o


def extend_sdpa(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens,
                extend_prefix_lens, extend_seq_lens, scaling=None,
                enable_gqa=False, causal=True) -> torch.Tensor:
    """o[extend_tokens,Hq,Dv].  Restates torch_native_backend.py:27-110
    (_run_sdpa_forward_extend).  The reference pads the query with
    *uninitialised* rows for the prefix and discards their outputs (:80-88,
    :108); zeros are used here, the kept rows are unaffected."""
    B = int(seq_lens.shape[0])
    Hq, D, Dv = q.shape[1], q.shape[2], v_cache.shape[-1]
    o = torch.empty(q.shape[0], Hq, Dv, dtype=q.dtype)
    qT = q.movedim(0, 1)
    start_q = 0
    for i in range(B):
        ext = int(extend_seq_lens[i])
        pre = int(extend_prefix_lens[i])
        S = int(seq_lens[i])
        end_q = start_q + ext
        q_red = torch.zeros(Hq, S, D, dtype=q.dtype)
        q_red[:, pre:, :] = qT[:, start_q:end_q, :]
        tok = req_to_token[int(req_pool_indices[i]), :S].to(torch.int64)
        key = k_cache[tok].movedim(0, 1)
        val = v_cache[tok].movedim(0, 1)
        out = scaled_dot_product_attention(
            q_red.unsqueeze(0), key.unsqueeze(0), val.unsqueeze(0),
            enable_gqa=enable_gqa, scale=scaling, is_causal=causal,
        ).squeeze(0).movedim(1, 0)
        o[start_q:end_q] = out[pre:]
        start_q = end_q
    return o


def forward_decode(q, k, v, k_cache, v_cache, req_to_token, req_pool_indices,
                   seq_lens, out_cache_loc, Hq, Hkv, scaling, save_kv_cache=True):
    """Restates TorchNativeAttnBackend.forward_decode, torch_native_backend.py:
    226-267: KV write at out_cache_loc first, then SDPA over the pool.
    q [tokens, Hq*D]; k,v [tokens, Hkv, D].  Returns o [tokens, Hq*Dv]."""
    D = q.shape[-1] // Hq
    if save_kv_cache:
        set_kv_buffer(k_cache, v_cache, out_cache_loc, k, v)
    o = decode_sdpa(q.view(-1, Hq, D), k_cache, v_cache, req_to_token,
                    req_pool_indices, seq_lens, scaling=scaling,
                    enable_gqa=(Hq != Hkv))
    return o.reshape(q.shape[0], -1)


def forward_extend(q, k, v, k_cache, v_cache, req_to_token, req_pool_indices,
                   seq_lens, extend_prefix_lens, extend_seq_lens, out_cache_loc,
                   Hq, Hkv, scaling, causal=True, save_kv_cache=True):
    """Restates TorchNativeAttnBackend.forward_extend, torch_native_backend.py:
    182-224 (write-then-read: the new tokens are read back from the pool)."""
    D = q.shape[-1] // Hq
    if save_kv_cache:
        set_kv_buffer(k_cache, v_cache, out_cache_loc, k, v)
    o = extend_sdpa(q.view(-1, Hq, D), k_cache, v_cache, req_to_token,
                    req_pool_indices, seq_lens, extend_prefix_lens,
                    extend_seq_lens, scaling=scaling, enable_gqa=(Hq != Hkv),
                    causal=causal)
    return o.reshape(q.shape[0], -1)


# ---------------------------------------------------------------------------
# fp32 flavour: the same math spelled out, accumulated in fp32, NOT rounded to
# the I/O dtype.  Used to state kernel error against exact arithmetic and for
# the options torch-native ignores (logit cap, sliding window) whose semantics
# come from the Triton kernels.
# ---------------------------------------------------------------------------
def _softmax_av(s: torch.Tensor, v: torch.Tensor):
    """s [Hq, Lq, S] fp32 (masked entries = -inf), v [Hq, S, Dv] fp32 -> o, lse."""
    m = s.max(dim=-1, keepdim=True).values
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    o = (p @ v) / l
    lse = (m + torch.log(l)).squeeze(-1)
    return o, lse


def _cap(s, logit_cap):
    # triton_ops/decode_attention.py:121-122, extend_attention.py:164-165
    if logit_cap and logit_cap > 0:
        return logit_cap * torch.tanh(s / logit_cap)
    return s


def decode_fp32(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens,
                scaling=None, logit_cap=0.0, return_lse=False):
    """fp32 restatement of decode (softmax(q K^T * scale) V per request); GQA by
    head // group like SDPA's enable_gqa (torch_native_backend.py:166-175)."""
    B = int(seq_lens.shape[0])
    Hq, D = q.shape[1], q.shape[2]
    Hkv, Dv = k_cache.shape[1], v_cache.shape[-1]
    g = Hq // Hkv
    scale = scaling if scaling is not None else 1.0 / math.sqrt(D)
    o = torch.empty(B, Hq, Dv, dtype=torch.float32)
    lse = torch.empty(B, Hq, dtype=torch.float32)
    for i in range(B):
        S = int(seq_lens[i])
        tok = req_to_token[int(req_pool_indices[i]), :S].to(torch.int64)
        key = k_cache[tok].float().movedim(0, 1).repeat_interleave(g, dim=0)
        val = v_cache[tok].float().movedim(0, 1).repeat_interleave(g, dim=0)
        s = torch.einsum("hd,hsd->hs", q[i].float(), key) * scale
        s = _cap(s, logit_cap).unsqueeze(1)
        oi, li = _softmax_av(s, val)
        o[i] = oi.squeeze(1)
        lse[i] = li.squeeze(1)
    return (o, lse) if return_lse else o


def extend_fp32(q, k_cache, v_cache, req_to_token, req_pool_indices, seq_lens,
                extend_prefix_lens, extend_seq_lens, scaling=None, causal=True,
                logit_cap=0.0, sliding_window=-1, custom_mask=None, mask_indptr=None,
                skip_prefix_custom_mask=True):
    """fp32 restatement of extend: row j of request i (global position
    pre_i + j) attends keys 0..pre_i+j (causal) or all S_i keys (non-causal).
    sliding_window >0 keeps keys with q_pos <= k_pos + window
    (triton_ops/extend_attention.py:182-187).
    custom_mask (speculative tree verification, extend_attention.py:93-94,168-178,245-257): flat bool;
    request i owns [ext_i, S_i] row-major at mask_indptr[i]; it REPLACES the causal rule on the new-token
    keys, and masks the prefix keys too unless skip_prefix_custom_mask (the reference's default)."""
    B = int(seq_lens.shape[0])
    Hq, D = q.shape[1], q.shape[2]
    Hkv, Dv = k_cache.shape[1], v_cache.shape[-1]
    g = Hq // Hkv
    scale = scaling if scaling is not None else 1.0 / math.sqrt(D)
    o = torch.empty(q.shape[0], Hq, Dv, dtype=torch.float32)
    start = 0
    for i in range(B):
        ext, pre, S = int(extend_seq_lens[i]), int(extend_prefix_lens[i]), int(seq_lens[i])
        tok = req_to_token[int(req_pool_indices[i]), :S].to(torch.int64)
        key = k_cache[tok].float().movedim(0, 1).repeat_interleave(g, dim=0)
        val = v_cache[tok].float().movedim(0, 1).repeat_interleave(g, dim=0)
        qi = q[start : start + ext].float().movedim(0, 1)  # [Hq, ext, D]
        s = _cap(torch.einsum("hqd,hsd->hqs", qi, key) * scale, logit_cap)
        qpos = torch.arange(pre, pre + ext).view(1, ext, 1)
        kpos = torch.arange(S).view(1, 1, S)
        mask = torch.ones(1, ext, S, dtype=torch.bool)
        if custom_mask is not None:
            m0 = int(mask_indptr[i])
            cm = custom_mask[m0: m0 + ext * S].to(torch.bool).view(1, ext, S).clone()
            if skip_prefix_custom_mask:
                cm[:, :, :pre] = True
            mask = mask & cm
        elif causal:
            mask = mask & (kpos <= qpos)
        if sliding_window is not None and sliding_window > 0:
            mask = mask & (qpos <= kpos + sliding_window)
        s = s.masked_fill(~mask, float("-inf"))
        oi, _ = _softmax_av(s, val)
        o[start : start + ext] = oi.movedim(0, 1)
        start += ext
    return o


def merge_state(o_a, lse_a, o_b, lse_b):
    """lse-weighted merge of two partial attention results.
    Restates sgl-kernel/csrc/attention/merge_attn_states.cu:63-104
    (inf lse -> -inf guard, fp32 arithmetic, output in o dtype, lse fp32).  Pinned by the reference's
    `merge_state_torch` (sgl-kernel/tests/test_merge_state_v2.py:101-135) through tests/golden/elementwise.pt."""
    la = torch.where(torch.isinf(lse_a), torch.full_like(lse_a, float("-inf")), lse_a).float()
    lb = torch.where(torch.isinf(lse_b), torch.full_like(lse_b, float("-inf")), lse_b).float()
    m = torch.maximum(la, lb)
    pa, pb = torch.exp(la - m), torch.exp(lb - m)
    se = pa + pb
    out = o_a.float() * (pa / se).unsqueeze(-1) + o_b.float() * (pb / se).unsqueeze(-1)
    return out.to(o_a.dtype), torch.log(se) + m


# ---------------------------------------------------------------------------
# fp8 (e4m3fn) KV cache -- SURVEY 8f row 1.  PARITY UNPINNED by reference-run vectors: the torch-native
# oracle backend cannot read an fp8 pool; the arithmetic below restates the CONVENTION of the backends that
# do (flashinfer_backend.py:474-555, flashattention_backend.py:643-671: store k / k_scale, attend with
# k_descale / v_descale) on top of MHATokenToKVPool.set_kv_buffer (memory_pool.py:432-440).
# ---------------------------------------------------------------------------
def set_kv_buffer_fp8(k_cache8, v_cache8, loc, k, v, k_scale: float = 1.0, v_scale: float = 1.0) -> None:
    """k_cache8/v_cache8: float8_e4m3fn [slots, Hkv, D].  cache_k.div_(k_scale) on the T-typed rows (the GPU
    kernel torch dispatches multiplies by the fp32 reciprocal and rounds to T), then .to(fp8); values beyond
    +-448 saturate here and in the HIP kernel (torch's own cast would give NaN)."""
    def q8(x, scale):
        if scale != 1.0:
            inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(scale, dtype=torch.float32)
            x = (x.float() * inv).to(x.dtype)
        return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    k_cache8[loc] = q8(k.reshape(loc.shape[0], *k_cache8.shape[1:]), k_scale)
    v_cache8[loc] = q8(v.reshape(loc.shape[0], *v_cache8.shape[1:]), v_scale)


def decode_fp32_fp8kv(q, k_cache8, v_cache8, req_to_token, req_pool_indices, seq_lens, scaling,
                      k_scale: float = 1.0, v_scale: float = 1.0, logit_cap: float = 0.0):
    """softmax(scaling * k_scale * q . k8) . v8 * v_scale, everything in fp32 (decode_fp32 on the dequantised pool)."""
    kf = k_cache8.float() * k_scale
    vf = v_cache8.float() * v_scale
    return decode_fp32(q, kf, vf, req_to_token, req_pool_indices, seq_lens, scaling, logit_cap=logit_cap)
