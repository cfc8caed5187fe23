"""`MiAttnBackend`: the reference's AttentionBackend plugin surface on hand-written gfx950 HIP.

Mirrors TritonAttnBackend's metadata protocol (python/sglang/srt/layers/attention/
triton_backend.py:160-336, 338-627) and TorchNativeAttnBackend's numerics
(torch_native_backend.py:182-267): KV write at out_cache_loc, then attention over the paged
pool indexed through req_to_token.  No Triton, no torch math: every step is a C-ABI call
(kv_indptr scan, kv_indices gather, kv_write scatter, decode / extend kernels).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from ._compat import AttentionBackend


@dataclass
class ForwardMetadata:
    """Same role as triton_backend.py:23-37 (fields we do not need are dropped)."""
    kv_indptr: torch.Tensor
    kv_indices: torch.Tensor
    qo_indptr: Optional[torch.Tensor]
    max_extend_len: Optional[int]
    num_kv_splits: int
    workspace: Optional[torch.Tensor]
    split_chunk: int = 0                            # keys per split (0: seq_len / num_kv_splits per request)
    work: Optional[torch.Tensor] = None             # ragged batches: (request, split) launch list, longest first
    custom_mask: Optional[torch.Tensor] = None      # TARGET_VERIFY: flat tree mask (triton_backend.py:253-260)
    mask_indptr: Optional[torch.Tensor] = None
    # decode, sliding-window layers: the same fields over the last min(S, window + 1) keys of every request
    # (window_kv_indptr / window_kv_indices / window_num_kv_splits of triton_backend.py:35-37)
    page_indptr: Optional[torch.Tensor] = None      # page-granular decode (page_size >= 16): one index per page
    page_indices: Optional[torch.Tensor] = None
    page_size: int = 1
    window: Optional["ForwardMetadata"] = None


class _GraphPlan:
    """Device buffer {num_work, num_splits, split_chunk, - | work list} read by the captured decode launches, with the
    rotating pinned staging it is rewritten from (an event guards reuse, so the host never waits unless it runs four
    steps ahead of the device)."""

    def __init__(self, cap: int, device):
        self.cap = cap
        self.buf = torch.zeros(4 + 2 * cap, dtype=torch.int32, device=device)
        pin = torch.device(device).type == "cuda"
        self.stage = [torch.zeros(4 + 2 * cap, dtype=torch.int32, pin_memory=pin) for _ in range(4)]
        self.ev = [None] * len(self.stage)
        self.i = 0

    def write(self, splits: int, chunk: int, work: torch.Tensor):
        n = work.shape[0]
        i = self.i
        self.i = (i + 1) % len(self.stage)
        stage, ev = self.stage[i], self.ev[i]
        if ev is not None:
            ev.synchronize()
        stage[0], stage[1], stage[2] = n, splits, chunk
        stage[4: 4 + 2 * n] = work.reshape(-1)
        self.buf[: 4 + 2 * n].copy_(stage[: 4 + 2 * n], non_blocking=True)
        if stage.is_pinned():
            ev = torch.cuda.Event()
            ev.record()
            self.ev[i] = ev

    def views(self, cap: int):
        assert cap <= self.cap
        return self.buf[4: 4 + 2 * cap].view(cap, 2), self.buf[:4]


class MiAttnBackend(AttentionBackend):
    def __init__(self, model_runner, skip_prefill: bool = False):
        super().__init__()
        self.device = model_runner.device
        self.req_to_token = model_runner.req_to_token_pool.req_to_token
        max_bs = model_runner.req_to_token_pool.size
        mc = model_runner.model_config
        tp = getattr(model_runner, "tp_size", 1) or 1
        self.num_head = mc.num_attention_heads // tp
        self.num_kv_head = mc.get_num_kv_heads(tp) if hasattr(mc, "get_num_kv_heads") else mc.num_key_value_heads
        self.max_context_len = mc.context_len
        self.v_head_dim = model_runner.token_to_kv_pool.get_value_buffer(0).shape[-1]
        sa = getattr(model_runner, "server_args", None)
        # --triton-attention-num-kv-splits (server_args.py:210) is reused as the split cap
        self.max_kv_splits = int(os.environ.get("MI_ATTN_MAX_KV_SPLITS",
                                                getattr(sa, "triton_attention_num_kv_splits", 8) or 8))
        # triton_backend.py:63-68: layers with layer.sliding_window_size > -1 attend the last window + 1 keys
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = int(sw) if sw is not None and int(sw) > 0 else None
        self.skip_prefill = skip_prefill
        # page-granular decode indices when the pool is paged (PagedTokenToKVPoolAllocator hands out page-aligned runs,
        # allocator.py:407-543): page_size a power of two >= 16; smaller pages keep the token-granular path
        ps = int(getattr(model_runner, "page_size", 1) or 1)
        self.page_size = ps if ps >= 16 and ps & (ps - 1) == 0 else 1
        self.page_indptr = torch.zeros(max_bs + 1, dtype=torch.int32, device=self.device) if self.page_size > 1 else None
        self.num_draft_tokens = getattr(sa, "speculative_num_draft_tokens", None)
        self.speculative_num_steps = getattr(sa, "speculative_num_steps", None)     # triton_backend.py:102
        self.mask_indptr = torch.zeros(max_bs + 1, dtype=torch.int64, device=self.device)
        self.kv_indptr = torch.zeros(max_bs + 1, dtype=torch.int32, device=self.device)
        self.window_kv_indptr = torch.zeros_like(self.kv_indptr) if self.sliding_window_size else None
        self.qo_indptr = torch.zeros(max_bs + 1, dtype=torch.int32, device=self.device)
        self.cu_count = max(ops.cu_count(), 1)
        self.forward_metadata: Optional[ForwardMetadata] = None
        self._ws: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ helpers
    def _heads_per_wg(self) -> int:
        return 8 if self.num_kv_head % 8 == 0 else 4 if self.num_kv_head % 4 == 0 else 2 if self.num_kv_head % 2 == 0 else 1

    def _split_cap(self, bs: int) -> int:
        """Most splits a request may get.  --triton-attention-num-kv-splits (8) is the cap at serving batch sizes; a
        few long requests need more to reach every CU (B=1, S=32768: 8 splits 490 us, 64 splits 81-107 us), so small
        batches may go up to 64 -- never beyond what the workspace of a graph (max_bs x max_kv_splits) can hold."""
        per_req = self.num_kv_head // self._heads_per_wg()
        return max(self.max_kv_splits, min(64, 512 // max(bs * per_req, 1)))

    def _choose_splits(self, bs: int, seq_lens_sum: int, cap: Optional[int] = None) -> int:
        """Split-KV count (replaces get_num_kv_splits_triton, triton_backend.py:875-924).  The decode
        kernel runs ONE 8-wave workgroup (8 kv heads of one (request, split)) per CU at a time, every
        workgroup doing the same work, so the time is  rounds x (keys per split + a fixed ramp) + the merge:
        pick the split count that minimises it -- i.e. a workgroup count that fills whole rounds of the
        chip (measured at B=128, S=2048: 2 splits = 256 workgroups 183 us, 3 splits 223 us, 4 splits
        194 us).  Never below 128 keys per split: a lone workgroup walks its keys at ~21 keys/us (dependent
        index -> row loads), so small batches want many short splits (B=1, S=2048: 8 splits 25.8 us, 16 splits
        19.7 us, 64 splits 22.0 us -- the merge costs ~0.13 us per split).  Any value gives the same math up to fp32
        reassociation."""
        heads_per_wg = self._heads_per_wg()
        wgs = bs * (self.num_kv_head // heads_per_wg)
        slots = self.cu_count * max(1, 8 // heads_per_wg)        # co-resident workgroups on the chip
        avg = max(seq_lens_sum // max(bs, 1), 1)
        ramp, merge = 96, 0.8                                   # launch + first-tile latency / merge step, in keys
        cap = self._split_cap(bs) if cap is None else cap
        best, best_cost = 1, None
        for s in range(1, max(1, cap) + 1):
            if s > 1 and avg // s < 128:
                break
            rounds = -(-wgs * s // slots)
            cost = rounds * (-(-avg // s) + ramp) + merge * s
            if best_cost is None or cost < best_cost:
                best, best_cost = s, cost
        return best

    def _ragged_chunk(self, lens: torch.Tensor, mx: int, cap: int) -> int:
        """Keys per split of a ragged batch.  Floor: `min_split_chunk` (512) and mx / cap (the split capacity of the
        workspace).  Above the floor the chunk is chosen from a closed-form estimate of the launch's makespan on the
        chip's co-resident workgroup slots, evaluated for 13 candidates at once (O(13 B) vectorised, ~0.1 ms at
        B = 512; the item-by-item heap simulation it replaces took 3-10 ms per decode step): the full chunks (equal
        cost chunk + ramp) run first in floor(n_full / slots) whole rounds plus a partial one, the remainders -- by
        decreasing length -- fill the slots the partial round leaves free, so the launch ends at
        max(end of the full chunks, start of the last whole round + the longest remainder, total work / slots, and --
        when there are more remainders than free slots -- the end of the largest remainder that has to wait for the
        smallest of the first batch); the merge walks ceil(mx / chunk) splits.  With 512-key chunks
        the 257 k keys of the B = 128, S_i ~ U[1, 4096] batch are 570 items = 2.2 rounds of 256."""
        floor = max(getattr(self, "min_split_chunk", 512), -(-mx // cap))
        floor = (floor + 15) // 16 * 16
        if "min_split_chunk" in self.__dict__:
            return floor                                    # pinned by the caller (tools/attn_bench.py FLOOR=)
        slots = self.cu_count * max(1, 8 // self._heads_per_wg())
        ramp, merge = 96.0, 0.8
        # candidates: the floor and up to 192 keys above it (measured on that batch: 512 -> 208.7 us, 544 -> 204.5,
        # 640 -> 220, 768 -> 243, 1024 -> 217 us: the model ranks the neighbourhood of the floor correctly and is too kind
        # to much larger chunks, whose few long items also lose memory-level parallelism)
        import numpy as np
        cand = np.arange(floor, floor + 193, 16, dtype=np.int64)[:, None]                 # [C, 1]
        L = np.asarray(lens, dtype=np.int64)[None, :]                                     # [1, B]
        full, rem = L // cand, L % cand                                                   # [C, B]
        n_full = full.sum(1).astype(np.float64)
        n_rem = (rem > 0).sum(1).astype(np.float64)
        item = cand[:, 0].astype(np.float64) + ramp
        rem_max = rem.max(1).astype(np.float64) + ramp
        work = n_full * item + rem.sum(1) + ramp * n_rem
        one_round = np.where(n_full > 0, item, rem_max)                                   # everything in one round
        whole = np.floor(n_full / slots) * item                                           # the whole rounds of full chunks
        part = np.where(n_full % slots > 0, item, 0.0)
        many = np.maximum(np.maximum(whole + part, whole + rem_max), work / slots)
        # remainders beyond the slots the partial round leaves free wait for a slot: the largest of them starts when the
        # smallest remainder of the first batch ends
        B = L.shape[1]
        kf = (slots - n_full % slots).astype(np.int64)                                    # free slots, 1 .. slots
        rs = -np.sort(-rem, axis=1)                                                       # remainders, decreasing
        idx = np.minimum(kf, B - 1)
        rows = np.arange(cand.shape[0])
        second = whole + (rs[rows, idx - 1] + ramp) + (rs[rows, idx] + ramp)
        many = np.where((kf < B) & (n_rem > kf), np.maximum(many, second), many)
        cost = np.where(n_full + n_rem <= slots, one_round, many) + merge * np.ceil(mx / cand[:, 0])
        return int(cand[int(np.argmin(cost)), 0])              # argmin returns the first (smallest) of equal costs

    def _plan_on_host(self, bs: int, seq_lens_sum: int, seq_lens_cpu=None, force_list: bool = False,
                      cap: Optional[int] = None):
        """(num_kv_splits, split_chunk, host work list or None).  Uniform batches: _choose_splits, no chunk, no list
        (with `force_list`: the full (request, split) grid, split index outermost).  RAGGED batches (longest request
        > 1.5x the mean): fixed-size splits of `chunk` keys and a launch list of the non-empty (request, split) pairs
        -- all full chunks first, then the remainders by decreasing length, so the last round of workgroups is filled
        with the short pieces (longest-processing-time-first packing).  Measured at B=128, S_i ~ U[1,4096] (257 k
        keys): per-request S/2 splits 280 us, S/8 245-270 us, fixed chunks in grid order 260-300 us, this list: see
        DESIGN.md section 3.1; the uniform batch of the same size takes 183 us."""
        cap = self._split_cap(bs) if cap is None else cap
        splits = self._choose_splits(bs, seq_lens_sum, cap)
        ragged = False
        if seq_lens_cpu is not None and bs > 1:
            lens = torch.as_tensor(seq_lens_cpu)[:bs].to(torch.int64)
            mx, avg = int(lens.max()), max(seq_lens_sum // max(bs, 1), 1)
            ragged = 2 * mx > 3 * avg and mx > 512
        if not ragged:
            if not force_list:
                return splits, 0, None
            work = torch.stack([torch.arange(bs).repeat(splits), torch.arange(splits).repeat_interleave(bs)], dim=1)
            return splits, 0, work.to(torch.int32)
        chunk = self._ragged_chunk(lens, mx, cap)
        nsplit = -(-mx // chunk)
        full, rem = lens // chunk, lens % chunk
        # full chunks: split index outermost so that neighbouring workgroups are different requests (XCD spread)
        fs = torch.arange(int(full.max()) if bs else 0).view(-1, 1)
        mask = fs < full.view(1, -1)
        b_idx = torch.arange(bs).view(1, -1).expand_as(mask)[mask]
        s_idx = fs.expand_as(mask)[mask]
        order = torch.argsort(rem, descending=True)
        order = order[rem[order] > 0]
        work = torch.stack([torch.cat([b_idx, order]), torch.cat([s_idx, full[order]])], dim=1).to(torch.int32)
        return nsplit, chunk, work.contiguous()

    def _choose_split_plan(self, bs: int, seq_lens_sum: int, seq_lens_cpu=None):
        """(num_kv_splits, split_chunk, device work list or None) of an eager decode step -- see _plan_on_host."""
        if torch.cuda.is_current_stream_capturing():
            seq_lens_cpu = None
        splits, chunk, work = self._plan_on_host(bs, seq_lens_sum, seq_lens_cpu)
        return splits, chunk, (work.to(self.device, non_blocking=True) if work is not None else None)

    def _window_lens(self, seq_lens, seq_lens_cpu, seq_lens_sum: int, bs: int):
        """update_sliding_window_buffer (triton_backend.py:927-955): the last min(S, window + 1) keys of every request.
        Returns (device lens int64, device start int32, host lens or None, their sum or an upper bound of it)."""
        w1 = self.sliding_window_size + 1
        wl = torch.clamp(seq_lens[:bs], max=w1)
        start = (seq_lens[:bs] - wl).to(torch.int32)
        wl_cpu = torch.clamp(torch.as_tensor(seq_lens_cpu)[:bs], max=w1) if seq_lens_cpu is not None else None
        wsum = int(wl_cpu.sum()) if wl_cpu is not None else min(seq_lens_sum, bs * w1)
        return wl, start, wl_cpu, wsum

    def _window_decode_metadata(self, bs, req_pool_indices, seq_lens, seq_lens_sum, seq_lens_cpu) -> ForwardMetadata:
        wl, start, wl_cpu, wsum = self._window_lens(seq_lens, seq_lens_cpu, seq_lens_sum, bs)
        indptr = ops.kv_indptr(wl, self.window_kv_indptr)
        indices = torch.empty(max(wsum, 1), dtype=torch.int32, device=self.device)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], wl, indptr, indices, kv_start_idx=start)
        splits, chunk, work = self._choose_split_plan(bs, wsum, wl_cpu)
        return ForwardMetadata(indptr, indices, None, None, splits, self._workspace(bs, splits), split_chunk=chunk,
                               work=work)

    def _choose_extend_splits(self, ext_lens, prefix_lens) -> int:
        """Split-KV count of an extend / verify launch.  The kernel puts one workgroup on every (request, 64 query rows
        = 64 / heads-per-group tokens, kv head): a verify batch of 4 x 8 draft tokens over 8 k cached keys is 32
        workgroups, each walking its 8 k keys alone (453 us measured; the bytes are worth 27 us).  The chip holds ~4 of
        these 4-wave workgroups per CU, so aim at ~1024 of them: split the key range (>= 512 keys per split, up to 32
        splits) and merge as decode does.  Measured: 4 x 8 over 8192 keys 453 -> 68 us (16 splits), 16 x 8 over 4096
        248 -> 75 us (8), one 64-token chunk over 32000 keys 1685 -> 149 us (16)."""
        group = max(self.num_head // max(self.num_kv_head, 1), 1)
        hg = 4 if group % 4 == 0 else 2 if group % 2 == 0 else 1
        bq = 64 // hg
        wgs = sum(-(-int(e) // bq) for e in ext_lens) * self.num_kv_head * (group // hg)
        keys = max((int(p_) + int(e) for p_, e in zip(prefix_lens, ext_lens)), default=0)
        if wgs <= 0 or wgs >= 768:
            return 1
        return max(1, min(32, 1024 // wgs, keys // 512))

    def _workspace(self, bs: int, splits: int) -> Optional[torch.Tensor]:
        n = ops.decode_workspace_numel(bs, self.num_head, self.v_head_dim, splits)
        if n == 0:
            return None
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._ws

    # ----------------------------------------------------------------- metadata
    def init_forward_metadata(self, forward_batch):
        bs = forward_batch.batch_size
        mode = forward_batch.forward_mode
        spec_info = getattr(forward_batch, "spec_info", None)
        if mode.is_draft_extend():
            self.forward_metadata = self._draft_extend_metadata(
                bs, forward_batch.req_pool_indices, forward_batch.seq_lens, int(forward_batch.seq_lens_sum), spec_info,
                getattr(forward_batch, "extend_seq_lens_cpu", None), getattr(forward_batch, "seq_lens_cpu", None))
            return
        if mode.is_decode_or_idle() and spec_info is not None:
            # draft decode (triton_backend.py:200-202): the draft worker hands the index arrays over ready-made, one row
            # per (request, top-k branch); the host does not know the lengths, so the split count is chosen from the total
            kv_indptr, kv_indices = spec_info.kv_indptr, spec_info.kv_indices
            rows = kv_indptr.shape[0] - 1
            splits = self._choose_splits(rows, max(int(kv_indices.shape[0]), rows))
            self.forward_metadata = ForwardMetadata(kv_indptr, kv_indices, None, None, splits,
                                                    self._workspace(rows, splits))
            return
        if mode.is_target_verify():
            # triton_backend.py:226-263: every request verifies num_draft_tokens tree nodes against its whole
            # committed sequence ("prefix" = seq_lens) under spec_info.custom_mask
            nd = int(self.num_draft_tokens or 0)
            if nd <= 0 or spec_info is None or getattr(spec_info, "custom_mask", None) is None:
                raise ValueError("TARGET_VERIFY needs server_args.speculative_num_draft_tokens and spec_info.custom_mask")
            qo_indptr = torch.arange(0, (1 + bs) * nd, step=nd, dtype=torch.int32, device=self.device)
            kv_indptr = ops.kv_indptr(forward_batch.seq_lens, self.kv_indptr)
            kv_indices = torch.empty(max(int(forward_batch.seq_lens_sum), 1), dtype=torch.int32, device=self.device)
            ops.kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr, kv_indices)
            seq_mask_len = nd * (forward_batch.seq_lens.to(torch.int64) + nd)
            mask_indptr = self.mask_indptr
            mask_indptr[1: bs + 1] = torch.cumsum(seq_mask_len[:bs], dim=0)     # plumbing, as the reference does it
            lens_cpu = getattr(forward_batch, "seq_lens_cpu", None)
            splits = self._choose_extend_splits([nd] * bs, lens_cpu.tolist()) if lens_cpu is not None else 1
            self.forward_metadata = ForwardMetadata(kv_indptr, kv_indices, qo_indptr, nd, splits,
                                                    self._workspace(bs * nd, splits),
                                                    custom_mask=spec_info.custom_mask, mask_indptr=mask_indptr[: bs + 1])
            return
        if mode.is_decode_or_idle():
            kv_indptr = ops.kv_indptr(forward_batch.seq_lens, self.kv_indptr)
            kv_indices = torch.empty(max(int(forward_batch.seq_lens_sum), 1), dtype=torch.int32, device=self.device)
            ops.kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens, kv_indptr,
                           kv_indices)
            splits, chunk, work = self._choose_split_plan(bs, int(forward_batch.seq_lens_sum),
                                                          getattr(forward_batch, "seq_lens_cpu", None))
            self.forward_metadata = ForwardMetadata(kv_indptr, kv_indices, None, None, splits,
                                                    self._workspace(bs, splits), split_chunk=chunk, work=work)
            if self.page_size > 1:
                pages = torch.empty(max(-(-int(forward_batch.seq_lens_sum) // self.page_size) + bs, 1), dtype=torch.int32,
                                    device=self.device)
                pi, px = ops.kv_page_tables(self.req_to_token, forward_batch.req_pool_indices, forward_batch.seq_lens,
                                            self.page_size, self.page_indptr, pages)
                self.forward_metadata.page_indptr, self.forward_metadata.page_indices = pi, px
                self.forward_metadata.page_size = self.page_size
            if self.sliding_window_size:
                self.forward_metadata.window = self._window_decode_metadata(
                    bs, forward_batch.req_pool_indices, forward_batch.seq_lens, int(forward_batch.seq_lens_sum),
                    getattr(forward_batch, "seq_lens_cpu", None))
        else:
            # extend / mixed: kv_indices cover the cached PREFIX only (triton_backend.py:302-321); the
            # host-side length lists avoid the reference's .item() syncs (:288,:320)
            prefix_cpu = forward_batch.extend_prefix_lens_cpu
            ext_cpu = forward_batch.extend_seq_lens_cpu
            prefix_sum = int(sum(prefix_cpu)) if prefix_cpu is not None else int(forward_batch.extend_prefix_lens.sum())
            max_ext = int(max(ext_cpu)) if ext_cpu is not None else int(forward_batch.extend_seq_lens.max())
            kv_indptr = ops.kv_indptr(forward_batch.extend_prefix_lens, self.kv_indptr)
            kv_indices = torch.empty(max(prefix_sum, 1), dtype=torch.int32, device=self.device)
            ops.kv_indices(self.req_to_token, forward_batch.req_pool_indices, forward_batch.extend_prefix_lens,
                           kv_indptr, kv_indices)
            qo_indptr = ops.kv_indptr(forward_batch.extend_seq_lens, self.qo_indptr)
            splits = self._choose_extend_splits(ext_cpu, prefix_cpu) if ext_cpu is not None and prefix_cpu is not None else 1
            tokens = int(sum(ext_cpu)) if ext_cpu is not None else 0
            self.forward_metadata = ForwardMetadata(kv_indptr, kv_indices, qo_indptr, max_ext, splits,
                                                    self._workspace(tokens, splits))
            if self.page_size > 1 and prefix_sum > 0:
                # paged pool: one index per page of the cached prefix for the long-extend kernel
                pages = torch.empty(-(-prefix_sum // self.page_size) + bs, dtype=torch.int32, device=self.device)
                pi, px = ops.kv_page_tables(self.req_to_token, forward_batch.req_pool_indices,
                                            forward_batch.extend_prefix_lens, self.page_size, self.page_indptr, pages)
                self.forward_metadata.page_indptr, self.forward_metadata.page_indices = pi, px
                self.forward_metadata.page_size = self.page_size

    def _draft_extend_metadata(self, bs, req_pool_indices, seq_lens, seq_lens_sum, spec_info, ext_cpu=None,
                               seq_lens_cpu=None, kv_indices=None, splits=None, workspace=None,
                               accept_length=None) -> ForwardMetadata:
        """DRAFT_EXTEND (triton_backend.py:265-283, EagleDraftInput.generate_attn_arg_prefill eagle_utils.py:134-163):
        after a verify step every request extends its draft-model cache by its `accept_length` newly accepted tokens;
        `seq_lens` already counts them (eagle_utils.py:593) and `qo_indptr` is the scan of `spec_info.accept_length`.
        The new rows are attended from the k/v arguments (causal) and everything before them from the pool, so the
        pool-side index list covers the first seq_len - accept_length keys of each request: every key once, the
        arithmetic of the torch-native oracle (torch_native_backend.py:27-110).  (The Triton backend lists the whole
        sequence on the pool side as well, its kernel then meets the new keys twice -- the FIXME at
        triton_backend.py:276-278; not reproduced.)"""
        if accept_length is None:
            if spec_info is None or getattr(spec_info, "accept_length", None) is None:
                raise ValueError("DRAFT_EXTEND needs spec_info.accept_length")
            accept_length = spec_info.accept_length
        acc = accept_length[:bs]
        acc = acc if acc.dtype in (torch.int32, torch.int64) else acc.to(torch.int32)
        prefix = (seq_lens[:bs] - acc.to(seq_lens.dtype)).contiguous()
        cap_ext = int(self.speculative_num_steps or 0) + 1
        if ext_cpu is not None:
            ext_l = [int(e) for e in ext_cpu][:bs]
            max_ext, tokens = max(ext_l), sum(ext_l)
        else:
            if cap_ext <= 1:
                raise ValueError("DRAFT_EXTEND needs forward_batch.extend_seq_lens_cpu or server_args.speculative_num_steps")
            ext_l, max_ext, tokens = None, cap_ext, bs * cap_ext
        kv_indptr = ops.kv_indptr(prefix, self.kv_indptr)
        if kv_indices is None:
            n = seq_lens_sum - tokens if ext_l is not None else seq_lens_sum
            kv_indices = torch.empty(max(int(n), 1), dtype=torch.int32, device=self.device)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], prefix, kv_indptr, kv_indices)
        qo_indptr = ops.kv_indptr(acc.contiguous(), self.qo_indptr)
        if splits is None:
            if ext_l is not None and seq_lens_cpu is not None:
                pre_l = [int(s_) - e for s_, e in zip(torch.as_tensor(seq_lens_cpu)[:bs].tolist(), ext_l)]
                splits = self._choose_extend_splits(ext_l, pre_l)
            else:
                splits = 1
        if workspace is None and splits > 1:
            workspace = self._workspace(tokens, splits)
        return ForwardMetadata(kv_indptr, kv_indices, qo_indptr, max_ext, splits, workspace if splits > 1 else None)

    def _graph_extend_splits(self, bs: int, tokens_per_req: int) -> int:
        """Split-KV count of a CAPTURED verify / draft-extend launch: the grid is frozen, the key ranges are not (each
        split takes its share of whatever the request holds at replay, empty splits cost a merge step), so the count is
        chosen from the workgroup count alone, within the workspace init_cuda_graph_state allocated."""
        group = max(self.num_head // max(self.num_kv_head, 1), 1)
        hg = 4 if group % 4 == 0 else 2 if group % 2 == 0 else 1
        wgs = bs * -(-tokens_per_req // (64 // hg)) * self.num_kv_head * (group // hg)
        return max(1, min(self.max_kv_splits, 1024 // max(wgs, 1)))

    def _verify_graph_metadata(self, bs, req_pool_indices, seq_lens, spec_info, splits=None) -> ForwardMetadata:
        """TARGET_VERIFY into the persistent buffers (capture triton_backend.py:445-475, replay :579-607)."""
        nd = int(self.num_draft_tokens or 0)
        if nd <= 0 or spec_info is None or getattr(spec_info, "custom_mask", None) is None:
            raise ValueError("TARGET_VERIFY needs server_args.speculative_num_draft_tokens and spec_info.custom_mask")
        qo_indptr = self.qo_indptr[: bs + 1]
        qo_indptr.copy_(torch.arange(0, (1 + bs) * nd, step=nd, dtype=torch.int32, device=self.device))
        kv_indptr = ops.kv_indptr(seq_lens[:bs], self.kv_indptr)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], kv_indptr, self.cuda_graph_kv_indices)
        cm = spec_info.custom_mask
        self.cuda_graph_custom_mask[: cm.shape[0]].copy_(cm.view(torch.uint8) if cm.dtype == torch.bool else cm)
        mask_indptr = self.mask_indptr[: bs + 1]
        mask_indptr[1: bs + 1] = torch.cumsum(nd * (seq_lens[:bs].to(torch.int64) + nd), dim=0)
        splits = self._graph_extend_splits(bs, nd) if splits is None else splits
        return ForwardMetadata(kv_indptr, self.cuda_graph_kv_indices, qo_indptr, nd, splits,
                               self.cuda_graph_workspace if splits > 1 else None,
                               custom_mask=self.cuda_graph_custom_mask.view(torch.bool), mask_indptr=mask_indptr)

    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int, kv_indices_buf: Optional[torch.Tensor] = None):
        """Preallocate everything replay touches (triton_backend.py:338-388).  Beyond the reference's buffers: the
        decode PLAN.  A captured launch bakes its grid and scalar arguments in, so the captured decode kernels take
        the split count, the split size and the (request, split) launch list from a device buffer instead
        (mi_decode_attn's `plan`), which replay rewrites from the host-side lengths with one small async copy."""
        self.cuda_graph_kv_indices = (kv_indices_buf if kv_indices_buf is not None else
                                      torch.zeros(max_num_tokens * self.max_context_len, dtype=torch.int32,
                                                  device=self.device))
        n = ops.decode_workspace_numel(max_num_tokens, self.num_head, self.v_head_dim, self.max_kv_splits)
        self.cuda_graph_workspace = torch.empty(max(n, 1), dtype=torch.float32, device=self.device)
        self._gplan = _GraphPlan(max_num_tokens * self.max_kv_splits, self.device)
        self.cuda_graph_plan_buf = self._gplan.buf
        if not self.skip_prefill:                                   # triton_backend.py:366-371
            self.cuda_graph_custom_mask = torch.zeros(max_num_tokens * self.max_context_len, dtype=torch.uint8,
                                                      device=self.device)
        if self.page_size > 1:
            self.cuda_graph_page_indices = torch.zeros(max_num_tokens * (-(-self.max_context_len // self.page_size)),
                                                       dtype=torch.int32, device=self.device)
        if self.sliding_window_size:
            # triton_backend.py:373-381
            wcap = max_num_tokens * min(self.max_context_len, self.sliding_window_size + 1)
            self.cuda_graph_window_kv_indices = torch.zeros(wcap, dtype=torch.int32, device=self.device)
            self._gplan_win = _GraphPlan(max_num_tokens * self.max_kv_splits, self.device)

    def _write_graph_plan(self, bs: int, seq_lens_sum: int, seq_lens_cpu, gplan=None):
        """Plan this replay on the host and ship {num_work, num_splits, split_chunk | work list} to the device buffer
        the captured kernels read (one small async copy from rotating pinned buffers, no host sync)."""
        cap = self._graph_split_cap(bs)
        splits, chunk, work = self._plan_on_host(bs, seq_lens_sum, seq_lens_cpu, force_list=True, cap=cap)
        assert splits <= cap and work.shape[0] <= bs * cap
        (gplan or self._gplan).write(splits, chunk, work)

    def _graph_split_cap(self, bs: int) -> int:
        """_split_cap within what init_cuda_graph_state allocated (workspace and work list hold max_bs x max_kv_splits)."""
        return max(self.max_kv_splits, min(self._split_cap(bs), self._gplan.cap // max(bs, 1)))

    def _graph_metadata(self, bs: int, kv_indptr, kv_indices=None, gplan=None) -> ForwardMetadata:
        cap = self._graph_split_cap(bs)
        work, plan = (gplan or self._gplan).views(bs * cap)
        md = ForwardMetadata(kv_indptr, self.cuda_graph_kv_indices if kv_indices is None else kv_indices, None, None,
                             cap, self.cuda_graph_workspace, split_chunk=0, work=(work, plan))
        return self._attach_graph_pages(md) if kv_indices is None else md

    def _attach_graph_pages(self, md: ForwardMetadata) -> ForwardMetadata:
        if self.page_size > 1:
            md.page_indptr, md.page_indices, md.page_size = self.page_indptr, self.cuda_graph_page_indices, self.page_size
        return md

    def _graph_page_tables(self, bs, req_pool_indices, seq_lens):
        """Capture and replay: page ids into the persistent buffer (two kernels, no allocation, no host sync)."""
        if self.page_size > 1:
            ops.kv_page_tables(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], self.page_size, self.page_indptr,
                               self.cuda_graph_page_indices)

    def _graph_window(self, bs, req_pool_indices, seq_lens, seq_lens_sum, seq_lens_cpu) -> Optional[ForwardMetadata]:
        """Capture and replay of the sliding-window set (update_sliding_window_buffer_cuda_graph,
        triton_backend.py:958-984): indices into the persistent buffer, plan into its own device buffer."""
        if not self.sliding_window_size:
            return None
        wl, start, wl_cpu, wsum = self._window_lens(seq_lens, seq_lens_cpu, seq_lens_sum, bs)
        indptr = ops.kv_indptr(wl, self.window_kv_indptr)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], wl, indptr, self.cuda_graph_window_kv_indices,
                       kv_start_idx=start)
        if self.max_kv_splits > 1:
            self._write_graph_plan(bs, wsum, wl_cpu, self._gplan_win)
            return self._graph_metadata(bs, indptr, self.cuda_graph_window_kv_indices, self._gplan_win)
        return ForwardMetadata(indptr, self.cuda_graph_window_kv_indices, None, None, 1, None)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens, encoder_lens,
                                                 forward_mode, spec_info):
        assert encoder_lens is None, "Not supported"
        if forward_mode.is_target_verify():
            self.forward_metadata = self._verify_graph_metadata(bs, req_pool_indices, seq_lens, spec_info)
            return
        if forward_mode.is_draft_extend():
            # triton_backend.py:476-503: speculative_num_steps + 1 query rows per request at capture time; replay
            # rewrites qo_indptr from spec_info.accept_length (shorter requests leave their last blocks idle)
            per = int(self.speculative_num_steps or 0) + 1
            acc = getattr(spec_info, "accept_length", None)
            if acc is None:
                acc = torch.full((bs,), per, dtype=torch.int32, device=self.device)
            self.forward_metadata = self._draft_extend_metadata(
                bs, req_pool_indices, seq_lens, 0, spec_info, None, None, kv_indices=self.cuda_graph_kv_indices,
                splits=self._graph_extend_splits(bs, per), workspace=self.cuda_graph_workspace, accept_length=acc)
            return
        if not forward_mode.is_decode_or_idle():
            raise ValueError(f"Invalid forward mode: {forward_mode=} for graph capture.")
        if spec_info is not None:                # draft decode: the draft worker's own persistent index arrays (:433-434)
            self.forward_metadata = ForwardMetadata(spec_info.kv_indptr, spec_info.kv_indices, None, None,
                                                    self.max_kv_splits, self.cuda_graph_workspace)
            return
        kv_indptr = ops.kv_indptr(seq_lens[:bs], self.kv_indptr)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], kv_indptr, self.cuda_graph_kv_indices)
        # the captured launch covers the whole capacity bs x max_kv_splits of the work list; which entries are live,
        # the split count and the split size come from the device-side plan.  A valid plan must be in place for the
        # capture-time warm-up runs too (the capture inputs are fill values: one split per request)
        self._graph_page_tables(bs, req_pool_indices, seq_lens)
        if self.max_kv_splits > 1:
            self._write_graph_plan(bs, bs * self.get_cuda_graph_seq_len_fill_value(), None)
            self.forward_metadata = self._graph_metadata(bs, kv_indptr)
        else:
            self.forward_metadata = self._attach_graph_pages(
                ForwardMetadata(kv_indptr, self.cuda_graph_kv_indices, None, None, 1, None))
        self.forward_metadata.window = self._graph_window(bs, req_pool_indices, seq_lens,
                                                          bs * self.get_cuda_graph_seq_len_fill_value(), None)

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum, encoder_lens,
                                                forward_mode, spec_info, seq_lens_cpu):
        if forward_mode.is_target_verify():
            self._verify_graph_metadata(len(req_pool_indices), req_pool_indices, seq_lens, spec_info)
            return
        if forward_mode.is_draft_extend():       # triton_backend.py:608-624: qo_indptr and the index list, no sync
            self._draft_extend_metadata(bs, req_pool_indices, seq_lens, 0, spec_info, None, None,
                                        kv_indices=self.cuda_graph_kv_indices, splits=1)
            return
        if not forward_mode.is_decode_or_idle():
            raise ValueError(f"Invalid forward mode: {forward_mode=} for graph replay.")
        if spec_info is not None:                # draft decode (triton_backend.py:575-578)
            self.kv_indptr[: spec_info.kv_indptr.shape[0]].copy_(spec_info.kv_indptr)
            self.cuda_graph_kv_indices[: spec_info.kv_indices.shape[0]].copy_(spec_info.kv_indices)
            return
        # no allocation, no host sync: two kernels into persistent buffers (triton_backend.py:544-566) and the plan
        kv_indptr = ops.kv_indptr(seq_lens[:bs], self.kv_indptr)
        ops.kv_indices(self.req_to_token, req_pool_indices[:bs], seq_lens[:bs], kv_indptr, self.cuda_graph_kv_indices)
        self._graph_page_tables(bs, req_pool_indices, seq_lens)
        if self.max_kv_splits > 1:
            self._write_graph_plan(bs, int(seq_lens_sum), seq_lens_cpu)
        self._graph_window(bs, req_pool_indices, seq_lens, int(seq_lens_sum), seq_lens_cpu)

    def get_cuda_graph_seq_len_fill_value(self):
        return 1  # triton_backend.py:629 -- padded rows attend one key (slot 0 sink via req_to_token)

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _pool_buffers(forward_batch, layer):
        pool = forward_batch.token_to_kv_pool
        return pool.get_key_buffer(layer.layer_id), pool.get_value_buffer(layer.layer_id)

    @staticmethod
    def _kv_scales(layer):
        """(k_scale, v_scale) floats of an fp8 KV cache layer (radix_attention.py:71-74, loaded by
        quantization/kv_cache.py:17-82); 1.0 when the checkpoint has none (plain cast, as Triton stores it)."""
        ks = getattr(layer, "k_scale_float", None)
        vs = getattr(layer, "v_scale_float", None)
        # a scales file (--quantization-param-path -> load_kv_cache_scales, llama.py:359-378) sets the plain
        # k_scale / v_scale attributes and leaves the *_float ones unset
        # (python floats; a device Parameter is never read here -- that would synchronise inside a capture)
        if ks is None and isinstance(getattr(layer, "k_scale", None), (int, float)) and layer.k_scale > 0:
            ks = float(layer.k_scale)
        if vs is None and isinstance(getattr(layer, "v_scale", None), (int, float)) and layer.v_scale > 0:
            vs = float(layer.v_scale)
        return (1.0 if ks is None else float(ks)), (1.0 if vs is None else float(vs))

    def _save_kv(self, forward_batch, layer, k, v):
        k_buf, v_buf = self._pool_buffers(forward_batch, layer)
        if k_buf.element_size() == 1:            # fp8 pool: divide by the scales and cast, one scatter kernel
            ks, vs = self._kv_scales(layer)
            ops.kv_write_fp8(k_buf, v_buf, forward_batch.out_cache_loc, k, v, ks, vs)
            return
        if k.dtype != k_buf.dtype:
            raise NotImplementedError(f"MiAttnBackend: KV pool dtype {k_buf.dtype} with {k.dtype} activations")
        ops.kv_write(k_buf, v_buf, forward_batch.out_cache_loc, k, v)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache=True, fp8_out_scale=None):
        """`fp8_out_scale` (extension, passed through AttentionBackend.forward's **kwargs): the static input
        scale of the following FP8 linear; the output is then returned already quantised (fp8), bit-identical
        to quantising the bf16 result -- the merge stage writes it directly."""
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        if layer.qk_head_dim != layer.v_head_dim:
            raise NotImplementedError("MiAttnBackend: qk_head_dim != v_head_dim (MLA) is out of scope")
        if save_kv_cache:
            self._save_kv(forward_batch, layer, k, v)
        k_buf, v_buf = self._pool_buffers(forward_batch, layer)
        md = self.forward_metadata
        lw = getattr(layer, "sliding_window_size", None)
        if lw is not None and lw > -1:           # triton_backend.py:711-713
            if md.window is None:
                raise ValueError("sliding-window layer, but model_runner.sliding_window_size was not set")
            md = md.window
        if md.page_indptr is not None:           # page-granular indices: same kernel, one index per page
            fp8_pool = k_buf.element_size() == 1
            ks, vs = self._kv_scales(layer) if fp8_pool else (1.0, 1.0)
            o8 = torch.empty(q.shape, dtype=ops.FP8_DTYPE, device=q.device) if fp8_out_scale is not None else None
            o = q.new_empty(q.shape) if o8 is None else None
            ops.decode_attention_paged(q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k_buf, v_buf, md.kv_indptr,
                                       md.page_indptr, md.page_indices, md.page_size, layer.scaling,
                                       getattr(layer, "logit_cap", 0.0) or 0.0, md.num_kv_splits, md.workspace,
                                       o=None if o is None else o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                                       o_fp8=o8, o_scale=fp8_out_scale, k_scale=ks, v_scale=vs,
                                       split_chunk=md.split_chunk, work=md.work)
            return o if o8 is None else o8
        if k_buf.element_size() == 1:            # fp8 KV cache
            ks, vs = self._kv_scales(layer)
            q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
            cap = getattr(layer, "logit_cap", 0.0) or 0.0
            if fp8_out_scale is not None:
                o8 = torch.empty(q.shape, dtype=ops.FP8_DTYPE, device=q.device)
                ops.decode_attention_fp8kv(q3, k_buf, v_buf, md.kv_indptr, md.kv_indices, layer.scaling, ks, vs, cap,
                                           md.num_kv_splits, md.workspace, o_fp8=o8, o_scale=fp8_out_scale,
                                           split_chunk=md.split_chunk, work=md.work)
                return o8
            o = q.new_empty(q.shape)
            ops.decode_attention_fp8kv(q3, k_buf, v_buf, md.kv_indptr, md.kv_indices, layer.scaling, ks, vs, cap,
                                       md.num_kv_splits, md.workspace, o=o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                                       split_chunk=md.split_chunk, work=md.work)
            return o
        if fp8_out_scale is not None:
            o8 = torch.empty(q.shape, dtype=ops.FP8_DTYPE, device=q.device)
            ops.decode_attention_fp8out(q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k_buf, v_buf, o8,
                                        fp8_out_scale, md.kv_indptr, md.kv_indices, layer.scaling,
                                        getattr(layer, "logit_cap", 0.0) or 0.0, md.num_kv_splits, md.workspace,
                                        split_chunk=md.split_chunk, work=md.work)
            return o8
        o = q.new_empty(q.shape)
        ops.decode_attention(q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k_buf, v_buf,
                             o.view(-1, layer.tp_q_head_num, layer.v_head_dim), md.kv_indptr, md.kv_indices,
                             layer.scaling, getattr(layer, "logit_cap", 0.0) or 0.0, md.num_kv_splits, md.workspace,
                             split_chunk=md.split_chunk, work=md.work)
        return o

    def extend_rotates_q(self, layer, fp8_out_scale) -> bool:
        """True when forward_extend(..., fp8_out_scale, q_rope=...) can rotate Q inside the attention kernel for the
        metadata at hand (long-extend kernel, plain EXTEND batch on a T-typed pool): the caller then runs the rope
        kernel on k alone (ops.rope_neox_k_)."""
        md = self.forward_metadata
        if fp8_out_scale is None or md is None or md.num_kv_splits > 1 or md.custom_mask is not None:
            return False
        window = getattr(layer, "sliding_window_size", -1)
        window = -1 if window is None else int(window)
        return ops.extend_fp8_out_is_fused(layer.qk_head_dim, md.max_extend_len, getattr(layer, "logit_cap", 0.0) or 0.0,
                                           window) and layer.qk_head_dim == layer.v_head_dim

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache=True, fp8_out_scale=None, q_rope=None):
        """`fp8_out_scale` (extension, as in forward_decode): the static input scale of the following FP8 linear; the
        output is then returned already quantised (fp8 [tokens, Hq * Dv]), bit-identical to quantising the T-typed
        result.  Long extends write it from the attention epilogue (mi_extend_attn_fp8out); the speculative / split /
        fp8-pool forms quantise their T-typed output in a second launch.  `q_rope` = (positions, cos_sin_cache_t): q
        is unrotated and the kernel applies NeoX RoPE as it loads Q -- only where extend_rotates_q() said so."""
        if layer.qk_head_dim != layer.v_head_dim:
            raise NotImplementedError("MiAttnBackend: qk_head_dim != v_head_dim (MLA) is out of scope")
        if save_kv_cache:
            self._save_kv(forward_batch, layer, k, v)
        k_buf, v_buf = self._pool_buffers(forward_batch, layer)
        md = self.forward_metadata
        if fp8_out_scale is not None:
            cap = getattr(layer, "logit_cap", 0.0) or 0.0
            window = getattr(layer, "sliding_window_size", -1)
            window = -1 if window is None else int(window)
            plain = md.num_kv_splits <= 1 and md.custom_mask is None and k_buf.element_size() != 1
            if q_rope is not None and not (plain and self.extend_rotates_q(layer, fp8_out_scale)):
                raise ValueError("MiAttnBackend: q_rope on a batch whose attention kernel cannot rotate Q")
            if not plain:
                o = self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=False)
                return ops.fp8_quant_per_tensor(o, fp8_out_scale)[0]
            causal = not (getattr(layer, "is_cross_attention", False)
                          or getattr(getattr(layer, "attn_type", None), "value", "decoder") == "encoder_only")
            fused = ops.extend_fp8_out_is_fused(layer.qk_head_dim, md.max_extend_len, cap, window)
            o = None if fused else q.new_empty(q.shape)
            o8 = torch.empty(q.shape, dtype=ops.FP8_DTYPE, device=q.device)
            ops.extend_attention_fp8out(q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
                                        k.reshape(-1, layer.tp_k_head_num, layer.qk_head_dim),
                                        v.reshape(-1, layer.tp_v_head_num, layer.v_head_dim), o8, fp8_out_scale, k_buf,
                                        v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices, md.max_extend_len,
                                        layer.scaling, cap, causal, window if window > 0 else -1,
                                        md.page_indptr, md.page_indices if md.page_indptr is not None else None,
                                        md.page_size if md.page_indptr is not None else 1,
                                        None if o is None else o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                                        None if q_rope is None else q_rope[0], None if q_rope is None else q_rope[1])
            return o8
        if q_rope is not None:
            raise ValueError("MiAttnBackend: q_rope needs fp8_out_scale (the fused prefill form)")
        o = q.new_empty(q.shape)
        causal = not (getattr(layer, "is_cross_attention", False)
                      or getattr(getattr(layer, "attn_type", None), "value", "decoder") == "encoder_only")
        window = getattr(layer, "sliding_window_size", -1)
        window = -1 if window is None else int(window)
        q3 = q.view(-1, layer.tp_q_head_num, layer.qk_head_dim)
        k3 = k.reshape(-1, layer.tp_k_head_num, layer.qk_head_dim)
        v3 = v.reshape(-1, layer.tp_v_head_num, layer.v_head_dim)
        o3 = o.view(-1, layer.tp_q_head_num, layer.v_head_dim)
        cap = getattr(layer, "logit_cap", 0.0) or 0.0
        fp8_pool = k_buf.element_size() == 1
        if md.num_kv_splits > 1 or (fp8_pool and md.custom_mask is not None):
            # short extends over long prefixes (speculative verify, chunk tails): split-KV form of the same kernel;
            # it is also the general entry point (fp8 pool + tree mask)
            ks, vs = self._kv_scales(layer) if fp8_pool else (1.0, 1.0)
            ops.extend_attention_splitkv(q3, k3, v3, o3, k_buf, v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices,
                                         md.max_extend_len, layer.scaling, md.num_kv_splits, md.workspace, cap,
                                         causal, window if window > 0 else -1, ks, vs, md.custom_mask, md.mask_indptr,
                                         True)
            return o
        if md.custom_mask is not None:
            ops.extend_attention_masked(q3, k3, v3, o3, k_buf, v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices,
                                        md.custom_mask, md.mask_indptr, md.max_extend_len, layer.scaling, cap, True,
                                        window if window > 0 else -1)
            return o
        if fp8_pool:
            # fp8 pool: the cached prefix is read (and converted) from the pool; the new tokens are attended from
            # the T-typed k/v arguments, exactly as in the bf16 case (they were just written to the pool in fp8)
            ks, vs = self._kv_scales(layer)
            ops.extend_attention_fp8kv(q3, k3, v3, o3, k_buf, v_buf, ks, vs, md.qo_indptr, md.kv_indptr, md.kv_indices,
                                       md.max_extend_len, layer.scaling, cap, causal, window if window > 0 else -1)
            return o
        if md.page_indptr is not None:
            ops.extend_attention_paged(q3, k3, v3, o3, k_buf, v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices,
                                       md.page_indptr, md.page_indices, md.page_size, md.max_extend_len, layer.scaling,
                                       cap, causal, window if window > 0 else -1)
            return o
        ops.extend_attention(q3, k3, v3, o3, k_buf, v_buf, md.qo_indptr, md.kv_indptr, md.kv_indices, md.max_extend_len,
                             layer.scaling, cap, causal, window if window > 0 else -1)
        return o

    def support_triton(self):
        return False  # scheduler-side helpers then use their torch forms (srt/utils.py:204-205)
