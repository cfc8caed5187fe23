"""Measurement / parity harness: the thinnest Llama-shaped caller of the hot path.

It plays the role of the reference's MockModelRunner (python/sglang/test/attention/
test_flashattn_backend.py:15-69) plus the per-layer call order of LlamaDecoderLayer
(python/sglang/srt/models/llama.py:94-98,180-191,245-268): the plugins are driven only through
their reference interfaces -- `quant_method.create_weights / process_weights_after_loading /
apply` and `attn_backend.init_forward_metadata / forward` -- so what is timed here is what the
unmodified srt model code would call.  Nothing in this file computes: it allocates, wires and
calls (torch is used for the bf16 lm_head GEMM, embedding gather, argmax and collectives).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import List, Optional

import torch

from . import ops
from ._compat import ForwardMode
from .attention_backend import MiAttnBackend
from .parallel import tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce


@dataclass
class ModelShape:
    name: str
    hidden: int
    layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate: int
    vocab: int
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    context_len: int = 8192


LLAMA3_8B = ModelShape("Llama-3-8B", 4096, 32, 32, 8, 128, 14336, 128256)
LLAMA2_7B = ModelShape("Llama-2-7B", 4096, 32, 32, 32, 128, 11008, 32000, rope_theta=10000.0, context_len=4096)
LLAMA3_70B = ModelShape("Llama-3-70B", 8192, 80, 64, 8, 128, 28672, 128256)
TINY = ModelShape("tiny-llama", 256, 2, 8, 2, 64, 512, 1000, context_len=512)


class AttnLayer:
    """The attributes of RadixAttention a backend reads (layers/radix_attention.py:37-80)."""

    def __init__(self, num_heads, head_dim, scaling, num_kv_heads, layer_id, logit_cap=0.0):
        self.tp_q_head_num, self.tp_k_head_num, self.tp_v_head_num = num_heads, num_kv_heads, num_kv_heads
        self.head_dim = self.qk_head_dim = self.v_head_dim = head_dim
        self.scaling, self.layer_id, self.logit_cap = scaling, layer_id, logit_cap
        self.sliding_window_size = -1
        self.is_cross_attention = False
        self.k_scale = self.v_scale = None


def make_kv_pool(size, layer_num, head_num, head_dim, dtype, device, fill_random=False, seed=0, kv_dtype=None,
                 page_size=1):
    """MHATokenToKVPool (mem_cache/memory_pool.py:176-260), optionally pre-filled with N(0,1) data so that a
    steady-state decode batch can be benchmarked without running the prefill first."""
    from .mem_cache import MHATokenToKVPool
    pool = MHATokenToKVPool(size, page_size, kv_dtype or dtype, head_num, head_dim, layer_num, device)
    if fill_random:
        g = torch.Generator(device=device).manual_seed(seed)
        for bufs in zip(pool.k_buffer, pool.v_buffer):
            for t in bufs:
                if t.dtype == torch.uint8:      # fp8 pool: N(0,1) values stored as e4m3fn bytes, filled in chunks
                    rows = t.shape[0]
                    step = max(1, (1 << 26) // (head_num * head_dim))
                    for r0 in range(0, rows, step):
                        blk = torch.empty(min(step, rows - r0), head_num, head_dim, dtype=torch.float32, device=device)
                        blk.normal_(generator=g)
                        t[r0:r0 + blk.shape[0]] = blk.to(torch.float8_e4m3fn).view(torch.uint8)
                else:
                    t.normal_(generator=g)
    return pool


class Linear(torch.nn.Module):
    """A LinearBase-shaped module: the quant method owns weights and forward (linear.py:208-229)."""

    def __init__(self, in_features, out_partition_sizes: List[int], quant_method, params_dtype, bias=False):
        super().__init__()
        self.output_partition_sizes = out_partition_sizes
        self.quant_method = quant_method
        self.bias = None
        quant_method.create_weights(self, in_features, out_partition_sizes, in_features, sum(out_partition_sizes),
                                    params_dtype, weight_loader=None)

    calibrating = False   # class-wide switch used by LlamaStack.calibrate_static_input_scales

    def fused_quant_scale(self):
        """Static input scale when the producer may hand this linear an fp8 activation directly."""
        if Linear.calibrating or not Linear.fuse_producer_quant:
            return None
        f = getattr(self.quant_method, "static_input_scale", None)
        return f(self) if f is not None else None

    def forward_prequantized(self, qx, out_dtype, out=None):
        if out is None:
            return self.quant_method.apply_prequantized(self, qx, out_dtype, self.bias)
        return self.quant_method.apply_prequantized(self, qx, out_dtype, self.bias, out)

    fuse_producer_quant = True   # norm/activation kernels emit fp8 for static-scale linears

    def forward(self, x):
        if Linear.calibrating and getattr(self, "input_scale", None) is not None:
            # one-off calibration of a static per-tensor activation scale (what an FP8 checkpoint ships):
            # scale = absmax / 448, per_tensor_quant_fp8.cu:42 -- torch here, never on the timed path
            self.input_scale.data.copy_((x.float().abs().max() / 448.0).reshape(1))
        return self.quant_method.apply(self, x, self.bias)


def make_runner(shape: ModelShape, max_reqs, ctx, pool_tokens, dtype, device, tp=1, fill_kv=False, seed=0,
                max_kv_splits=8, kv_dtype=None, page_size=1):
    """MockModelRunner-equivalent namespace (test_flashattn_backend.py:15-69, SURVEY 8b list)."""
    Hkv = max(1, shape.num_kv_heads // tp)
    mc = SimpleNamespace(num_attention_heads=shape.num_heads, num_key_value_heads=shape.num_kv_heads,
                         context_len=ctx, head_dim=shape.head_dim, is_encoder_decoder=False,
                         get_num_kv_heads=lambda tp_size: max(1, shape.num_kv_heads // tp_size))
    from .mem_cache import ReqToTokenPool, TokenToKVPoolAllocator
    r2t = ReqToTokenPool(max_reqs, ctx, device)
    pool = make_kv_pool(pool_tokens, shape.layers, Hkv, shape.head_dim, dtype, device, fill_random=fill_kv, seed=seed,
                        kv_dtype=kv_dtype, page_size=page_size)
    allocator = TokenToKVPoolAllocator(pool_tokens, pool.dtype, device, pool)
    sa = SimpleNamespace(triton_attention_num_kv_splits=max_kv_splits, page_size=page_size,
                         speculative_num_draft_tokens=None, speculative_num_steps=None)
    return SimpleNamespace(device=device, dtype=dtype, model_config=mc, req_to_token_pool=r2t,
                           token_to_kv_pool=pool, token_to_kv_pool_allocator=allocator, sliding_window_size=None,
                           server_args=sa, tp_size=tp, gpu_id=0,
                           kv_cache_dtype="auto" if kv_dtype is None else "fp8_e4m3", page_size=page_size)


def rope_cache(head_dim, max_pos, base, device):
    """RotaryEmbedding._compute_cos_sin_cache (layers/rotary_embedding.py:108-125), fp32 [max_pos, D]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float, device=device) / head_dim))
    t = torch.arange(max_pos, dtype=torch.float, device=device)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).contiguous()


class LlamaStack:
    """qkv_proj -> rope -> attention -> o_proj -> (add+norm) -> gate_up -> silu*mul -> down, L times,
    then final norm + lm_head.  `make_method()` returns a fresh LinearMethodBase per linear."""

    def __init__(self, shape: ModelShape, make_method, dtype, device, tp=1, rank=0, group=None,
                 weight_range=1e-3, seed=1234, weights_cpu_seeded=False, custom_ar=None):
        self.shape, self.dtype, self.device = shape, dtype, device
        self.tp, self.rank, self.group, self.custom_ar = tp, rank, group, custom_ar
        s = shape
        self.Hq, self.Hkv = s.num_heads // tp, max(1, s.num_kv_heads // tp)
        D = s.head_dim
        self.q_size, self.kv_size = self.Hq * D, self.Hkv * D
        inter = s.intermediate // tp
        self.inter = inter
        gen = torch.Generator(device="cpu" if weights_cpu_seeded else device).manual_seed(seed + rank)

        def dummy(*size):
            # model_loader/weight_utils.py:750-779: uniform(-1e-3, 1e-3) dummy weights
            if weights_cpu_seeded:
                t = torch.rand(*size, generator=gen, dtype=torch.float32) * (2 * weight_range) - weight_range
                return t.to(dtype).to(device)
            t = torch.empty(*size, dtype=torch.float32, device=device).uniform_(-weight_range, weight_range, generator=gen)
            return t.to(dtype)

        def linear(k, outs):
            lin = Linear(k, outs, make_method(), dtype).to(device)
            self._init_linear(lin, dummy)
            lin.quant_method.process_weights_after_loading(lin)
            return lin

        self.layers = []
        for i in range(s.layers):
            self.layers.append(SimpleNamespace(
                input_norm=(torch.ones(s.hidden, dtype=dtype, device=device) + dummy(s.hidden)),
                post_norm=(torch.ones(s.hidden, dtype=dtype, device=device) + dummy(s.hidden)),
                qkv=linear(s.hidden, [self.q_size, self.kv_size, self.kv_size]),
                o=linear(self.q_size, [s.hidden]),
                gate_up=linear(s.hidden, [inter, inter]),
                down=linear(inter, [s.hidden]),
                attn=AttnLayer(self.Hq, D, D ** -0.5, self.Hkv, i)))
        self.final_norm = torch.ones(s.hidden, dtype=dtype, device=device) + dummy(s.hidden)
        self.vocab_shard = s.vocab // tp
        self.lm_head = dummy(self.vocab_shard, s.hidden)           # bf16, unquantised (logits_processor.py:423-480)
        self.embed = dummy(s.vocab, s.hidden)                       # token ids -> hidden (plumbing, replicated)
        self.cos_sin = rope_cache(D, s.context_len, s.rope_theta, device)
        # the cache as RotaryEmbedding.forward_native uses it (cast to the activation dtype per call,
        # rotary_embedding.py:150-151): the fused prefill forms read this copy
        self.cos_sin_t = self.cos_sin.to(dtype).contiguous()

    @staticmethod
    def _init_linear(lin, dummy):
        """Fill whatever checkpoint-layout parameters the method created with synthetic data."""
        names = dict(lin.named_parameters())
        if "weight" in names:
            w = names["weight"]
            if w.dtype in (torch.bfloat16, torch.float16):   # fp8 from a bf16 checkpoint: plain weight
                w.data.copy_(dummy(*w.shape))
            else:                                            # serialized FP8 checkpoint: fp8 weight + scales
                qw, ws = ops.fp8_quant_per_tensor(dummy(*w.shape), weight_mode=True)
                w.data.copy_(qw)
                names["weight_scale"].data.copy_(ws.expand_as(names["weight_scale"]))
                if names.get("input_scale") is not None:
                    names["input_scale"].data.fill_(1.0)     # calibrated later
        if "qweight" in names:                     # int4: random nibbles, small scales
            dev = names["qweight"].device
            g = torch.Generator(device=dev).manual_seed(1234)
            for n in ("qweight", "qzeros"):
                p = names[n]
                p.data.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, p.shape, dtype=torch.int32, device=dev, generator=g))
            if type(lin.quant_method).__name__.startswith("GPTQ"):
                names["qzeros"].data.bitwise_and_(0x66666666)  # stored zero <= 14 so z+1 is a nibble
            sc = names["scales"]
            sc.data.copy_((torch.rand(sc.shape, device=dev, generator=g) * 1e-2 / 8).to(sc.dtype))

    def _all_reduce(self, x):
        return tensor_model_parallel_all_reduce(x, self.tp, self.group, self.custom_ar)   # linear.py:1376-1378

    def _row_parallel_add_norm(self, lin, qx, residual, norm_w, q_scale, want_out=False):
        """Row-parallel linear -> all-reduce -> add + RMSNorm (+ fp8 quant for the next linear), TP > 1
        (linear.py:1360-1382 then layernorm.py:128-146).  With the native all-reduce: the GEMM writes straight into
        the registered staging buffer and ONE kernel reduces over the ranks, adds the residual, norms and quantises
        (3 launches: GEMM partials, slab reduce, fused all-reduce); otherwise GEMM, RCCL/gloo all-reduce, norm kernel.
        Returns (out or None, fp8 or None); residual is updated in place.  Same bits either way."""
        M, H = qx.shape[0], self.shape.hidden
        ca = self.custom_ar
        if self.comm_disabled:      # timing only (bench: exposed communication = step - this): every rank keeps its partial sum
            h = lin.forward_prequantized(qx, self.dtype)
            if q_scale is not None:
                return None, ops.rmsnorm_fp8(h, norm_w, self.shape.rms_eps, q_scale, residual=residual)
            return ops.rmsnorm(h, norm_w, self.shape.rms_eps, residual=residual), None
        if ca is not None and ca.should_fuse_norm(M, H, self.dtype):
            h = lin.forward_prequantized(qx, self.dtype, out=ca.staging((M, H), self.dtype))
            return ca.all_reduce_add_rmsnorm(h, residual, norm_w, self.shape.rms_eps, q_scale=q_scale,
                                             want_out=want_out or q_scale is None)
        h = self._all_reduce(lin.forward_prequantized(qx, self.dtype))
        out = q = None
        if q_scale is not None:
            q = ops.rmsnorm_fp8(h, norm_w, self.shape.rms_eps, q_scale, residual=residual)
        else:
            out = ops.rmsnorm(h, norm_w, self.shape.rms_eps, residual=residual)
        return out, q

    def calibrate_static_input_scales(self, hidden, positions, fb, backend):
        """Give every static-activation linear a realistic `input_scale` (amax/448 of one forward pass),
        standing in for the calibrated scales a serialized FP8 checkpoint carries."""
        Linear.calibrating = True
        try:
            self.forward(hidden, positions, fb, backend)
        finally:
            Linear.calibrating = False

    fuse_decode_layer = True   # linear + consumer fused forms on decode batches
    comm_disabled = False      # bench only: drop the TP all-reduces to time the step without communication

    def _fused_decode_ok(self, hidden, fb):
        if not (LlamaStack.fuse_decode_layer and Linear.fuse_producer_quant) or Linear.calibrating:
            return False
        if not fb.forward_mode.is_decode() or hidden.shape[0] > 512:
            return False
        M = hidden.shape[0]
        for L in self.layers:
            for lin in (L.qkv, L.o, L.gate_up, L.down):
                ok = getattr(lin.quant_method, "fused_decode_ok", None)
                if ok is None or not ok(lin, M):
                    return False
        return True

    def _fused_decode16_ok(self, hidden, fb):
        """int4 linears (16-bit activations): every linear has the fused-consumer forms at this batch size, TP = 1,
        and the KV pool holds the activation dtype."""
        if not LlamaStack.fuse_decode_layer or self.tp != 1 or not fb.forward_mode.is_decode() or hidden.shape[0] > 128:
            return False
        if fb.token_to_kv_pool.get_key_buffer(0).dtype != self.dtype:
            return False
        M = hidden.shape[0]
        for L in self.layers:
            for lin in (L.qkv, L.o, L.gate_up, L.down):
                ok = getattr(lin.quant_method, "fused_decode16_ok", None)
                if ok is None or not ok(lin, M):
                    return False
        return True

    def forward_decode_fused16(self, hidden, positions, fb, backend):
        """The layer sequence of forward() for linears with 16-bit activations (AWQ / GPTQ int4), each linear fused
        with its consumer: norm (first layer) | qkv + rope + kv-write | attention | o + add + norm | gate_up + silu*mul |
        down + add + norm(next layer) -- 7 launches per layer (+ the split merge) instead of 17, bit-identical."""
        s, D = self.shape, self.shape.head_dim
        pool = fb.token_to_kv_pool
        residual = hidden
        x = ops.rmsnorm(hidden, self.layers[0].input_norm, s.rms_eps)
        for i, L in enumerate(self.layers):
            q = L.qkv.quant_method.apply_rope_kvwrite16(L.qkv, x, positions, self.cos_sin, pool.get_key_buffer(i),
                                                        pool.get_value_buffer(i), fb.out_cache_loc, self.Hq, self.Hkv, D)
            a = backend.forward(q, None, None, L.attn, fb, save_kv_cache=False)
            x = L.o.quant_method.apply_add_rmsnorm16(L.o, a, residual, L.post_norm, s.rms_eps)
            act = L.gate_up.quant_method.apply_silu_mul16(L.gate_up, x)
            last = i + 1 == len(self.layers)
            nw = self.final_norm if last else self.layers[i + 1].input_norm
            x = L.down.quant_method.apply_add_rmsnorm16(L.down, act, residual, nw, s.rms_eps)
        logits = torch.matmul(x, self.lm_head.t())
        return tensor_model_parallel_all_gather(logits, self.tp, self.group)

    def forward_decode_fused(self, hidden, positions, fb, backend, return_hidden=False):
        """The same layer sequence as forward() with each FP8 linear fused with its consumer:
        norm+quant | qkv+rope+kv-write | attention(+quant) | o+add+norm+quant | gate_up+silu*mul+quant |
        down+add+norm(next layer)+quant  -- 7 launches per layer instead of 15, bit-identical results.
        With TP>1 the row-parallel o/down outputs need the all-reduce first: there the add + norm + quant is fused
        into the all-reduce kernel instead (_row_parallel_add_norm)."""
        s, D = self.shape, self.shape.head_dim
        pool = fb.token_to_kv_pool
        residual = hidden
        L0 = self.layers[0]
        qx = ops.rmsnorm_fp8(hidden, L0.input_norm, s.rms_eps, L0.qkv.input_scale)
        x = None
        fp8_pool = pool.get_key_buffer(0).element_size() == 1
        for i, L in enumerate(self.layers):
            if fp8_pool:   # the fused rope + kv-write consumer stores T-typed rows: fp8 pools take the scatter kernel
                qkv = L.qkv.forward_prequantized(qx, self.dtype)
                q, k, v = (qkv[:, : self.q_size], qkv[:, self.q_size: self.q_size + self.kv_size],
                           qkv[:, self.q_size + self.kv_size:])
                ops.rope_neox_(q, k, positions, self.cos_sin, D)
                a8 = backend.forward(q, k.reshape(-1, self.Hkv, D), v.reshape(-1, self.Hkv, D), L.attn, fb,
                                     fp8_out_scale=L.o.input_scale)
            else:
                q = L.qkv.quant_method.apply_rope_kvwrite(L.qkv, qx, positions, self.cos_sin, pool.get_key_buffer(i),
                                                          pool.get_value_buffer(i), fb.out_cache_loc, self.Hq, self.Hkv, D)
                a8 = backend.forward(q, None, None, L.attn, fb, save_kv_cache=False, fp8_out_scale=L.o.input_scale)
            if self.tp == 1:
                _, qx = L.o.quant_method.apply_add_rmsnorm(L.o, a8, residual, L.post_norm, s.rms_eps,
                                                           L.gate_up.input_scale)
            else:
                _, qx = self._row_parallel_add_norm(L.o, a8, residual, L.post_norm, L.gate_up.input_scale)
            act8 = L.gate_up.quant_method.apply_silu_mul(L.gate_up, qx, L.down.input_scale, self.dtype)
            last = i + 1 == len(self.layers)
            nw = self.final_norm if last else self.layers[i + 1].input_norm
            ns = None if last else self.layers[i + 1].qkv.input_scale
            if self.tp == 1:
                x, qx = L.down.quant_method.apply_add_rmsnorm(L.down, act8, residual, nw, s.rms_eps, ns)
            else:
                x, qx = self._row_parallel_add_norm(L.down, act8, residual, nw, ns)
        if return_hidden:          # final-norm output; the caller runs the lm_head (two-micro-batch step: once, after the join)
            return x
        logits = torch.matmul(x, self.lm_head.t())
        return tensor_model_parallel_all_gather(logits, self.tp, self.group)

    def forward_decode_two_batch(self, hidden, positions, fbs, backends, streams, custom_ars=None, return_hidden=False):
        """Two-micro-batch overlap of a decode step (the behaviour of two_batch_overlap.py:361-615 for a dense model):
        the batch is split into halves A and B (`fbs`, `backends`: one ForwardBatch + attention backend each, rows
        [0, nA) and [nA, nA + nB) of `hidden`), each half runs the whole fused layer sequence on its own HIP stream,
        and the two streams only meet at the start and at the end of the step.  With TP > 1 every half has its own
        native all-reduce communicator (`custom_ars`: separate staging buffer and barrier flags), so the row-parallel
        all-reduce of half A -- a latency-bound kernel on at most 64 workgroups that mostly waits for its peers --
        runs while half B's attention and GEMMs keep the rest of the chip busy, and vice versa.
        Every kernel of the step computes rows independently, so the logits are bit-identical to the serial step.
        The lm_head (a library GEMM, outside the hot path) and the vocabulary all-gather run ONCE on the joined stream
        over the whole batch (one M = B GEMM reads the lm_head weights once instead of twice).
        INVARIANT of the per-stream region: only this repository's kernels, each with scratch of its own stream.
        Round 2 saw the capture of this step hang twice (gpurun_out/tbo_a.err: the watchdog found the host in
        torch.cuda.synchronize() under the capture barrier) while each half still ran its own lm_head GEMM and while
        ops._gemm_workspace handed BOTH streams the same split-K slab buffer; both were changed in the same commit, so
        which of the two removed the hang was never established.  Neither can come back silently:
        the fused decode path is asserted below (no library GEMM can run in it) and the slab scratch is keyed by
        stream (asserted below as well)."""
        nA = fbs[0].batch_size
        cur = torch.cuda.current_stream()
        outs = []
        saved = self.custom_ar
        assert streams[0] != streams[1] and cur not in streams, "two-batch overlap needs two side streams"
        for i, (lo, hi) in enumerate(((0, nA), (nA, hidden.shape[0]))):
            if not self._fused_decode_ok(hidden[lo:hi], fbs[i]):
                raise RuntimeError("forward_decode_two_batch: the per-stream region must be the fused FP8 decode path "
                                   "(a library GEMM on two streams side by side is not allowed here)")
        try:
            for i, (lo, hi) in enumerate(((0, nA), (nA, hidden.shape[0]))):
                streams[i].wait_stream(cur)
                with torch.cuda.stream(streams[i]):
                    ws_probe = ops._gemm_workspace(16, hidden.device)
                    assert i == 0 or ws_probe.data_ptr() != ws_first, "split-K scratch must be private to a stream"
                    ws_first = ws_probe.data_ptr()
                    if custom_ars is not None:
                        self.custom_ar = custom_ars[i]
                    backends[i].init_forward_metadata(fbs[i])
                    outs.append(self.forward_decode_fused(hidden[lo:hi], positions[lo:hi], fbs[i], backends[i],
                                                          return_hidden=True))
        finally:
            self.custom_ar = saved
        for st in streams:
            cur.wait_stream(st)
        for o in outs:
            o.record_stream(cur)       # allocated on a side stream, consumed on the joined one
        if return_hidden:              # final-norm output of the whole batch (tests capture the step without the lm_head)
            return torch.cat(outs, dim=0)
        logits = torch.matmul(torch.cat(outs, dim=0), self.lm_head.t())
        return tensor_model_parallel_all_gather(logits, self.tp, self.group)

    def forward(self, hidden, positions, fb, backend, last_token_logits=None):
        """hidden [T, H] -> logits [T, vocab]; follows llama.py:245-268 with the fused add+norm form.
        last_token_logits = extend_seq_lens: logits only for each request's last token, as
        LogitsProcessor does for extend batches (logits_processor.py:308-330)."""
        s = self.shape
        if last_token_logits is None and self._fused_decode_ok(hidden, fb):
            return self.forward_decode_fused(hidden, positions, fb, backend)
        if last_token_logits is None and self._fused_decode16_ok(hidden, fb):
            return self.forward_decode_fused16(hidden, positions, fb, backend)
        residual = None
        for L in self.layers:
            first = residual is None
            if first:
                residual = hidden
            qs = L.qkv.fused_quant_scale()
            if qs is not None:
                qkv = L.qkv.forward_prequantized(
                    ops.rmsnorm_fp8(hidden, L.input_norm, s.rms_eps, qs, residual=None if first else residual),
                    self.dtype)
            else:
                x = ops.rmsnorm(hidden, L.input_norm, s.rms_eps, residual=None if first else residual)
                qkv = L.qkv(x)
            q, k, v = qkv[:, : self.q_size], qkv[:, self.q_size: self.q_size + self.kv_size], qkv[:, self.q_size + self.kv_size:]
            qs_o = L.o.fused_quant_scale()
            # prefill: where this batch's attention kernel can rotate Q as it loads it (long extends, fp8 o_proj input),
            # only k goes through the rope kernel -- its read + write of q (4/5 of its traffic) disappears
            q_rope = None
            if (qs_o is not None and LlamaStack.fuse_decode_layer and not fb.forward_mode.is_decode()
                    and fb.token_to_kv_pool.get_key_buffer(L.attn.layer_id).dtype == self.dtype
                    and getattr(backend, "extend_rotates_q", lambda *_: False)(L.attn, qs_o)):
                ops.rope_neox_k_(k, positions, self.cos_sin, s.head_dim)
                q_rope = (positions, self.cos_sin_t)
            else:
                ops.rope_neox_(q, k, positions, self.cos_sin, s.head_dim)
            if qs_o is not None:     # attention hands o_proj its fp8 input (decode: the split merge; extend: the epilogue)
                a8 = backend.forward(q, k.reshape(-1, self.Hkv, s.head_dim), v.reshape(-1, self.Hkv, s.head_dim),
                                     L.attn, fb, fp8_out_scale=qs_o,
                                     **({"q_rope": q_rope} if q_rope is not None else {}))
                hidden = self._all_reduce(L.o.forward_prequantized(a8, self.dtype))
            else:
                a = backend.forward(q, k.reshape(-1, self.Hkv, s.head_dim), v.reshape(-1, self.Hkv, s.head_dim),
                                    L.attn, fb)
                hidden = self._all_reduce(L.o(a))
            qs = L.gate_up.fused_quant_scale()
            qs_down = L.down.fused_quant_scale()
            silu_ok = getattr(L.gate_up.quant_method, "fused_silu_ok", None)
            if (qs is not None and qs_down is not None and LlamaStack.fuse_decode_layer and silu_ok is not None
                    and silu_ok(L.gate_up, hidden.shape[0])):
                # gate_up GEMM with SiLU*mul + FP8 quant in its epilogue (prefill: the tile kernel's; the [T, 2I]
                # intermediate is never written) -- bit-identical to the two-step form below
                act8 = L.gate_up.quant_method.apply_silu_mul(
                    L.gate_up, ops.rmsnorm_fp8(hidden, L.post_norm, s.rms_eps, qs, residual=residual), qs_down, self.dtype)
                hidden = self._all_reduce(L.down.forward_prequantized(act8, self.dtype))
                continue
            if qs is not None:
                gu = L.gate_up.forward_prequantized(
                    ops.rmsnorm_fp8(hidden, L.post_norm, s.rms_eps, qs, residual=residual), self.dtype)
            else:
                gu = L.gate_up(ops.rmsnorm(hidden, L.post_norm, s.rms_eps, residual=residual))
            qs = L.down.fused_quant_scale()
            if qs is not None:
                hidden = self._all_reduce(L.down.forward_prequantized(ops.silu_and_mul_fp8(gu, qs), self.dtype))
            else:
                hidden = self._all_reduce(L.down(ops.silu_and_mul(gu)))
        x = ops.rmsnorm(hidden, self.final_norm, s.rms_eps, residual=residual)
        if last_token_logits is not None:
            x = x[torch.cumsum(last_token_logits.to(torch.int64), 0) - 1]
        logits = torch.matmul(x, self.lm_head.t())
        return tensor_model_parallel_all_gather(logits, self.tp, self.group)   # logits_processor.py:464-477


def split_decode_batch(fb, backend_b, n_first):
    """The two halves of a decode ForwardBatch for forward_decode_two_batch: requests [0, n_first) keep `fb.attn_backend`,
    the rest get `backend_b` (each backend owns the metadata of its half)."""
    halves = []
    for lo, hi, be in ((0, n_first, fb.attn_backend), (n_first, fb.batch_size, backend_b)):
        lens_cpu = fb.seq_lens_cpu[lo:hi]
        halves.append(SimpleNamespace(forward_mode=fb.forward_mode, batch_size=hi - lo,
                                      req_pool_indices=fb.req_pool_indices[lo:hi].contiguous(),
                                      seq_lens=fb.seq_lens[lo:hi].contiguous(), seq_lens_sum=int(lens_cpu.sum()),
                                      seq_lens_cpu=lens_cpu, out_cache_loc=fb.out_cache_loc[lo:hi].contiguous(),
                                      req_to_token_pool=fb.req_to_token_pool, token_to_kv_pool=fb.token_to_kv_pool,
                                      attn_backend=be, spec_info=None, positions=fb.positions[lo:hi].contiguous()))
    return halves


def make_decode_batch(runner, backend, batch, seq_len, device, scattered=True, seed=0, ragged=None):
    """A decode ForwardBatch at steady state: every request already holds `seq_len` tokens
    (the token being generated included, schedule_batch.py:1541-1546), slots scattered over the
    pool by a random permutation (worst-case gather) or contiguous."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lens = torch.full((batch,), seq_len, dtype=torch.int64) if ragged is None else ragged.to(torch.int64)
    total = int(lens.sum())
    P = int(getattr(runner, "page_size", 1) or 1)
    r2t = runner.req_to_token_pool.req_to_token
    out_loc = []
    if P > 1:
        # paged pool (PagedTokenToKVPoolAllocator, allocator.py:407-543): every request owns whole pages, page 0 is the
        # padding sink, the slots inside a page are consecutive; the pages themselves are scattered (or in order)
        npages = [-(-int(n) // P) for n in lens]
        assert (sum(npages) + 1) * P <= runner.token_to_kv_pool.size + P
        pages = (torch.randperm(sum(npages), generator=g) if scattered else torch.arange(sum(npages))) + 1
        off = 0
        for i in range(batch):
            L = int(lens[i])
            sl = (pages[off: off + npages[i]].view(-1, 1) * P + torch.arange(P).view(1, -1)).reshape(-1)[:L]
            r2t[i, :L] = sl.to(torch.int32).to(device)
            out_loc.append(int(sl[L - 1]))
            off += npages[i]
    else:
        assert total <= runner.token_to_kv_pool.size
        slots = (torch.randperm(total, generator=g) if scattered else torch.arange(total)) + 1
        off = 0
        for i in range(batch):
            L = int(lens[i])
            r2t[i, :L] = slots[off: off + L].to(torch.int32).to(device)
            out_loc.append(int(slots[off + L - 1]))
            off += L
    fb = SimpleNamespace(forward_mode=ForwardMode.DECODE, batch_size=batch,
                         req_pool_indices=torch.arange(batch, dtype=torch.int64, device=device),
                         seq_lens=lens.to(device), seq_lens_sum=total, seq_lens_cpu=lens,
                         out_cache_loc=torch.tensor(out_loc, dtype=torch.int64, device=device),
                         req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                         attn_backend=backend, spec_info=None,
                         positions=(lens - 1).to(device))
    return fb


def make_extend_batch(runner, backend, prefix_lens, extend_lens, device, seed=0, req_offset=0, slot_offset=0):
    """An EXTEND ForwardBatch (schedule_batch.py:1119-1309): prefix slots already in req_to_token,
    new tokens get fresh slots written to req_to_token[pre:seq] and out_cache_loc."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = len(prefix_lens)
    lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    total = sum(lens)
    assert slot_offset + total <= runner.token_to_kv_pool.size
    slots = torch.randperm(total, generator=g) + 1 + slot_offset     # request rows req_offset.., a slot range of its own
    r2t = runner.req_to_token_pool.req_to_token
    off, out_loc, pos = 0, [], []
    for i in range(B):
        r2t[req_offset + i, : lens[i]] = slots[off: off + lens[i]].to(torch.int32).to(device)
        out_loc.append(slots[off + prefix_lens[i]: off + lens[i]])
        pos.append(torch.arange(prefix_lens[i], lens[i]))
        off += lens[i]
    return SimpleNamespace(forward_mode=ForwardMode.EXTEND, batch_size=B,
                           req_pool_indices=torch.arange(req_offset, req_offset + B, dtype=torch.int64, device=device),
                           seq_lens=torch.tensor(lens, dtype=torch.int64, device=device), seq_lens_sum=total,
                           seq_lens_cpu=torch.tensor(lens, dtype=torch.int64),
                           extend_prefix_lens=torch.tensor(prefix_lens, dtype=torch.int32, device=device),
                           extend_seq_lens=torch.tensor(extend_lens, dtype=torch.int32, device=device),
                           extend_prefix_lens_cpu=list(prefix_lens), extend_seq_lens_cpu=list(extend_lens),
                           out_cache_loc=torch.cat(out_loc).to(torch.int64).to(device),
                           req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                           attn_backend=backend, spec_info=None, positions=torch.cat(pos).to(device))
