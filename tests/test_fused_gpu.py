"""Fused decode-layer forms vs the unfused call sequence through the same C ABI:
the fused entry points promise BIT-IDENTICAL results (every rounding of the unfused sequence is
reproduced), so every comparison here is torch.equal.  The unfused pieces are themselves pinned to the
oracle in test_quant_gpu.py / test_elementwise_gpu.py / test_attention_gpu.py / test_e2e_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
FP8 = torch.float8_e4m3fn


def _fp8_operands(M, N, K, g, amp=1.0):
    from iaas_sglang_amd import ops
    x = (torch.randn(M, K, generator=g) * amp).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    xs = (x.float().abs().max() / 448.0).reshape(1)
    qx, _ = ops.fp8_quant_per_tensor(x, xs)
    qw, ws = ops.fp8_quant_per_tensor(w, weight_mode=True)
    return qx, qw.t(), xs, ws


def _bits(t):
    return t.view(torch.uint8) if t.dtype == FP8 else t.view(torch.int16)


@pytest.mark.parametrize("M,N,K", [(128, 4096, 4096), (128, 4096, 14336), (37, 4096, 4096), (1, 256, 512), (16, 1024, 256),
                                   (128, 4096, 2048), (128, 4096, 512), (128, 4096, 1792)])     # o / down per rank at TP = 2, 8
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_residual", [True, False])
def test_gemm_add_rmsnorm_fp8_bit_identical(M, N, K, dtype, with_residual):
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    qx, w, xs, ws = _fp8_operands(M, N, K, g)
    res = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(DEV)
    qscale = torch.tensor([0.02], device=DEV)
    # unfused: GEMM -> (add) rmsnorm with fp8 output
    y = ops.fp8_gemm(qx, w, xs, ws, dtype)
    r1 = res.clone()
    out1 = ops.rmsnorm(y, nw, 1e-5, residual=r1 if with_residual else None)
    q1, _ = ops.fp8_quant_per_tensor(out1, qscale)
    r2 = res.clone()
    out2, q2 = ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, r2 if with_residual else None, nw, 1e-5, qscale, want_out=True)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out1), _bits(out2))
    assert torch.equal(_bits(q1), _bits(q2))
    assert torch.equal(_bits(r1), _bits(r2))
    # fp8-only and out-only forms
    _, q3 = ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, res.clone() if with_residual else None, nw, 1e-5, qscale)
    out4, q4 = ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, res.clone() if with_residual else None, nw, 1e-5, None)
    assert torch.equal(_bits(q1), _bits(q3)) and q4 is None and torch.equal(_bits(out1), _bits(out4))


@pytest.mark.parametrize("M,Hq,Hkv,D,K", [(128, 32, 8, 128, 4096), (5, 32, 8, 128, 4096), (64, 8, 2, 64, 256),
                                          (17, 4, 4, 128, 512)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_rope_kvwrite_bit_identical(M, Hq, Hkv, D, K, dtype):
    from iaas_sglang_amd import harness as H, ops
    g = torch.Generator().manual_seed(M + Hq + K)
    N = (Hq + 2 * Hkv) * D
    qx, w, xs, ws = _fp8_operands(M, N, K, g)
    slots = 4 * M + 1
    cache = H.rope_cache(D, 4096, 10000.0, DEV)
    pos = torch.randint(0, 4096, (M,), generator=g).to(DEV)
    loc = (torch.randperm(slots - 1, generator=g)[:M] + 1).to(DEV)
    kc1 = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    vc1 = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    kc2, vc2 = kc1.clone(), vc1.clone()
    qkv = ops.fp8_gemm(qx, w, xs, ws, dtype)
    q1, k1, v1 = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    ops.rope_neox_(q1, k1, pos, cache, D)
    ops.kv_write(kc1, vc1, loc, k1, v1)
    q2 = ops.fp8_gemm_rope_kvwrite(qx, w, xs, ws, pos, cache, kc2, vc2, loc, Hq, Hkv, D)
    torch.cuda.synchronize()
    assert torch.equal(_bits(q1.contiguous()), _bits(q2))
    assert torch.equal(_bits(kc1), _bits(kc2)) and torch.equal(_bits(vc1), _bits(vc2))   # written rows AND untouched rows


@pytest.mark.parametrize("M,I,K", [(128, 14336, 4096), (77, 14336, 4096), (128, 1792, 4096), (3, 512, 256), (16, 64, 128),
                                   (128, 1000 * 8, 512),
                                   (2048, 14336, 1024), (700, 1792, 4096), (513, 128, 128),    # prefill: tile-kernel epilogue
                                   (4300, 2176, 256)])      # 17 x 17 tiles on 256 CUs: persistent loop, ragged last row block
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_silu_mul_fp8_bit_identical(M, I, K, dtype):
    """Covers both routes: the in-kernel epilogue (N large, no split-K: 14336) and the slab consumer."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(M + I + K)
    qx, w, xs, ws = _fp8_operands(M, 2 * I, K, g, amp=2.0)
    qscale = torch.tensor([0.05], device=DEV)
    gu = ops.fp8_gemm(qx, w, xs, ws, dtype)
    q1 = ops.silu_and_mul_fp8(gu, qscale)
    q2 = ops.fp8_gemm_silu_mul(qx, w, xs, ws, qscale, dtype)
    torch.cuda.synchronize()
    if M <= 512:
        assert torch.equal(_bits(q1), _bits(q2))
    else:
        # prefill: same tile kernel, but the unfused call may pick split-K slabs (another fp32 summation order), so a
        # few values land on the neighbouring T / fp8 code
        diff = (_bits(q1) != _bits(q2)).float().mean().item()
        assert diff < 5e-3, diff
        torch.testing.assert_close(q2.float(), q1.float(), rtol=0.13, atol=2 ** -6)
    assert int((_bits(q1) != 0).sum()) > q1.numel() // 4      # not a trivially-zero comparison


@pytest.mark.parametrize("B,Hq,Hkv,D,splits", [(128, 32, 8, 128, 4), (7, 32, 8, 128, 1), (16, 8, 8, 64, 3), (5, 4, 1, 64, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_attention_fp8out_bit_identical(B, Hq, Hkv, D, splits, dtype):
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(B + Hq + D)
    lens = torch.randint(1, 300, (B,), generator=g)
    lens[0] = 1
    total = int(lens.sum())
    kb = torch.randn(total + 1, Hkv, D, generator=g).to(dtype).to(DEV)
    vb = torch.randn(total + 1, Hkv, D, generator=g).to(dtype).to(DEV)
    q = torch.randn(B, Hq, D, generator=g).to(dtype).to(DEV)
    indptr = ops.kv_indptr(lens.to(DEV))
    idx = (torch.randperm(total, generator=g) + 1).to(torch.int32).to(DEV)
    ws = torch.empty(max(ops.decode_workspace_numel(B, Hq, D, splits), 1), dtype=torch.float32, device=DEV)
    scale = torch.tensor([0.01], device=DEV)
    o1 = torch.empty_like(q)
    ops.decode_attention(q, kb, vb, o1, indptr, idx, D ** -0.5, 0.0, splits, ws)
    q1, _ = ops.fp8_quant_per_tensor(o1.view(B, Hq * D), scale)
    o8 = torch.empty(B, Hq * D, dtype=FP8, device=DEV)
    o2 = torch.empty_like(q)
    ops.decode_attention_fp8out(q, kb, vb, o8, scale, indptr, idx, D ** -0.5, 0.0, splits, ws, o=o2)
    o8b = torch.empty(B, Hq * D, dtype=FP8, device=DEV)
    ops.decode_attention_fp8out(q, kb, vb, o8b, scale, indptr, idx, D ** -0.5, 0.0, splits, ws)
    torch.cuda.synchronize()
    assert torch.equal(_bits(o1), _bits(o2))
    assert torch.equal(_bits(q1), _bits(o8)) and torch.equal(_bits(q1), _bits(o8b))


@pytest.mark.parametrize("M", [129, 256, 300, 512])
def test_fused_forms_beyond_128_rows_equal_the_per_chunk_sequence(M):
    """Decode batches of 129..512 rows (graph batch sizes; C5 = batch 256) run the fused pair once per 128-row
    chunk: every row must come out exactly as from the unfused sequence applied to its chunk."""
    from iaas_sglang_amd import harness as H, ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M)
    chunks = [(m0, min(M, m0 + 128)) for m0 in range(0, M, 128)]
    # o/down: GEMM -> add -> RMSNorm -> fp8
    N, K = 4096, 1024
    qx, w, xs, ws = _fp8_operands(M, N, K, g)
    res = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(DEV)
    qscale = torch.tensor([0.02], device=DEV)
    r1, r2 = res.clone(), res.clone()
    outs, qs = [], []
    for a, b in chunks:
        y = ops.fp8_gemm(qx[a:b], w, xs, ws, dtype)
        o = ops.rmsnorm(y, nw, 1e-5, residual=r1[a:b])
        outs.append(o); qs.append(ops.fp8_quant_per_tensor(o, qscale)[0])
    out2, q2 = ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, r2, nw, 1e-5, qscale, want_out=True)
    assert torch.equal(_bits(torch.cat(outs)), _bits(out2)) and torch.equal(_bits(torch.cat(qs)), _bits(q2))
    assert torch.equal(_bits(r1), _bits(r2))
    # qkv: GEMM -> RoPE -> q + KV-pool rows
    Hq, Hkv, D, K = 8, 2, 128, 512
    qx, w, xs, ws = _fp8_operands(M, (Hq + 2 * Hkv) * D, K, g)
    slots = 2 * M + 1
    cache = H.rope_cache(D, 4096, 10000.0, DEV)
    pos = torch.randint(0, 4096, (M,), generator=g).to(DEV)
    loc = (torch.randperm(slots - 1, generator=g)[:M] + 1).to(DEV)
    kc1 = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    vc1 = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    kc2, vc2 = kc1.clone(), vc1.clone()
    q1 = []
    for a, b in chunks:
        qkv = ops.fp8_gemm(qx[a:b], w, xs, ws, dtype)
        qq, kk, vv = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
        ops.rope_neox_(qq, kk, pos[a:b], cache, D)
        ops.kv_write(kc1, vc1, loc[a:b], kk, vv)
        q1.append(qq.contiguous())
    q2 = ops.fp8_gemm_rope_kvwrite(qx, w, xs, ws, pos, cache, kc2, vc2, loc, Hq, Hkv, D)
    assert torch.equal(_bits(torch.cat(q1)), _bits(q2))
    assert torch.equal(_bits(kc1), _bits(kc2)) and torch.equal(_bits(vc1), _bits(vc2))
    # gate_up: both routes (in-kernel epilogue at I = 14336, slab consumer at I = 512)
    for I, K in ((14336, 512), (512, 256)):
        qx, w, xs, ws = _fp8_operands(M, 2 * I, K, g, amp=2.0)
        qscale = torch.tensor([0.05], device=DEV)
        q1 = torch.cat([ops.silu_and_mul_fp8(ops.fp8_gemm(qx[a:b], w, xs, ws, dtype), qscale) for a, b in chunks])
        q2 = ops.fp8_gemm_silu_mul(qx, w, xs, ws, qscale, dtype)
        assert torch.equal(_bits(q1), _bits(q2))
    torch.cuda.synchronize()


def _decode_logits(stack, runner, backend, fb, hidden, fused):
    from iaas_sglang_amd import harness as H
    H.LlamaStack.fuse_decode_layer = fused
    try:
        backend.init_forward_metadata(fb)
        return stack.forward(hidden.clone(), fb.positions, fb, backend)
    finally:
        H.LlamaStack.fuse_decode_layer = True


@pytest.mark.parametrize("batch,seq,kv_dtype", [(128, 64, None), (9, 200, None), (16, 100, FP8)])
def test_llama_layer_stack_fused_equals_unfused(batch, seq, kv_dtype):
    """Two Llama-3-8B-shaped layers, static FP8 scheme: logits and KV-pool contents of the fused decode step
    are bit-identical to the unfused plugin-surface sequence (15 launches per layer vs 7)."""
    import dataclasses
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config

    shape = dataclasses.replace(H.LLAMA3_8B, layers=2, vocab=4096)
    dtype = torch.bfloat16
    cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
    runner = H.make_runner(shape, max_reqs=batch, ctx=seq + 8, pool_tokens=batch * seq, dtype=dtype, device=DEV,
                           fill_kv=True, kv_dtype=kv_dtype)      # FP8: fp8 KV pool (qkv stays unfused on that path)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weight_range=0.02)
    fb = H.make_decode_batch(runner, backend, batch, seq, DEV, seed=3)
    g = torch.Generator().manual_seed(1)
    hidden = torch.randn(batch, shape.hidden, generator=g).to(dtype).to(DEV)
    backend.init_forward_metadata(fb)
    stack.calibrate_static_input_scales(hidden.clone(), fb.positions, fb, backend)
    pool = runner.token_to_kv_pool
    snap = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    assert stack._fused_decode_ok(hidden, fb)
    l_unfused = _decode_logits(stack, runner, backend, fb, hidden, fused=False)
    kv_unfused = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    for b, s0 in zip(pool.k_buffer + pool.v_buffer, snap):
        b.copy_(s0)
    l_fused = _decode_logits(stack, runner, backend, fb, hidden, fused=True)
    torch.cuda.synchronize()
    assert torch.equal(_bits(l_unfused), _bits(l_fused))
    for b, ref in zip(pool.k_buffer + pool.v_buffer, kv_unfused):
        assert torch.equal(b.view(torch.uint8), ref.view(torch.uint8))
    assert float(l_fused.float().abs().max()) > 0


@pytest.mark.parametrize("pre,ext", [([0, 64, 300], [600, 1000, 129]), ([0, 0, 40], [300, 200, 20])])
def test_llama_layer_stack_prefill_fused_equals_unfused(pre, ext):
    """Two Llama-3-8B-shaped layers, static FP8 scheme, an EXTEND batch: the prefill path with its fused forms (Q rotated
    inside the attention kernel, fp8 o_proj input from the attention epilogue, SiLU*mul in the gate_up GEMM's epilogue, fp8
    norm outputs) against the plugin-surface sequence (every op its own launch): logits and the whole KV pool bit for
    bit.  The second case holds a request whose extend is shorter than the 32-row block of the attention kernel."""
    import dataclasses
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config

    shape = dataclasses.replace(H.LLAMA3_8B, layers=2, vocab=4096)
    dtype = torch.bfloat16
    cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
    total = sum(pre) + sum(ext)
    runner = H.make_runner(shape, max_reqs=len(pre), ctx=max(p + e for p, e in zip(pre, ext)) + 8, pool_tokens=total + 64,
                           dtype=dtype, device=DEV, fill_kv=True)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weight_range=0.02)
    fb = H.make_extend_batch(runner, backend, pre, ext, DEV, seed=5)
    g = torch.Generator().manual_seed(2)
    hidden = torch.randn(sum(ext), shape.hidden, generator=g).to(dtype).to(DEV)
    pool = runner.token_to_kv_pool
    snap = [b.clone() for b in pool.k_buffer + pool.v_buffer]      # before the calibration pass writes the new rows
    backend.init_forward_metadata(fb)
    stack.calibrate_static_input_scales(hidden.clone(), fb.positions, fb, backend)
    lens = fb.extend_seq_lens

    def run(fused):
        for b, s0 in zip(pool.k_buffer + pool.v_buffer, snap):
            b.copy_(s0)
        H.LlamaStack.fuse_decode_layer = fused
        H.Linear.fuse_producer_quant = fused
        try:
            backend.init_forward_metadata(fb)
            out = stack.forward(hidden.clone(), fb.positions, fb, backend, last_token_logits=lens)
        finally:
            H.LlamaStack.fuse_decode_layer = True
            H.Linear.fuse_producer_quant = True
        torch.cuda.synchronize()
        return out, [b.clone() for b in pool.k_buffer + pool.v_buffer]

    l0, kv0 = run(False)
    l1, kv1 = run(True)
    assert torch.equal(_bits(l0), _bits(l1))
    for a, b in zip(kv0, kv1):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    assert float(l1.float().abs().max()) > 0 and not torch.equal(kv1[0], snap[0])


def test_fused_entry_points_reject_bad_arguments():
    from iaas_sglang_amd import ops
    from iaas_sglang_amd._lib import MiHotpathError
    g = torch.Generator().manual_seed(0)
    qx, w, xs, ws = _fp8_operands(600, 128, 256, g)          # M > 512 and I = 64: neither a decode shape nor tile-aligned halves
    with pytest.raises(MiHotpathError):
        ops.fp8_gemm_silu_mul(qx, w, xs, ws, torch.tensor([0.05], device=DEV), torch.bfloat16)
    qx, w, xs, ws = _fp8_operands(600, 256, 256, g)          # the slab consumers are decode forms: M <= 512
    with pytest.raises(MiHotpathError):
        ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, None, torch.ones(256, dtype=torch.bfloat16, device=DEV), 1e-5)
    qx, w, xs, ws = _fp8_operands(8, 256, 192, g)            # K % 128 != 0
    with pytest.raises(MiHotpathError):
        ops.fp8_gemm_add_rmsnorm(qx, w, xs, ws, None, torch.ones(256, dtype=torch.bfloat16, device=DEV), 1e-5)


# ---------------------------------------------------------------- int4 (AWQ) linears fused with their consumer
def _awq_weight(N, K, g, dtype, group=128):
    from iaas_sglang_amd import ops
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K, N // 8), dtype=torch.int32, generator=g)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // group, N // 8), dtype=torch.int32, generator=g)
    scales = (torch.rand(K // group, N, generator=g) * 0.004 + 0.001).to(dtype)
    qw, zs, _ = ops.w4_repack(qweight.to(DEV), qzeros.to(DEV), scales.to(DEV), group, ops.MI_W4_AWQ)
    return qw, zs


@pytest.mark.parametrize("M,N,K", [(64, 4096, 4096), (64, 4096, 11008), (128, 4096, 4096), (7, 4096, 4096), (33, 1024, 512)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_w4_gemm_add_rmsnorm_bit_identical(M, N, K, dtype):
    """mi_w4a16_gemm_add_rmsnorm == mi_w4a16_gemm (awq.py:199-203) then mi_rmsnorm with residual (layernorm.py:128-146)."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    qw, zs = _awq_weight(N, K, g, dtype)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    res = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).to(DEV)
    assert ops.w4a16_fused_ok(M, N, K, 128)
    y = ops.w4a16_gemm(x, qw, zs, N, 128)
    r1 = res.clone()
    out1 = ops.rmsnorm(y, nw, 1e-5, residual=r1)
    r2 = res.clone()
    out2 = ops.w4a16_gemm_add_rmsnorm(x, qw, zs, N, 128, r2, nw, 1e-5)
    out3 = ops.w4a16_gemm_add_rmsnorm(x, qw, zs, N, 128, None, nw, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(_bits(out1), _bits(out2)) and torch.equal(_bits(r1), _bits(r2))
    assert torch.equal(_bits(ops.rmsnorm(y, nw, 1e-5)), _bits(out3))


@pytest.mark.parametrize("M,Hq,Hkv,D,K", [(64, 32, 32, 128, 4096), (5, 32, 8, 128, 4096), (128, 8, 8, 64, 512)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_w4_gemm_rope_kvwrite_bit_identical(M, Hq, Hkv, D, K, dtype):
    """mi_w4a16_gemm_rope_kvwrite == mi_w4a16_gemm, NeoX RoPE (rotary_embedding.py:49-166), set_kv_buffer."""
    from iaas_sglang_amd import harness as H, ops
    g = torch.Generator().manual_seed(M + Hq + K)
    N = (Hq + 2 * Hkv) * D
    qw, zs = _awq_weight(N, K, g, dtype)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    slots = 4 * M + 8
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    loc = (torch.randperm(slots - 1, generator=g)[:M] + 1).to(torch.int64).to(DEV)
    pos = torch.randint(0, 4000, (M,), generator=g).to(torch.int64).to(DEV)
    cs = H.rope_cache(D, 4096, 10000.0, DEV)
    qkv = ops.w4a16_gemm(x, qw, zs, N, 128)
    q1, k1, v1 = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    ops.rope_neox_(q1, k1, pos, cs, D)
    kc1, vc1 = kc.clone(), vc.clone()
    ops.kv_write(kc1, vc1, loc, k1.reshape(M, Hkv, D).contiguous(), v1.reshape(M, Hkv, D).contiguous())
    kc2, vc2 = kc.clone(), vc.clone()
    q2 = ops.w4a16_gemm_rope_kvwrite(x, qw, zs, 128, pos, cs, kc2, vc2, loc, Hq, Hkv, D)
    torch.cuda.synchronize()
    assert torch.equal(_bits(q1.contiguous()), _bits(q2))
    assert torch.equal(_bits(kc1), _bits(kc2)) and torch.equal(_bits(vc1), _bits(vc2))


@pytest.mark.parametrize("M,I,K", [(64, 11008, 4096), (128, 11008, 4096), (3, 512, 256)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_w4_gemm_silu_mul_bit_identical(M, I, K, dtype):
    """mi_w4a16_gemm_silu_mul == mi_w4a16_gemm then mi_silu_and_mul (activation.py:56-58)."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(M + I + K)
    qw, zs = _awq_weight(2 * I, K, g, dtype)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    want = ops.silu_and_mul(ops.w4a16_gemm(x, qw, zs, 2 * I, 128))
    got = ops.w4a16_gemm_silu_mul(x, qw, zs, 2 * I, 128)
    torch.cuda.synchronize()
    assert torch.equal(_bits(want), _bits(got))


def test_llama2_awq_layer_stack_fused_equals_unfused():
    """Two Llama-2-7B-shaped AWQ layers (configuration C4 at its batch 64, KV 512): logits and KV-pool contents of the
    fused decode step (7 launches per layer) are bit-identical to the unfused plugin-surface sequence (17)."""
    import dataclasses
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import AWQConfig
    shape = dataclasses.replace(H.LLAMA2_7B, layers=2, vocab=4096)
    dtype, batch, seq = torch.float16, 64, 512
    awq = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    runner = H.make_runner(shape, max_reqs=batch, ctx=seq + 8, pool_tokens=batch * seq, dtype=dtype, device=DEV, fill_kv=True)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: awq.get_quant_method(None, ""), dtype, DEV)
    fb = H.make_decode_batch(runner, backend, batch, seq, DEV, seed=3)
    hidden = torch.randn(batch, shape.hidden, generator=torch.Generator().manual_seed(1)).to(dtype).to(DEV)
    backend.init_forward_metadata(fb)
    pool = runner.token_to_kv_pool
    snap = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    assert stack._fused_decode16_ok(hidden, fb)
    l_unfused = _decode_logits(stack, runner, backend, fb, hidden, fused=False)
    kv_unfused = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    for b, s0 in zip(pool.k_buffer + pool.v_buffer, snap):
        b.copy_(s0)
    l_fused = _decode_logits(stack, runner, backend, fb, hidden, fused=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(l_fused.float()).all())
    assert torch.equal(_bits(l_unfused), _bits(l_fused))
    for b, ref in zip(pool.k_buffer + pool.v_buffer, kv_unfused):
        assert torch.equal(_bits(b), _bits(ref))
