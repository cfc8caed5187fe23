"""fp8 (e4m3fn) KV cache -- SURVEY 8f row 1: the scatter kernel (bit-exact vs the oracle restatement of
MHATokenToKVPool.set_kv_buffer, memory_pool.py:432-455) and decode attention over the fp8 pool (fp32 oracle on
the dequantised pool; tolerance as for bf16 KV: |d| <= 2e-3 + 2 ulp of the output dtype).
PARITY UNPINNED by reference-run vectors (the torch-native oracle backend cannot read an fp8 pool); the
convention is flashinfer_backend.py:474-555: store k / k_scale, attend with k_scale / v_scale."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oa  # noqa: E402

DEV = "cuda"
FP8 = torch.float8_e4m3fn


def _tol(dtype):
    return 2e-3, (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k_scale,v_scale", [(1.0, 1.0), (0.37, 2.5)])
def test_kv_write_fp8_bit_exact(dtype, k_scale, v_scale):
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(0)
    T, Hkv, D, slots = 37, 8, 128, 200
    k = (torch.randn(T, Hkv, D, generator=g) * 3).to(dtype)
    v = (torch.randn(T, Hkv, D, generator=g) * 300).to(dtype)       # some values beyond +-448: saturate
    loc = (torch.randperm(slots - 1, generator=g)[:T] + 1)
    kc = torch.randint(0, 255, (slots, Hkv, D), generator=g, dtype=torch.uint8)
    vc = torch.randint(0, 255, (slots, Hkv, D), generator=g, dtype=torch.uint8)
    kr, vr = kc.clone().view(FP8), vc.clone().view(FP8)
    oa.set_kv_buffer_fp8(kr, vr, loc, k, v, k_scale, v_scale)
    kd, vd = kc.to(DEV), vc.to(DEV)
    ops.kv_write_fp8(kd, vd, loc.to(DEV), k.to(DEV), v.to(DEV), k_scale, v_scale)
    torch.cuda.synchronize()
    assert torch.equal(kd.cpu(), kr.view(torch.uint8)) and torch.equal(vd.cpu(), vr.view(torch.uint8))
    assert not torch.isnan(vr[loc].float()).any()                      # written rows saturate, never NaN


@pytest.mark.parametrize("Hq,Hkv,lens,splits", [(32, 8, [1, 37, 128, 300, 16], 1), (32, 8, [700, 5, 256], 3),
                                                (16, 2, [64, 65], 2), (4, 4, [33, 1, 90], 1), (8, 1, [129], 2)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k_scale,v_scale,cap", [(1.0, 1.0, 0.0), (0.5, 2.0, 0.0), (0.8, 1.25, 30.0)])
def test_decode_attention_fp8kv_vs_oracle(Hq, Hkv, lens, splits, dtype, k_scale, v_scale, cap):
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(Hq + len(lens) + splits)
    D, B = 128, len(lens)
    total = sum(lens)
    k8 = (torch.randn(total + 1, Hkv, D, generator=g) / k_scale).clamp(-448, 448).to(FP8)
    v8 = (torch.randn(total + 1, Hkv, D, generator=g) / v_scale).clamp(-448, 448).to(FP8)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    perm = torch.randperm(total, generator=g) + 1
    r2t = torch.zeros(B, max(lens), dtype=torch.int32)
    off = 0
    for i, L in enumerate(lens):
        r2t[i, :L] = perm[off:off + L].to(torch.int32)
        off += L
    rpi, sl = torch.arange(B), torch.tensor(lens)
    ref = oa.decode_fp32_fp8kv(q, k8, v8, r2t, rpi, sl, 1 / math.sqrt(D), k_scale, v_scale, logit_cap=cap)
    indptr = ops.kv_indptr(sl.to(DEV))
    idx = torch.empty(total, dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi.to(DEV), sl.to(DEV), indptr, idx)
    ws = torch.empty(max(ops.decode_workspace_numel(B, Hq, D, splits), 1), dtype=torch.float32, device=DEV)
    o = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention_fp8kv(q.to(DEV), k8.view(torch.uint8).to(DEV), v8.to(DEV), indptr, idx, 1 / math.sqrt(D),
                               k_scale, v_scale, cap, splits, ws, o=o)
    # fused fp8 output of the same call == quantising the T output
    scale = torch.tensor([0.02], device=DEV)
    o8 = torch.empty(B, Hq * D, dtype=FP8, device=DEV)
    ops.decode_attention_fp8kv(q.to(DEV), k8.to(DEV), v8.to(DEV), indptr, idx, 1 / math.sqrt(D), k_scale, v_scale, cap,
                               splits, ws, o_fp8=o8, o_scale=scale)
    q1, _ = ops.fp8_quant_per_tensor(o.view(B, Hq * D), scale)
    torch.cuda.synchronize()
    atol, rtol = _tol(dtype)
    torch.testing.assert_close(o.cpu().float(), ref, atol=atol * max(1.0, v_scale), rtol=rtol)
    assert torch.equal(o8.view(torch.uint8), q1.view(torch.uint8))


def test_backend_with_fp8_pool_decode_and_prefill():
    """MiAttnBackend over an fp8 MHATokenToKVPool: prefill writes fp8 rows and attends the bf16 arguments; decode
    steps read the fp8 pool; an extend over a cached prefix reads (and converts) the prefix from the fp8 pool."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.mem_cache import MHATokenToKVPool, ReqToTokenPool

    shape, dtype = H.LLAMA3_8B, torch.bfloat16
    Hq, Hkv, D = 32, 8, 128
    runner = H.make_runner(shape, max_reqs=4, ctx=256, pool_tokens=600, dtype=dtype, device=DEV)
    runner.token_to_kv_pool = MHATokenToKVPool(600, 1, FP8, Hkv, D, 1, DEV)
    runner.req_to_token_pool = ReqToTokenPool(4, 256, DEV)
    backend = MiAttnBackend(runner)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    layer.k_scale_float, layer.v_scale_float = 0.5, 2.0
    g = torch.Generator().manual_seed(3)
    ext = [40, 7, 100]
    fb = H.make_extend_batch(runner, backend, [0, 0, 0], ext, DEV, seed=1)
    E = sum(ext)
    q = torch.randn(E, Hq * D, generator=g).to(dtype)
    k = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v = torch.randn(E, Hkv, D, generator=g).to(dtype)
    backend.init_forward_metadata(fb)
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    # oracle: bf16 pool for the prefill attention itself (new tokens are attended at full precision) ...
    kc, vc = torch.zeros(601, Hkv, D, dtype=dtype), torch.zeros(601, Hkv, D, dtype=dtype)
    r2t = runner.req_to_token_pool.req_to_token.cpu()
    ref = oa.forward_extend(q, k, v, kc, vc, r2t, fb.req_pool_indices.cpu(), fb.seq_lens.cpu(),
                            fb.extend_prefix_lens.cpu(), fb.extend_seq_lens.cpu(), fb.out_cache_loc.cpu(), Hq, Hkv, D ** -0.5)
    torch.testing.assert_close(o.cpu().float(), ref.float(), atol=2e-2, rtol=2e-2)
    # ... and the fp8 pool contents bit-exact
    k8 = torch.zeros(601, Hkv, D).to(FP8)
    v8 = torch.zeros(601, Hkv, D).to(FP8)
    oa.set_kv_buffer_fp8(k8, v8, fb.out_cache_loc.cpu(), k, v, 0.5, 2.0)
    pool = runner.token_to_kv_pool
    assert torch.equal(pool.k_buffer[0].cpu(), k8.view(torch.uint8)) and torch.equal(pool.v_buffer[0].cpu(), v8.view(torch.uint8))
    # decode step on top
    lens = [e + 1 for e in ext]
    loc = torch.arange(400, 403, dtype=torch.int64)
    for i in range(3):
        runner.req_to_token_pool.req_to_token[i, lens[i] - 1] = int(loc[i])
    fb.forward_mode = H.ForwardMode.DECODE
    fb.seq_lens = torch.tensor(lens, dtype=torch.int64, device=DEV)
    fb.seq_lens_sum, fb.out_cache_loc = sum(lens), loc.to(DEV)
    qd = torch.randn(3, Hq * D, generator=g).to(dtype)
    kd = torch.randn(3, Hkv, D, generator=g).to(dtype)
    vd = torch.randn(3, Hkv, D, generator=g).to(dtype)
    backend.init_forward_metadata(fb)
    od = backend.forward(qd.to(DEV), kd.to(DEV), vd.to(DEV), layer, fb)
    oa.set_kv_buffer_fp8(k8, v8, loc, kd, vd, 0.5, 2.0)
    refd = oa.decode_fp32_fp8kv(qd.view(3, Hq, D), k8, v8, runner.req_to_token_pool.req_to_token.cpu(),
                                fb.req_pool_indices.cpu(), fb.seq_lens.cpu(), D ** -0.5, 0.5, 2.0)
    torch.testing.assert_close(od.view(3, Hq, D).cpu().float(), refd, atol=4e-3, rtol=2 ** -7)
    # extend over the cached prefix (prefix rows come from the fp8 pool, new rows from the arguments)
    lens2 = [l for l in lens]                               # requests 0..2 now hold lens2 tokens in the pool
    ext2 = [9, 33, 1]
    pre2 = lens2
    r2t = runner.req_to_token_pool.req_to_token
    new_slots, pos = [], 450
    for i in range(3):
        sl = torch.arange(pos, pos + ext2[i], dtype=torch.int32)
        r2t[i, pre2[i]: pre2[i] + ext2[i]] = sl.to(DEV)
        new_slots.append(sl.to(torch.int64))
        pos += ext2[i]
    from types import SimpleNamespace
    E2 = sum(ext2)
    fb2 = SimpleNamespace(forward_mode=H.ForwardMode.EXTEND, batch_size=3, req_pool_indices=fb.req_pool_indices,
                          seq_lens=torch.tensor([p + e for p, e in zip(pre2, ext2)], dtype=torch.int64, device=DEV),
                          seq_lens_sum=sum(pre2) + E2, seq_lens_cpu=None,
                          extend_prefix_lens=torch.tensor(pre2, dtype=torch.int32, device=DEV),
                          extend_seq_lens=torch.tensor(ext2, dtype=torch.int32, device=DEV),
                          extend_prefix_lens_cpu=pre2, extend_seq_lens_cpu=ext2,
                          out_cache_loc=torch.cat(new_slots).to(DEV), req_to_token_pool=runner.req_to_token_pool,
                          token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend, spec_info=None, positions=None)
    q2 = torch.randn(E2, Hq * D, generator=g).to(dtype)
    k2 = torch.randn(E2, Hkv, D, generator=g).to(dtype)
    v2 = torch.randn(E2, Hkv, D, generator=g).to(dtype)
    backend.init_forward_metadata(fb2)
    o2 = backend.forward(q2.to(DEV), k2.to(DEV), v2.to(DEV), layer, fb2)
    # oracle: an fp32 pool holding the DEQUANTISED prefix rows and the full-precision new rows
    kf, vf = k8.float() * 0.5, v8.float() * 2.0
    kf[torch.cat(new_slots)] = k2.float()
    vf[torch.cat(new_slots)] = v2.float()
    ref2 = oa.extend_fp32(q2.view(E2, Hq, D), kf, vf, r2t.cpu(), fb2.req_pool_indices.cpu(), fb2.seq_lens.cpu(),
                          fb2.extend_prefix_lens.cpu(), fb2.extend_seq_lens.cpu(), D ** -0.5)
    torch.testing.assert_close(o2.view(E2, Hq, D).cpu().float(), ref2, atol=2e-2, rtol=2e-2)
    # and the new rows landed in the fp8 pool, quantised with the scales
    oa.set_kv_buffer_fp8(k8, v8, torch.cat(new_slots), k2, v2, 0.5, 2.0)
    assert torch.equal(pool.k_buffer[0].cpu(), k8.view(torch.uint8)) and torch.equal(pool.v_buffer[0].cpu(), v8.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k_scale,v_scale,cap,causal", [(1.0, 1.0, 0.0, True), (0.6, 1.7, 0.0, True), (0.6, 1.7, 20.0, True),
                                                         (0.8, 1.2, 0.0, False)])
def test_extend_attention_fp8_prefix_vs_oracle(dtype, k_scale, v_scale, cap, causal):
    """Ragged extend with prefixes in an fp8 pool (through the C ABI) vs the fp32 oracle on the dequantised prefix."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(11)
    Hq, Hkv, D = 32, 8, 128
    pre, ext = [0, 70, 33, 129, 5], [17, 1, 64, 40, 100]
    B, E, P = len(pre), sum(ext), sum(pre)
    slots = P + E + 1
    k8 = (torch.randn(slots, Hkv, D, generator=g) / k_scale).clamp(-448, 448).to(FP8)
    v8 = (torch.randn(slots, Hkv, D, generator=g) / v_scale).clamp(-448, 448).to(FP8)
    q = torch.randn(E, Hq, D, generator=g).to(dtype)
    k = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v = torch.randn(E, Hkv, D, generator=g).to(dtype)
    perm = torch.randperm(slots - 1, generator=g) + 1
    r2t = torch.zeros(B, max(p + e for p, e in zip(pre, ext)), dtype=torch.int32)
    off, new_loc = 0, []
    for i in range(B):
        n = pre[i] + ext[i]
        r2t[i, :n] = perm[off: off + n].to(torch.int32)
        new_loc.append(perm[off + pre[i]: off + n])
        off += n
    new_loc = torch.cat(new_loc)
    kf, vf = k8.float() * k_scale, v8.float() * v_scale
    kf[new_loc], vf[new_loc] = k.float(), v.float()
    rpi = torch.arange(B)
    sl = torch.tensor([p + e for p, e in zip(pre, ext)])
    ref = oa.extend_fp32(q, kf, vf, r2t, rpi, sl, torch.tensor(pre), torch.tensor(ext), D ** -0.5, causal=causal, logit_cap=cap)
    pre_t = torch.tensor(pre, dtype=torch.int32, device=DEV)
    ext_t = torch.tensor(ext, dtype=torch.int32, device=DEV)
    kvp = ops.kv_indptr(pre_t).clone()
    idx = torch.empty(max(P, 1), dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi.to(DEV), pre_t, kvp, idx)
    qo = ops.kv_indptr(ext_t)
    o = torch.empty(E, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fp8kv(q.to(DEV), k.to(DEV), v.to(DEV), o, k8.to(DEV), v8.view(torch.uint8).to(DEV), k_scale, v_scale,
                               qo, kvp, idx, max(ext), D ** -0.5, cap, causal, -1)
    torch.cuda.synchronize()
    torch.testing.assert_close(o.cpu().float(), ref, atol=2e-2 * max(1.0, v_scale), rtol=2e-2)
    o2 = torch.empty_like(o)          # split-KV form over the fp8 pool
    ops.extend_attention_splitkv(q.to(DEV), k.to(DEV), v.to(DEV), o2, k8.to(DEV), v8.view(torch.uint8).to(DEV), qo, kvp, idx,
                                 max(ext), D ** -0.5, 3, None, cap, causal, -1, k_scale, v_scale)
    torch.cuda.synchronize()
    torch.testing.assert_close(o2.cpu().float(), ref, atol=2e-2 * max(1.0, v_scale), rtol=2e-2)
    if causal:       # tree mask over an fp8 pool (TARGET_VERIFY with --kv-cache-dtype fp8): a lower-triangular mask == causal
        masks, mptr = [], [0]
        for i in range(B):
            m = torch.ones(ext[i], pre[i] + ext[i], dtype=torch.bool)
            m[:, pre[i]:] = torch.tril(torch.ones(ext[i], ext[i], dtype=torch.bool))
            masks.append(m.reshape(-1)); mptr.append(mptr[-1] + m.numel())
        for splits in (1, 3):
            o3 = torch.empty_like(o)
            ops.extend_attention_splitkv(q.to(DEV), k.to(DEV), v.to(DEV), o3, k8.to(DEV), v8.view(torch.uint8).to(DEV), qo, kvp,
                                         idx, max(ext), D ** -0.5, splits, None, cap, True, -1, k_scale, v_scale,
                                         torch.cat(masks).to(DEV), torch.tensor(mptr, dtype=torch.int64, device=DEV), True)
            torch.cuda.synchronize()
            torch.testing.assert_close(o3.cpu().float(), ref, atol=2e-2 * max(1.0, v_scale), rtol=2e-2)


def test_fp8kv_full_size_constant_v_identity():
    """B=128, S=2048 (BASELINE size): with every V row equal to one vector c the output must be v_scale * c for any
    K, any split count (softmax weights sum to 1) -- a size-independent property of the full-size launch."""
    from iaas_sglang_amd import ops
    B, S, Hq, Hkv, D = 128, 2048, 32, 8, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    k8 = torch.randn(B * S + 1, Hkv, D, device=DEV, generator=g).to(FP8)
    c = torch.randn(Hkv, D, device=DEV, generator=g).to(FP8)
    v8 = c.unsqueeze(0).expand(B * S + 1, Hkv, D).contiguous()
    q = torch.randn(B, Hq, D, device=DEV, generator=g).to(torch.bfloat16)
    idx = (torch.randperm(B * S, device=DEV, generator=g) + 1).to(torch.int32)
    indptr = ops.kv_indptr(torch.full((B,), S, dtype=torch.int64, device=DEV))
    want = (c.float() * 1.5).repeat_interleave(Hq // Hkv, dim=0).to(torch.bfloat16)
    for splits in (1, 2, 4):
        ws = torch.empty(max(ops.decode_workspace_numel(B, Hq, D, splits), 1), dtype=torch.float32, device=DEV)
        o = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
        ops.decode_attention_fp8kv(q, k8, v8, indptr, idx, D ** -0.5, 0.7, 1.5, 0.0, splits, ws, o=o)
        torch.cuda.synchronize()
        torch.testing.assert_close(o.float(), want.float().unsqueeze(0).expand(B, Hq, D), atol=1e-5, rtol=2 ** -7)


def test_mem_cache_set_kv_buffer_bf16_matches_oracle():
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.mem_cache import MHATokenToKVPool
    g = torch.Generator().manual_seed(1)
    pool = MHATokenToKVPool(50, 1, torch.bfloat16, 2, 64, 2, DEV)
    layer = H.AttnLayer(4, 64, 0.125, 2, 1)
    k = torch.randn(9, 2, 64, generator=g).to(torch.bfloat16)
    v = torch.randn(9, 2, 64, generator=g).to(torch.bfloat16)
    loc = torch.randperm(50, generator=g)[:9] + 1
    pool.set_kv_buffer(layer, loc.to(DEV), k.to(DEV), v.to(DEV))
    kc, vc = torch.zeros(51, 2, 64, dtype=torch.bfloat16), torch.zeros(51, 2, 64, dtype=torch.bfloat16)
    oa.set_kv_buffer(kc, vc, loc, k, v)
    torch.cuda.synchronize()
    assert torch.equal(pool.get_key_buffer(1).cpu().view(torch.int16), kc.view(torch.int16))
    assert torch.equal(pool.get_value_buffer(1).cpu().view(torch.int16), vc.view(torch.int16))
    assert int(pool.k_buffer[0].abs().sum()) == 0                     # the other layer is untouched
