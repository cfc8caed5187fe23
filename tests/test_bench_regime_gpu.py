"""Parity at the sizes bench.py actually runs: the kernels the prefill leg and the decode step
launch, compared with the CPU oracle in THEIR regime -- 2048-token extends (32 q-blocks, more work items than CUs, 16
key tiles), C2 at its stated size (batch 32, extends from {1, 17, 128, 1000, 2048} + random over prefixes
{0, 64, 1024}, then a ragged decode step on the same pool), the persistent tile GEMM with the fused SiLU epilogue at
4096 x 28672 x 4096, and the north-star logit bound on the Llama-3-8B-wide fused decode stack at B = 128, S = 2048."""
import dataclasses

import pytest
import torch

from oracle import attention as oa
from oracle import elementwise as oe
from oracle import quant as oq

pytestmark = pytest.mark.gpu
DEV = "cuda"
Hq, Hkv, D = 32, 8, 128          # Llama-3-8B heads


def _scattered_r2t(lens, g):
    tot = sum(lens)
    perm = (torch.randperm(tot, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(len(lens), max(lens), dtype=torch.int32)
    off = 0
    for i, n in enumerate(lens):
        r2t[i, :n] = perm[off: off + n]
        off += n
    return r2t, tot + 1


def _extend_case(pre, ext, seed, dtype=torch.bfloat16):
    """The backend's extend sequence through the C ABI (KV write, prefix kv_indices, extend kernel) on a scattered
    pool, and the fp32 oracle on the pool the oracle itself wrote."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(seed)
    pre_t, ext_t = torch.tensor(pre), torch.tensor(ext)
    lens = [p + e for p, e in zip(pre, ext)]
    B = len(lens)
    r2t, slots = _scattered_r2t(lens, g)
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    E = sum(ext)
    q = torch.randn(E, Hq, D, generator=g).to(dtype)
    k_new = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v_new = torch.randn(E, Hkv, D, generator=g).to(dtype)
    rpi = torch.arange(B, dtype=torch.int64)
    loc = torch.cat([r2t[i, pre[i]: lens[i]] for i in range(B)]).to(torch.int64)
    d = lambda t: t.to(DEV)
    kcd, vcd = d(kc), d(vc)
    ops.kv_write(kcd, vcd, d(loc), d(k_new), d(v_new))
    pre_d, ext_d = d(pre_t.to(torch.int32)), d(ext_t.to(torch.int32))
    kv_indptr, qo_indptr = ops.kv_indptr(pre_d).clone(), ops.kv_indptr(ext_d).clone()
    idx = torch.empty(max(1, sum(pre)), dtype=torch.int32, device=DEV)
    ops.kv_indices(d(r2t), d(rpi), pre_d, kv_indptr, idx)
    out = torch.empty_like(d(q))
    ops.extend_attention(d(q), d(k_new), d(v_new), out, kcd, vcd, qo_indptr, kv_indptr, idx, max(ext), D ** -0.5, 0.0, True, -1)
    torch.cuda.synchronize()
    oa.set_kv_buffer(kc, vc, loc, k_new, v_new)
    assert torch.equal(kcd.cpu().view(torch.int16), kc.view(torch.int16))
    ref = oa.extend_fp32(q, kc, vc, r2t, rpi, torch.tensor(lens), pre_t, ext_t, scaling=D ** -0.5, causal=True)
    return out.cpu().float(), ref


@pytest.mark.parametrize("name,pre,ext", [
    ("4x2048 causal, no prefix (the prefill leg's chunk shape)", [0] * 4, [2048] * 4),
    ("2x1024 over a 1024-key scattered prefix", [1024, 1024], [1024, 1024]),
    ("ragged: long and short extends, more work items than CUs", [0, 1024, 0, 300, 64, 0, 2000, 17],
     [2048, 64, 700, 1, 333, 1500, 65, 128]),
])
def test_extend_attention_at_the_prefill_legs_regime(name, pre, ext):
    """extend_attn32_kernel (+ the 16x16 kernel for the short rows of the ragged case) against oa.extend_fp32
    (torch_native_backend.py:27-110 arithmetic): up to 32 q-blocks per request, 256..1024 work items, 16+ key tiles."""
    out, ref = _extend_case(pre, ext, seed=len(pre) * 7 + 1)
    # P is rounded to bf16 before the PV MFMA (as the Triton kernel does): same tolerance as the small-shape tests
    torch.testing.assert_close(out, ref, atol=4e-3, rtol=2 ** -6)


def test_config_c2_at_its_stated_size_extend_then_ragged_decode():
    """SURVEY 8d, configuration C2: batch 32, extends drawn from {1, 17, 128, 1000, 2048} + random over cached
    prefixes {0, 64, 1024} through MiAttnBackend (EXTEND), then one ragged decode step on the same pool (DECODE): the
    pool contents bit-exact, both outputs against the fp32 oracle."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    g = torch.Generator().manual_seed(2)
    base_ext, base_pre = [1, 17, 128, 1000, 2048], [0, 64, 1024]
    B = 32
    ext = [base_ext[i % 5] if i % 6 else int(torch.randint(1, 600, (1,), generator=g)) for i in range(B)]
    pre = [base_pre[(i * 7) % 3] for i in range(B)]
    lens = [p + e for p, e in zip(pre, ext)]
    tot = sum(lens) + B + 8
    runner = H.make_runner(H.LLAMA3_8B, max_reqs=B, ctx=4096, pool_tokens=tot, dtype=torch.bfloat16, device=DEV)
    runner.token_to_kv_pool = H.make_kv_pool(tot, 1, Hkv, D, torch.bfloat16, DEV, fill_random=True)
    backend = MiAttnBackend(runner)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    fb = H.make_extend_batch(runner, backend, pre, ext, DEV, seed=3)
    E = sum(ext)
    q = torch.randn(E, Hq * D, generator=g).to(torch.bfloat16)
    k = torch.randn(E, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(E, Hkv, D, generator=g).to(torch.bfloat16)
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu().clone(), pool.v_buffer[0].cpu().clone()
    backend.init_forward_metadata(fb)
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    torch.cuda.synchronize()
    oa.set_kv_buffer(kc, vc, fb.out_cache_loc.cpu(), k, v)
    assert torch.equal(pool.k_buffer[0].cpu().view(torch.int16), kc.view(torch.int16))
    assert torch.equal(pool.v_buffer[0].cpu().view(torch.int16), vc.view(torch.int16))
    r2t = runner.req_to_token_pool.req_to_token.cpu()
    ref = oa.extend_fp32(q.view(-1, Hq, D), kc, vc, r2t, fb.req_pool_indices.cpu(), torch.tensor(lens), torch.tensor(pre),
                         torch.tensor(ext), scaling=D ** -0.5, causal=True)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)
    # ragged decode step on the same pool: every request grows by one token
    next_slot = sum(lens) + 1
    new_loc = torch.arange(next_slot, next_slot + B, dtype=torch.int64)
    for i in range(B):
        runner.req_to_token_pool.req_to_token[i, lens[i]] = int(new_loc[i])
    lens1 = torch.tensor([n + 1 for n in lens])
    fb.forward_mode = H.ForwardMode.DECODE
    fb.seq_lens, fb.seq_lens_sum, fb.seq_lens_cpu = lens1.to(DEV), int(lens1.sum()), lens1
    fb.out_cache_loc, fb.positions = new_loc.to(DEV), (lens1 - 1).to(DEV)
    qd = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16)
    kd = torch.randn(B, Hkv, D, generator=g).to(torch.bfloat16)
    vd = torch.randn(B, Hkv, D, generator=g).to(torch.bfloat16)
    backend.init_forward_metadata(fb)
    assert backend.forward_metadata.work is not None            # ragged: the launch-list plan
    od = backend.forward(qd.to(DEV), kd.to(DEV), vd.to(DEV), layer, fb)
    torch.cuda.synchronize()
    oa.set_kv_buffer(kc, vc, new_loc, kd, vd)
    want = oa.decode_fp32(qd.view(B, Hq, D), kc, vc, runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(),
                          lens1, scaling=D ** -0.5)
    torch.testing.assert_close(od.view(B, Hq, D).cpu().float(), want, atol=2e-3, rtol=2 ** -7)
    assert torch.equal(pool.k_buffer[0].cpu().view(torch.int16), kc.view(torch.int16))


def test_tile_gemm_fused_silu_epilogue_at_the_prefill_chunk_shape():
    """fp8_gemm_tile_kernel<EPI=1> at 4096 x 28672 x 4096 (the gate_up of a prefill chunk): bit-identical to the plain
    tile GEMM followed by mi_silu_and_mul_fp8, and a sample of rows against the oracle (fp8_utils.py:715-723 scaled-mm
    semantics, activation.py:56-58, static per-tensor quant)."""
    from iaas_sglang_amd import ops
    FP8 = torch.float8_e4m3fn
    M, I, K = 4096, 14336, 4096
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, K, generator=g) * 0.5).to(FP8)
    w = (torch.randn(2 * I, K, generator=g) * 0.5).to(FP8)
    sa, sb = torch.tensor([0.02]), torch.tensor([0.015])
    qs = torch.tensor([0.05])
    xd, wd = x.to(DEV), w.to(DEV)
    fused = ops.fp8_gemm_silu_mul(xd, wd.t(), sa.to(DEV), sb.to(DEV), qs.to(DEV), torch.bfloat16)
    gu = ops.fp8_gemm(xd, wd.t(), sa.to(DEV), sb.to(DEV), torch.bfloat16)
    unfused = ops.silu_and_mul_fp8(gu, qs.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.uint8), unfused.view(torch.uint8))
    rows = torch.tensor([0, 1, 255, 256, 1000, 2047, 3000, 4095])
    ref_gu = oq.scaled_mm(x[rows], w.t(), sa, sb, torch.bfloat16)
    torch.testing.assert_close(gu[rows.to(DEV)].cpu().float(), ref_gu.float(), rtol=2 ** -7, atol=2e-2)
    act = oe.silu_and_mul(ref_gu).float()
    ref_q, _ = oq.per_tensor_quant_fp8(act.to(torch.bfloat16), qs)
    got, want = fused[rows.to(DEV)].cpu().float(), ref_q.float()
    # the fp8 code of an element flips when the bf16 activation sits on a rounding boundary: one e4m3 step at most,
    # and only for a small fraction of the elements
    step = torch.maximum(want.abs(), torch.tensor(2.0 ** -6)) * 2 ** -3 + 2 ** -9
    assert bool(((got - want).abs() <= step).all())
    assert float((got != want).float().mean()) < 0.02


@pytest.mark.parametrize("weight_range,tol", [(1e-3, 1e-3), (0.02, None)])
def test_north_star_logit_bound_on_the_8b_wide_fused_decode_stack(weight_range, tol):
    """The north-star criterion at its own shape: 2 layers of the Llama-3-8B-wide stack (hidden 4096, 32/8 heads,
    intermediate 14336, vocabulary 128256), B = 128, KV 2048, per-tensor FP8 linears with static activation scales, the
    FUSED decode step (what bench.py captures) against the oracle's layer sequence (models/llama.py:245-268 order):
    max-abs logit error < 1e-3 with the reference's +-1e-3 dummy weights; with +-0.02 weights (logits of order 1..4)
    within 3 % of the largest logit, 0.3 % on average."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config
    from test_e2e_gpu import _oracle_forward
    shape = dataclasses.replace(H.LLAMA3_8B, layers=2)
    dtype, B, S = torch.bfloat16, 128, 2048
    cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
    runner = H.make_runner(shape, max_reqs=B, ctx=S + 8, pool_tokens=B * S + 8, dtype=dtype, device=DEV, fill_kv=True)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weight_range=weight_range,
                         weights_cpu_seeded=True)
    fb = H.make_decode_batch(runner, backend, B, S, DEV, seed=1)
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(B, shape.hidden, generator=g).to(dtype)
    backend.init_forward_metadata(fb)
    stack.calibrate_static_input_scales(hidden.to(DEV), fb.positions, fb, backend)
    pool = runner.token_to_kv_pool
    kc = [b.cpu().clone() for b in pool.k_buffer]
    vc = [b.cpu().clone() for b in pool.v_buffer]
    W = {"layers": [], "final_norm": stack.final_norm.cpu(), "lm_head": stack.lm_head.cpu()}
    for L in stack.layers:
        d = {"input_norm": L.input_norm.cpu(), "post_norm": L.post_norm.cpu()}
        for key, lin in (("qkv", L.qkv), ("o", L.o), ("gu", L.gate_up), ("down", L.down)):
            d[key + "_w"], d[key + "_s"], d[key + "_i"] = lin.weight.cpu(), lin.weight_scale.cpu(), lin.input_scale.cpu()
        W["layers"].append(d)
    assert stack._fused_decode_ok(hidden.to(DEV), fb)           # the step below IS the fused path
    logits = stack.forward(hidden.to(DEV), fb.positions, fb, backend)
    torch.cuda.synchronize()
    ref = _oracle_forward(W, shape, hidden, fb.positions.cpu(), kc, vc, runner.req_to_token_pool.req_to_token.cpu(),
                          fb.req_pool_indices.cpu(), fb.seq_lens.cpu(), None, None, fb.out_cache_loc.cpu(), decode=True)
    diff = (logits.float().cpu() - ref).abs()
    err, mean_err = float(diff.max()), float(diff.mean())
    # +-0.02 weights give logits of order 1..4: the HIP path and the oracle share the quantised weights and differ by
    # accumulation order and one rounding per op, plus the odd fp8 ACTIVATION code that flips on such a difference (one
    # e4m3 step = 6-12 % of that element, four quantisation points per layer): the worst of 16 M logits within 3 % of
    # the largest logit, the mean error within 0.3 % of it
    bound = tol if tol is not None else 0.03 * float(ref.abs().max())
    print(f"8B-wide fused decode, B={B} S={S}: max|logit|={float(ref.abs().max()):.4f} err={err:.3e} "
          f"mean err={mean_err:.3e} bound={bound:.3e}")
    assert err < bound
    assert tol is not None or mean_err < 0.003 * float(ref.abs().max())
    # the new token's K/V rows in the pool are what the oracle wrote (bf16 rounding of slightly different fp8 sums aside)
    loc = fb.out_cache_loc.cpu()
    for li in range(shape.layers):
        want_k = kc[li][loc].float()
        torch.testing.assert_close(pool.k_buffer[li].cpu()[loc].float(), want_k, atol=tol or 0.03 * float(want_k.abs().max()),
                                   rtol=2e-2)
