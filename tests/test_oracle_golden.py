"""Pin the CPU oracle against vectors produced by the reference's own files
(tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from oracle import attention as oa
from oracle import elementwise as oe
from oracle import quant as oq
from oracle import sched as osch

FP8 = torch.float8_e4m3fn

DECODE = ["decode_gqa4_d128_bf16", "decode_gqa8_d128_bf16", "decode_mha_d128_fp16",
          "decode_mha_d64_fp16", "decode_gqa4_d128_bf16_shifted"]
EXTEND = ["extend_gqa4_d128_bf16_noprefix", "extend_gqa4_d128_bf16_prefix",
          "extend_mha_d64_fp16_prefix", "extend_gqa4_d128_bf16_noncausal"]


@pytest.mark.parametrize("name", DECODE)
def test_decode_oracle_matches_reference(golden_attention, name):
    c = golden_attention[name]
    Hq, Hkv = c["q"].shape[1], c["k_cache"].shape[1]
    kc, vc = c["k_cache"].clone(), c["v_cache"].clone()
    o = oa.forward_decode(c["q"].reshape(c["q"].shape[0], -1), c["k_new"], c["v_new"], kc, vc,
                          c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["out_cache_loc"],
                          Hq, Hkv, float(c["scaling"]))
    # same gather + same SDPA call on the same host => bit-identical
    assert torch.equal(o.view_as(c["o"]), c["o"])
    # fp32 restatement agrees up to the I/O dtype's rounding
    o32 = oa.decode_fp32(c["q"], kc, vc, c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                         scaling=float(c["scaling"]))
    torch.testing.assert_close(o32, c["o"].float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("name", EXTEND)
def test_extend_oracle_matches_reference(golden_attention, name):
    c = golden_attention[name]
    Hq, Hkv = c["q"].shape[1], c["k_cache"].shape[1]
    kc, vc = c["k_cache"].clone(), c["v_cache"].clone()
    o = oa.forward_extend(c["q"].reshape(c["q"].shape[0], -1), c["k_new"], c["v_new"], kc, vc,
                          c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                          c["extend_prefix_lens"], c["extend_seq_lens"], c["out_cache_loc"],
                          Hq, Hkv, float(c["scaling"]), causal=bool(c["causal"]))
    assert torch.equal(o.view_as(c["o"]), c["o"])
    o32 = oa.extend_fp32(c["q"], kc, vc, c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                         c["extend_prefix_lens"], c["extend_seq_lens"], scaling=float(c["scaling"]),
                         causal=bool(c["causal"]))
    torch.testing.assert_close(o32, c["o"].float(), atol=1e-2, rtol=1e-2)


def test_kv_indices_oracle():
    # restates test/srt/test_create_kvindices.py:18-49 at small size
    g = torch.Generator().manual_seed(3)
    max_batch, ctx, batch = 64, 96, 37
    req_to_token = torch.arange(max_batch * ctx, dtype=torch.int32).reshape(max_batch, ctx)
    rpi = torch.randperm(max_batch, generator=g)[:batch]
    lens = torch.randperm(ctx, generator=g)[:batch].to(torch.int32)   # includes a zero length
    indptr, idx = oa.kv_indices(req_to_token, rpi, lens)
    assert indptr.dtype == torch.int32 and idx.dtype == torch.int32
    assert int(indptr[-1]) == int(lens.sum()) == idx.numel()
    for i in range(batch):
        seg = idx[int(indptr[i]): int(indptr[i + 1])]
        assert torch.equal(seg, req_to_token[rpi[i], : int(lens[i])])


def test_merge_state_oracle_is_consistent():
    # merging two halves of the key set == attention over the whole set
    g = torch.Generator().manual_seed(5)
    q = torch.randn(3, 4, 32, generator=g)
    k = torch.randn(50, 4, 32, generator=g)
    v = torch.randn(50, 4, 32, generator=g)
    r2t = torch.arange(50, dtype=torch.int32).repeat(3, 1)
    rpi = torch.arange(3)
    full, lse = oa.decode_fp32(q, k, v, r2t, rpi, torch.tensor([50, 50, 50]), return_lse=True)
    a, la = oa.decode_fp32(q, k[:20], v[:20], r2t, rpi, torch.tensor([20] * 3), return_lse=True)
    r2t_b = (torch.arange(30, dtype=torch.int32) + 20).repeat(3, 1)
    b, lb = oa.decode_fp32(q, k, v, r2t_b, rpi, torch.tensor([30] * 3), return_lse=True)
    m, lm = oa.merge_state(a, la, b, lb)
    torch.testing.assert_close(m, full, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(lm, lse, atol=1e-5, rtol=1e-5)
    # inf guard (merge_attn_states.cu:66-67)
    m2, _ = oa.merge_state(a, la, b, torch.full_like(lb, float("inf")))
    torch.testing.assert_close(m2, a)


@pytest.mark.parametrize("name", ["awq_g128_fp16", "awq_gK_bf16"])
def test_awq_dequant_oracle(golden_quant, name):
    c = golden_quant[name]
    W = oq.awq_dequantize(c["qweight"], c["scales"], c["qzeros"], int(c["group"]))
    assert torch.equal(W, c["W"])


@pytest.mark.parametrize("name", ["scaled_mm_bf16_bias", "scaled_mm_fp16"])
def test_scaled_mm_oracle(golden_quant, name):
    c = golden_quant[name]
    o = oq.scaled_mm(c["a"].view(FP8), c["b_nk"].view(FP8).t(), c["scale_a"], c["scale_b"],
                     c["o"].dtype, c.get("bias"))
    assert torch.equal(o, c["o"])


def test_per_tensor_and_token_quant_oracle(golden_quant):
    c = golden_quant["per_tensor_dynamic"]
    q, s = oq.per_tensor_quant_fp8(c["x"])
    assert torch.equal(s, c["scale"]) and torch.equal(q.view(torch.uint8), c["q"])
    c = golden_quant["per_tensor_static"]
    q, s = oq.per_tensor_quant_fp8(c["x"], c["scale"])
    assert torch.equal(q.view(torch.uint8), c["q"])
    c = golden_quant["per_token"]
    q, s = oq.per_token_quant_fp8(c["x"])
    assert torch.equal(s.flatten(), c["scale"]) and torch.equal(q.view(torch.uint8), c["q"])


def test_input_to_float8_oracle(golden_quant):
    c = golden_quant["input_to_float8"]
    q, inv = oq.input_to_float8(c["x"])
    assert torch.equal(q.view(torch.uint8), c["q"]) and torch.equal(inv, c["inv_scale"])


def test_fp8_linear_oracle(golden_quant):
    # per-tensor x per-tensor: the reference runs torch._scaled_mm (fp8_utils.py:715); its bf16 result
    # can differ from the fp32-spelled product by bf16 rounding of the epilogue order only.
    c = golden_quant["fp8_linear_per_tensor"]
    y = oq.fp8_linear(c["x"], c["w_nk"].view(FP8).t(), c["w_scale"], None, c["bias"])
    torch.testing.assert_close(y.float(), c["y"].float(), atol=2e-3, rtol=1.6e-2)
    c = golden_quant["fp8_linear_per_token_fallback"]
    y = oq.fp8_linear(c["x"], c["w_nk"].view(FP8).t(), c["w_scale"], None, c["bias"], per_token=True)
    torch.testing.assert_close(y.float(), c["y"].float(), atol=2e-3, rtol=1.6e-2)


def test_gptq_dequant_self_consistency():
    # PARITY UNPINNED (vllm arithmetic not under /root/reference): check only the packing convention
    g = torch.Generator().manual_seed(9)
    K, N, gs = 256, 32, 128
    w = torch.randint(0, 16, (K, N), generator=g, dtype=torch.int32)
    z = torch.randint(0, 15, (K // gs, N), generator=g, dtype=torch.int32)
    s = torch.rand(K // gs, N, generator=g).to(torch.float16)
    qweight = torch.zeros(K // 8, N, dtype=torch.int32)
    for i in range(8):
        qweight |= w[i::8] << (4 * i)
    qzeros = torch.zeros(K // gs, N // 8, dtype=torch.int32)
    for j in range(8):
        qzeros |= z[:, j::8] << (4 * j)
    W = oq.gptq_dequantize(qweight, s, qzeros, None, gs)
    ref = (w - (z + 1).repeat_interleave(gs, 0)) * s.repeat_interleave(gs, 0)
    assert torch.equal(W, ref)


# ---------------------------------------------------------------- tests/golden/elementwise.pt (make_golden_elementwise.py)
RMS = ["rmsnorm_bf16_4096", "rmsnorm_fp16_1024", "rmsnorm_bf16_8192", "rmsnorm_fp16_128"]
ROPE = ["rope_bf16_d128", "rope_fp16_d128", "rope_fp16_d64"]


@pytest.mark.parametrize("name", RMS)
def test_rmsnorm_oracle_matches_reference(golden_elementwise, name):
    c = golden_elementwise[name]
    eps = float(c["eps"])
    # layernorm.py:128-146 (RMSNorm.forward_native), executed by the generator: bit-identical
    assert torch.equal(oe.rmsnorm(c["x"], c["weight"], eps), c["y"])
    y, r = oe.rmsnorm(c["x"], c["weight"], eps, c["residual"])
    assert torch.equal(y, c["y_add"]) and torch.equal(r, c["residual_out"])
    # the sgl-kernel tests' torch forms (test_norm.py:8-15,40-50) are the same arithmetic
    assert torch.equal(c["y_kernel_test"], c["y"]) and torch.equal(c["y_add_kernel_test"], c["y_add"])
    assert torch.equal(c["residual_out_kernel_test"], c["residual_out"])


@pytest.mark.parametrize("name", ["silu_mul_bf16", "silu_mul_fp16"])
def test_silu_and_mul_oracle_matches_reference(golden_elementwise, name):
    c = golden_elementwise[name]
    assert torch.equal(oe.silu_and_mul(c["x"]), c["y"])


@pytest.mark.parametrize("name", ROPE)
def test_rope_oracle_matches_reference(golden_elementwise, name):
    c = golden_elementwise[name]
    D = int(c["head_dim"])
    cache = oe.rope_cos_sin_cache(D, int(c["max_pos"]), float(c["base"]))
    assert torch.equal(cache, c["cos_sin_cache_f32"])           # _compute_cos_sin_cache, fp32
    q, k = oe.rope_neox(c["positions"], c["q"].clone(), c["k"].clone(), cache, D)
    # rotary_embedding.py:138-166 with the cache cast to the model dtype (:104-105): bit-identical
    assert torch.equal(q, c["q_out"]) and torch.equal(k, c["k_out"])
    # the sgl-kernel test's form rotates in fp32 and rounds once: within 2 ulp of the model dtype of the native form
    ulp = 2 ** -8 if c["q"].dtype == torch.bfloat16 else 2 ** -11
    torch.testing.assert_close(q.float(), c["q_out_kernel_test"].float(), atol=4 * ulp, rtol=4 * ulp)
    torch.testing.assert_close(k.float(), c["k_out_kernel_test"].float(), atol=4 * ulp, rtol=4 * ulp)


@pytest.mark.parametrize("name", ["merge_bf16", "merge_fp16"])
def test_merge_state_oracle_matches_reference(golden_elementwise, name):
    c = golden_elementwise[name]
    o, lse = oa.merge_state(c["o_a"], c["lse_a"].clone(), c["o_b"], c["lse_b"].clone())
    # merge_state_torch leaves its result in fp32 (bf16 * fp32 promotes); the op rounds to the I/O dtype
    assert c["o"].dtype == torch.float32
    assert torch.equal(o, c["o"].to(o.dtype))
    assert torch.equal(lse, c["lse"])


def test_weight_scale_utils_oracle_matches_reference(golden_elementwise):
    for name in ("requantize_unfused", "requantize_fused"):
        c = golden_elementwise[name]
        s, w = oq.requantize_with_max_scale(c["weight"].view(FP8), c["weight_scale"], c["widths"].tolist())
        assert torch.equal(s, c["max_scale"]) and torch.equal(w.view(torch.uint8), c["weight_out"]), name
    c = golden_elementwise["convert_to_channelwise"]
    assert torch.equal(oq.convert_to_channelwise(c["weight_scale"], c["widths"].tolist()), c["out"])
    c = golden_elementwise["per_tensor_dequantize"]
    assert torch.equal(oq.per_tensor_dequantize(c["weight"].view(FP8), c["scale"]), c["out"])


# ---------------------------------------------------------------- tests/golden/sched.pt (make_golden_sched.py)
@pytest.mark.parametrize("name", ["small", "one", "wide", "long"])
def test_sched_oracle_matches_reference(golden_sched, name):
    c = golden_sched[name]
    assert torch.equal(osch.get_last_loc(c["req_to_token"], c["req_pool_indices"], c["prefix_lens"]), c["last_loc"])
    r2t = osch.write_req_to_token(c["req_to_token"].clone(), c["req_pool_indices"], c["prefix_lens"], c["seq_lens"],
                                  c["extend_lens"], c["out_cache_loc"])
    assert torch.equal(r2t, c["req_to_token_after"])
    pos, start = osch.compute_position(c["prefix_lens"].to(torch.int32), c["extend_lens"].to(torch.int32))
    assert torch.equal(pos, c["positions"]) and torch.equal(start, c["extend_start_loc"])
    assert start.dtype == c["extend_start_loc"].dtype == torch.int32
