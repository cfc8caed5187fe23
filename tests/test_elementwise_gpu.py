"""GPU parity for the layer-glue kernels vs their native-torch restatement."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import elementwise as oe  # noqa: E402

DEV = "cuda"


def ops():
    from iaas_sglang_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,H", [(128, 4096), (3, 768), (17, 8192), (1, 14336)])
def test_rmsnorm(dtype, M, H):
    o_ = ops()
    g = torch.Generator().manual_seed(H)
    x = torch.randn(M, H, generator=g).to(dtype)
    r = torch.randn(M, H, generator=g).to(dtype)
    w = (torch.rand(H, generator=g) + 0.5).to(dtype)
    out = o_.rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    torch.testing.assert_close(out.cpu().float(), oe.rmsnorm(x, w, 1e-5).float(), rtol=2 * ulp, atol=1e-6)
    rd = r.to(DEV).clone()
    out = o_.rmsnorm(x.to(DEV), w.to(DEV), 1e-5, residual=rd)
    ref, rref = oe.rmsnorm(x, w, 1e-5, r)
    assert torch.equal(rd.cpu(), rref)                     # the residual stream is bit-exact
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=2 * ulp, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (12, 12, 64)])
def test_rope_neox_bit_exact(dtype, Hq, Hkv, D):
    o_ = ops()
    g = torch.Generator().manual_seed(D)
    T = 37
    cache = oe.rope_cos_sin_cache(D, 4096)
    pos = torch.randint(0, 4096, (T,), generator=g)
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, generator=g).to(dtype)
    q, k = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D]
    rq, rk = oe.rope_neox(pos, q.clone(), k.clone(), cache, D)
    qkv_d = qkv.to(DEV)
    o_.rope_neox_(qkv_d[:, : Hq * D], qkv_d[:, Hq * D: (Hq + Hkv) * D], pos.to(DEV), cache.to(DEV), D)
    got = qkv_d.cpu()
    assert torch.equal(got[:, : Hq * D], rq) and torch.equal(got[:, Hq * D: (Hq + Hkv) * D], rk)
    assert torch.equal(got[:, (Hq + Hkv) * D:], qkv[:, (Hq + Hkv) * D:])  # v untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul(dtype):
    o_ = ops()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(65, 2 * 14336, generator=g) * 2).to(dtype)
    out = o_.silu_and_mul(x.to(DEV))
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    # two roundings (silu, then the product): allow 2 ulp of the output dtype
    torch.testing.assert_close(out.cpu().float(), oe.silu_and_mul(x).float(), rtol=4 * ulp, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_fp8_producers_equal_unfused_pair(dtype):
    """rmsnorm_fp8 / silu_and_mul_fp8 == producer followed by the static per-tensor quant kernel, bit for bit."""
    o_ = ops()
    g = torch.Generator().manual_seed(5)
    M, H = 128, 4096
    x = torch.randn(M, H, generator=g).to(dtype).to(DEV)
    r = torch.randn(M, H, generator=g).to(dtype).to(DEV)
    w = (torch.rand(H, generator=g) + 0.5).to(dtype).to(DEV)
    scale = torch.tensor([0.011], device=DEV)
    r1, r2 = r.clone(), r.clone()
    q_fused = o_.rmsnorm_fp8(x, w, 1e-5, scale, residual=r1)
    q_ref, _ = o_.fp8_quant_per_tensor(o_.rmsnorm(x, w, 1e-5, residual=r2), scale)
    assert torch.equal(r1, r2)
    assert torch.equal(q_fused.view(torch.uint8), q_ref.view(torch.uint8))
    q_fused = o_.rmsnorm_fp8(x, w, 1e-5, scale)
    q_ref, _ = o_.fp8_quant_per_tensor(o_.rmsnorm(x, w, 1e-5), scale)
    assert torch.equal(q_fused.view(torch.uint8), q_ref.view(torch.uint8))
    gu = (torch.randn(M, 2 * 14336, generator=g) * 2).to(dtype).to(DEV)
    q_fused = o_.silu_and_mul_fp8(gu, scale)
    q_ref, _ = o_.fp8_quant_per_tensor(o_.silu_and_mul(gu), scale)
    assert torch.equal(q_fused.view(torch.uint8), q_ref.view(torch.uint8))


# ---------------------------------------------------------------- reference-run vectors (tests/golden/elementwise.pt)
@pytest.mark.parametrize("name", ["rmsnorm_bf16_4096", "rmsnorm_fp16_1024", "rmsnorm_bf16_8192", "rmsnorm_fp16_128"])
def test_rmsnorm_golden(golden_elementwise, name):
    """RMSNorm.forward_native (layernorm.py:128-146) run by the generator: the residual stream is bit-exact, the normed
    output within 2 ulp of the model dtype (the kernel reduces the sum of squares in a different order)."""
    o_ = ops()
    c = golden_elementwise[name]
    eps = float(c["eps"])
    ulp = 2 ** -8 if c["x"].dtype == torch.bfloat16 else 2 ** -11
    out = o_.rmsnorm(c["x"].to(DEV), c["weight"].to(DEV), eps)
    torch.testing.assert_close(out.cpu().float(), c["y"].float(), rtol=2 * ulp, atol=1e-6)
    rd = c["residual"].to(DEV).clone()
    out = o_.rmsnorm(c["x"].to(DEV), c["weight"].to(DEV), eps, residual=rd)
    assert torch.equal(rd.cpu(), c["residual_out"])
    torch.testing.assert_close(out.cpu().float(), c["y_add"].float(), rtol=2 * ulp, atol=1e-6)


@pytest.mark.parametrize("name", ["rope_bf16_d128", "rope_fp16_d128", "rope_fp16_d64"])
def test_rope_golden_bit_exact(golden_elementwise, name):
    """RotaryEmbedding.forward_native (rotary_embedding.py:138-166) run by the generator: bit-exact."""
    o_ = ops()
    c = golden_elementwise[name]
    q, k = c["q"].to(DEV).clone(), c["k"].to(DEV).clone()
    o_.rope_neox_(q, k, c["positions"].to(DEV), c["cos_sin_cache_f32"].to(DEV), int(c["head_dim"]))
    assert torch.equal(q.cpu(), c["q_out"]) and torch.equal(k.cpu(), c["k_out"])


@pytest.mark.parametrize("name", ["silu_mul_bf16", "silu_mul_fp16"])
def test_silu_and_mul_golden(golden_elementwise, name):
    """SiluAndMul.forward_native (activation.py:56-58) run by the generator: within 2 roundings of the model dtype."""
    o_ = ops()
    c = golden_elementwise[name]
    out = o_.silu_and_mul(c["x"].to(DEV))
    ulp = 2 ** -8 if c["x"].dtype == torch.bfloat16 else 2 ** -11
    torch.testing.assert_close(out.cpu().float(), c["y"].float(), rtol=4 * ulp, atol=1e-6)


@pytest.mark.parametrize("name", ["merge_bf16", "merge_fp16"])
def test_merge_state_golden(golden_elementwise, name):
    """merge_state_torch (sgl-kernel/tests/test_merge_state_v2.py:101-135) run by the generator."""
    o_ = ops()
    c = golden_elementwise[name]
    out, lse = o_.merge_state(c["o_a"].to(DEV), c["lse_a"].to(DEV), c["o_b"].to(DEV), c["lse_b"].to(DEV))
    torch.testing.assert_close(out.cpu().float(), c["o"], atol=1e-3, rtol=2 ** -7)
    torch.testing.assert_close(lse.cpu(), c["lse"], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("name", ["requantize_unfused", "requantize_fused"])
def test_requantize_with_max_scale_golden(golden_elementwise, name):
    """Fp8LinearMethod.process_weights_after_loading on a fused module against requantize_with_max_scale
    (quantization/utils.py:94-119) run by the generator: weight bytes and the scale bit-exact."""
    from iaas_sglang_amd.quantization import Fp8Config
    c = golden_elementwise[name]
    widths = c["widths"].tolist()
    cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="dynamic")

    class FakeLinear(torch.nn.Module):
        output_partition_sizes = widths
    lin = FakeLinear()
    method = cfg.get_quant_method(lin, "model.layers.0.self_attn.qkv_proj")
    K = c["weight"].shape[1]
    method.create_weights(lin, K, widths, K, sum(widths), torch.bfloat16, weight_loader=None)
    lin = lin.to(DEV)
    lin.weight.data.copy_(c["weight"].view(torch.float8_e4m3fn))
    lin.weight_scale.data.copy_(c["weight_scale"])
    method.process_weights_after_loading(lin)
    assert torch.equal(lin.weight.data.t().contiguous().cpu().view(torch.uint8), c["weight_out"])
    assert torch.equal(lin.weight_scale.data.cpu().reshape(()), c["max_scale"])
