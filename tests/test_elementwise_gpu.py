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
