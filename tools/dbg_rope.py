import sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from iaas_sglang_amd import harness as H, ops
from test_fused_gpu import _fp8_operands, _bits
DEV = "cuda"
for dtype in (torch.float16, torch.bfloat16):
    M, Hq, Hkv, D, K = 128, 32, 8, 128, 4096
    g = torch.Generator().manual_seed(M + Hq + K)
    N = (Hq + 2 * Hkv) * D
    qx, w, xs, ws = _fp8_operands(M, N, K, g)
    slots = 4 * M + 1
    cache = H.rope_cache(D, 4096, 10000.0, DEV)
    pos = torch.randint(0, 4096, (M,), generator=g).to(DEV)
    loc = (torch.randperm(slots - 1, generator=g)[:M] + 1).to(DEV)
    kc2 = torch.zeros(slots, Hkv, D, dtype=dtype, device=DEV); vc2 = torch.zeros_like(kc2)
    qkv = ops.fp8_gemm(qx, w, xs, ws, dtype)
    qkv0 = qkv.clone()
    q1, k1 = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D]
    ops.rope_neox_(q1, k1, pos, cache, D)
    q2 = ops.fp8_gemm_rope_kvwrite(qx, w, xs, ws, pos, cache, kc2, vc2, loc, Hq, Hkv, D)
    torch.cuda.synchronize()
    d = (_bits(q1.contiguous()) != _bits(q2))
    print(dtype, "mismatch", int(d.sum()), "of", d.numel())
    if d.any():
        idx = d.nonzero()[:10]
        for r, c in idx.tolist():
            h, e = c // D, c % D
            e1 = e % (D // 2)
            print(r, c, "h", h, "e", e, float(q1[r, c]), float(q2[r, c]), "x1", float(qkv0[r, h * D + e1]), "x2", float(qkv0[r, h * D + e1 + D // 2]),
                  "cos", float(cache[pos[r], e1]), "sin", float(cache[pos[r], D // 2 + e1]))
