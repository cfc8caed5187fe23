"""GPU parity: HIP decode / indices / kv-write / merge vs the oracle and the golden vectors.
All calls go through the C ABI (iaas_sglang_amd.ops -> libmi_hotpath.so)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oa  # noqa: E402  (checker only)

DEV = "cuda"


def ops():
    from iaas_sglang_amd import ops as _ops
    return _ops


def _tol(dtype):
    # |hip - oracle| <= atol + rtol*|oracle| : 2 ulp of the I/O dtype + 1e-3 absolute
    return dict(atol=2e-3, rtol=2 ** -7 if dtype == torch.bfloat16 else 2 ** -9)


def _run_decode(c, num_splits):
    o_ = ops()
    q = c["q"].to(DEV)
    kc, vc = c["k_cache"].to(DEV).clone(), c["v_cache"].to(DEV).clone()
    r2t, rpi, sl = c["req_to_token"].to(DEV), c["req_pool_indices"].to(DEV), c["seq_lens"].to(DEV)
    loc = c["out_cache_loc"].to(DEV)
    o_.kv_write(kc, vc, loc, c["k_new"].to(DEV), c["v_new"].to(DEV))
    indptr = o_.kv_indptr(sl)
    idx = torch.empty(int(sl.sum()), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t, rpi, sl, indptr, idx)
    B, Hq, D = q.shape
    out = torch.empty_like(q)
    ws = torch.empty(max(1, o_.decode_workspace_numel(B, Hq, D, num_splits)), dtype=torch.float32, device=DEV)
    o_.decode_attention(q, kc, vc, out, indptr, idx, float(c["scaling"]), 0.0, num_splits, ws)
    torch.cuda.synchronize()
    return out.cpu(), kc.cpu(), vc.cpu(), indptr.cpu(), idx.cpu()


DECODE = ["decode_gqa4_d128_bf16", "decode_gqa8_d128_bf16", "decode_mha_d128_fp16",
          "decode_mha_d64_fp16", "decode_gqa4_d128_bf16_shifted"]


@pytest.mark.parametrize("name", DECODE)
@pytest.mark.parametrize("num_splits", [1, 3, 8])
def test_decode_matches_reference_golden(golden_attention, name, num_splits):
    c = golden_attention[name]
    out, kc, vc, indptr, idx = _run_decode(c, num_splits)
    # integer path bit-exact
    ip_ref, idx_ref = oa.kv_indices(c["req_to_token"], c["req_pool_indices"], c["seq_lens"])
    assert torch.equal(indptr, ip_ref) and torch.equal(idx, idx_ref)
    # pool contents after the write: bit-exact, every other slot untouched
    k_exp, v_exp = c["k_cache"].clone(), c["v_cache"].clone()
    k_exp[c["out_cache_loc"]] = c["k_new"]
    v_exp[c["out_cache_loc"]] = c["v_new"]
    assert torch.equal(kc, k_exp) and torch.equal(vc, v_exp)
    # values: vs the reference's own bf16/fp16 output and vs exact fp32 math
    torch.testing.assert_close(out.float(), c["o"].float(), atol=2e-2, rtol=2e-2)
    o32 = oa.decode_fp32(c["q"], k_exp, v_exp, c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                         scaling=float(c["scaling"]))
    torch.testing.assert_close(out.float(), o32, **_tol(out.dtype))


def _synthetic(B, Hq, Hkv, D, lens, dtype, seed=0, scattered=True):
    g = torch.Generator().manual_seed(seed)
    total = int(sum(lens))
    slots = total + 1
    k = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    v = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    perm = (torch.randperm(total, generator=g) + 1) if scattered else (torch.arange(total) + 1)
    ctx = max(lens) + 3
    r2t = torch.zeros(B + 2, ctx, dtype=torch.int32)
    rpi = torch.randperm(B + 2, generator=g)[:B]
    off = 0
    for i, L in enumerate(lens):
        r2t[rpi[i], :L] = perm[off:off + L].to(torch.int32)
        off += L
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    return q, k, v, r2t, rpi.to(torch.int64), torch.tensor(lens, dtype=torch.int64)


@pytest.mark.parametrize("Hq,Hkv,D,dtype", [(32, 8, 128, torch.bfloat16), (8, 8, 128, torch.float16),
                                            (12, 12, 64, torch.float16), (14, 2, 128, torch.bfloat16),
                                            (8, 1, 128, torch.bfloat16),      # C5 per rank: Llama-3-70B at TP=8
                                            (4, 1, 128, torch.bfloat16),      # C3 at TP=8
                                            (32, 32, 128, torch.float16)])    # C4: Llama-2-7B (MHA)
def test_decode_ragged_vs_oracle(Hq, Hkv, D, dtype):
    lens = [1, 2, 15, 16, 17, 31, 32, 33, 100, 511, 512, 700]
    q, k, v, r2t, rpi, sl = _synthetic(len(lens), Hq, Hkv, D, lens, dtype, seed=1)
    o_ = ops()
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    indptr = o_.kv_indptr(sl.to(DEV))
    idx = torch.empty(int(sl.sum()), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t.to(DEV), rpi.to(DEV), sl.to(DEV), indptr, idx)
    ref = oa.decode_fp32(q, k, v, r2t, rpi, sl, scaling=1.0 / math.sqrt(D))
    for ns in (1, 4, 16):
        out = torch.empty_like(qd)
        ws = torch.empty(max(1, o_.decode_workspace_numel(len(lens), Hq, D, ns)), dtype=torch.float32, device=DEV)
        o_.decode_attention(qd, kd, vd, out, indptr, idx, 1.0 / math.sqrt(D), 0.0, ns, ws)
        torch.testing.assert_close(out.cpu().float(), ref, **_tol(dtype))


def test_decode_logit_cap_and_spiked_max():
    # a spiked key forces the deferred-rescale branch late in the sequence; logit cap as in Triton
    D, Hq, Hkv = 128, 8, 2
    lens = [300, 90]
    q, k, v, r2t, rpi, sl = _synthetic(2, Hq, Hkv, D, lens, torch.bfloat16, seed=4)
    k[r2t[rpi[0], 250].item()] = (q[0, 0] * 3).to(k.dtype)  # huge logit for head 0 at token 250
    o_ = ops()
    indptr = o_.kv_indptr(sl.to(DEV))
    idx = torch.empty(int(sl.sum()), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t.to(DEV), rpi.to(DEV), sl.to(DEV), indptr, idx)
    for cap in (0.0, 30.0):
        ref = oa.decode_fp32(q, k, v, r2t, rpi, sl, scaling=1.0 / math.sqrt(D), logit_cap=cap)
        for ns in (1, 2):
            out = torch.empty(2, Hq, D, dtype=torch.bfloat16, device=DEV)
            ws = torch.empty(max(1, o_.decode_workspace_numel(2, Hq, D, ns)), dtype=torch.float32, device=DEV)
            o_.decode_attention(q.to(DEV), k.to(DEV), v.to(DEV), out, indptr, idx, 1.0 / math.sqrt(D), cap, ns, ws)
            torch.testing.assert_close(out.cpu().float(), ref, **_tol(torch.bfloat16))


def test_kv_indices_like_reference_test():
    # test/srt/test_create_kvindices.py:18-71: batch in {1, 37, 1786}, 4096 x 4096 table
    o_ = ops()
    max_batch = ctx = 4096
    r2t = torch.arange(max_batch * ctx, dtype=torch.int32, device=DEV).reshape(max_batch, ctx)
    g = torch.Generator().manual_seed(0)
    for batch in (1, 37, 1786):
        rpi = torch.randperm(max_batch, generator=g)[:batch].to(torch.int64)
        lens = torch.randperm(ctx, generator=g)[:batch].to(torch.int32)
        for lens_t in (lens, lens.to(torch.int64)):
            indptr = o_.kv_indptr(lens_t.to(DEV))
            out = torch.empty(int(lens.sum()), dtype=torch.int32, device=DEV)
            o_.kv_indices(r2t, rpi.to(DEV), lens_t.to(DEV), indptr, out)
            ip_ref, ref = oa.kv_indices(r2t.cpu(), rpi, lens)
            assert torch.equal(indptr.cpu(), ip_ref)
            assert torch.equal(out.cpu(), ref)
    # kv_start_idx variant (sliding-window builder, attention/utils.py:22-26)
    rpi = torch.tensor([5, 9], dtype=torch.int64)
    lens = torch.tensor([100, 7], dtype=torch.int32)
    start = torch.tensor([11, 0], dtype=torch.int32)
    indptr = o_.kv_indptr(lens.to(DEV))
    out = torch.empty(107, dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t, rpi.to(DEV), lens.to(DEV), indptr, out, start.to(DEV))
    _, ref = oa.kv_indices(r2t.cpu(), rpi, lens, start)
    assert torch.equal(out.cpu(), ref)


def test_merge_state_vs_oracle():
    o_ = ops()
    g = torch.Generator().manual_seed(2)
    for dtype in (torch.bfloat16, torch.float16):
        a = torch.randn(37, 6, 128, generator=g).to(dtype)
        b = torch.randn(37, 6, 128, generator=g).to(dtype)
        la = torch.randn(37, 6, generator=g) * 3
        lb = torch.randn(37, 6, generator=g) * 3
        lb[3, 2] = float("inf")
        la[5, 1] = float("-inf")
        out, lse = o_.merge_state(a.to(DEV), la.to(DEV), b.to(DEV), lb.to(DEV))
        ro, rl = oa.merge_state(a, la, b, lb)
        torch.testing.assert_close(out.cpu().float(), ro.float(), atol=1e-3, rtol=2 ** -7)
        torch.testing.assert_close(lse.cpu(), rl, atol=1e-5, rtol=1e-5)


def test_decode_full_size_properties():
    """BASELINE config (B=128, S=2048, Hq=32, Hkv=8, D=128) -- too big for the CPU oracle in
    seconds, so check size-independent properties: split-count invariance, permutation
    invariance of the slot assignment, and a constant-V identity (softmax weights sum to 1)."""
    o_ = ops()
    B, S, Hq, Hkv, D = 128, 2048, 32, 8, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    slots = B * S + 1
    k = torch.randn(slots, Hkv, D, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn(slots, Hkv, D, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn(B, Hq, D, device=DEV, generator=g, dtype=torch.float32).to(torch.bfloat16)
    perm = torch.randperm(B * S, device=DEV, generator=g).to(torch.int32) + 1
    sl = torch.full((B,), S, dtype=torch.int64, device=DEV)
    indptr = o_.kv_indptr(sl)
    scale = 1.0 / math.sqrt(D)

    def run(idx, vbuf, ns):
        out = torch.empty_like(q)
        ws = torch.empty(max(1, o_.decode_workspace_numel(B, Hq, D, ns)), dtype=torch.float32, device=DEV)
        o_.decode_attention(q, k, vbuf, out, indptr, idx, scale, 0.0, ns, ws)
        return out.float()

    base = run(perm, v, 1)
    for ns in (2, 4, 8):
        torch.testing.assert_close(run(perm, v, ns), base, atol=4e-3, rtol=2 ** -7)
    # attention is invariant to the ORDER of a request's kv_indices
    shuffled = perm.view(B, S)[:, torch.randperm(S, device=DEV, generator=g)].contiguous().view(-1)
    torch.testing.assert_close(run(shuffled, v, 4), base, atol=4e-3, rtol=2 ** -7)
    # constant V => output equals that constant (weights sum to one)
    vconst = torch.full_like(v, 0.5)
    out = run(perm, vconst, 4)
    torch.testing.assert_close(out, torch.full_like(out, 0.5), atol=1e-3, rtol=0)
    # spot-check 3 requests against the fp32 oracle
    sel = [0, 63, 127]
    r2t = perm.view(B, S).cpu()
    ref = oa.decode_fp32(q[sel].cpu(), k.cpu(), v.cpu(), r2t[sel], torch.arange(3), torch.full((3,), S),
                         scaling=scale)
    torch.testing.assert_close(base[sel].cpu(), ref, **_tol(torch.bfloat16))


# ------------------------------------------------------------------------------ extend
EXTEND = ["extend_gqa4_d128_bf16_noprefix", "extend_gqa4_d128_bf16_prefix",
          "extend_mha_d64_fp16_prefix", "extend_gqa4_d128_bf16_noncausal"]


def _run_extend(c, q, k_new, v_new, kc, vc, r2t, rpi, pre, ext, scaling, causal, logit_cap=0.0, window=-1, splits=1):
    """The backend's extend sequence through the C ABI: KV write, prefix kv_indices, kernel (splits > 1: split-KV form)."""
    o_ = ops()
    d = lambda t: t.to(DEV)
    kc, vc = d(kc).clone(), d(vc).clone()
    B = len(pre)
    lens = pre + ext
    loc = torch.cat([r2t[rpi[i], int(pre[i]):int(lens[i])] for i in range(B)]).to(torch.int64)
    o_.kv_write(kc, vc, d(loc), d(k_new), d(v_new))
    pre_d, ext_d = d(pre.to(torch.int32)), d(ext.to(torch.int32))
    kv_indptr = o_.kv_indptr(pre_d)
    qo_indptr = o_.kv_indptr(ext_d)
    idx = torch.empty(max(1, int(pre.sum())), dtype=torch.int32, device=DEV)
    o_.kv_indices(d(r2t), d(rpi), pre_d, kv_indptr, idx)
    out = torch.empty_like(d(q))
    if splits > 1:
        o_.extend_attention_splitkv(d(q), d(k_new), d(v_new), out, kc, vc, qo_indptr, kv_indptr, idx, int(ext.max()),
                                    scaling, splits, None, logit_cap, causal, window)
    else:
        o_.extend_attention(d(q), d(k_new), d(v_new), out, kc, vc, qo_indptr, kv_indptr, idx, int(ext.max()),
                            scaling, logit_cap, causal, window)
    torch.cuda.synchronize()
    return out.cpu(), kc.cpu(), vc.cpu()


@pytest.mark.parametrize("name", EXTEND)
def test_extend_matches_reference_golden(golden_attention, name):
    c = golden_attention[name]
    pre, ext = c["extend_prefix_lens"].to(torch.int64), c["extend_seq_lens"].to(torch.int64)
    out, kc, vc = _run_extend(c, c["q"], c["k_new"], c["v_new"], c["k_cache"], c["v_cache"], c["req_to_token"],
                              c["req_pool_indices"], pre, ext, float(c["scaling"]), bool(c["causal"]))
    k_exp, v_exp = c["k_cache"].clone(), c["v_cache"].clone()
    k_exp[c["out_cache_loc"]] = c["k_new"]
    v_exp[c["out_cache_loc"]] = c["v_new"]
    assert torch.equal(kc, k_exp) and torch.equal(vc, v_exp)
    torch.testing.assert_close(out.float(), c["o"].float(), atol=2e-2, rtol=2e-2)
    o32 = oa.extend_fp32(c["q"], k_exp, v_exp, c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                         c["extend_prefix_lens"], c["extend_seq_lens"], scaling=float(c["scaling"]),
                         causal=bool(c["causal"]))
    # P is rounded to the I/O dtype before the PV MFMA (as the Triton kernel does, extend_attention.py:205)
    torch.testing.assert_close(out.float(), o32, atol=4e-3, rtol=2 ** -6 if out.dtype == torch.bfloat16 else 2 ** -8)


@pytest.mark.parametrize("Hq,Hkv,D,dtype", [(32, 8, 128, torch.bfloat16), (4, 4, 128, torch.float16),
                                            (12, 12, 64, torch.float16), (8, 4, 128, torch.bfloat16),
                                            (8, 1, 128, torch.bfloat16), (6, 2, 64, torch.bfloat16)])
def test_extend_ragged_vs_oracle(Hq, Hkv, D, dtype):
    # MIXED-style batch: long prefill chunks next to 1-token extends with long prefixes
    pre = torch.tensor([0, 0, 64, 300, 33, 1, 129, 0])
    ext = torch.tensor([1, 130, 17, 1, 64, 200, 31, 65])
    lens = (pre + ext).tolist()
    g = torch.Generator().manual_seed(7)
    slots = sum(lens) + 1
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    perm = torch.randperm(slots - 1, generator=g) + 1
    B = len(lens)
    r2t = torch.zeros(B + 1, max(lens) + 2, dtype=torch.int32)
    rpi = torch.randperm(B + 1, generator=g)[:B].to(torch.int64)
    off = 0
    for i, L in enumerate(lens):
        r2t[rpi[i], :L] = perm[off:off + L].to(torch.int32)
        off += L
    r2t[rpi[4], :33] = r2t[rpi[3], :33]   # shared radix prefix
    E = int(ext.sum())
    q = torch.randn(E, Hq, D, generator=g).to(dtype)
    k_new = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v_new = torch.randn(E, Hkv, D, generator=g).to(dtype)
    scaling = D ** -0.5
    for causal, cap, window in [(True, 0.0, -1), (False, 0.0, -1), (True, 50.0, -1), (True, 0.0, 40)]:
        out, kca, vca = _run_extend(None, q, k_new, v_new, kc, vc, r2t, rpi, pre, ext, scaling, causal, cap, window)
        ref = oa.extend_fp32(q, kca, vca, r2t, rpi, torch.tensor(lens), pre, ext, scaling=scaling, causal=causal,
                             logit_cap=cap, sliding_window=window)
        torch.testing.assert_close(out.float(), ref, atol=4e-3, rtol=2 ** -6 if dtype == torch.bfloat16 else 2 ** -8)
        # split-KV form: 3 splits, and 16 (more splits than key tiles for most blocks: empty splits must merge away)
        for splits in (3, 16):
            out_s, _, _ = _run_extend(None, q, k_new, v_new, kc, vc, r2t, rpi, pre, ext, scaling, causal, cap, window, splits)
            torch.testing.assert_close(out_s.float(), ref, atol=4e-3, rtol=2 ** -6 if dtype == torch.bfloat16 else 2 ** -8)


def test_extend_reference_test_shape():
    # shape family of test/srt/test_triton_attention_kernels.py:44-181 (B=19, Hq=12, Hkv=4, D=128), reduced N_CTX
    g = torch.Generator().manual_seed(0)
    B, Hq, Hkv, D, dtype = 19, 12, 4, 128, torch.bfloat16
    pre = torch.randint(1, 300, (B,), generator=g)
    ext = torch.randint(1, 300, (B,), generator=g)
    lens = (pre + ext).tolist()
    slots = sum(lens) + 1
    kc = (torch.randn(slots, Hkv, D, generator=g) * 0.2 + 0.1).to(dtype)
    vc = (torch.randn(slots, Hkv, D, generator=g) * 0.2 + 0.1).to(dtype)
    r2t = torch.zeros(B, max(lens), dtype=torch.int32)
    off = 1
    for i, L in enumerate(lens):   # contiguous slots, like the reference test's b_start_loc layout
        r2t[i, :L] = torch.arange(off, off + L, dtype=torch.int32)
        off += L
    rpi = torch.arange(B, dtype=torch.int64)
    E = int(ext.sum())
    q = (torch.randn(E, Hq, D, generator=g) * 0.2 + 0.1).to(dtype)
    k_new = (torch.randn(E, Hkv, D, generator=g) * 0.2 + 0.1).to(dtype)
    v_new = (torch.randn(E, Hkv, D, generator=g) * 0.2 + 0.1).to(dtype)
    out, kca, vca = _run_extend(None, q, k_new, v_new, kc, vc, r2t, rpi, pre, ext, D ** -0.5, True)
    ref = oa.extend_fp32(q, kca, vca, r2t, rpi, torch.tensor(lens), pre, ext, scaling=D ** -0.5)
    # the reference's own tolerance for this test is rtol 1e-2 (:171-181)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=2e-3)


def test_extend_len1_equals_decode():
    # an extend of one token over a prefix must equal token (decode) attention over prefix+1 keys
    o_ = ops()
    g = torch.Generator().manual_seed(3)
    B, Hq, Hkv, D, dtype = 5, 32, 8, 128, torch.bfloat16
    pre = torch.tensor([7, 100, 513, 64, 1])
    ext = torch.ones(B, dtype=torch.int64)
    lens = (pre + ext).tolist()
    slots = sum(lens) + 1
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    perm = torch.randperm(slots - 1, generator=g) + 1
    r2t = torch.zeros(B, max(lens), dtype=torch.int32)
    off = 0
    for i, L in enumerate(lens):
        r2t[i, :L] = perm[off:off + L].to(torch.int32)
        off += L
    rpi = torch.arange(B, dtype=torch.int64)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    k_new = torch.randn(B, Hkv, D, generator=g).to(dtype)
    v_new = torch.randn(B, Hkv, D, generator=g).to(dtype)
    out_e, kca, vca = _run_extend(None, q, k_new, v_new, kc, vc, r2t, rpi, pre, ext, D ** -0.5, True)
    sl = torch.tensor(lens, dtype=torch.int64).to(DEV)
    indptr = o_.kv_indptr(sl)
    idx = torch.empty(sum(lens), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t.to(DEV), rpi.to(DEV), sl, indptr, idx)
    out_d = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    o_.decode_attention(q.to(DEV), kca.to(DEV), vca.to(DEV), out_d, indptr, idx, D ** -0.5)
    torch.testing.assert_close(out_e.float(), out_d.cpu().float(), atol=4e-3, rtol=2 ** -6)


@pytest.mark.parametrize("page_size", [16, 64])
def test_decode_and_extend_with_paged_slots(page_size):
    """page_size > 1 (test_flashattn_backend.py:324-346 cases): the allocator hands out page-aligned
    runs of slots, req_to_token still holds one slot id per token, so the token-granular kernels
    must give the same answers.  Slots here: request i owns whole pages, pages in random order."""
    o_ = ops()
    g = torch.Generator().manual_seed(page_size)
    Hq, Hkv, D, dtype = 32, 8, 128, torch.bfloat16
    lens = [1, page_size - 1, page_size, page_size + 1, 3 * page_size + 7, 200]
    pages_needed = [-(-L // page_size) for L in lens]
    n_pages = sum(pages_needed)
    page_order = torch.randperm(n_pages, generator=g) + 1          # page 0 is the padding page
    slots = (n_pages + 1) * page_size
    k = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    v = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    B = len(lens)
    r2t = torch.zeros(B, max(lens) + 1, dtype=torch.int32)
    off = 0
    for i, L in enumerate(lens):
        pg = page_order[off: off + pages_needed[i]]
        off += pages_needed[i]
        tok = (pg[:, None] * page_size + torch.arange(page_size)[None, :]).reshape(-1)[:L]
        r2t[i, :L] = tok.to(torch.int32)
    rpi = torch.arange(B, dtype=torch.int64)
    sl = torch.tensor(lens, dtype=torch.int64)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    indptr = o_.kv_indptr(sl.to(DEV))
    idx = torch.empty(sum(lens), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t.to(DEV), rpi.to(DEV), sl.to(DEV), indptr, idx)
    out = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    ws = torch.empty(o_.decode_workspace_numel(B, Hq, D, 4), dtype=torch.float32, device=DEV)
    o_.decode_attention(q.to(DEV), k.to(DEV), v.to(DEV), out, indptr, idx, D ** -0.5, 0.0, 4, ws)
    ref = oa.decode_fp32(q, k, v, r2t, rpi, sl, scaling=D ** -0.5)
    torch.testing.assert_close(out.cpu().float(), ref, **_tol(dtype))
    # extend: half of every request cached (page-aligned prefix slots), half new
    pre = torch.tensor([L // 2 for L in lens])
    ext = torch.tensor([L - L // 2 for L in lens])
    E = int(ext.sum())
    qe = torch.randn(E, Hq, D, generator=g).to(dtype)
    kn = torch.randn(E, Hkv, D, generator=g).to(dtype)
    vn = torch.randn(E, Hkv, D, generator=g).to(dtype)
    oute, kca, vca = _run_extend(None, qe, kn, vn, k, v, r2t, rpi, pre, ext, D ** -0.5, True)
    refe = oa.extend_fp32(qe, kca, vca, r2t, rpi, sl, pre, ext, scaling=D ** -0.5)
    torch.testing.assert_close(oute.float(), refe, atol=4e-3, rtol=2 ** -6)


# ---------------------------------------------------------------------------------------------
# speculative-decode tree mask (custom_mask) in extend -- SURVEY 8a row extend_attention_fwd / 8f row 4
def _tree_case(g, B, nd, Hq, Hkv, D, dtype, max_seq):
    from iaas_sglang_amd import ops
    seq = torch.randint(1, max_seq, (B,), generator=g)
    P, E = int(seq.sum()), B * nd
    slots = P + E + 1
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    q = torch.randn(E, Hq, D, generator=g).to(dtype)
    k = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v = torch.randn(E, Hkv, D, generator=g).to(dtype)
    perm = torch.randperm(slots - 1, generator=g) + 1
    r2t = torch.zeros(B, max_seq + nd, dtype=torch.int32)
    off, new_loc = 0, []
    for i in range(B):
        n = int(seq[i]) + nd
        r2t[i, :n] = perm[off: off + n].to(torch.int32)
        new_loc.append(perm[off + int(seq[i]): off + n])
        off += n
    new_loc = torch.cat(new_loc)
    kc[new_loc], vc[new_loc] = k, v                       # the oracle reads the new rows back from the pool
    return seq, q, k, v, kc, vc, r2t, new_loc


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nd,Hq,Hkv,D,skip_prefix", [(5, 8, 32, 8, 128, True), (3, 4, 8, 8, 64, True), (4, 16, 16, 4, 128, False),
                                                       (2, 64, 4, 2, 128, True)])
def test_extend_custom_mask_tree_verify_vs_oracle(dtype, B, nd, Hq, Hkv, D, skip_prefix):
    """Random tree masks (every draft row sees itself so no row is fully masked), prefix visible or masked."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(B * nd + Hq)
    seq, q, k, v, kc, vc, r2t, _ = _tree_case(g, B, nd, Hq, Hkv, D, dtype, 90)
    masks, mptr = [], [0]
    for i in range(B):
        S = int(seq[i]) + nd
        m = torch.rand(nd, S, generator=g) < 0.6
        m[torch.arange(nd), int(seq[i]) + torch.arange(nd)] = True         # a node always sees itself
        if not skip_prefix:
            m[:, 0] = True                                                 # keep at least one prefix key per row
        masks.append(m.reshape(-1))
        mptr.append(mptr[-1] + nd * S)
    custom_mask = torch.cat(masks)
    mask_indptr = torch.tensor(mptr, dtype=torch.int64)
    rpi = torch.arange(B)
    ext = torch.full((B,), nd, dtype=torch.int64)
    ref = oa.extend_fp32(q, kc, vc, r2t, rpi, seq + nd, seq, ext, D ** -0.5, causal=True, custom_mask=custom_mask,
                         mask_indptr=mask_indptr, skip_prefix_custom_mask=skip_prefix)
    seq_d = seq.to(torch.int32).to(DEV)
    kvp = ops.kv_indptr(seq_d).clone()
    idx = torch.empty(int(seq.sum()), dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi.to(DEV), seq_d, kvp, idx)
    qo = torch.arange(0, (B + 1) * nd, nd, dtype=torch.int32, device=DEV)
    o = torch.empty_like(q).to(DEV)
    ops.extend_attention_masked(q.to(DEV), k.to(DEV), v.to(DEV), o, kc.to(DEV), vc.to(DEV), qo, kvp, idx, custom_mask.to(DEV),
                                mask_indptr.to(DEV), nd, D ** -0.5, 0.0, skip_prefix)
    torch.cuda.synchronize()
    torch.testing.assert_close(o.cpu().float(), ref, atol=2e-2, rtol=2e-2)
    o2 = torch.empty_like(o)
    ops.extend_attention_splitkv(q.to(DEV), k.to(DEV), v.to(DEV), o2, kc.to(DEV), vc.to(DEV), qo, kvp, idx, nd, D ** -0.5, 4,
                                 None, 0.0, True, -1, 1.0, 1.0, custom_mask.to(DEV), mask_indptr.to(DEV), skip_prefix)
    torch.cuda.synchronize()
    torch.testing.assert_close(o2.cpu().float(), ref, atol=2e-2, rtol=2e-2)


def test_extend_custom_mask_equal_to_causal_reproduces_causal():
    """The reference's own check (test_triton_attention_kernels.py: a lower-triangular custom mask == causal)."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(5)
    B, nd, Hq, Hkv, D, dtype = 3, 40, 32, 8, 128, torch.bfloat16
    seq, q, k, v, kc, vc, r2t, _ = _tree_case(g, B, nd, Hq, Hkv, D, dtype, 70)
    masks, mptr = [], [0]
    for i in range(B):
        S = int(seq[i]) + nd
        m = torch.zeros(nd, S, dtype=torch.bool)
        m[:, : int(seq[i])] = True
        m[:, int(seq[i]):] = torch.tril(torch.ones(nd, nd, dtype=torch.bool))
        masks.append(m.reshape(-1)); mptr.append(mptr[-1] + nd * S)
    rpi = torch.arange(B)
    seq_d = seq.to(torch.int32).to(DEV)
    kvp = ops.kv_indptr(seq_d).clone()
    idx = torch.empty(int(seq.sum()), dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi.to(DEV), seq_d, kvp, idx)
    qo = torch.arange(0, (B + 1) * nd, nd, dtype=torch.int32, device=DEV)
    args = (q.to(DEV), k.to(DEV), v.to(DEV))
    o1, o2 = torch.empty_like(q).to(DEV), torch.empty_like(q).to(DEV)
    ops.extend_attention(*args, o1, kc.to(DEV), vc.to(DEV), qo, kvp, idx, nd, D ** -0.5, 0.0, True, -1)
    ops.extend_attention_masked(*args, o2, kc.to(DEV), vc.to(DEV), qo, kvp, idx, torch.cat(masks).to(DEV),
                                torch.tensor(mptr, dtype=torch.int64, device=DEV), nd, D ** -0.5, 0.0, False)
    torch.cuda.synchronize()
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))


def test_backend_target_verify_mode():
    """MiAttnBackend in TARGET_VERIFY: metadata as triton_backend.py:226-263, draft K/V written at out_cache_loc, tree mask."""
    from types import SimpleNamespace
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype, nd = H.TINY, torch.bfloat16, 4
    Hq, Hkv, D = shape.num_heads, shape.num_kv_heads, shape.head_dim
    runner = H.make_runner(shape, max_reqs=4, ctx=128, pool_tokens=400, dtype=dtype, device=DEV, fill_kv=True)
    runner.server_args.speculative_num_draft_tokens = nd
    backend = MiAttnBackend(runner)
    g = torch.Generator().manual_seed(9)
    seq = [30, 7, 64]
    B = len(seq)
    r2t = runner.req_to_token_pool.req_to_token
    slot, loc = 1, []
    for i, s_ in enumerate(seq):
        r2t[i, : s_ + nd] = torch.arange(slot, slot + s_ + nd, dtype=torch.int32, device=DEV)
        loc.append(torch.arange(slot + s_, slot + s_ + nd)); slot += s_ + nd
    loc = torch.cat(loc)
    masks = []
    for s_ in seq:
        m = torch.rand(nd, s_ + nd, generator=g) < 0.5
        m[torch.arange(nd), s_ + torch.arange(nd)] = True
        masks.append(m.reshape(-1))
    cm = torch.cat(masks)
    fb = SimpleNamespace(forward_mode=H.ForwardMode.TARGET_VERIFY, batch_size=B,
                         req_pool_indices=torch.arange(B, dtype=torch.int64, device=DEV),
                         seq_lens=torch.tensor(seq, dtype=torch.int64, device=DEV), seq_lens_sum=sum(seq), seq_lens_cpu=None,
                         out_cache_loc=loc.to(DEV), req_to_token_pool=runner.req_to_token_pool,
                         token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend,
                         spec_info=SimpleNamespace(custom_mask=cm.to(DEV)), positions=None)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    q = torch.randn(B * nd, Hq * D, generator=g).to(dtype)
    k = torch.randn(B * nd, Hkv, D, generator=g).to(dtype)
    v = torch.randn(B * nd, Hkv, D, generator=g).to(dtype)
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu().clone(), pool.v_buffer[0].cpu().clone()
    backend.init_forward_metadata(fb)
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    oa.set_kv_buffer(kc, vc, loc, k, v)
    mptr = torch.tensor([0] + [nd * (s_ + nd) for s_ in seq]).cumsum(0)
    ref = oa.extend_fp32(q.view(-1, Hq, D), kc, vc, r2t.cpu(), torch.arange(B), torch.tensor(seq) + nd, torch.tensor(seq),
                         torch.full((B,), nd), D ** -0.5, custom_mask=cm, mask_indptr=mptr)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), ref, atol=2e-2, rtol=2e-2)
    assert torch.equal(pool.k_buffer[0].cpu().view(torch.int16), kc.view(torch.int16))


@pytest.mark.parametrize("chunk,splits", [(16, 8), (64, 8), (256, 4), (64, 2), (1024, 3)])
def test_decode_fixed_chunk_splits_on_ragged_batch(chunk, splits):
    """split_chunk: every split covers `chunk` keys; short requests leave trailing splits empty, a request longer
    than splits * chunk falls back to S / splits (case (64, 2) with 700 keys).  Same math as any other split plan."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(chunk + splits)
    Hq, Hkv, D, dtype = 32, 8, 128, torch.bfloat16
    lens = [1, 700, 130, 17, 64, 65, 300, 511]
    B, total = len(lens), sum(lens)
    kb = torch.randn(total + 1, Hkv, D, generator=g).to(dtype)
    vb = torch.randn(total + 1, Hkv, D, generator=g).to(dtype)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    perm = torch.randperm(total, generator=g) + 1
    r2t = torch.zeros(B, max(lens), dtype=torch.int32)
    off = 0
    for i, L in enumerate(lens):
        r2t[i, :L] = perm[off:off + L].to(torch.int32); off += L
    rpi, sl = torch.arange(B), torch.tensor(lens)
    ref = oa.decode_fp32(q, kb, vb, r2t, rpi, sl, scaling=D ** -0.5)
    indptr = ops.kv_indptr(sl.to(DEV))
    idx = torch.empty(total, dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi.to(DEV), sl.to(DEV), indptr, idx)
    ws = torch.empty(ops.decode_workspace_numel(B, Hq, D, splits), dtype=torch.float32, device=DEV)
    o = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    ops.decode_attention(q.to(DEV), kb.to(DEV), vb.to(DEV), o, indptr, idx, D ** -0.5, 0.0, splits, ws, split_chunk=chunk)
    torch.cuda.synchronize()
    torch.testing.assert_close(o.cpu().float(), ref, atol=2e-3, rtol=2 ** -7)


def test_backend_ragged_decode_uses_work_list_and_matches_oracle():
    """A ragged decode batch through MiAttnBackend: the plan is fixed chunks + a launch list (longest first); the
    result is the oracle's, and the KV write still happens."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype = H.LLAMA3_8B, torch.bfloat16
    Hq, Hkv, D = 32, 8, 128
    lens = [1500, 40, 700, 513, 90, 1024, 2, 333]
    runner = H.make_runner(shape, max_reqs=8, ctx=2048, pool_tokens=sum(lens) + 8, dtype=dtype, device=DEV, fill_kv=True)
    runner.token_to_kv_pool = H.make_kv_pool(sum(lens) + 8, 1, Hkv, D, dtype, DEV, fill_random=True)
    backend = MiAttnBackend(runner)
    fb = H.make_decode_batch(runner, backend, len(lens), 0, DEV, seed=2, ragged=torch.tensor(lens))
    backend.init_forward_metadata(fb)
    md = backend.forward_metadata
    assert md.work is not None and md.split_chunk == 512 and md.num_kv_splits == 3
    assert md.work.shape[0] == sum(-(-L // 512) for L in lens)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    g = torch.Generator().manual_seed(8)
    q = torch.randn(len(lens), Hq * D, generator=g).to(dtype)
    k = torch.randn(len(lens), Hkv, D, generator=g).to(dtype)
    v = torch.randn(len(lens), Hkv, D, generator=g).to(dtype)
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu().clone(), pool.v_buffer[0].cpu().clone()
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    ref = oa.forward_decode(q, k, v, kc, vc, runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(),
                            fb.seq_lens.cpu(), fb.out_cache_loc.cpu(), Hq, Hkv, D ** -0.5)
    torch.testing.assert_close(o.cpu().float(), ref.float(), atol=2e-2, rtol=2e-2)
    oa_f = oa.decode_fp32(q.view(-1, Hq, D), kc, vc, runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(),
                          fb.seq_lens.cpu(), scaling=D ** -0.5)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), oa_f, atol=4e-3, rtol=2 ** -7)


@pytest.mark.parametrize("graph", [False, True])
def test_backend_sliding_window_decode_matches_oracle(graph):
    """Sliding-window layers (layer.sliding_window_size = W > -1) decode over the last min(S, W + 1) keys
    (triton_backend.py:186-205, 711-713); full-attention layers of the same model keep the whole sequence.  The
    expectation is the oracle's extend form with its window mask (q_pos <= k_pos + W) for one new token."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype = H.LLAMA3_8B, torch.bfloat16
    Hq, Hkv, D, W = 32, 8, 128, 100
    trials = [[300, 17, 1500, 64, 101, 102], [2000, 2000, 2000, 2000, 2000, 2000], [1, 2, 3, 100, 101, 3000]]
    bs = len(trials[0])
    runner = H.make_runner(shape, max_reqs=8, ctx=4096, pool_tokens=13000, dtype=dtype, device=DEV, fill_kv=True)
    runner.token_to_kv_pool = H.make_kv_pool(13000, 1, Hkv, D, dtype, DEV, fill_random=True)
    runner.sliding_window_size = W
    backend = MiAttnBackend(runner)
    win_layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    win_layer.sliding_window_size = W
    full_layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    pool = runner.token_to_kv_pool
    g = torch.Generator().manual_seed(8)
    rpi = torch.zeros(bs, dtype=torch.int64, device=DEV)
    seq_lens = torch.ones(bs, dtype=torch.int64, device=DEV)
    q = torch.zeros(bs, Hq * D, dtype=dtype, device=DEV)
    if graph:
        backend.init_cuda_graph_state(bs, bs)
        fbg = H.make_decode_batch(runner, backend, bs, 5, DEV, seed=1)
        fbg.req_pool_indices, fbg.seq_lens = rpi, seq_lens
        backend.init_forward_metadata_capture_cuda_graph(bs, bs, rpi, seq_lens, None, H.ForwardMode.DECODE, None)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            o_win_g = backend.forward(q, None, None, win_layer, fbg, save_kv_cache=False)
            o_full_g = backend.forward(q, None, None, full_layer, fbg, save_kv_cache=False)
    for t, lens in enumerate(trials):
        fb = H.make_decode_batch(runner, backend, bs, 0, DEV, seed=20 + t, ragged=torch.tensor(lens))
        qc = torch.randn(bs, Hq * D, generator=g).to(dtype)
        q.copy_(qc)
        if graph:
            rpi.copy_(fb.req_pool_indices); seq_lens.copy_(fb.seq_lens)
            backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, sum(lens), None, H.ForwardMode.DECODE,
                                                            None, fb.seq_lens_cpu)
            cg.replay()
            torch.cuda.synchronize()
            o_win, o_full = o_win_g.clone(), o_full_g.clone()
        else:
            backend.init_forward_metadata(fb)
            md = backend.forward_metadata
            assert md.window is not None and int(md.window.kv_indptr[bs]) == sum(min(L, W + 1) for L in lens)
            o_win = backend.forward(q, None, None, win_layer, fb, save_kv_cache=False)
            o_full = backend.forward(q, None, None, full_layer, fb, save_kv_cache=False)
        kc, vc = pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu()
        r2t = runner.req_to_token_pool.req_to_token.cpu()
        sl = torch.tensor(lens)
        want_full = oa.decode_fp32(qc.view(bs, Hq, D), kc, vc, r2t, fb.req_pool_indices.cpu(), sl, scaling=D ** -0.5)
        want_win = oa.extend_fp32(qc.view(bs, Hq, D), kc, vc, r2t, fb.req_pool_indices.cpu(), sl, sl - 1,
                                  torch.ones(bs, dtype=torch.int64), scaling=D ** -0.5, causal=True, sliding_window=W)
        torch.testing.assert_close(o_full.view(bs, Hq, D).cpu().float(), want_full, atol=4e-3, rtol=2 ** -7)
        torch.testing.assert_close(o_win.view(bs, Hq, D).cpu().float(), want_win, atol=4e-3, rtol=2 ** -7)
        assert not torch.allclose(want_win[2], want_full[2], atol=1e-2) or lens[2] <= W + 1   # the window really matters


@pytest.mark.parametrize("lens", [[12000], [9000, 300], [4096, 4096, 4096, 4096]])
@pytest.mark.parametrize("graph", [False, True])
def test_backend_small_batch_long_context_splits_beyond_the_serving_cap(lens, graph):
    """A few long requests: more than --triton-attention-num-kv-splits splits (and fewer kv heads per workgroup) so
    that the launch reaches every CU; eager plan and the device-side plan of a captured launch; vs the oracle."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype = H.LLAMA3_8B, torch.bfloat16
    Hq, Hkv, D = 32, 8, 128
    bs, tot = len(lens), sum(lens)
    runner = H.make_runner(shape, max_reqs=4, ctx=12288, pool_tokens=tot + 8, dtype=dtype, device=DEV, fill_kv=True)
    runner.token_to_kv_pool = H.make_kv_pool(tot + 8, 1, Hkv, D, dtype, DEV, fill_random=True)
    backend = MiAttnBackend(runner)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    fb = H.make_decode_batch(runner, backend, bs, 0, DEV, seed=5, ragged=torch.tensor(lens))
    g = torch.Generator().manual_seed(3)
    q = torch.randn(bs, Hq * D, generator=g).to(dtype).to(DEV)
    if graph:
        backend.init_cuda_graph_state(bs, bs)
        rpi, seq_lens = fb.req_pool_indices.clone(), torch.ones(bs, dtype=torch.int64, device=DEV)
        fbg = H.make_decode_batch(runner, backend, bs, 0, DEV, seed=5, ragged=torch.tensor(lens))
        fbg.req_pool_indices, fbg.seq_lens = rpi, seq_lens
        backend.init_forward_metadata_capture_cuda_graph(bs, bs, rpi, seq_lens, None, H.ForwardMode.DECODE, None)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            og = backend.forward(q, None, None, layer, fbg, save_kv_cache=False)
        seq_lens.copy_(fb.seq_lens)
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, tot, None, H.ForwardMode.DECODE, None,
                                                        fb.seq_lens_cpu)
        cg.replay()
        torch.cuda.synchronize()
        o = og.clone()
        assert int(backend.cuda_graph_plan_buf[1]) <= backend._graph_split_cap(bs)
    else:
        backend.init_forward_metadata(fb)
        md = backend.forward_metadata
        assert md.num_kv_splits >= backend.max_kv_splits and (bs > 2 or md.num_kv_splits > backend.max_kv_splits)
        o = backend.forward(q, None, None, layer, fb, save_kv_cache=False)
    pool = runner.token_to_kv_pool
    want = oa.decode_fp32(q.view(bs, Hq, D).cpu(), pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu(),
                          runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(), torch.tensor(lens),
                          scaling=D ** -0.5)
    torch.testing.assert_close(o.view(bs, Hq, D).cpu().float(), want, atol=4e-3, rtol=2 ** -7)


def test_backend_verify_over_long_prefixes_splits_the_key_range():
    """TARGET_VERIFY of a few requests with long committed sequences: the backend picks a split-KV launch (the
    unsplit one has batch x kv-heads workgroups); result = the oracle's masked extend."""
    from types import SimpleNamespace
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype, nd = H.LLAMA3_8B, torch.bfloat16, 8
    Hq, Hkv, D = 32, 8, 128
    lens = [5000, 3000, 4100]
    bs, tot = len(lens), sum(lens)
    runner = H.make_runner(shape, max_reqs=4, ctx=8192, pool_tokens=tot + bs * nd + 8, dtype=dtype, device=DEV, fill_kv=True)
    runner.token_to_kv_pool = H.make_kv_pool(tot + bs * nd + 8, 1, Hkv, D, dtype, DEV, fill_random=True)
    runner.server_args.speculative_num_draft_tokens = nd
    backend = MiAttnBackend(runner)
    g = torch.Generator().manual_seed(12)
    fb = H.make_decode_batch(runner, backend, bs, 0, DEV, seed=3, ragged=torch.tensor(lens))
    r2t = runner.req_to_token_pool.req_to_token
    new_loc = torch.arange(tot + 1, tot + 1 + bs * nd, dtype=torch.int64)
    for i in range(bs):                                     # the draft nodes' slots follow the committed sequence
        r2t[i, lens[i]: lens[i] + nd] = new_loc[i * nd: (i + 1) * nd].to(torch.int32).to(DEV)
    masks, mptr = [], [0]
    for i in range(bs):
        S = lens[i] + nd
        m = torch.ones(nd, S, dtype=torch.bool)
        m[:, lens[i]:] = torch.rand(nd, nd, generator=g) < 0.5
        m[torch.arange(nd), lens[i] + torch.arange(nd)] = True
        masks.append(m.reshape(-1)); mptr.append(mptr[-1] + nd * S)
    custom_mask = torch.cat(masks)
    fb.forward_mode = H.ForwardMode.TARGET_VERIFY
    fb.spec_info = SimpleNamespace(custom_mask=custom_mask.to(DEV))
    fb.out_cache_loc = new_loc.to(DEV)
    backend.init_forward_metadata(fb)
    assert backend.forward_metadata.num_kv_splits > 1
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    q = torch.randn(bs * nd, Hq * D, generator=g).to(dtype)
    k = torch.randn(bs * nd, Hkv, D, generator=g).to(dtype)
    v = torch.randn(bs * nd, Hkv, D, generator=g).to(dtype)
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu()
    assert torch.equal(kc[new_loc], k) and torch.equal(vc[new_loc], v)
    sl = torch.tensor(lens)
    ref = oa.extend_fp32(q.view(-1, Hq, D), kc, vc, r2t.cpu(), fb.req_pool_indices.cpu(), sl + nd, sl,
                         torch.full((bs,), nd), D ** -0.5, causal=True, custom_mask=custom_mask,
                         mask_indptr=torch.tensor(mptr, dtype=torch.int64), skip_prefix_custom_mask=True)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)


# ---------------------------------------------------------------- page-granular decode
@pytest.mark.parametrize("page_size", [16, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_paged_decode_matches_token_granular(page_size, dtype):
    """One index per page instead of one per token: the page tables are bit-exact against the python form, and the
    decode kernel walks the same slots in the same order -- outputs identical to the token-granular call bit for bit
    (and so within the oracle tolerance), for unsplit, split and ragged-plan launches."""
    o_ = ops()
    g = torch.Generator().manual_seed(page_size)
    B, Hq, Hkv, D = 7, 8, 2, 128
    lens = torch.tensor([1, page_size, page_size + 1, 5 * page_size - 3, 700, 33, 2 * page_size], dtype=torch.int64)
    npages = [-(-int(n) // page_size) for n in lens]
    pages = torch.randperm(sum(npages) + 5, generator=g)[: sum(npages)] + 1            # page 0 = padding sink
    ctx = int(lens.max()) + 7
    r2t = torch.zeros(B + 2, ctx, dtype=torch.int32)
    rpi = torch.randperm(B + 2, generator=g)[:B].to(torch.int64)
    off = 0
    for i in range(B):
        sl = (pages[off: off + npages[i]].view(-1, 1) * page_size + torch.arange(page_size).view(1, -1)).reshape(-1)
        r2t[rpi[i], : int(lens[i])] = sl[: int(lens[i])].to(torch.int32)
        off += npages[i]
    slots = (int(pages.max()) + 1) * page_size
    k = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    v = torch.randn(slots, Hkv, D, generator=g).to(dtype)
    q = torch.randn(B, Hq, D, generator=g).to(dtype)
    # page tables: bit-exact against the python form
    pi, px = o_.kv_page_tables(r2t.to(DEV), rpi.to(DEV), lens.to(DEV), page_size)
    want_pi = torch.tensor([0] + list(torch.tensor(npages).cumsum(0)), dtype=torch.int32)
    assert torch.equal(pi.cpu(), want_pi)
    want_px = torch.cat([r2t[rpi[i], : int(lens[i]): page_size] // page_size for i in range(B)])
    assert torch.equal(px.cpu()[: want_px.numel()], want_px)
    indptr = o_.kv_indptr(lens.to(DEV))
    idx = torch.empty(int(lens.sum()), dtype=torch.int32, device=DEV)
    o_.kv_indices(r2t.to(DEV), rpi.to(DEV), lens.to(DEV), indptr, idx)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    sm = D ** -0.5
    for splits in (1, 3):
        ws = torch.empty(max(o_.decode_workspace_numel(B, Hq, D, splits), 1), dtype=torch.float32, device=DEV)
        o_tok = torch.empty_like(qd)
        o_.decode_attention(qd, kd, vd, o_tok, indptr, idx, sm, 0.0, splits, ws)
        o_pg = torch.empty_like(qd)
        o_.decode_attention_paged(qd, kd, vd, indptr, pi, px, page_size, sm, 0.0, splits, ws, o=o_pg)
        assert torch.equal(o_pg, o_tok)
    ref = oa.decode_fp32(q, k, v, r2t, rpi, lens, scaling=sm)
    torch.testing.assert_close(o_pg.cpu().float(), ref, atol=4e-3, rtol=2 ** -7)
    # the fp8-output form and the backend's own metadata path
    qs = torch.tensor([0.05], device=DEV)
    o8_tok = torch.empty(B, Hq * D, dtype=torch.float8_e4m3fn, device=DEV)
    o_.decode_attention_fp8out(qd, kd, vd, o8_tok, qs, indptr, idx, sm, 0.0, 1, None)
    o8_pg = torch.empty(B, Hq * D, dtype=torch.float8_e4m3fn, device=DEV)
    o_.decode_attention_paged(qd, kd, vd, indptr, pi, px, page_size, sm, 0.0, 1, None, o_fp8=o8_pg, o_scale=qs)
    assert torch.equal(o8_pg.view(torch.uint8), o8_tok.view(torch.uint8))


@pytest.mark.parametrize("page_size", [16, 64])
def test_backend_decode_on_a_paged_pool(page_size):
    """MiAttnBackend on a runner with page_size >= 16: decode metadata carries page tables (eager and graph replay) and
    forward_decode gives the bits of the token-granular backend on the same pool."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype, B = H.TINY, torch.bfloat16, 6
    lens = torch.tensor([40, 3, 129, 64, 17, 100])
    out = {}
    for ps in (page_size, 1):
        runner = H.make_runner(shape, max_reqs=B, ctx=160, pool_tokens=(sum(-(-int(n) // page_size) for n in lens) + 2) * page_size,
                               dtype=dtype, device=DEV, fill_kv=True, page_size=page_size)
        runner.page_size = ps               # same paged pool; ps = 1 makes the backend take the token-granular path
        backend = MiAttnBackend(runner)
        assert (backend.page_size > 1) == (ps > 1)
        fb = H.make_decode_batch(SimpleNamespaceLike(runner, page_size), backend, B, 0, DEV, seed=3, ragged=lens)
        layer = H.AttnLayer(shape.num_heads, shape.head_dim, shape.head_dim ** -0.5, shape.num_kv_heads, 0)
        g = torch.Generator().manual_seed(1)
        q = torch.randn(B, shape.num_heads * shape.head_dim, generator=g).to(dtype).to(DEV)
        kn = torch.randn(B, shape.num_kv_heads, shape.head_dim, generator=g).to(dtype).to(DEV)
        vn = torch.randn(B, shape.num_kv_heads, shape.head_dim, generator=g).to(dtype).to(DEV)
        backend.init_forward_metadata(fb)
        assert (backend.forward_metadata.page_indptr is not None) == (ps > 1)
        eager = backend.forward(q, kn, vn, layer, fb)
        # graph-mode metadata: capture-time fill values, then a replay with the real lengths
        backend.init_cuda_graph_state(B, B)
        rpi, sl = fb.req_pool_indices.clone(), torch.ones(B, dtype=torch.int64, device=DEV)
        backend.init_forward_metadata_capture_cuda_graph(B, B, rpi, sl, None, H.ForwardMode.DECODE, None)
        o_buf = torch.empty_like(eager)
        graph = torch.cuda.CUDAGraph()
        o_buf.copy_(backend.forward(q, kn, vn, layer, fb))
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            o_buf.copy_(backend.forward(q, kn, vn, layer, fb))
        sl.copy_(fb.seq_lens)
        backend.init_forward_metadata_replay_cuda_graph(B, rpi, sl, int(lens.sum()), None, H.ForwardMode.DECODE, None, lens)
        graph.replay()
        torch.cuda.synchronize()
        out[ps] = (eager.clone(), o_buf.clone())
    assert torch.equal(out[page_size][0], out[1][0])
    torch.testing.assert_close(out[page_size][1].float(), out[1][1].float(), atol=4e-3, rtol=2 ** -7)
    torch.testing.assert_close(out[page_size][1].float(), out[page_size][0].float(), atol=4e-3, rtol=2 ** -7)


class SimpleNamespaceLike:
    """A runner view with another page_size for make_decode_batch (the pool layout stays paged)."""

    def __init__(self, runner, page_size):
        self.__dict__.update(runner.__dict__)
        self.page_size = page_size


def test_extend_output_row_pitch_fallback():
    """The 32x32 extend kernel stores 16-byte pieces of whole head rows: an output whose row pitch is not a multiple of
    8 elements (or is misaligned) must take the 16x16 kernel and give the same answer."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(11)
    B, S, Hq, Hkv, D, dtype = 3, 160, 8, 2, 128, torch.bfloat16
    E = B * S
    q = torch.randn(E, Hq, D, generator=g).to(dtype).to(DEV)
    k = torch.randn(E, Hkv, D, generator=g).to(dtype).to(DEV)
    v = torch.randn(E, Hkv, D, generator=g).to(dtype).to(DEV)
    kb = torch.zeros(2, Hkv, D, dtype=dtype, device=DEV)
    ext = torch.full((B,), S, dtype=torch.int32, device=DEV)
    pre = torch.zeros(B, dtype=torch.int32, device=DEV)
    qo, kvp = ops.kv_indptr(ext), ops.kv_indptr(pre).clone()
    idx = torch.ones(1, dtype=torch.int32, device=DEV)
    o_ref = torch.empty_like(q)
    ops.extend_attention(q, k, v, o_ref, kb, kb, qo, kvp, idx, S, D ** -0.5, 0.0, True, -1)
    big = torch.zeros(E, Hq * D + 4, dtype=dtype, device=DEV)          # row pitch 1028: a multiple of 4, not of 8
    o_pitch = big[:, : Hq * D].unflatten(1, (Hq, D))
    assert o_pitch.stride(0) == Hq * D + 4 and o_pitch.stride(1) == D
    ops.extend_attention(q, k, v, o_pitch, kb, kb, qo, kvp, idx, S, D ** -0.5, 0.0, True, -1)
    torch.cuda.synchronize()
    torch.testing.assert_close(o_pitch.float(), o_ref.float(), atol=4e-3, rtol=2 ** -6)
    assert float(big[:, Hq * D:].abs().max()) == 0.0                   # nothing written past the rows


# ---------------------------------------------------------------- page-granular prefix in extend (SURVEY 8f-3 remainder)
def _paged_prefix_case(g, pre, ext, page_size, Hq=32, Hkv=8, D=128, dtype=torch.bfloat16):
    """Requests own whole pages in random order (PagedTokenToKVPoolAllocator layout); returns everything both forms need."""
    from iaas_sglang_amd import ops
    lens = [p + e for p, e in zip(pre, ext)]
    need = [-(-L // page_size) for L in lens]
    order = torch.randperm(sum(need), generator=g) + 1
    slots = (sum(need) + 1) * page_size
    B = len(lens)
    r2t = torch.zeros(B, max(lens) + 1, dtype=torch.int32)
    off = 0
    for i, L in enumerate(lens):
        pg = order[off: off + need[i]]
        off += need[i]
        r2t[i, :L] = (pg[:, None] * page_size + torch.arange(page_size)[None, :]).reshape(-1)[:L].to(torch.int32)
    kc = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    vc = torch.randn(slots, Hkv, D, generator=g).to(dtype).to(DEV)
    E = sum(ext)
    q = torch.randn(E, Hq, D, generator=g).to(dtype).to(DEV)
    kn = torch.randn(E, Hkv, D, generator=g).to(dtype).to(DEV)
    vn = torch.randn(E, Hkv, D, generator=g).to(dtype).to(DEV)
    rpi = torch.arange(B, dtype=torch.int64, device=DEV)
    pre_d = torch.tensor(pre, dtype=torch.int32, device=DEV)
    ext_d = torch.tensor(ext, dtype=torch.int32, device=DEV)
    kvp, qop = ops.kv_indptr(pre_d).clone(), ops.kv_indptr(ext_d).clone()
    idx = torch.empty(max(1, sum(pre)), dtype=torch.int32, device=DEV)
    ops.kv_indices(r2t.to(DEV), rpi, pre_d, kvp, idx)
    pi, px = ops.kv_page_tables(r2t.to(DEV), rpi, pre_d, page_size)
    return q, kn, vn, kc, vc, qop, kvp, idx, pi.clone(), px


@pytest.mark.parametrize("page_size", [16, 64])
@pytest.mark.parametrize("pre,ext", [([1024] * 4, [256] * 4), ([0, 17, 2048, 63, 640], [2048, 64, 100, 700, 129])])
def test_extend_page_granular_prefix_is_bit_identical_to_token_granular(page_size, pre, ext):
    """mi_extend_attn_paged (one page id per page of the cached prefix in extend_attn32_kernel, kv_indices for the rows
    the 16x16 kernel takes) against mi_extend_attn on the same page-aligned pool: the same bits."""
    from iaas_sglang_amd import ops
    g = torch.Generator().manual_seed(page_size + len(pre))
    q, kn, vn, kc, vc, qop, kvp, idx, pi, px = _paged_prefix_case(g, pre, ext, page_size)
    D = q.shape[2]
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    ops.extend_attention(q, kn, vn, o1, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1)
    ops.extend_attention_paged(q, kn, vn, o2, kc, vc, qop, kvp, idx, pi, px, page_size, max(ext), D ** -0.5, 0.0, True, -1)
    torch.cuda.synchronize()
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    # a wrong page table must change the result (the kernel really reads it) when some request has a long extend + prefix
    if any(p > 0 and e >= 64 for p, e in zip(pre, ext)):
        o3 = torch.empty_like(q)
        used = int(pi[-1])                                  # only VALID page ids, in another order (stays in bounds)
        wrong = px.clone()
        wrong[:used] = torch.roll(px[:used], 1)
        ops.extend_attention_paged(q, kn, vn, o3, kc, vc, qop, kvp, idx, pi, wrong, page_size,
                                   max(ext), D ** -0.5, 0.0, True, -1)
        torch.cuda.synchronize()
        assert not torch.equal(o1.view(torch.int16), o3.view(torch.int16))


@pytest.mark.parametrize("page_size", [16, 64])
def test_backend_extend_on_a_paged_pool_uses_page_tables(page_size):
    """MiAttnBackend on a runner with page_size >= 16: EXTEND metadata carries the prefix page tables and forward_extend
    takes mi_extend_attn_paged; result = the fp32 oracle."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    Hq, Hkv, D, dtype = 32, 8, 128, torch.bfloat16
    pre, ext = [512, 0, 1000], [300, 128, 65]
    lens = [p + e for p, e in zip(pre, ext)]
    pool_tokens = (sum(-(-L // page_size) for L in lens) + 2) * page_size
    runner = H.make_runner(H.LLAMA3_8B, max_reqs=4, ctx=2048, pool_tokens=pool_tokens, dtype=dtype, device=DEV)
    runner.token_to_kv_pool = H.make_kv_pool(pool_tokens, 1, Hkv, D, dtype, DEV, fill_random=True)
    runner.page_size = page_size
    backend = MiAttnBackend(runner)
    assert backend.page_size == page_size
    g = torch.Generator().manual_seed(3)
    # page-aligned layout: request i owns whole pages in random order
    need = [-(-L // page_size) for L in lens]
    order = torch.randperm(sum(need), generator=g) + 1
    r2t = runner.req_to_token_pool.req_to_token
    off, loc = 0, []
    for i, L in enumerate(lens):
        pg = order[off: off + need[i]]
        off += need[i]
        sl = (pg[:, None] * page_size + torch.arange(page_size)[None, :]).reshape(-1)[:L]
        r2t[i, :L] = sl.to(torch.int32).to(DEV)
        loc.append(sl[pre[i]:])
    loc = torch.cat(loc).to(torch.int64)
    from types import SimpleNamespace
    fb = SimpleNamespace(forward_mode=H.ForwardMode.EXTEND, batch_size=3,
                         req_pool_indices=torch.arange(3, dtype=torch.int64, device=DEV),
                         seq_lens=torch.tensor(lens, dtype=torch.int64, device=DEV), seq_lens_sum=sum(lens),
                         extend_prefix_lens=torch.tensor(pre, dtype=torch.int64, device=DEV),
                         extend_seq_lens=torch.tensor(ext, dtype=torch.int64, device=DEV),
                         extend_prefix_lens_cpu=pre, extend_seq_lens_cpu=ext, out_cache_loc=loc.to(DEV),
                         req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                         attn_backend=backend, spec_info=None, positions=None)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    E = sum(ext)
    q = torch.randn(E, Hq * D, generator=g).to(dtype)
    k = torch.randn(E, Hkv, D, generator=g).to(dtype)
    v = torch.randn(E, Hkv, D, generator=g).to(dtype)
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu().clone(), pool.v_buffer[0].cpu().clone()
    backend.init_forward_metadata(fb)
    assert backend.forward_metadata.page_indptr is not None and backend.forward_metadata.page_size == page_size
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    torch.cuda.synchronize()
    oa.set_kv_buffer(kc, vc, loc, k, v)
    ref = oa.extend_fp32(q.view(-1, Hq, D), kc, vc, r2t.cpu(), torch.arange(3), torch.tensor(lens), torch.tensor(pre),
                         torch.tensor(ext), scaling=D ** -0.5, causal=True)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("pre,ext,paged", [
    ([0] * 3, [2048, 300, 64], False),                       # long extends, no prefix: the epilogue writes the fp8 rows
    ([1024, 17, 0, 640], [256, 1000, 65, 129], True),        # page-granular prefix, ragged tails inside a 32-row block
    ([512, 0], [40, 63], False),                             # short extends: T-typed kernel + quant launch (o required)
])
def test_extend_fp8_output_is_bit_identical_to_extend_then_static_quant(dtype, pre, ext, paged):
    """mi_extend_attn_fp8out (prefill's form of mi_decode_attn_fp8out: the o_proj input quantised by the attention
    kernel) against mi_extend_attn[_paged] followed by mi_fp8_quant_per_tensor(static): the same bytes; with `o` given
    the T-typed result equals the plain call too."""
    from iaas_sglang_amd import ops
    page_size = 16
    g = torch.Generator().manual_seed(11 + len(pre))
    q, kn, vn, kc, vc, qop, kvp, idx, pi, px = _paged_prefix_case(g, pre, ext, page_size)
    q, kn, vn, kc, vc = (t.to(dtype) for t in (q, kn, vn, kc, vc))
    E, Hq, D = q.shape
    scale = torch.tensor([0.0123], dtype=torch.float32, device=DEV)
    ref = torch.empty_like(q)
    if paged:
        ops.extend_attention_paged(q, kn, vn, ref, kc, vc, qop, kvp, idx, pi, px, page_size, max(ext), D ** -0.5, 0.0, True, -1)
    else:
        ops.extend_attention(q, kn, vn, ref, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1)
    ref8, _ = ops.fp8_quant_per_tensor(ref.view(E, Hq * D), scale)
    fused = ops.extend_fp8_out_is_fused(D, max(ext), 0.0, -1)
    assert fused == (max(ext) >= 64)
    pargs = dict(page_indptr=pi, page_indices=px, page_size=page_size) if paged else {}
    # fp8 only (where the kernel writes it itself), and fp8 + T-typed output
    outs = []
    if fused:
        o8 = torch.full((E, Hq * D), 0x7f, dtype=torch.uint8, device=DEV).view(ops.FP8_DTYPE)
        ops.extend_attention_fp8out(q, kn, vn, o8, scale, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1, **pargs)
        outs.append((o8, None))
    else:
        with pytest.raises(RuntimeError):
            ops.extend_attention_fp8out(q, kn, vn, torch.empty(E, Hq * D, dtype=ops.FP8_DTYPE, device=DEV), scale, kc, vc,
                                        qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1, **pargs)
    o8 = torch.full((E, Hq * D), 0x7f, dtype=torch.uint8, device=DEV).view(ops.FP8_DTYPE)
    o = torch.empty_like(q)
    ops.extend_attention_fp8out(q, kn, vn, o8, scale, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1, o=o, **pargs)
    outs.append((o8, o))
    torch.cuda.synchronize()
    for o8, o in outs:
        assert torch.equal(o8.view(torch.uint8), ref8.view(torch.uint8))
        if o is not None:
            assert torch.equal(o.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("pre,ext,paged", [([0] * 2, [1500, 200], False), ([1024, 17, 0], [256, 1000, 65], True)])
def test_extend_q_rotation_on_load_is_bit_identical_to_rope_then_extend(dtype, pre, ext, paged):
    """mi_extend_attn_fp8out with q_positions + the T-typed rotary cache (Q rotated in registers as the long-extend kernel
    loads it; the partner of the qkv GEMM epilogue's rotate_q = 0) against mi_rope_neox on q followed by the same call
    without them: the same fp8 bytes and the same T-typed output."""
    from iaas_sglang_amd import harness as H, ops
    page_size = 16
    g = torch.Generator().manual_seed(23 + len(pre))
    q, kn, vn, kc, vc, qop, kvp, idx, pi, px = _paged_prefix_case(g, pre, ext, page_size)
    q, kn, vn, kc, vc = (t.to(dtype) for t in (q, kn, vn, kc, vc))
    E, Hq, D = q.shape
    cache = H.rope_cache(D, 4096, 10000.0, DEV)
    cache_t = cache.to(dtype)
    pos = torch.cat([torch.arange(p, p + e) for p, e in zip(pre, ext)]).to(DEV)
    pos[3] = 4000                                    # positions are READ, not derived from the prefix length
    scale = torch.tensor([0.0123], dtype=torch.float32, device=DEV)
    pargs = dict(page_indptr=pi, page_indices=px, page_size=page_size) if paged else {}
    q_rot = q.clone()
    k_dummy = kn.clone()
    ops.rope_neox_(q_rot.view(E, Hq * D), k_dummy.view(E, -1), pos, cache, D)
    ref8 = torch.empty(E, Hq * D, dtype=ops.FP8_DTYPE, device=DEV)
    ref = torch.empty_like(q)
    ops.extend_attention_fp8out(q_rot, kn, vn, ref8, scale, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1, o=ref, **pargs)
    o8 = torch.full((E, Hq * D), 0x7f, dtype=torch.uint8, device=DEV).view(ops.FP8_DTYPE)
    o = torch.empty_like(q)
    ops.extend_attention_fp8out(q, kn, vn, o8, scale, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1, o=o,
                                q_positions=pos, cos_sin_cache_t=cache_t, **pargs)
    torch.cuda.synchronize()
    assert torch.equal(o8.view(torch.uint8), ref8.view(torch.uint8))
    assert torch.equal(o.view(torch.int16), ref.view(torch.int16))
    assert not torch.equal(q.view(torch.int16), q_rot.view(torch.int16))


@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (4, 2), (3, 3), (16, 2)])       # GQA 8 (70B per rank at TP = 8), 2, MHA, 8
def test_extend_fused_prefill_forms_other_head_geometries(Hq, Hkv):
    """fp8 output + Q rotation on load of the long-extend kernel at the head groupings its launch distinguishes (8 / 4 / 2 /
    1 q heads per kv head and workgroup): against rope on q + mi_extend_attn + static quant, bit for bit."""
    from iaas_sglang_amd import harness as H, ops
    pre, ext, page_size, dtype = [300, 0, 64], [200, 700, 65], 16, torch.bfloat16
    g = torch.Generator().manual_seed(Hq * 10 + Hkv)
    q, kn, vn, kc, vc, qop, kvp, idx, pi, px = _paged_prefix_case(g, pre, ext, page_size, Hq=Hq, Hkv=Hkv)
    E, _, D = q.shape
    cache = H.rope_cache(D, 4096, 10000.0, DEV)
    pos = torch.cat([torch.arange(p, p + e) for p, e in zip(pre, ext)]).to(DEV)
    scale = torch.tensor([0.02], dtype=torch.float32, device=DEV)
    q_rot, k_dummy = q.clone(), kn.clone()
    ops.rope_neox_(q_rot.view(E, Hq * D), k_dummy.view(E, -1), pos, cache, D)
    ref = torch.empty_like(q)
    ops.extend_attention(q_rot, kn, vn, ref, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1)
    ref8, _ = ops.fp8_quant_per_tensor(ref.view(E, Hq * D), scale)
    o8 = torch.full((E, Hq * D), 0x7f, dtype=torch.uint8, device=DEV).view(ops.FP8_DTYPE)
    ops.extend_attention_fp8out(q, kn, vn, o8, scale, kc, vc, qop, kvp, idx, max(ext), D ** -0.5, 0.0, True, -1,
                                q_positions=pos, cos_sin_cache_t=cache.to(dtype))
    torch.cuda.synchronize()
    assert torch.equal(o8.view(torch.uint8), ref8.view(torch.uint8))
