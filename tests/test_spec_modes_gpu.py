"""Speculative-decoding forward modes of MiAttnBackend: DRAFT_EXTEND and draft decode metadata
(triton_backend.py:200-202, 265-283), and TARGET_VERIFY / DRAFT_EXTEND under graph capture + replay
(triton_backend.py:445-520, 579-627).  Expectation: the fp32 oracle (torch_native_backend.py:27-180 arithmetic)."""
from types import SimpleNamespace

import pytest
import torch

from oracle import attention as oa

pytestmark = pytest.mark.gpu
DEV = "cuda"
Hq, Hkv, D = 32, 8, 128


def _setup(lens, extra, steps=None, nd=None, ctx=4096):
    """Runner + backend with a random pool; request i owns lens[i] + extra consecutive slots of req_to_token."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    tot = sum(lens) + extra * len(lens)
    runner = H.make_runner(H.LLAMA3_8B, max_reqs=max(8, len(lens)), ctx=ctx, pool_tokens=tot + 8, dtype=torch.bfloat16, device=DEV,
                           fill_kv=True)
    runner.token_to_kv_pool = H.make_kv_pool(tot + 8, 1, Hkv, D, torch.bfloat16, DEV, fill_random=True)
    runner.server_args.speculative_num_steps = steps
    runner.server_args.speculative_num_draft_tokens = nd
    backend = MiAttnBackend(runner)
    layer = H.AttnLayer(Hq, D, D ** -0.5, Hkv, 0)
    return H, runner, backend, layer


def _assign_slots(runner, lens_with_new, g):
    """Scattered slots: request i gets lens_with_new[i] slots of a random permutation."""
    r2t = runner.req_to_token_pool.req_to_token
    r2t.zero_()
    tot = sum(lens_with_new)
    perm = (torch.randperm(tot, generator=g) + 1).to(torch.int32)
    off = 0
    for i, n in enumerate(lens_with_new):
        r2t[i, :n] = perm[off: off + n].to(DEV)
        off += n
    return r2t


def _draft_extend_batch(H, runner, backend, seq, acc, g):
    """forward_batch of a DRAFT_EXTEND step: seq_lens count the accepted tokens already (eagle_utils.py:593)."""
    bs = len(seq)
    r2t = _assign_slots(runner, seq, g)
    loc = torch.cat([r2t[i, seq[i] - acc[i]: seq[i]] for i in range(bs)]).to(torch.int64)
    fb = SimpleNamespace(forward_mode=H.ForwardMode.DRAFT_EXTEND, batch_size=bs,
                         req_pool_indices=torch.arange(bs, dtype=torch.int64, device=DEV),
                         seq_lens=torch.tensor(seq, dtype=torch.int64, device=DEV), seq_lens_sum=sum(seq),
                         seq_lens_cpu=torch.tensor(seq), extend_seq_lens_cpu=list(acc), out_cache_loc=loc,
                         req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                         attn_backend=backend, positions=None,
                         spec_info=SimpleNamespace(accept_length=torch.tensor(acc, dtype=torch.int32, device=DEV)))
    return fb, loc


def _extend_ref(runner, q, k, v, loc, seq, acc, **kw):
    pool = runner.token_to_kv_pool
    kc, vc = pool.k_buffer[0].cpu().clone(), pool.v_buffer[0].cpu().clone()
    oa.set_kv_buffer(kc, vc, loc.cpu(), k, v)
    bs = len(seq)
    sl, ext = torch.tensor(seq), torch.tensor(acc)
    return oa.extend_fp32(q.view(-1, Hq, D), kc, vc, runner.req_to_token_pool.req_to_token.cpu(), torch.arange(bs), sl,
                          sl - ext, ext, D ** -0.5, **kw), kc, vc


@pytest.mark.parametrize("with_cpu_lens", [True, False])
def test_backend_draft_extend_mode(with_cpu_lens):
    """qo_indptr = scan(spec_info.accept_length); the pool side covers seq_len - accept_length keys, the new rows come
    from k/v and are written at out_cache_loc; every key is attended once (the torch-native arithmetic)."""
    seq, acc = [37, 5, 300, 1024, 64], [3, 1, 4, 2, 4]
    H, runner, backend, layer = _setup(seq, 0, steps=3)
    g = torch.Generator().manual_seed(21)
    fb, loc = _draft_extend_batch(H, runner, backend, seq, acc, g)
    if not with_cpu_lens:
        fb.extend_seq_lens_cpu, fb.seq_lens_cpu = None, None       # bound = speculative_num_steps + 1 rows per request
    T = sum(acc)
    q = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    backend.init_forward_metadata(fb)
    md = backend.forward_metadata
    assert md.qo_indptr.tolist() == torch.tensor([0] + acc).cumsum(0).tolist()
    assert md.kv_indptr.tolist() == torch.tensor([0] + [s - a for s, a in zip(seq, acc)]).cumsum(0).tolist()
    assert md.max_extend_len == (max(acc) if with_cpu_lens else 4)
    o = backend.forward(q.to(DEV), k.to(DEV), v.to(DEV), layer, fb)
    ref, kc, _ = _extend_ref(runner, q, k, v, loc, seq, acc, causal=True)
    torch.testing.assert_close(o.view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)
    assert torch.equal(runner.token_to_kv_pool.k_buffer[0].cpu().view(torch.int16), kc.view(torch.int16))


def test_backend_draft_decode_mode_uses_the_draft_workers_index_arrays():
    """Decode with spec_info (triton_backend.py:200-202): kv_indptr / kv_indices arrive ready-made, one row per
    (request, top-k branch); rows of one request share its committed prefix and differ in their draft tail."""
    from iaas_sglang_amd import ops
    seq, topk, step = [130, 900, 17], 2, 2
    H, runner, backend, layer = _setup(seq, topk * step)
    g = torch.Generator().manual_seed(4)
    r2t = _assign_slots(runner, [s + topk * step for s in seq], g).cpu()
    rows, lists = len(seq) * topk, []
    for i, s in enumerate(seq):
        for b in range(topk):
            lists.append(torch.cat([r2t[i, :s], r2t[i, s + b * step: s + (b + 1) * step]]))
    kv_indptr = torch.tensor([0] + [len(l) for l in lists]).cumsum(0).to(torch.int32).to(DEV)
    kv_indices = torch.cat(lists).to(torch.int32).to(DEV)
    fb = SimpleNamespace(forward_mode=H.ForwardMode.DECODE, batch_size=rows,
                         req_pool_indices=torch.arange(len(seq), dtype=torch.int64, device=DEV).repeat_interleave(topk),
                         seq_lens=torch.tensor([s + step for s in seq for _ in range(topk)], dtype=torch.int64, device=DEV),
                         seq_lens_sum=int(kv_indices.numel()), seq_lens_cpu=None, out_cache_loc=None,
                         req_to_token_pool=runner.req_to_token_pool, token_to_kv_pool=runner.token_to_kv_pool,
                         attn_backend=backend, positions=None,
                         spec_info=SimpleNamespace(kv_indptr=kv_indptr, kv_indices=kv_indices))
    backend.init_forward_metadata(fb)
    assert backend.forward_metadata.kv_indices is kv_indices
    q = torch.randn(rows, Hq * D, generator=g).to(torch.bfloat16)
    o = backend.forward(q.to(DEV), None, None, layer, fb, save_kv_cache=False)
    pool = runner.token_to_kv_pool
    fake_r2t = torch.zeros(rows, max(len(l) for l in lists), dtype=torch.int32)
    for r, l in enumerate(lists):
        fake_r2t[r, : len(l)] = l
    want = oa.decode_fp32(q.view(rows, Hq, D), pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu(), fake_r2t, torch.arange(rows),
                          torch.tensor([len(l) for l in lists]), scaling=D ** -0.5)
    torch.testing.assert_close(o.view(rows, Hq, D).cpu().float(), want, atol=4e-3, rtol=2 ** -7)


def test_draft_extend_captured_and_replayed():
    """Capture with speculative_num_steps + 1 rows per request and fill lengths, replay twice with other accept
    lengths / sequences (triton_backend.py:476-503, 608-624): each replay = the eager result = the oracle."""
    steps, bs = 3, 4
    per = steps + 1
    trials = [([200, 31, 1500, 64], [4, 1, 2, 3]), ([9, 2000, 40, 700], [1, 4, 4, 2])]
    H, runner, backend, layer = _setup([2000] * bs, 0, steps=steps)
    g = torch.Generator().manual_seed(33)
    backend.init_cuda_graph_state(bs, bs * per)
    rpi = torch.arange(bs, dtype=torch.int64, device=DEV)
    seq_lens = torch.full((bs,), per, dtype=torch.int64, device=DEV)
    acc_buf = torch.full((bs,), per, dtype=torch.int32, device=DEV)
    spec = SimpleNamespace(accept_length=acc_buf)
    q = torch.zeros(bs * per, Hq * D, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros(bs * per, Hkv, D, dtype=torch.bfloat16, device=DEV)
    v = torch.zeros(bs * per, Hkv, D, dtype=torch.bfloat16, device=DEV)
    loc = torch.zeros(bs * per, dtype=torch.int64, device=DEV)          # padding rows write to the sink slot 0
    _assign_slots(runner, [per] * bs, g)
    fbg = SimpleNamespace(forward_mode=H.ForwardMode.DRAFT_EXTEND, batch_size=bs, req_pool_indices=rpi, seq_lens=seq_lens,
                          out_cache_loc=loc, req_to_token_pool=runner.req_to_token_pool,
                          token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend, positions=None, spec_info=spec)
    backend.init_forward_metadata_capture_cuda_graph(bs, bs * per, rpi, seq_lens, None, H.ForwardMode.DRAFT_EXTEND, spec)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        og = backend.forward(q, k, v, layer, fbg)
    for seq, acc in trials:
        fb, new_loc = _draft_extend_batch(H, runner, backend, seq, acc, g)
        T = sum(acc)
        qc = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16)
        kc_ = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
        vc_ = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
        ref, _, _ = _extend_ref(runner, qc, kc_, vc_, new_loc, seq, acc, causal=True)
        q.zero_(); k.zero_(); v.zero_(); loc.zero_()
        q[:T], k[:T], v[:T], loc[:T] = qc.to(DEV), kc_.to(DEV), vc_.to(DEV), new_loc
        seq_lens.copy_(fb.seq_lens); acc_buf.copy_(fb.spec_info.accept_length)
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, sum(seq), None, H.ForwardMode.DRAFT_EXTEND,
                                                        spec, torch.tensor(seq))
        cg.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(og[:T].view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)


def _tree_masks(seq, nd, g):
    masks = []
    for s in seq:
        m = torch.ones(nd, s + nd, dtype=torch.bool)
        m[:, s:] = torch.rand(nd, nd, generator=g) < 0.5
        m[torch.arange(nd), s + torch.arange(nd)] = True
        masks.append(m.reshape(-1))
    return torch.cat(masks)


@pytest.mark.parametrize("bs,nd", [(3, 8), (16, 4)])
def test_target_verify_captured_and_replayed(bs, nd):
    """TARGET_VERIFY under capture (triton_backend.py:445-475) and replay (:579-607): kv_indptr / kv_indices /
    custom_mask / mask_indptr rewritten in the persistent buffers; replay == eager (bit for bit when the split count
    agrees) == the oracle's masked extend."""
    g = torch.Generator().manual_seed(7 + bs)
    trials = [[int(x) for x in torch.randint(1, 3000, (bs,), generator=g)] for _ in range(2)]
    H, runner, backend, layer = _setup([3000] * bs, nd, nd=nd)
    backend.init_cuda_graph_state(bs, bs * nd)
    rpi = torch.arange(bs, dtype=torch.int64, device=DEV)
    seq_lens = torch.ones(bs, dtype=torch.int64, device=DEV)
    q = torch.zeros(bs * nd, Hq * D, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros(bs * nd, Hkv, D, dtype=torch.bfloat16, device=DEV)
    v = torch.zeros(bs * nd, Hkv, D, dtype=torch.bfloat16, device=DEV)
    loc = torch.zeros(bs * nd, dtype=torch.int64, device=DEV)
    _assign_slots(runner, [1 + nd] * bs, g)
    spec = SimpleNamespace(custom_mask=_tree_masks([1] * bs, nd, g).to(DEV))
    fbg = SimpleNamespace(forward_mode=H.ForwardMode.TARGET_VERIFY, batch_size=bs, req_pool_indices=rpi, seq_lens=seq_lens,
                          out_cache_loc=loc, req_to_token_pool=runner.req_to_token_pool,
                          token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend, positions=None, spec_info=spec)
    backend.init_forward_metadata_capture_cuda_graph(bs, bs * nd, rpi, seq_lens, None, H.ForwardMode.TARGET_VERIFY, spec)
    captured_splits = backend.forward_metadata.num_kv_splits
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        og = backend.forward(q, k, v, layer, fbg)
    for seq in trials:
        r2t = _assign_slots(runner, [s + nd for s in seq], g)
        new_loc = torch.cat([r2t[i, seq[i]: seq[i] + nd] for i in range(bs)]).to(torch.int64)
        cm = _tree_masks(seq, nd, g)
        qc = torch.randn(bs * nd, Hq * D, generator=g).to(torch.bfloat16)
        kc_ = torch.randn(bs * nd, Hkv, D, generator=g).to(torch.bfloat16)
        vc_ = torch.randn(bs * nd, Hkv, D, generator=g).to(torch.bfloat16)
        q.copy_(qc); k.copy_(kc_); v.copy_(vc_); loc.copy_(new_loc)
        seq_lens.copy_(torch.tensor(seq))
        spec_r = SimpleNamespace(custom_mask=cm.to(DEV))
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, sum(seq), None, H.ForwardMode.TARGET_VERIFY,
                                                        spec_r, torch.tensor(seq))
        cg.replay()
        torch.cuda.synchronize()
        got = og.clone()
        pool = runner.token_to_kv_pool
        kc, vc = pool.k_buffer[0].cpu(), pool.v_buffer[0].cpu()
        assert torch.equal(kc[new_loc.cpu()], kc_)
        sl = torch.tensor(seq)
        mptr = torch.tensor([0] + [nd * (s + nd) for s in seq]).cumsum(0)
        ref = oa.extend_fp32(qc.view(-1, Hq, D), kc, vc, r2t.cpu(), torch.arange(bs), sl + nd, sl, torch.full((bs,), nd),
                             D ** -0.5, causal=True, custom_mask=cm, mask_indptr=mptr, skip_prefix_custom_mask=True)
        torch.testing.assert_close(got.view(-1, Hq, D).cpu().float(), ref, atol=4e-3, rtol=2 ** -6)
        # eager step on the same inputs
        fbe = SimpleNamespace(forward_mode=H.ForwardMode.TARGET_VERIFY, batch_size=bs, req_pool_indices=rpi,
                              seq_lens=seq_lens.clone(), seq_lens_sum=sum(seq), seq_lens_cpu=torch.tensor(seq),
                              out_cache_loc=loc, req_to_token_pool=runner.req_to_token_pool,
                              token_to_kv_pool=runner.token_to_kv_pool, attn_backend=backend, positions=None, spec_info=spec_r)
        backend.init_forward_metadata(fbe)
        oe = backend.forward(q, k, v, layer, fbe)
        torch.cuda.synchronize()
        if backend.forward_metadata.num_kv_splits == captured_splits:
            assert torch.equal(oe.view(torch.int16), got.view(torch.int16))
        else:
            torch.testing.assert_close(oe.float(), got.float(), atol=4e-3, rtol=2 ** -6)
