"""Host-side bookkeeping classes (SURVEY 8a rows ReqToTokenPool / MHATokenToKVPool / TokenToKVPoolAllocator):
integer behaviour restated from mem_cache/memory_pool.py:49-96,176-260 and mem_cache/allocator.py:36-153 and
checked on the CPU (no kernel is launched: set_kv_buffer is covered by the gpu tests)."""
import torch

from iaas_sglang_amd.mem_cache import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator


def test_req_to_token_pool_free_list_order():
    p = ReqToTokenPool(size=4, max_context_len=16, device="cpu")
    assert p.req_to_token.shape == (4, 16) and p.req_to_token.dtype == torch.int32 and int(p.req_to_token.sum()) == 0
    assert p.alloc(2) == [0, 1] and p.available_size() == 2
    assert p.alloc(3) is None                       # cannot be served: None, nothing taken (memory_pool.py:80-82)
    assert p.available_size() == 2
    p.free(0)
    p.free([1])
    assert p.alloc(4) == [2, 3, 0, 1]               # FIFO reuse, exactly as the reference's list slicing
    p.clear()
    assert p.alloc(1) == [0]
    p.write((0, slice(0, 3)), torch.tensor([5, 6, 7], dtype=torch.int32))
    assert p.req_to_token[0, :4].tolist() == [5, 6, 7, 0]


def test_allocator_hands_out_slots_from_one_and_never_slot_zero():
    pool = MHATokenToKVPool(size=10, page_size=1, dtype=torch.bfloat16, head_num=2, head_dim=8, layer_num=2, device="cpu")
    a = TokenToKVPoolAllocator(10, torch.bfloat16, "cpu", pool)
    assert a.available_size() == 10 and a.page_size == 1 and a.get_kvcache() is pool
    x = a.alloc(4)
    assert x.dtype == torch.int64 and x.tolist() == [1, 2, 3, 4]          # slot 0 is the padding sink (allocator.py:120-124)
    assert a.alloc(7) is None and a.available_size() == 6
    y = a.alloc(6)
    assert y.tolist() == [5, 6, 7, 8, 9, 10] and a.available_size() == 0
    a.free(x[1:3])
    a.free(torch.empty(0, dtype=torch.int64))                              # no-op (allocator.py:139-140)
    assert a.alloc(2).tolist() == [2, 3]
    # grouped frees are applied at free_group_end, in order (allocator.py:80-87)
    a.free_group_begin()
    a.free(y[:2]); a.free(x[:1])
    assert a.available_size() == 0
    a.free_group_end()
    assert a.alloc(3).tolist() == [5, 6, 1]
    state = a.backup_state()
    a.free(y[2:])
    a.restore_state(state)
    assert a.available_size() == 0
    a.clear()
    assert a.alloc(10).tolist() == list(range(1, 11))
    for f in (a.alloc_extend, a.alloc_decode):
        try:
            f()
            raise AssertionError("paged entry points must raise on the page_size=1 allocator")
        except NotImplementedError:
            pass


def test_kv_pool_layout_and_fp8_storage():
    pool = MHATokenToKVPool(size=6, page_size=1, dtype=torch.bfloat16, head_num=2, head_dim=8, layer_num=3, device="cpu")
    assert len(pool.k_buffer) == 3 and pool.k_buffer[0].shape == (7, 2, 8)   # size + page_size rows (memory_pool.py:236-249)
    assert pool.get_key_buffer(1) is pool.k_buffer[1] and pool.token_stride == 16
    k, v = pool.get_kv_size_bytes()
    assert k == v == 3 * 7 * 16 * 2
    p8 = MHATokenToKVPool(size=6, page_size=1, dtype=torch.float8_e4m3fn, head_num=2, head_dim=8, layer_num=1, device="cpu")
    assert p8.store_dtype == torch.uint8 and p8.k_buffer[0].dtype == torch.uint8     # stored as bytes (memory_pool.py:113-117)
    kb, vb = p8.get_kv_buffer(0)
    assert kb.dtype == torch.float8_e4m3fn and kb.data_ptr() == p8.k_buffer[0].data_ptr() and vb.shape == (7, 2, 8)
    try:
        MHATokenToKVPool(size=6, page_size=1, dtype=torch.float32, head_num=2, head_dim=8, layer_num=1, device="cpu")
        raise AssertionError("fp32 KV has no kernel")
    except NotImplementedError:
        pass


# ---- pinned by vectors produced by RUNNING the reference's allocator.py (tests/golden/make_golden_alloc.py)
import os  # noqa: E402

GOLD = torch.load(os.path.join(os.path.dirname(__file__), "golden", "allocator.pt"), weights_only=True)


def _replay_token_allocator(a):
    log = []
    x = a.alloc(5); log.append(("alloc5", x.clone()))
    log.append(("alloc9_none", torch.tensor([-1 if a.alloc(9) is None else 0])))
    y = a.alloc(7); log.append(("alloc7", y.clone()))
    a.free(x[1:4]); log.append(("avail", torch.tensor([a.available_size()])))
    log.append(("alloc2", a.alloc(2).clone()))
    a.free_group_begin(); a.free(y[:3]); a.free(x[:1]); a.free_group_end()
    log.append(("alloc4", a.alloc(4).clone()))
    a.clear(); log.append(("alloc12", a.alloc(12).clone()))
    return log


def test_token_allocator_equals_reference_run():
    a = TokenToKVPoolAllocator(12, torch.bfloat16, "cpu", None)
    got = _replay_token_allocator(a)
    want = GOLD["token_allocator_log"]
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, g), (_, w) in zip(got, want):
        assert torch.equal(g, w), n


def test_paged_allocator_torch_parts_equal_reference_run():
    from iaas_sglang_amd.mem_cache import PagedTokenToKVPoolAllocator
    a = PagedTokenToKVPoolAllocator(64, 4, torch.bfloat16, "cpu", None)
    log = []
    x = a.alloc(12); log.append(("alloc12", x.clone()))
    log.append(("alloc_too_many", torch.tensor([-1 if a.alloc(64) is None else 0])))
    a.free(x[2:9]); log.append(("free_pages_after_free", a.free_pages.clone()))
    log.append(("alloc8", a.alloc(8).clone()))
    a.clear(); log.append(("free_pages_after_clear", a.free_pages.clone()))
    for (n, g), (n2, w) in zip(log, GOLD["paged_log"]):
        assert n == n2 and torch.equal(g, w), n
    assert a.available_size() == 64


def test_alloc_oracle_equals_reference_torch_forms():
    """oracle/alloc.py (restatement of the Triton kernels) == the reference's own torch forms, on every golden case."""
    from oracle import alloc as oal
    n = 0
    for c in GOLD["paged_cases"]:
        if c["kind"] == "extend":
            out, pages, toks = oal.alloc_extend(c["prefix_lens"], c["seq_lens"], c["last_loc"], c["free_pages"], c["page_size"])
            assert toks == int((c["seq_lens"] - c["prefix_lens"]).sum())
        else:
            out, pages = oal.alloc_decode(c["seq_lens"], c["last_loc"], c["free_pages"], c["page_size"])
        assert torch.equal(out, c["out_indices"]) and pages == c["num_new_pages"]
        n += 1
    assert n == 12
    # the case the reference's torch form mishandles: an extension that stays inside the old partial page
    out, pages, toks = oal.alloc_extend(torch.tensor([5]), torch.tensor([7]), torch.tensor([4 * 8 + 4]), torch.tensor([9]), 8)
    assert out.tolist() == [37, 38] and pages == 0 and toks == 2
