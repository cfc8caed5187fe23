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
