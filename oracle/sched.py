"""CPU oracle: scheduler-side request bookkeeping of an extend batch ("K10").

TEST INFRASTRUCTURE ONLY.  Restates, citations relative to /root/reference/python/sglang/srt:
  * get_last_loc            managers/schedule_batch.py:1900-1909 (get_last_loc_torch)
  * write_req_to_token      managers/schedule_batch.py:1303-1309 over mem_cache/memory_pool.py:77-78
                            (= what write_req_to_token_pool_triton :1848-1882 computes)
  * compute_position        model_executor/forward_batch_info.py:734-750 (compute_position_torch)
PINNED by tests/golden/sched.pt (tests/golden/make_golden_sched.py executes the reference's two torch functions
and replays its inline write loop); tests/test_oracle_golden.py asserts bit-identity.
"""
import torch


def get_last_loc(req_to_token, req_pool_indices, prefix_lens):
    return torch.where(prefix_lens > 0, req_to_token[req_pool_indices, prefix_lens - 1].to(prefix_lens.dtype),
                       torch.full_like(prefix_lens, -1))


def write_req_to_token(req_to_token, req_pool_indices, prefix_lens, seq_lens, extend_lens, out_cache_loc):
    """In place; request i receives out_cache_loc[sum(extend_lens[:i]) ...] at columns prefix_i .. seq_i."""
    pt = 0
    for i in range(len(req_pool_indices)):
        n = int(extend_lens[i])
        req_to_token[int(req_pool_indices[i]), int(prefix_lens[i]): int(seq_lens[i])] = \
            out_cache_loc[pt: pt + n].to(req_to_token.dtype)
        pt += n
    return req_to_token


def compute_position(extend_prefix_lens, extend_seq_lens):
    positions = torch.cat([torch.arange(int(p), int(p) + int(e)) for p, e in zip(extend_prefix_lens, extend_seq_lens)])
    start = torch.zeros_like(extend_seq_lens)
    start[1:] = torch.cumsum(extend_seq_lens[:-1], dim=0)
    return positions.to(torch.int64), start
