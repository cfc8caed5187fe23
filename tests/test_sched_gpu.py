"""GPU parity (bit-exact) of the scheduler-side integer kernels against reference-run vectors
(tests/golden/sched.pt) and the oracle at the sizes a C3 prefill batch has."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sched as osch  # noqa: E402  (checker only)

DEV = "cuda"


def ops():
    from iaas_sglang_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("name", ["small", "one", "wide", "long"])
def test_sched_kernels_golden(golden_sched, name):
    o_ = ops()
    c = {k: v.to(DEV) for k, v in golden_sched[name].items()}
    assert torch.equal(o_.get_last_loc(c["req_to_token"], c["req_pool_indices"], c["prefix_lens"]), c["last_loc"])
    r2t = c["req_to_token"].clone()
    o_.write_req_to_token(r2t, c["req_pool_indices"], c["prefix_lens"], c["seq_lens"], c["extend_lens"], c["out_cache_loc"])
    assert torch.equal(r2t, c["req_to_token_after"])                     # written columns AND every untouched one
    pre32, ext32 = c["prefix_lens"].to(torch.int32), c["extend_lens"].to(torch.int32)
    pos, start = o_.compute_position(pre32, ext32, int(c["extend_lens"].sum()))
    assert pos.dtype == torch.int64 and start.dtype == torch.int32
    assert torch.equal(pos, c["positions"]) and torch.equal(start, c["extend_start_loc"])
    # int64 lengths, and "no prefixes" = an empty prefix tensor (forward_batch_info.py:683)
    pos64, start64 = o_.compute_position(c["prefix_lens"], c["extend_lens"], int(c["extend_lens"].sum()))
    assert torch.equal(pos64, c["positions"]) and torch.equal(start64, c["extend_start_loc"])
    pos0, _ = o_.compute_position(pre32[:0], ext32, int(c["extend_lens"].sum()))
    want0, _ = osch.compute_position(torch.zeros_like(golden_sched[name]["extend_lens"]), golden_sched[name]["extend_lens"])
    assert torch.equal(pos0.cpu(), want0)


def test_sched_kernels_full_prefill_batch():
    """128 requests x 2048 new tokens (the metric's prefill batch) + a ragged one: vs the oracle, bit-exact."""
    o_ = ops()
    g = torch.Generator().manual_seed(8)
    for bs, lens in ((128, torch.full((128,), 2048, dtype=torch.int64)),
                     (97, torch.randint(1, 4096, (97,), generator=g, dtype=torch.int64))):
        ctx = 4096 + 4
        pre = torch.minimum(torch.randint(0, 1024, (bs,), generator=g, dtype=torch.int64), ctx - lens)
        seq = pre + lens
        rpi = torch.randperm(256, generator=g)[:bs].to(torch.int64)
        r2t = torch.randint(1, 1 << 20, (256, ctx), generator=g, dtype=torch.int32)
        loc = torch.randperm(1 << 22, generator=g)[: int(lens.sum())].to(torch.int64) + 1
        want = osch.write_req_to_token(r2t.clone(), rpi, pre, seq, lens, loc)
        got = r2t.to(DEV)
        o_.write_req_to_token(got, rpi.to(DEV), pre.to(DEV), seq.to(DEV), lens.to(DEV), loc.to(DEV))
        assert torch.equal(got.cpu(), want)
        pos, start = o_.compute_position(pre.to(torch.int32).to(DEV), lens.to(torch.int32).to(DEV), int(lens.sum()))
        wp, ws = osch.compute_position(pre.to(torch.int32), lens.to(torch.int32))
        assert torch.equal(pos.cpu(), wp) and torch.equal(start.cpu(), ws)
        assert torch.equal(o_.get_last_loc(got, rpi.to(DEV), pre.to(DEV)).cpu(), osch.get_last_loc(want, rpi, pre))


def test_sched_kernels_empty_batch():
    o_ = ops()
    e64 = torch.empty(0, dtype=torch.int64, device=DEV)
    r2t = torch.zeros(4, 8, dtype=torch.int32, device=DEV)
    o_.write_req_to_token(r2t, e64, e64, e64, e64, e64)
    assert o_.get_last_loc(r2t, e64, e64).numel() == 0
    pos, start = o_.compute_position(e64.to(torch.int32), e64.to(torch.int32), 0)
    assert pos.numel() == 0 and start.numel() == 0 and int(r2t.abs().sum()) == 0
