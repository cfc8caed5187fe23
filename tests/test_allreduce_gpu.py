"""Native xGMI all-reduce: N processes on the one GPU of the test box, peers mapped through hipIpc
(same protocol as N GPUs).  Integer-valued inputs so the expected sum is exact, sizes 512 B .. 2 MiB
as in sgl-kernel/tests/test_custom_allreduce.py:42-55."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaas_sglang_amd.custom_all_reduce import CustomAllreduce
    ok = True
    ca = CustomAllreduce(dist.group.WORLD, torch.device("cuda", 0), max_size=4 * 1024 * 1024)
    try:
        assert not ca.disabled
        g = torch.Generator().manual_seed(1234)               # same stream of random numbers on every rank
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            for nbytes in (512, 4096, 32768, 262144, 524288, 1048576, 2097152, 4 * 1024 * 1024):
                n = nbytes // torch.empty(0, dtype=dtype).element_size()
                allx = torch.randint(-3, 4, (world, n), generator=g).to(dtype)
                x = allx[rank].cuda()
                dist.barrier()                                  # keep host-side skew between ranks small
                for _ in range(3):                              # back-to-back calls exercise flag reuse
                    y = ca.all_reduce(x)
                torch.cuda.synchronize()
                good = (not ca.timed_out()) and torch.equal(y.cpu().float(), allx.float().sum(0))
                if not good:
                    bad = (y.cpu().float() != allx.float().sum(0)).nonzero().flatten()
                    print(f"[rank {rank}] FAIL dtype={dtype} bytes={nbytes} timed_out={ca.timed_out()} "
                          f"n_bad={bad.numel()} first_bad={bad[:4].tolist()}", flush=True)
                ok = ok and good
        ok = ok and ca.custom_all_reduce(torch.zeros(5, dtype=torch.bfloat16, device="cuda")) is None  # 10 B: not 16-B multiple
    finally:
        ca.close()
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_custom_allreduce_same_gpu(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)
