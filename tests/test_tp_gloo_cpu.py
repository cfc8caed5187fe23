"""world_size-2 gloo test of the N>1 call sites (CPU): column-parallel -> row-parallel + ONE
all-reduce reproduces the unsharded product, the vocab all-gather reassembles logits, and the
max-over-ranks timing rule.  Compute is plain torch here: the subject is the exchange pattern
(iaas_sglang_amd/parallel.py), which is byte-for-byte what runs over RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaas_sglang_amd import parallel as par
    torch.manual_seed(0)                       # same full weights on every rank
    H, I, V, T = 64, 96, 40, 5
    x = torch.randn(T, H, dtype=torch.float64)
    w_up = torch.randn(I, H, dtype=torch.float64)      # column parallel: shard rows (output dim)
    w_down = torch.randn(H, I, dtype=torch.float64)    # row parallel: shard columns (input dim)
    w_lm = torch.randn(V, H, dtype=torch.float64)      # vocab parallel
    sz = par.shard_sizes(8, 2, I, V, world)
    i0, i1 = rank * sz["intermediate"], (rank + 1) * sz["intermediate"]
    partial = (x @ w_up[i0:i1].t()) @ w_down[:, i0:i1].t()
    y = par.tensor_model_parallel_all_reduce(partial.clone(), world, None)
    v0, v1 = rank * sz["vocab"], (rank + 1) * sz["vocab"]
    logits = par.tensor_model_parallel_all_gather(y @ w_lm[v0:v1].t(), world, None)
    full = ((x @ w_up.t()) @ w_down.t()) @ w_lm.t()
    ok = torch.allclose(logits, full, rtol=1e-10, atol=1e-10)
    ok = ok and sz == dict(q_heads=8 // world, kv_heads=1, intermediate=I // world, vocab=V // world)
    t = par.max_over_ranks(1.0 + rank, world, "cpu")
    ok = ok and t == float(world)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_tp2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_tp8_gloo():
    """The same exchange pattern with eight ranks (the driver's N = 8 scaling run; kv heads replicated: 2 kv heads on 8
    ranks).  A one-GPU box admits at most 6 processes on its card, so world 8 can only be rehearsed here, on the CPU."""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)
