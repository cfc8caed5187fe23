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
        # fused all-reduce + add + RMSNorm (+ static fp8 quant): bit-identical to the unfused sequence; the producer
        # writes straight into the registered staging buffer in half of the cases
        from iaas_sglang_amd import ops
        for dtype in (torch.bfloat16, torch.float16):
            for rows, H in ((1, 4096), (7, 1024), (128, 4096), (128, 8192), (33, 4096), (256, 8192), (5, 16384)):
                allx = (torch.randn(world, rows, H, generator=g) * 0.5).to(dtype)
                res0 = torch.randn(rows, H, generator=g).to(dtype)
                w = torch.randn(H, generator=g).to(dtype).cuda()
                qs = torch.tensor([0.02], device="cuda")
                for staged in (False, True):
                    x = allx[rank].cuda()
                    if staged:
                        buf = ca.staging((rows, H), dtype)
                        buf.copy_(x)
                        x = buf
                    dist.barrier()
                    res_a = res0.cuda()
                    h = ca.all_reduce(x)
                    want_o = ops.rmsnorm(h, w, 1e-5, residual=res_a)
                    res_b = res0.cuda()
                    want_q = ops.rmsnorm_fp8(h, w, 1e-5, qs, residual=res_b)
                    res_c = res0.cuda()
                    # leave OTHER values in staging and in both tmp layouts first: a fused call that picked up a
                    # peer's tmp before it was rewritten would otherwise read the very sums it is about to compute
                    junk = (allx[(rank + 1) % world] * 3 + 1).to(dtype).cuda()
                    ca.all_reduce(junk)
                    ca.all_reduce_add_rmsnorm(junk, None, w, 1e-5, want_out=True)
                    if staged:
                        buf.copy_(allx[rank].cuda())
                    got_o, got_q = ca.all_reduce_add_rmsnorm(x, res_c, w, 1e-5, q_scale=qs, want_out=True)
                    torch.cuda.synchronize()
                    good = (not ca.timed_out() and torch.equal(got_o, want_o) and torch.equal(res_c, res_a)
                            and torch.equal(got_q.view(torch.uint8), want_q.view(torch.uint8))
                            and torch.equal(h.cpu().float(), allx.float().sum(0).to(dtype).float()))
                    if not good:
                        print(f"[rank {rank}] FAIL fused dtype={dtype} rows={rows} H={H} staged={staged} "
                              f"timed_out={ca.timed_out()} o={torch.equal(got_o, want_o)} res={torch.equal(res_c, res_a)}",
                              flush=True)
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


def _stack_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.custom_all_reduce import CustomAllreduce
    from iaas_sglang_amd.quantization import Fp8Config
    ca = CustomAllreduce(dist.group.WORLD, dev, max_size=1 << 20)
    ok = not ca.disabled
    try:
        shape, dtype, B, S = H.TINY, torch.bfloat16, 8, 40
        cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
        logits = {}
        for name, car in (("rccl_or_gloo", None), ("native_fused", ca)):
            runner = H.make_runner(shape, max_reqs=B, ctx=128, pool_tokens=B * S + 8, dtype=dtype, device=dev, tp=world,
                                   fill_kv=True, seed=rank)
            backend = MiAttnBackend(runner)
            stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, dev, tp=world, rank=rank,
                                 group=dist.group.WORLD, custom_ar=car, weight_range=0.05)
            fb = H.make_decode_batch(runner, backend, B, S, dev, seed=0)
            hidden = torch.randn(B, shape.hidden, generator=torch.Generator().manual_seed(3)).to(dtype).to(dev)
            backend.init_forward_metadata(fb)
            stack.calibrate_static_input_scales(hidden, fb.positions, fb, backend)
            backend.init_forward_metadata(fb)
            assert stack._fused_decode_ok(hidden, fb)
            logits[name] = stack.forward(hidden, fb.positions, fb, backend).float().cpu()
        torch.cuda.synchronize()
        # world 2: a two-operand sum is the same in any order and precision >= bf16, so the collective-based and the
        # native fused step agree bit for bit
        same = torch.equal(logits["rccl_or_gloo"], logits["native_fused"])
        if not same:
            print(f"[rank {rank}] TP stack: max diff {(logits['rccl_or_gloo'] - logits['native_fused']).abs().max()}", flush=True)
        ok = ok and same and not ca.timed_out() and bool(torch.isfinite(logits["native_fused"]).all())
    finally:
        ca.close()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_tp2_decode_step_with_fused_allreduce_norm_matches_collective():
    """One TP=2 decode step of the fused layer sequence: row-parallel GEMM -> registered staging buffer -> fused
    all-reduce + add + RMSNorm + fp8 quant, against the same step through torch.distributed's all-reduce."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_stack_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert all(out.get(r) for r in range(2)), dict(out)


def _graph_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iaas_sglang_amd import ops
    from iaas_sglang_amd.custom_all_reduce import CustomAllreduce
    ca = CustomAllreduce(dist.group.WORLD, torch.device("cuda", 0), max_size=4 * 1024 * 1024)
    ok = not ca.disabled
    try:
        g = torch.Generator().manual_seed(77)
        for dtype, rows, H in ((torch.bfloat16, 128, 4096), (torch.float16, 16, 1024), (torch.bfloat16, 256, 8192)):
            x = torch.zeros(rows, H, dtype=dtype, device="cuda")           # persistent graph inputs
            res = torch.zeros(rows, H, dtype=dtype, device="cuda")
            st = ca.staging((rows, H), dtype)
            w = torch.randn(H, generator=g).to(dtype).cuda()
            qs = torch.tensor([0.02], device="cuda")
            # warm-up inside capture() but outside a stream capture only mimics the allocation (custom_all_reduce.py:476)
            with ca.capture():
                warm = ca.custom_all_reduce(x)
                ok = ok and warm is not None and warm.shape == x.shape
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                dist.barrier()
                with torch.cuda.graph(graph):
                    y = ca.custom_all_reduce(x)                            # staged by a captured copy node
                    st.copy_(x)                                            # "producer" writing into staging
                    o, q = ca.all_reduce_add_rmsnorm(st, res, w, 1e-5, q_scale=qs, want_out=True)
            ok = ok and ca.register_graph_buffers() == 0 and not ca._IS_CAPTURING
            for it in range(3):
                allx = (torch.randn(world, rows, H, generator=g) * 0.5).to(dtype)
                res0 = torch.randn(rows, H, generator=g).to(dtype)
                x.copy_(allx[rank])
                res.copy_(res0)
                dist.barrier()
                graph.replay()
                torch.cuda.synchronize()
                want_h = allx.float().sum(0).to(dtype)                     # world 2: one rounding, any order
                res_ref = res0.cuda()
                want_o = ops.rmsnorm(want_h.cuda(), w, 1e-5, residual=res_ref)
                res_ref2 = res0.cuda()
                want_q = ops.rmsnorm_fp8(want_h.cuda(), w, 1e-5, qs, residual=res_ref2)
                torch.cuda.synchronize()
                good = (not ca.timed_out() and torch.equal(y.cpu(), want_h) and torch.equal(o, want_o)
                        and torch.equal(res, res_ref) and torch.equal(q.view(torch.uint8), want_q.view(torch.uint8)))
                if not good:
                    print(f"[rank {rank}] FAIL graph replay {it} dtype={dtype} rows={rows} H={H} y={torch.equal(y.cpu(), want_h)} "
                          f"o={torch.equal(o, want_o)} res={torch.equal(res, res_ref)}", flush=True)
                ok = ok and good
            del graph
    finally:
        ca.close()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_allreduce_captured_in_graph_and_replayed():
    """`ca_comm.capture()` around a hipGraph capture (parallel_state.py:377-378) holding the plain and the fused
    all-reduce; three replays with new inputs against the unfused eager ops."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_graph_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert all(out.get(r) for r in range(2)), dict(out)


def _tbo_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.custom_all_reduce import CustomAllreduce
    from iaas_sglang_amd.quantization import Fp8Config
    cas = [CustomAllreduce(dist.new_group(backend="gloo"), dev, max_size=1 << 20) for _ in range(2)]
    ok = not any(c.disabled for c in cas)
    try:
        shape, dtype, B, S = H.TINY, torch.bfloat16, 8, 40
        cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
        runner = H.make_runner(shape, max_reqs=B, ctx=128, pool_tokens=B * S + 8, dtype=dtype, device=dev, tp=world,
                               fill_kv=True, seed=rank)
        backend, backend_b = MiAttnBackend(runner), MiAttnBackend(runner)
        stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, dev, tp=world, rank=rank,
                             group=dist.group.WORLD, custom_ar=cas[0], weight_range=0.05)
        fb = H.make_decode_batch(runner, backend, B, S, dev, seed=0)
        hidden = torch.randn(B, shape.hidden, generator=torch.Generator().manual_seed(3)).to(dtype).to(dev)
        backend.init_forward_metadata(fb)
        stack.calibrate_static_input_scales(hidden, fb.positions, fb, backend)
        pool = runner.token_to_kv_pool
        kv0 = [b.clone() for b in pool.k_buffer + pool.v_buffer]
        backend.init_forward_metadata(fb)
        assert stack._fused_decode_ok(hidden, fb)
        hidden0 = hidden.clone()             # the step updates its input in place (it is the residual stream)
        want = stack.forward_decode_fused(hidden, fb.positions, fb, backend).float().cpu()
        for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
            b.copy_(b0)
        hidden.copy_(hidden0)
        halves = H.split_decode_batch(fb, backend_b, B // 2)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        dist.barrier()
        got = stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams, cas)
        torch.cuda.synchronize()
        same = torch.equal(got.float().cpu(), want)
        if not same:
            print(f"[rank {rank}] two-batch TP step: max diff {(got.float().cpu() - want).abs().max()}", flush=True)
        ok = ok and same and not any(c.timed_out() for c in cas) and bool(torch.isfinite(want).all())
        # the configuration bench.py times for world > 1 (--tbo auto): the two-branch step captured into ONE hipGraph
        # inside both communicators' capture() contexts, replayed with new inputs; both ranks replay in lock-step (the
        # all-reduce kernels of the two branches spin on their peers), every replay bit-identical to the eager serial step
        import contextlib
        for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
            b.copy_(b0)
        hidden.copy_(hidden0)
        stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams, cas, return_hidden=True)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with contextlib.ExitStack() as es:
            for c in cas:
                es.enter_context(c.capture())
            with torch.cuda.graph(g):
                got_h = stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams, cas,
                                                       return_hidden=True)
        for trial in range(3):
            h_in = torch.randn(B, shape.hidden, generator=torch.Generator().manual_seed(10 + trial)).to(dtype).to(dev)
            for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
                b.copy_(b0)
            hidden.copy_(h_in)
            backend.init_forward_metadata(fb)
            want_h = stack.forward_decode_fused(hidden, fb.positions, fb, backend, return_hidden=True).clone()
            for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
                b.copy_(b0)
            hidden.copy_(h_in)
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            same = torch.equal(got_h, want_h)
            if not same:
                print(f"[rank {rank}] captured two-batch TP step, replay {trial}: max diff "
                      f"{(got_h.float() - want_h.float()).abs().max()}", flush=True)
            ok = ok and same and not any(c.timed_out() for c in cas)
    finally:
        for c in cas:
            c.close()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_tp2_two_batch_overlap_matches_serial_step():
    """C5's overlap on a TP=2 decode step (two processes on one GPU): each micro-batch on its own stream with its own
    native all-reduce communicator, against the serial fused step -- bit-identical logits; then the same two-branch
    step captured into one hipGraph under both communicators' capture() and replayed three times with new inputs."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tbo_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert all(out.get(r) for r in range(2)), dict(out)
