"""End-to-end parity through the plugin surfaces: a tiny Llama-shaped stack (extend, then decode
steps) on the HIP path vs the same stack spelled with the CPU oracle.  North-star bar: max-abs
logit error < 1e-3 with the reference's dummy weights (uniform +-1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attention as oa  # noqa: E402
from oracle import elementwise as oe  # noqa: E402
from oracle import quant as oq  # noqa: E402

DEV = "cuda"


def _fp8_lin(x, W, key):
    return oq.fp8_linear(x, W[key + "_w"], W[key + "_s"], W.get(key + "_i"))


def _oracle_forward(stack_w, shape, hidden, positions, kc, vc, r2t, rpi, seq_lens, pre, ext, loc, decode, lin=_fp8_lin):
    """Same layer order as harness.LlamaStack.forward, every op from oracle/ (CPU); `lin(x, W, key)` is the
    oracle form of the stack's linear method."""
    D, Hq, Hkv = shape.head_dim, shape.num_heads, shape.num_kv_heads
    cache = oe.rope_cos_sin_cache(D, shape.context_len, shape.rope_theta)
    residual = None
    for li, W in enumerate(stack_w["layers"]):
        if residual is None:
            residual = hidden
            x = oe.rmsnorm(hidden, W["input_norm"], shape.rms_eps)
        else:
            x, residual = oe.rmsnorm(hidden, W["input_norm"], shape.rms_eps, residual)
        qkv = lin(x, W, "qkv")
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        q, k = oe.rope_neox(positions, q.contiguous(), k.contiguous(), cache, D)
        k3, v3 = k.reshape(-1, Hkv, D), v.reshape(-1, Hkv, D)
        if decode:
            a = oa.forward_decode(q, k3, v3, kc[li], vc[li], r2t, rpi, seq_lens, loc, Hq, Hkv, D ** -0.5)
        else:
            a = oa.forward_extend(q, k3, v3, kc[li], vc[li], r2t, rpi, seq_lens, pre, ext, loc, Hq, Hkv, D ** -0.5)
        hidden = lin(a, W, "o")
        x, residual = oe.rmsnorm(hidden, W["post_norm"], shape.rms_eps, residual)
        gu = lin(x, W, "gu")
        hidden = lin(oe.silu_and_mul(gu), W, "down")
    x, _ = oe.rmsnorm(hidden, stack_w["final_norm"], shape.rms_eps, residual)
    return (x.float() @ stack_w["lm_head"].float().t())


@pytest.mark.parametrize("scheme", ["dynamic", "static"])
@pytest.mark.parametrize("weight_range,tol", [(1e-3, 1e-3), (0.05, None)])
def test_tiny_llama_fp8_extend_then_decode(weight_range, tol, scheme):
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config

    shape, dtype = H.TINY, torch.bfloat16
    cfg = Fp8Config(is_checkpoint_fp8_serialized=(scheme == "static"), activation_scheme=scheme)
    runner = H.make_runner(shape, max_reqs=8, ctx=128, pool_tokens=600, dtype=dtype, device=DEV)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weight_range=weight_range,
                         weights_cpu_seeded=True)
    g = torch.Generator().manual_seed(0)
    prefix, extend = [0, 0, 0], [37, 5, 64]
    fb = H.make_extend_batch(runner, backend, prefix, extend, DEV, seed=1)
    E = sum(extend)
    hidden = torch.randn(E, shape.hidden, generator=g).to(dtype)
    backend.init_forward_metadata(fb)
    if scheme == "static":   # calibrated per-tensor input scales, as a serialized FP8 checkpoint carries
        stack.calibrate_static_input_scales(hidden.to(DEV), fb.positions, fb, backend)
        for b in runner.token_to_kv_pool.k_buffer + runner.token_to_kv_pool.v_buffer:
            b.zero_()
    # the quantised weights as the oracle sees them ([K,N] fp8 view + scales), copied off the device
    W = {"layers": [], "final_norm": stack.final_norm.cpu(), "lm_head": stack.lm_head.cpu()}
    for L in stack.layers:
        d = {"input_norm": L.input_norm.cpu(), "post_norm": L.post_norm.cpu()}
        for key, lin in (("qkv", L.qkv), ("o", L.o), ("gu", L.gate_up), ("down", L.down)):
            d[key + "_w"], d[key + "_s"] = lin.weight.cpu(), lin.weight_scale.cpu()
            if scheme == "static":
                d[key + "_i"] = lin.input_scale.cpu()
        W["layers"].append(d)
    logits = stack.forward(hidden.to(DEV), fb.positions, fb, backend)
    kc = [torch.zeros_like(b).cpu() for b in runner.token_to_kv_pool.k_buffer]
    vc = [torch.zeros_like(b).cpu() for b in runner.token_to_kv_pool.v_buffer]
    r2t = runner.req_to_token_pool.req_to_token.cpu()
    ref = _oracle_forward(W, shape, hidden, fb.positions.cpu(), kc, vc, r2t, fb.req_pool_indices.cpu(),
                          fb.seq_lens.cpu(), fb.extend_prefix_lens.cpu(), fb.extend_seq_lens.cpu(),
                          fb.out_cache_loc.cpu(), decode=False)
    # +-1e-3 dummy weights: the north-star bar, max-abs < 1e-3.  Weights of +-0.05 give logits of order 0.5: HIP and
    # oracle share the quantised weights and differ by accumulation order and one rounding per op, plus the odd fp8
    # activation code that flips on such a difference -- bounded relative to the logits: 2 % of max |logit|
    def bound(r):
        return tol if tol is not None else 0.02 * float(r.abs().max())
    err = float((logits.float().cpu() - ref).abs().max())
    print(f"e2e extend: max|logit|={float(ref.abs().max()):.4f} err={err:.3e} bound={bound(ref):.3e}")
    assert err < bound(ref)
    tol_kv = tol if tol is not None else 5e-2
    # the KV pool after prefill is what the oracle wrote (bf16 rounding of slightly different fp8 sums aside)
    for li in range(shape.layers):
        torch.testing.assert_close(runner.token_to_kv_pool.k_buffer[li].cpu().float(), kc[li].float(),
                                   atol=tol_kv, rtol=2e-2)
    # two decode steps on top (seq_lens grow by one, new slot per request)
    lens = [p + e for p, e in zip(prefix, extend)]
    next_slot = sum(lens) + 1
    for step in range(2):
        lens = [L + 1 for L in lens]
        B = len(lens)
        loc = torch.arange(next_slot, next_slot + B, dtype=torch.int64)
        next_slot += B
        for i in range(B):
            runner.req_to_token_pool.req_to_token[i, lens[i] - 1] = int(loc[i])
        fb.forward_mode = H.ForwardMode.DECODE
        fb.seq_lens = torch.tensor(lens, dtype=torch.int64, device=DEV)
        fb.seq_lens_sum = sum(lens)
        fb.out_cache_loc = loc.to(DEV)
        fb.positions = (fb.seq_lens - 1)
        hidden = torch.randn(B, shape.hidden, generator=g).to(dtype)
        backend.init_forward_metadata(fb)
        logits = stack.forward(hidden.to(DEV), fb.positions, fb, backend)
        ref = _oracle_forward(W, shape, hidden, fb.positions.cpu(), kc, vc,
                              runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(),
                              fb.seq_lens.cpu(), None, None, loc, decode=True)
        err = float((logits.float().cpu() - ref).abs().max())
        print(f"e2e decode {step}: max|logit|={float(ref.abs().max()):.4f} err={err:.3e} bound={bound(ref):.3e}")
        assert err < bound(ref)


@pytest.mark.parametrize("ctx,trials", [
    (512, [[300, 17, 450, 1], [5, 5, 5, 5], [511, 300, 2, 64]]),
    # long enough for the ragged plan (fixed chunks + launch list) and for several uniform split counts: the captured
    # launch must follow whatever plan replay writes
    (4096, [[3000, 40, 700, 513], [2048, 2048, 2048, 2048], [1, 1, 1, 1], [4000, 3999, 16, 2100], [600, 600, 600, 600]]),
])
def test_backend_graph_capture_replay_matches_eager(ctx, trials):
    """Decode under hipGraph: capture with one batch, replay with other seq_lens / slots
    (cuda_graph_runner.py:456-696 protocol: persistent inputs, metadata replay, padded rows)."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend

    shape, dtype = H.TINY, torch.bfloat16
    runner = H.make_runner(shape, max_reqs=8, ctx=ctx, pool_tokens=max(3000, 4 * ctx), dtype=dtype, device=DEV,
                           fill_kv=True)
    backend = MiAttnBackend(runner)
    bs = 4
    backend.init_cuda_graph_state(bs, bs)
    layer = H.AttnLayer(shape.num_heads, shape.head_dim, shape.head_dim ** -0.5, shape.num_kv_heads, 0)
    g = torch.Generator().manual_seed(0)
    # persistent graph inputs
    rpi = torch.zeros(bs, dtype=torch.int64, device=DEV)
    seq_lens = torch.full((bs,), backend.get_cuda_graph_seq_len_fill_value(), dtype=torch.int64, device=DEV)
    out_loc = torch.zeros(bs, dtype=torch.int64, device=DEV)
    q = torch.zeros(bs, shape.num_heads * shape.head_dim, dtype=dtype, device=DEV)
    k = torch.zeros(bs, shape.num_kv_heads, shape.head_dim, dtype=dtype, device=DEV)
    v = torch.zeros_like(k)
    fb = H.make_decode_batch(runner, backend, bs, 300, DEV, seed=2)
    fb.req_pool_indices, fb.seq_lens, fb.out_cache_loc = rpi, seq_lens, out_loc
    backend.init_forward_metadata_capture_cuda_graph(bs, bs, rpi, seq_lens, None, H.ForwardMode.DECODE, None)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = backend.forward(q, k, v, layer, fb)
    for trial, lens in enumerate(trials):
        fb2 = H.make_decode_batch(runner, MiAttnBackend(runner), bs, 0, DEV, seed=10 + trial,
                                  ragged=torch.tensor(lens))
        rpi.copy_(fb2.req_pool_indices); seq_lens.copy_(fb2.seq_lens); out_loc.copy_(fb2.out_cache_loc)
        q.copy_(torch.randn(q.shape, generator=g).to(dtype)); k.copy_(torch.randn(k.shape, generator=g).to(dtype))
        v.copy_(torch.randn(v.shape, generator=g).to(dtype))
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, sum(lens), None, H.ForwardMode.DECODE,
                                                        None, fb2.seq_lens_cpu)
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        eager_backend = fb2.attn_backend
        eager_backend.init_forward_metadata(fb2)
        want = eager_backend.forward(q, k, v, layer, fb2)     # KV write is idempotent (same k,v,slots)
        torch.testing.assert_close(got.float(), want.float(), atol=4e-3, rtol=2 ** -7)
    # back-to-back replays WITHOUT host syncs in between: every replay must see its own plan (the pinned staging
    # buffers rotate; a plan overwritten before its copy ran would show up as a wrong result)
    gots = []
    qs = torch.randn(q.shape, generator=g).to(dtype).to(DEV)
    q.copy_(qs)
    for it in range(12):
        lens = torch.randint(1, ctx - 1, (bs,), generator=torch.Generator().manual_seed(500 + it)).tolist()
        fbk = H.make_decode_batch(runner, MiAttnBackend(runner), bs, 0, DEV, seed=100 + it, ragged=torch.tensor(lens))
        rpi.copy_(fbk.req_pool_indices); seq_lens.copy_(fbk.seq_lens); out_loc.copy_(fbk.out_cache_loc)
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq_lens, sum(lens), None, H.ForwardMode.DECODE, None,
                                                        fbk.seq_lens_cpu)
        graph.replay()
        gots.append(out.clone())                              # stream-ordered, no sync
        eb = fbk.attn_backend
        eb.init_forward_metadata(fbk)
        gots.append(eb.forward(q, k, v, layer, fbk))          # eager result for the same inputs, also enqueued now
    torch.cuda.synchronize()
    for it in range(12):
        torch.testing.assert_close(gots[2 * it].float(), gots[2 * it + 1].float(), atol=4e-3, rtol=2 ** -7)


def _extend_then_decode(stack, runner, backend, shape, dtype, W, lin, prefix, extend, tol, decode_steps=2):
    """Ragged EXTEND (with cached prefixes) and then decode steps through the plugin surfaces vs the oracle stack."""
    from iaas_sglang_amd import harness as H
    g = torch.Generator().manual_seed(0)
    pool = runner.token_to_kv_pool
    fb = H.make_extend_batch(runner, backend, prefix, extend, DEV, seed=1)
    hidden = torch.randn(sum(extend), shape.hidden, generator=g).to(dtype)
    kc = [b.cpu().clone() for b in pool.k_buffer]          # prefixes are already cached (random rows)
    vc = [b.cpu().clone() for b in pool.v_buffer]
    backend.init_forward_metadata(fb)
    logits = stack.forward(hidden.to(DEV), fb.positions, fb, backend)
    r2t = runner.req_to_token_pool.req_to_token.cpu()
    ref = _oracle_forward(W, shape, hidden, fb.positions.cpu(), kc, vc, r2t, fb.req_pool_indices.cpu(),
                          fb.seq_lens.cpu(), fb.extend_prefix_lens.cpu(), fb.extend_seq_lens.cpu(),
                          fb.out_cache_loc.cpu(), decode=False, lin=lin)
    assert float((logits.float().cpu() - ref).abs().max()) < tol
    lens = [p + e for p, e in zip(prefix, extend)]
    next_slot = int(runner.req_to_token_pool.req_to_token.max()) + 1
    for step in range(decode_steps):
        lens = [L + 1 for L in lens]
        B = len(lens)
        loc = torch.arange(next_slot, next_slot + B, dtype=torch.int64)
        next_slot += B
        for i in range(B):
            runner.req_to_token_pool.req_to_token[i, lens[i] - 1] = int(loc[i])
        fb.forward_mode = H.ForwardMode.DECODE
        fb.seq_lens = torch.tensor(lens, dtype=torch.int64, device=DEV)
        fb.seq_lens_cpu = torch.tensor(lens, dtype=torch.int64)
        fb.seq_lens_sum = sum(lens)
        fb.out_cache_loc = loc.to(DEV)
        fb.positions = (fb.seq_lens - 1)
        hidden = torch.randn(B, shape.hidden, generator=g).to(dtype)
        backend.init_forward_metadata(fb)
        logits = stack.forward(hidden.to(DEV), fb.positions, fb, backend)
        ref = _oracle_forward(W, shape, hidden, fb.positions.cpu(), kc, vc,
                              runner.req_to_token_pool.req_to_token.cpu(), fb.req_pool_indices.cpu(),
                              fb.seq_lens.cpu(), None, None, loc, decode=True, lin=lin)
        assert float((logits.float().cpu() - ref).abs().max()) < tol
    for li in range(shape.layers):      # the pool holds exactly what the oracle wrote (same rows, rounding aside)
        torch.testing.assert_close(pool.k_buffer[li].cpu().float(), kc[li].float(), atol=tol, rtol=2e-2)


def test_tiny_llama_bf16_ragged_extend_with_prefix_then_decode():
    """Config C2 in miniature: bf16 linears (library GEMM, not ours), OUR paged-KV extend + decode attention on a
    ragged batch with cached prefixes; glue kernels ours."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd._compat import UnquantizedLinearMethod
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    shape, dtype = H.TINY, torch.bfloat16
    runner = H.make_runner(shape, max_reqs=8, ctx=512, pool_tokens=1200, dtype=dtype, device=DEV, fill_kv=True)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, UnquantizedLinearMethod, dtype, DEV, weight_range=0.05, weights_cpu_seeded=True)
    W = {"layers": [], "final_norm": stack.final_norm.cpu(), "lm_head": stack.lm_head.cpu()}
    for L in stack.layers:
        d = {"input_norm": L.input_norm.cpu(), "post_norm": L.post_norm.cpu()}
        for key, lin in (("qkv", L.qkv), ("o", L.o), ("gu", L.gate_up), ("down", L.down)):
            d[key + "_w"] = lin.weight.detach().cpu()
        W["layers"].append(d)

    def lin(x, Wl, key):       # F.linear in fp32, one rounding to bf16 (what a bf16 GEMM with fp32 accumulation gives)
        return (x.float() @ Wl[key + "_w"].float().t()).to(dtype)

    _extend_then_decode(stack, runner, backend, shape, dtype, W, lin, prefix=[64, 0, 200, 17, 0],
                        extend=[1, 37, 128, 5, 300], tol=5e-2)


def test_tiny_llama_awq_extend_then_decode(monkeypatch):
    """Config C4 in miniature: AWQ int4 g128 linears (fused dequant GEMM), fp16 activations, through
    AWQConfig.get_quant_method -> AWQLinearMethod, vs the oracle's awq_dequantize + matmul."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import AWQConfig
    shape, dtype = H.TINY, torch.float16
    cfg = AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    ckpt = []
    orig = H.LlamaStack._init_linear

    def recording_init(lin, dummy):          # keep the checkpoint-layout tensors: the method repacks them afterwards
        orig(lin, dummy)
        ckpt.append({k: p.detach().cpu().clone() for k, p in lin.named_parameters()})

    monkeypatch.setattr(H.LlamaStack, "_init_linear", staticmethod(recording_init))
    runner = H.make_runner(shape, max_reqs=8, ctx=512, pool_tokens=1200, dtype=dtype, device=DEV, fill_kv=True)
    backend = MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weights_cpu_seeded=True)
    assert type(stack.layers[0].qkv.quant_method).__name__ == "AWQLinearMethod" and len(ckpt) == 4 * shape.layers
    W = {"layers": [], "final_norm": stack.final_norm.cpu(), "lm_head": stack.lm_head.cpu()}
    for li, L in enumerate(stack.layers):
        d = {"input_norm": L.input_norm.cpu(), "post_norm": L.post_norm.cpu()}
        for j, key in enumerate(("qkv", "o", "gu", "down")):
            d[key] = ckpt[4 * li + j]
        W["layers"].append(d)

    def lin(x, Wl, key):
        c = Wl[key]
        return oq.awq_linear(x, c["qweight"], c["scales"], c["qzeros"])

    _extend_then_decode(stack, runner, backend, shape, dtype, W, lin, prefix=[0, 33, 0], extend=[37, 5, 64], tol=5e-2)


@pytest.mark.parametrize("n_first", [4, 5])
def test_two_batch_overlap_step_is_bit_identical_to_serial(n_first):
    """forward_decode_two_batch (each half of the batch on its own stream, its own attention metadata and GEMM scratch)
    against the serial fused step: every kernel computes rows independently, so the logits and the KV pool agree bit
    for bit -- eagerly and from a captured graph with two parallel branches."""
    from iaas_sglang_amd import harness as H
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import Fp8Config
    shape, dtype, B, S = H.TINY, torch.bfloat16, 8, 70
    cfg = Fp8Config(is_checkpoint_fp8_serialized=True, activation_scheme="static")
    runner = H.make_runner(shape, max_reqs=B, ctx=128, pool_tokens=B * S + 8, dtype=dtype, device=DEV, fill_kv=True)
    backend, backend_b = MiAttnBackend(runner), MiAttnBackend(runner)
    stack = H.LlamaStack(shape, lambda: cfg.get_quant_method(None, ""), dtype, DEV, weight_range=0.05)
    lens = torch.tensor([70, 3, 41, 64, 17, 70, 1, 33])
    fb = H.make_decode_batch(runner, backend, B, 0, DEV, seed=0, ragged=lens)
    hidden = torch.randn(B, shape.hidden, generator=torch.Generator().manual_seed(3)).to(dtype).to(DEV)
    backend.init_forward_metadata(fb)
    stack.calibrate_static_input_scales(hidden, fb.positions, fb, backend)
    pool = runner.token_to_kv_pool
    kv0 = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    backend.init_forward_metadata(fb)
    assert stack._fused_decode_ok(hidden, fb)
    hidden0 = hidden.clone()                 # the step updates its input in place (it is the residual stream)
    want = stack.forward_decode_fused(hidden, fb.positions, fb, backend)
    kv_want = [b.clone() for b in pool.k_buffer + pool.v_buffer]
    for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
        b.copy_(b0)
    halves = H.split_decode_batch(fb, backend_b, n_first)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    hidden.copy_(hidden0)
    got = stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert all(torch.equal(a, b) for a, b in zip(pool.k_buffer + pool.v_buffer, kv_want))
    # the same step as a hipGraph with two parallel branches
    out = torch.empty_like(want)
    graph = torch.cuda.CUDAGraph()
    hidden.copy_(hidden0)
    with torch.cuda.graph(graph):
        out.copy_(stack.forward_decode_two_batch(hidden, fb.positions, halves, [backend, backend_b], streams))
    for b, b0 in zip(pool.k_buffer + pool.v_buffer, kv0):
        b.copy_(b0)
    hidden.copy_(hidden0)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert all(torch.equal(a, b) for a, b in zip(pool.k_buffer + pool.v_buffer, kv_want))
