"""Tensor-parallel call sites of the hot path: head/column sharding needs no
exchange; row-parallel linears are followed by ONE all-reduce of [tokens, hidden]
(python/sglang/srt/layers/linear.py:1376-1378 -> distributed/communication_op.py:11-13) and the
vocab-sharded lm_head by an all-gather (layers/logits_processor.py:464-477).  The collective is
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def shard_sizes(num_heads: int, num_kv_heads: int, intermediate: int, vocab: int, tp: int):
    """Per-rank sizes (llama.py:116-130: kv heads are replicated when tp > num_kv_heads)."""
    assert num_heads % tp == 0 and intermediate % tp == 0 and vocab % tp == 0
    return dict(q_heads=num_heads // tp, kv_heads=max(1, num_kv_heads // tp), intermediate=intermediate // tp,
                vocab=vocab // tp)


def tensor_model_parallel_all_reduce(x: torch.Tensor, tp: int, group: Optional[dist.ProcessGroup],
                                     custom_ar=None) -> torch.Tensor:
    """GroupCoordinator.all_reduce's dispatch (distributed/parallel_state.py:426-500): the native xGMI
    kernel out of place when it applies, otherwise RCCL in place."""
    if tp > 1:
        if custom_ar is not None:
            out = custom_ar.custom_all_reduce(x)
            if out is not None:
                return out
        dist.all_reduce(x, group=group)
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, tp: int, group: Optional[dist.ProcessGroup]) -> torch.Tensor:
    """Gather the last dim (vocab shards) from every rank."""
    if tp == 1:
        return x
    x = x.contiguous()
    if dist.get_backend(group) == "nccl":                       # RCCL: one collective, graph-capturable
        gathered = torch.empty((tp,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(gathered, x, group=group)
        return gathered.movedim(0, -2).reshape(*x.shape[:-1], tp * x.shape[-1])
    parts = [torch.empty_like(x) for _ in range(tp)]            # gloo (CPU tests, single-GPU rehearsals)
    dist.all_gather(parts, x, group=group)
    return torch.cat(parts, dim=-1)


def max_over_ranks(seconds: float, world: int, device) -> float:
    """bench.py's timing rule: the slowest rank defines the step time."""
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
