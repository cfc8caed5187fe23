"""`CustomAllreduce`: native xGMI all-reduce for the TP row-parallel linears.

Mirrors the reference wrapper (python/sglang/srt/distributed/device_communicators/
custom_all_reduce.py:147-151 sizes, :360-385 handle exchange, :414-435 should_custom_ar,
:446-497 eager staging) on top of the C ABI `mi_ar_*` (csrc/allreduce.hip).  IPC handles are
exchanged through a CPU (gloo) process group with all_gather_object, exactly like the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import MI_BF16, MI_F32, MI_FP16, MiHotpathError, check, lib

_DT = {torch.bfloat16: MI_BF16, torch.float16: MI_FP16, torch.float32: MI_F32}


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]          # custom_all_reduce.py:147
    _IS_CAPTURING = False                          # custom_all_reduce.py:53, set by capture()

    def __init__(self, cpu_group: dist.ProcessGroup, device: torch.device, max_size: int = 8 * 1024 * 1024):
        self.disabled = True
        self.group = cpu_group
        self.rank = dist.get_rank(group=cpu_group)
        self.world_size = dist.get_world_size(group=cpu_group)
        self.device = torch.device(device)
        self.max_size = max_size
        self._ctx = None
        self._own = C.c_void_p()
        self._peers: List[Optional[int]] = []
        self._stage_bytes = None
        self.init_error = ""
        if self.world_size == 1 or self.world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        torch.cuda.set_device(self.device)
        # Every rank takes part in every collective below even if a local step fails, and the ranks
        # agree on the outcome at the end: one failing rank disables the path everywhere (-> RCCL)
        # instead of leaving its peers stuck in a collective.
        ok, why = True, ""
        handle = (C.c_char * 64)()
        try:
            nbytes = lib.mi_ar_shared_bytes(max_size)
            check(lib.mi_ar_alloc_shared(nbytes, C.byref(self._own)), "mi_ar_alloc_shared")
            check(lib.mi_ar_ipc_get(self._own, handle), "mi_ar_ipc_get")
        except Exception as e:  # noqa: BLE001
            ok, why = False, str(e)
        handles: List[Optional[bytes]] = [None] * self.world_size
        dist.all_gather_object(handles, bytes(handle.raw) if ok else None, group=cpu_group)
        ptrs = (C.c_void_p * self.world_size)()
        if ok and all(h is not None for h in handles):
            try:
                for r, h in enumerate(handles):
                    if r == self.rank:
                        ptrs[r] = self._own.value
                        self._peers.append(None)
                    else:
                        p = C.c_void_p()
                        check(lib.mi_ar_ipc_open(C.create_string_buffer(h, 64), C.byref(p)), "mi_ar_ipc_open")
                        ptrs[r] = p.value
                        self._peers.append(p.value)
                self._ctx = lib.mi_ar_create(ptrs, max_size, self.rank, self.world_size)
                if not self._ctx:
                    raise MiHotpathError(f"mi_ar_create failed: {lib.mi_last_error().decode()}")
            except Exception as e:  # noqa: BLE001
                ok, why = False, str(e)
        else:
            ok = False
        status: List[Optional[bool]] = [None] * self.world_size
        dist.all_gather_object(status, ok, group=cpu_group)   # also the "everyone has mapped everything" barrier
        if not all(status):
            self.init_error = why or "a peer rank failed to set up the shared buffers"
            self._release()
            return
        self.disabled = False

    def _release(self):
        self._stage_bytes = None
        if self._ctx:
            lib.mi_ar_destroy(self._ctx)
            self._ctx = None
        for p in self._peers:
            if p is not None:
                lib.mi_ar_ipc_close(p)
        self._peers = []
        if self._own.value:
            lib.mi_ar_free_shared(self._own)
            self._own = C.c_void_p()
        self.disabled = True

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """custom_all_reduce.py:414-435: 16-byte multiple, contiguous, within the registered size."""
        if self.disabled:
            return False
        nbytes = inp.numel() * inp.element_size()
        return (nbytes % 16 == 0 and inp.is_contiguous() and nbytes <= self.max_size and inp.dtype in _DT)

    def all_reduce(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(inp) if out is None else out
        check(lib.mi_ar_all_reduce(self._ctx, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                                   _DT[inp.dtype], torch.cuda.current_stream().cuda_stream), "mi_ar_all_reduce")
        return out

    def staging(self, shape, dtype: torch.dtype) -> torch.Tensor:
        """A tensor over the rank's own IPC-mapped staging buffer: a producer (the row-parallel GEMM) that writes its
        output here saves the staging copy of the next all_reduce / all_reduce_add_rmsnorm on it -- the job of the
        reference's register_buffer / register_graph_buffers (custom_all_reduce.py:387-412), without re-registration:
        the buffer is persistent, so captured graphs can use it as it is."""
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        if self.disabled or nbytes > self.max_size:
            raise MiHotpathError(f"CustomAllreduce.staging: {nbytes} bytes do not fit the registered {self.max_size}")
        if self._stage_bytes is None:
            class _Dev:          # the CUDA array interface is how torch wraps foreign device memory without a copy
                __cuda_array_interface__ = {"shape": (self.max_size,), "typestr": "|u1", "version": 2,
                                            "data": (int(lib.mi_ar_staging(self._ctx)), False)}
            self._stage_owner = _Dev()
            self._stage_bytes = torch.as_tensor(self._stage_owner, device=self.device)
            assert self._stage_bytes.data_ptr() == int(lib.mi_ar_staging(self._ctx))
        return self._stage_bytes[:nbytes].view(dtype).view(*shape)

    def all_reduce_add_rmsnorm(self, inp: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                               eps: float, q_scale: Optional[torch.Tensor] = None, want_out: bool = True):
        """sum over ranks of inp [rows, H], then `x = sum + residual; residual <- x; out = rmsnorm(x) * weight` in the
        same kernel (see mi_ar_all_reduce_add_rmsnorm).  Returns (out or None, fp8(out) or None): bit-identical to
        all_reduce followed by ops.rmsnorm / ops.rmsnorm_fp8."""
        assert inp.dim() == 2 and inp.is_contiguous() and inp.dtype in (torch.bfloat16, torch.float16)
        rows, H = inp.shape
        assert residual is None or (residual.shape == inp.shape and residual.stride(1) == 1 and residual.dtype == inp.dtype)
        assert weight.dtype == inp.dtype and weight.is_contiguous() and weight.numel() == H
        out = torch.empty_like(inp) if (want_out or q_scale is None) else None
        q = torch.empty((rows, H), dtype=torch.float8_e4m3fn, device=inp.device) if q_scale is not None else None
        check(lib.mi_ar_all_reduce_add_rmsnorm(
            self._ctx, inp.data_ptr(), residual.data_ptr() if residual is not None else None, weight.data_ptr(),
            out.data_ptr() if out is not None else None, q.data_ptr() if q is not None else None,
            q_scale.data_ptr() if q_scale is not None else None, rows, H,
            residual.stride(0) if residual is not None else H, H, float(eps), _DT[inp.dtype],
            torch.cuda.current_stream().cuda_stream), "mi_ar_all_reduce_add_rmsnorm")
        return out, q

    fuse_norm = os.environ.get("MI_AR_FUSE_NORM", "1") != "0"   # switch for A/B runs and for a failed self-check

    def should_fuse_norm(self, rows: int, H: int, dtype: torch.dtype) -> bool:
        return (self.fuse_norm and not self.disabled and dtype in (torch.bfloat16, torch.float16) and H % 8 == 0 and H <= 16384
                and rows * H * 2 <= self.max_size)

    @contextmanager
    def capture(self):
        """`GroupCoordinator.graph_capture` wraps every hipGraph capture in this (parallel_state.py:377-378;
        custom_all_reduce.py:336-348).  The reference needs it to collect the addresses of the captured inputs and
        register them with the peers afterwards; here every captured all-reduce goes through the one persistent
        IPC-mapped staging buffer (its copy node is part of the graph, or the producer wrote there directly), and the
        barrier flags live in device memory and only ever grow, so a captured launch replays as it is."""
        try:
            self._IS_CAPTURING = True
            yield
        finally:
            self._IS_CAPTURING = False
            if not self.disabled:
                self.register_graph_buffers()

    def register_graph_buffers(self) -> int:
        """custom_all_reduce.py:388-412 exchanges the IPC handles of the buffers recorded during capture.  Nothing was
        recorded (see capture()); returns the number of addresses registered, always 0."""
        return 0

    def custom_all_reduce(self, inp: torch.Tensor) -> Optional[torch.Tensor]:
        """None means: fall through to RCCL (parallel_state.py:495-500).  Inside capture() but outside an actual
        stream capture (the warm-up run) only the allocation pattern is mimicked, as custom_all_reduce.py:476-485."""
        if self.disabled or not self.should_custom_ar(inp):
            return None
        if self._IS_CAPTURING and not torch.cuda.is_current_stream_capturing():
            return torch.empty_like(inp)
        return self.all_reduce(inp)

    def timed_out(self) -> bool:
        return (not self.disabled) and lib.mi_ar_error(self._ctx) != 0

    def close(self):
        if self._ctx:
            torch.cuda.synchronize()
            dist.barrier(group=self.group)      # nobody is still reading my buffers
            self._release()

    def __del__(self):
        try:
            if self._ctx:
                lib.mi_ar_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass
