"""register(): plug the MI355X hot path into an UNMODIFIED SGLang install at import time.

This snapshot has no attention-backend registry (the dispatch is an if/elif in
ModelRunner._get_attention_backend, python/sglang/srt/model_executor/model_runner.py:1259-1337)
and gates names through argparse `choices` (srt/server_args.py:1125-1141, 693-712), so
registration wraps those seams without editing any reference file:

  1. ModelRunner._get_attention_backend  -> returns MiAttnBackend for --attention-backend mi355x
  2. ServerArgs argparse choices          -> accepts the new backend name
  3. the scheduler-side Triton helpers    -> our HIP kernels under the names the scheduler calls:
       schedule_batch.write_req_to_token_pool_triton[(bs,)](...), schedule_batch.get_last_loc_triton,
       forward_batch_info.compute_position_triton  (support_triton() stays True for our backend name, so the
       scheduler takes its one-launch branch instead of the per-request python loop, schedule_batch.py:1290-1309)
  4. QUANTIZATION_METHODS["fp8"|"awq"|"gptq"] -> our configs; awq/gptq are moved out of the
     vllm-gated table (quantization/__init__.py:75-120)

Usage (in the process that launches the server, before arguments are parsed):
    import iaas_sglang_amd.register as r; r.register()
    # then: python -m sglang.launch_server --attention-backend mi355x --quantization fp8 ...
"""
from __future__ import annotations

BACKEND_NAME = "mi355x"
_registered = False


def register(override_quantization: bool = True) -> bool:
    """Returns True when SGLang was found and patched, False when SGLang is not importable
    (the package is then usable stand-alone through its own mirrors of the interfaces)."""
    global _registered
    if _registered:
        return True
    try:
        import sglang.srt.model_executor.model_runner as mr
        import sglang.srt.server_args as sargs
    except Exception:
        return False

    from .attention_backend import MiAttnBackend

    # 1. attention dispatch
    orig_get = mr.ModelRunner._get_attention_backend

    def _get_attention_backend(self):
        if self.server_args.attention_backend == BACKEND_NAME:
            return MiAttnBackend(self)
        return orig_get(self)

    mr.ModelRunner._get_attention_backend = _get_attention_backend

    # 2. argparse choices: wrap add_cli_args so the name is accepted
    orig_add = sargs.ServerArgs.add_cli_args

    def add_cli_args(parser):
        orig_add(parser)
        for action in parser._actions:
            if "--attention-backend" in action.option_strings and action.choices is not None:
                if BACKEND_NAME not in action.choices:
                    action.choices = list(action.choices) + [BACKEND_NAME]

    sargs.ServerArgs.add_cli_args = staticmethod(add_cli_args)

    # 3. scheduler-side helpers: same names, same call forms, HIP kernels behind them
    install_scheduler_helpers()

    # 4. quantization registry
    if override_quantization:
        import sglang.srt.layers.quantization as q

        from .quantization import MI_QUANTIZATION_METHODS
        for name, cfg in MI_QUANTIZATION_METHODS.items():
            q.QUANTIZATION_METHODS[name] = cfg
            q.BASE_QUANTIZATION_METHODS[name] = cfg
            q.VLLM_QUANTIZATION_METHODS.pop(name, None)
    _registered = True
    return True


class _GridLaunch:
    """`kernel[grid](*args)` call form of a Triton launch for a plain host function (the grid is implied)."""

    def __init__(self, fn):
        self._fn = fn

    def __getitem__(self, grid):
        return self._fn


def write_req_to_token_pool(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc,
                            req_to_token_ptr_stride=None):
    """Argument list of write_req_to_token_pool_triton (schedule_batch.py:1848-1857)."""
    from . import ops
    ops.write_req_to_token(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc)


def get_last_loc(req_to_token, req_pool_indices_tensor, prefix_lens_tensor):
    """get_last_loc_triton (schedule_batch.py:1935-1956)."""
    from . import ops
    return ops.get_last_loc(req_to_token, req_pool_indices_tensor, prefix_lens_tensor)


def compute_position(extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum):
    """compute_position_triton (forward_batch_info.py:678-701)."""
    from . import ops
    return ops.compute_position(extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum)


def install_scheduler_helpers(schedule_batch=None, forward_batch_info=None) -> bool:
    """Bind the three helpers into the scheduler's modules (given, or imported from an installed SGLang)."""
    try:
        if schedule_batch is None:
            import sglang.srt.managers.schedule_batch as schedule_batch
        if forward_batch_info is None:
            import sglang.srt.model_executor.forward_batch_info as forward_batch_info
    except Exception:
        return False
    schedule_batch.write_req_to_token_pool_triton = _GridLaunch(write_req_to_token_pool)
    schedule_batch.get_last_loc_triton = get_last_loc
    forward_batch_info.compute_position_triton = compute_position
    return True
