"""register(): plug the MI355X hot path into an UNMODIFIED SGLang install at import time.

This snapshot has no attention-backend registry (the dispatch is an if/elif in
ModelRunner._get_attention_backend, python/sglang/srt/model_executor/model_runner.py:1259-1337)
and gates names through argparse `choices` (srt/server_args.py:1125-1141, 693-712), so
registration wraps those seams without editing any reference file:

  1. ModelRunner._get_attention_backend  -> returns MiAttnBackend for --attention-backend mi355x
  2. ServerArgs argparse choices          -> accepts the new backend name
  3. srt.utils.support_triton             -> False for our name (scheduler helpers use torch forms)
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
        import sglang.srt.utils as sutils
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

    # 3. scheduler-side triton helpers off for our backend
    orig_support = sutils.support_triton

    def support_triton(backend: str) -> bool:
        return False if backend == BACKEND_NAME else orig_support(backend)

    sutils.support_triton = support_triton
    for modname in ("sglang.srt.managers.schedule_batch", "sglang.srt.model_executor.forward_batch_info"):
        try:
            mod = __import__(modname, fromlist=["support_triton"])
            if hasattr(mod, "support_triton"):
                mod.support_triton = support_triton
        except Exception:
            pass

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
