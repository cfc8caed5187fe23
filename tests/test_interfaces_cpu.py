"""CPU-only: the plugin surface against the reference's, as dumped from the reference's own classes by
tests/golden/make_golden_interfaces.py (interfaces.json: signatures; interfaces_param.pt: loader behaviour).

  * the stand-ins of iaas_sglang_amd/_compat.py (used when SGLang is not importable) carry the SAME members with
    the SAME parameter lists and the same abstract-method sets as the reference's classes;
  * every plugin class implements all abstract methods, and every method it overrides can be called the way the
    reference's callers call the base method (same leading parameter names; extras must be optional);
  * the parameter stand-ins reproduce the reference classes' weight-loader results on the recorded vectors.
"""
import inspect
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "interfaces.json")))

from iaas_sglang_amd import _compat as C  # noqa: E402

pytestmark = pytest.mark.skipif(C.HAVE_SGLANG, reason="the real SGLang classes are in use; nothing to mirror")


def _params(fn):
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
            for p in inspect.signature(fn).parameters.values()]


def _member(cls, name):
    for k in cls.__mro__:
        if name in k.__dict__:
            return k.__dict__[name]
    raise AttributeError(f"{cls.__name__}.{name}")


def _describe(obj):
    if isinstance(obj, property):
        return {"kind": "property"}
    if isinstance(obj, classmethod):
        return {"kind": "classmethod", "params": _params(obj.__func__)}
    if isinstance(obj, staticmethod):
        return {"kind": "staticmethod", "params": _params(obj.__func__)}
    return {"kind": "method", "params": _params(obj)}


MIRRORED = ["AttentionBackend", "QuantizeMethodBase", "QuantizationConfig", "LinearMethodBase", "BasevLLMParameter",
            "_ColumnvLLMParameter", "RowvLLMParameter", "ModelWeightParameter", "GroupQuantScaleParameter",
            "ChannelQuantScaleParameter", "PerTensorScaleParameter", "PackedvLLMParameter"]


@pytest.mark.parametrize("name", MIRRORED)
def test_mirror_has_the_reference_members(name):
    ref, cls = REF[name], getattr(C, name)
    assert [b.__name__ for b in cls.__bases__] == ref["bases"]
    assert sorted(getattr(cls, "__abstractmethods__", ())) == ref["abstract"]
    for member, want in ref["members"].items():
        if member.startswith("_") and member not in ("__init__", "__new__"):
            continue                                             # private helpers are not part of the surface
        assert member in cls.__dict__, f"{name}.{member} missing from the mirror"
        got = _describe(cls.__dict__[member])
        assert got["kind"] == want["kind"], f"{name}.{member}: {got['kind']} != {want['kind']}"
        if "params" in want:
            assert got["params"] == want["params"], f"{name}.{member}: {got['params']} != {want['params']}"


def _accepts_reference_call(impl_params, ref_params, what):
    """An override is drop-in if the reference's parameters come first, under the same names, with defaults wherever
    the reference has them, and anything extra is optional."""
    named_ref = [p for p in ref_params if p[1] not in ("VAR_POSITIONAL", "VAR_KEYWORD")]
    named_impl = [p for p in impl_params if p[1] not in ("VAR_POSITIONAL", "VAR_KEYWORD")]
    ref_var_pos = any(p[1] == "VAR_POSITIONAL" for p in ref_params)
    assert len(named_impl) >= len(named_ref), f"{what}: fewer parameters than the reference"
    for r, i in zip(named_ref, named_impl):
        assert r[0] == i[0], f"{what}: parameter {i[0]!r} where the reference has {r[0]!r}"
        if r[2] is not None:
            assert i[2] == r[2], f"{what}: default of {r[0]!r} is {i[2]} (reference {r[2]})"
    if not ref_var_pos:          # base takes *weight_args: a subclass naming its positionals is the reference's own pattern
        for extra in named_impl[len(named_ref):]:
            assert extra[2] is not None, f"{what}: extra parameter {extra[0]!r} has no default"


def _plugins():
    from iaas_sglang_amd.attention_backend import MiAttnBackend
    from iaas_sglang_amd.quantization import (AWQConfig, CompressedTensorsConfig, Fp8Config, GPTQConfig)
    from iaas_sglang_amd.quantization.awq import AWQLinearMethod
    from iaas_sglang_amd.quantization.compressed_tensors import CompressedTensorsLinearMethod
    from iaas_sglang_amd.quantization.fp8 import Fp8LinearMethod
    from iaas_sglang_amd.quantization.gptq import GPTQLinearMethod
    from iaas_sglang_amd.quantization.kv_cache import BaseKVCacheMethod
    return [(MiAttnBackend, ["AttentionBackend"]),
            (Fp8Config, ["QuantizationConfig"]), (AWQConfig, ["QuantizationConfig"]), (GPTQConfig, ["QuantizationConfig"]),
            (CompressedTensorsConfig, ["QuantizationConfig"]),
            (Fp8LinearMethod, ["LinearMethodBase", "QuantizeMethodBase"]),
            (AWQLinearMethod, ["LinearMethodBase", "QuantizeMethodBase"]),
            (GPTQLinearMethod, ["LinearMethodBase", "QuantizeMethodBase"]),
            (CompressedTensorsLinearMethod, ["LinearMethodBase", "QuantizeMethodBase"]),
            (BaseKVCacheMethod, ["QuantizeMethodBase"])]


def test_plugins_are_drop_in_for_the_reference_interfaces():
    for cls, bases in _plugins():
        assert not getattr(cls, "__abstractmethods__", None), f"{cls.__name__} leaves {cls.__abstractmethods__} abstract"
        assert issubclass(cls, getattr(C, bases[0]))
        seen = set()
        for base in bases:
            for member, want in REF[base]["members"].items():
                if member in seen or member.startswith("__") or "params" not in want:
                    continue
                seen.add(member)
                impl = _member(cls, member)
                got = _describe(impl)
                # a classmethod may stand in for a method or a staticmethod (the reference's own Fp8Config does both:
                # get_name, get_config_filenames, fp8.py:124-143): `obj.name(...)` / `cls.name(...)` work either way;
                # what is compared is the argument list a caller passes
                binds = {"method": 1, "classmethod": 1, "staticmethod": 0}
                assert got["kind"] == want["kind"] or got["kind"] == "classmethod", f"{cls.__name__}.{member}"
                _accepts_reference_call(got["params"][binds[got["kind"]]:], want["params"][binds[want["kind"]]:],
                                        f"{cls.__name__}.{member}")


def test_parameter_mirrors_reproduce_reference_loaders():
    V = torch.load(os.path.join(ROOT, "tests", "golden", "interfaces_param.pt"), weights_only=True)
    c = V["model_weight_column_rank1of2"]
    p = C.ModelWeightParameter(data=torch.zeros(12, 16), input_dim=1, output_dim=0, weight_loader=None)
    p.load_column_parallel_weight(c["loaded"], tp_rank=1)
    assert torch.equal(p.data, c["data"]) and p.input_dim == 1 and p.output_dim == 0 and p.weight_loader is None
    c = V["model_weight_row_rank1of2"]
    p = C.ModelWeightParameter(data=torch.zeros(24, 8), input_dim=1, output_dim=0, weight_loader=None)
    p.load_row_parallel_weight(c["loaded"], tp_rank=1)
    assert torch.equal(p.data, c["data"])
    c = V["model_weight_merged_shard1_rank1of2"]
    p = C.ModelWeightParameter(data=torch.zeros(12, 16), input_dim=1, output_dim=0, weight_loader=None)
    p.load_merged_column_weight(c["loaded"], shard_offset=6, shard_size=6, tp_rank=1, use_presharded_weights=False)
    assert torch.equal(p.data, c["data"])
    c = V["model_weight_qkv_k_rank3"]
    p = C.ModelWeightParameter(data=torch.zeros(16, 16), input_dim=1, output_dim=0, weight_loader=None)
    p.load_qkv_weight(c["loaded"], tp_rank=3, shard_offset=8, shard_size=4, shard_id="k", num_heads=2)
    assert torch.equal(p.data, c["data"])
    p = C.PerTensorScaleParameter(data=torch.full((3,), -1.0), weight_loader=None)
    p.load_qkv_weight(torch.tensor(0.5), shard_id="k")
    p.load_merged_column_weight(torch.tensor([0.25]), shard_id=2)
    assert torch.equal(p.data, V["per_tensor_scale_shards"]["data"])
    p = C.PerTensorScaleParameter(data=torch.zeros(1), weight_loader=None)
    p.load_row_parallel_weight(torch.tensor([0.75]), tp_rank=1, use_presharded_weights=False)
    assert torch.equal(p.data, V["per_tensor_scale_row"]["data"])
    c = V["packed_merged_shard_rank1of2"]
    p = C.PackedvLLMParameter(data=torch.zeros(16, 6, dtype=torch.int32), input_dim=0, output_dim=1, packed_dim=1,
                              packed_factor=8, weight_loader=None)
    p.load_merged_column_weight(c["loaded"], shard_offset=16, shard_size=16, tp_rank=1, use_presharded_weights=False)
    assert torch.equal(p.data, c["data"])
    assert list(p.adjust_shard_indexes_for_packing(shard_size=16, shard_offset=16)) == c["adjusted"].tolist()
    assert p.packed_dim == 1 and p.packed_factor == 8 and p.marlin_tile_size is None
    c = V["group_scale_column_rank1of2"]
    p = C.GroupQuantScaleParameter(data=torch.zeros(2, 8), input_dim=0, output_dim=1, weight_loader=None)
    p.load_column_parallel_weight(c["loaded"], tp_rank=1)
    assert torch.equal(p.data, c["data"])
    c = V["channel_scale_column_rank0of2"]
    p = C.ChannelQuantScaleParameter(data=torch.zeros(12, 1), output_dim=0, weight_loader=None)
    p.load_column_parallel_weight(c["loaded"], tp_rank=0)
    assert torch.equal(p.data, c["data"])
    with pytest.raises(AssertionError):       # shape mismatches stay loud (parameter.py:56-58)
        C.BasevLLMParameter(data=torch.zeros(3), weight_loader=None).load_column_parallel_weight(torch.zeros(4))
