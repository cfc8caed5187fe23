#!/usr/bin/env python3
"""Dump the reference's plugin surface to tests/golden/interfaces.json (+ interfaces_param.pt).

Build container only (needs /root/reference).  With the namespace-stub recipe of make_golden.py the
reference's own files are imported -- no reference `__init__.py` runs:
  * python/sglang/srt/layers/attention/base_attn_backend.py   AttentionBackend
  * python/sglang/srt/layers/quantization/base_config.py      QuantizeMethodBase, QuantizationConfig
  * python/sglang/srt/layers/linear.py:111-149                LinearMethodBase (class statement taken by name with `ast`
                                                              at run time: the module itself imports the distributed stack)
  * python/sglang/srt/layers/parameter.py:29-441              the parameter classes the linear methods register
interfaces.json: for every class, its bases, the abstract-method set and `inspect.signature` of every method /
property it defines (parameter name, kind, default).  interfaces_param.pt: the reference's parameter classes driven
through their loaders on small tensors (inputs + resulting parameter data), so the stand-ins in
iaas_sglang_amd/_compat.py can be checked for behaviour too.  tests/test_interfaces_cpu.py compares both with
iaas_sglang_amd/_compat.py and the plugin classes.

Usage:  python tests/golden/make_golden_interfaces.py
"""
import ast
import importlib
import inspect
import json
import os
import sys
import types
from abc import abstractmethod
from typing import Any, Dict, List, Optional

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF


def describe_callable(fn):
    sig = inspect.signature(fn)
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
            for p in sig.parameters.values()]


def describe_class(cls):
    members = {}
    for name, obj in cls.__dict__.items():
        if name.startswith("__") and name not in ("__init__", "__new__"):
            continue
        if isinstance(obj, property):
            members[name] = {"kind": "property"}
        elif isinstance(obj, classmethod):
            members[name] = {"kind": "classmethod", "params": describe_callable(obj.__func__)}
        elif isinstance(obj, staticmethod):
            members[name] = {"kind": "staticmethod", "params": describe_callable(obj.__func__)}
        elif inspect.isfunction(obj):
            members[name] = {"kind": "method", "params": describe_callable(obj)}
    return {"bases": [b.__name__ for b in cls.__bases__], "mro": [c.__name__ for c in cls.__mro__[:-1]],
            "abstract": sorted(getattr(cls, "__abstractmethods__", ())), "members": members}


def class_from_file(path, name, ns):
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def param_vectors(P):
    """Drive the reference's parameter classes through the loaders the TP linears call (linear.py weight_loader_v2)."""
    g = torch.Generator().manual_seed(5)
    out = {}
    full = torch.randn(24, 16, generator=g)
    # column-parallel weight: rank 1 of 2 takes rows 12..24
    p = P.ModelWeightParameter(data=torch.zeros(12, 16), input_dim=1, output_dim=0, weight_loader=None)
    p.load_column_parallel_weight(full, tp_rank=1)
    out["model_weight_column_rank1of2"] = dict(loaded=full, data=p.data.clone())
    # row-parallel weight: rank 1 of 2 takes columns 8..16
    p = P.ModelWeightParameter(data=torch.zeros(24, 8), input_dim=1, output_dim=0, weight_loader=None)
    p.load_row_parallel_weight(full, tp_rank=1)
    out["model_weight_row_rank1of2"] = dict(loaded=full, data=p.data.clone())
    # merged column (gate_up): shard 1 at offset 6, rank 1 of 2
    p = P.ModelWeightParameter(data=torch.zeros(12, 16), input_dim=1, output_dim=0, weight_loader=None)
    up = torch.randn(12, 16, generator=g)
    p.load_merged_column_weight(up, shard_offset=6, shard_size=6, tp_rank=1, use_presharded_weights=False)
    out["model_weight_merged_shard1_rank1of2"] = dict(loaded=up, data=p.data.clone())
    # qkv: k shard (2 kv heads replicated over 4 ranks -> num_heads = 2 ranks per kv head), rank 3
    p = P.ModelWeightParameter(data=torch.zeros(8 + 4 + 4, 16), input_dim=1, output_dim=0, weight_loader=None)
    kfull = torch.randn(8, 16, generator=g)
    p.load_qkv_weight(kfull, tp_rank=3, shard_offset=8, shard_size=4, shard_id="k", num_heads=2)
    out["model_weight_qkv_k_rank3"] = dict(loaded=kfull, data=p.data.clone())
    # per-tensor scales of a fused module: one scalar per shard id
    p = P.PerTensorScaleParameter(data=torch.full((3,), -1.0), weight_loader=None)
    p.load_qkv_weight(torch.tensor(0.5), shard_id="k")
    p.load_merged_column_weight(torch.tensor([0.25]), shard_id=2)
    out["per_tensor_scale_shards"] = dict(data=p.data.clone())
    p = P.PerTensorScaleParameter(data=torch.zeros(1), weight_loader=None)
    p.load_row_parallel_weight(torch.tensor([0.75]), tp_rank=1, use_presharded_weights=False)
    out["per_tensor_scale_row"] = dict(data=p.data.clone())
    # packed int4 weight (AWQ qweight [K, N/8], packed along the output dim): merged shard offsets shrink by the factor
    p = P.PackedvLLMParameter(data=torch.zeros(16, 6, dtype=torch.int32), input_dim=0, output_dim=1, packed_dim=1,
                              packed_factor=8, weight_loader=None)
    qw = torch.randint(0, 1 << 30, (16, 4), generator=g, dtype=torch.int32)
    p.load_merged_column_weight(qw, shard_offset=16, shard_size=16, tp_rank=1, use_presharded_weights=False)
    out["packed_merged_shard_rank1of2"] = dict(loaded=qw, data=p.data.clone(),
                                               adjusted=torch.tensor(p.adjust_shard_indexes_for_packing(shard_size=16, shard_offset=16)))
    # group scales [K/g, N]: column then row sharding
    p = P.GroupQuantScaleParameter(data=torch.zeros(2, 8), input_dim=0, output_dim=1, weight_loader=None)
    sc = torch.randn(2, 16, generator=g)
    p.load_column_parallel_weight(sc, tp_rank=1)
    out["group_scale_column_rank1of2"] = dict(loaded=sc, data=p.data.clone())
    p = P.ChannelQuantScaleParameter(data=torch.zeros(12, 1), output_dim=0, weight_loader=None)
    ch = torch.randn(24, 1, generator=g)
    p.load_column_parallel_weight(ch, tp_rank=0)
    out["channel_scale_column_rank0of2"] = dict(loaded=ch, data=p.data.clone())
    return out


def main():
    m = mg.load_reference()                      # imports base_attn_backend.py and base_config.py on the way
    su = sys.modules["sglang.srt.utils"]
    su.is_cpu = lambda: False
    # parameter.py imports one helper from model_loader/weight_utils.py inside its loaders and calls it on the CPU-AMX
    # path only (`_is_cpu`): the name resolves to a stub so that model_loader/__init__.py never runs
    mg._ns("sglang.srt.model_loader")
    wu = types.ModuleType("sglang.srt.model_loader.weight_utils")
    wu.narrow_padded_param_and_loaded_weight = None
    sys.modules[wu.__name__] = wu
    bab = importlib.import_module("sglang.srt.layers.attention.base_attn_backend")
    bc = importlib.import_module("sglang.srt.layers.quantization.base_config")
    par = importlib.import_module("sglang.srt.layers.parameter")
    ns = {"QuantizeMethodBase": bc.QuantizeMethodBase, "abstractmethod": abstractmethod, "torch": torch,
          "List": List, "Optional": Optional, "Dict": Dict, "Any": Any}
    lmb = class_from_file(f"{REF}/python/sglang/srt/layers/linear.py", "LinearMethodBase", ns)
    classes = {"AttentionBackend": bab.AttentionBackend, "QuantizeMethodBase": bc.QuantizeMethodBase,
               "QuantizationConfig": bc.QuantizationConfig, "LinearMethodBase": lmb}
    for n in ["BasevLLMParameter", "_ColumnvLLMParameter", "RowvLLMParameter", "ModelWeightParameter",
              "GroupQuantScaleParameter", "ChannelQuantScaleParameter", "PerTensorScaleParameter", "PackedvLLMParameter"]:
        classes[n] = getattr(par, n)
    doc = {k: describe_class(v) for k, v in classes.items()}
    with open(os.path.join(HERE, "interfaces.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    torch.save(param_vectors(par), os.path.join(HERE, "interfaces_param.pt"))
    for k, v in doc.items():
        print(k, v["bases"], sorted(v["members"]))


if __name__ == "__main__":
    main()
