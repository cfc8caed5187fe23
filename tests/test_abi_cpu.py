"""CPU-only: the C-ABI library loads and exports every symbol include/mi_hotpath.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "mi_hotpath.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import iaas_sglang_amd._lib as L
    names = _header_symbols()
    assert len(names) >= 14
    raw = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in mi_hotpath.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert set(L.SIGNATURES) == set(names)
    hdr = open(os.path.join(ROOT, "include", "mi_hotpath.h")).read()
    assert L.lib.mi_abi_version() == int(re.search(r"#define MI_ABI_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: no file of the product package may reference it
    pkg = os.path.join(ROOT, "iaas_sglang_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
