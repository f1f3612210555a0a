"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares (no compute)."""
import ctypes
import glob
import os
import re
import sys

from conftest import ROOT


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(rtx_[a-z0-9_]+)\s*\(", src):
            syms.add(m.group(1))
    # typedef'd callback type, not a function
    syms.discard("rtx_layer_cb")
    return syms


def test_library_builds_and_exports_every_declared_symbol():
    from rectorch_amd import _lib
    path = _lib.build()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    # the ctypes signature table covers exactly the declared symbols
    assert set(_lib.SIGNATURES) == syms
    assert _lib.lib().rtx_abi_version() == 8


def test_struct_layouts_match_the_header():
    from rectorch_amd import _lib
    # int32 x2, int32[9] x2, int32 x2, float, int32 x3 (max_batch, splitk, cond_dim) = 26 * 4 bytes
    assert ctypes.sizeof(_lib.Cfg) == 26 * 4
    assert ctypes.sizeof(_lib.Batch) == 5 * 8 + 8
    # 8 floats, int32, pad to 8, 2 x u64, 2 x ptr
    assert ctypes.sizeof(_lib.Step) == 8 * 4 + 8 + 16 + 16
    assert _lib.Step.seed.offset == 40


def test_no_cpu_fallback_without_device():
    import pytest
    import torch
    from rectorch_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    with pytest.raises(_lib.RtxError):
        _lib.require_gpu()
    from rectorch_amd.engine import Engine
    with pytest.raises(_lib.RtxError):
        Engine([8, 4, 2], [2, 4, 8], "vae", 0.5)


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under rectorch_amd/ may reference it."""
    for path in glob.glob(os.path.join(ROOT, "rectorch_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", ".cpp")):
            src = open(path).read()
            assert "oracle" not in src.replace("the oracle", "").lower() or "import oracle" not in src, path
            assert "from oracle" not in src and "import oracle" not in src and "mvae_oracle" not in src, path


def test_graft_entry_build_succeeds():
    """the driver's build check: __graft_entry__.build() (make is a no-op when everything is up to date)"""
    import importlib
    sys_path_added = ROOT not in sys.path
    if sys_path_added:
        sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_integration_stub_structs_match_the_binding():
    """the ctypes structs shown in INTEGRATION.md list the same fields, in the same order, as rectorch_amd/_lib.py"""
    from rectorch_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, struct in (("Cfg", _lib.Cfg), ("Batch", _lib.Batch), ("Step", _lib.Step)):
        m = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\s*(?:#.*)?\n(?:class|\n|def)" % name, text, re.S)
        assert m, "struct %s not found in INTEGRATION.md" % name
        fields = re.findall(r'\("(\w+)"', m.group(1))
        assert fields == [f[0] for f in struct._fields_], (name, fields)


def _header_struct_fields(name):
    text = open(os.path.join(ROOT, "include", "rectorch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    bodies = {n: b for b, n in re.findall(r"typedef struct \{([^{}]*)\}\s*(\w+);", text)}
    assert name in bodies, name
    out = []
    for decl in bodies[name].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # drop the type: everything up to the last space before the first declarator
        decl = re.sub(r"^(?:const\s+)?\w+(?:\s*\*)?\s+", "", decl)
        for d in decl.split(","):
            out.append(re.sub(r"[\*\s]|\[.*?\]", "", d))
    return out


def test_header_struct_fields_match_the_binding():
    """field names and order of the C structs in include/rectorch_hip.h equal the ctypes mirrors"""
    from rectorch_amd import _lib
    for cname, struct in (("rtx_cfg", _lib.Cfg), ("rtx_batch", _lib.Batch), ("rtx_step", _lib.Step), ("rtx_svae_cfg", _lib.SvaeCfg),
                          ("rtx_dp_cfg", _lib.DpCfg)):
        assert _header_struct_fields(cname) == [f[0] for f in struct._fields_], cname
