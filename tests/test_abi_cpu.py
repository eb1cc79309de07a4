"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/bp_layer.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import broadphase_rs_amd as bp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bp_layer.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    bp.build_library()
    lib = bp.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "symbol %s declared in include/bp_layer.h is not exported" % name
    assert sorted(declared) == sorted(bp.layer.EXPORTS)
    assert lib.bp_version().decode().startswith("broadphase_amd")


def test_status_strings():
    lib = bp.load_library()
    assert lib.bp_status_string(0) == b"BP_OK"
    assert lib.bp_status_string(-2) == b"BP_ERR_NO_DEVICE"
    assert lib.bp_status_string(-6) == b"BP_ERR_KERNEL"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(bp.BroadphaseError) as ei:
        bp.LayerBuilder.new().build(bp.Index64_3D)
    assert ei.value.status == "BP_ERR_NO_DEVICE"


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "broadphase_rs_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "bp_oracle" not in src and "liboracle" not in src, f
