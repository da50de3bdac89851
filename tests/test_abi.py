"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol that
include/oess.h declares, and the Python binding lists exactly those symbols.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "oess.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(oess_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from openess_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "liboess.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in oess.h but not exported"


def test_binding_matches_header():
    from openess_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.oess_abi_version() >= 1
    assert b"gfx950" in lib.oess_build_info()
    assert lib.oess_strerror(-22) == b"invalid argument"


def test_workspace_query_and_argument_validation():
    """Host-only entry points / argument checks (return before any launch)."""
    from openess_amd import _lib
    lib = _lib.load()
    n = lib.oess_voxelize_workspace_bytes(1000, 4, 250, 5, 48, 64, 8)
    assert n >= 1000 * 4 * 16
    assert lib.oess_voxelize_workspace_bytes(-1, 4, 250, 5, 48, 64, 8) == 0
    # null pointers / bad shapes are rejected with OESS_EINVAL without touching the device
    assert lib.oess_voxelize_trilinear_f32(None, None, None, None, None, 1, 0, 5, 48, 64, 0, 0, None, None, 0, None) == -22
    assert lib.oess_task_loss_fwd(None, 0, None, 10, 10, 0, 0, 0, 11, 255, 3, None, None, None) == -22
    assert lib.oess_confusion_accumulate(None, None, 10, 11, 255, None, None) == -22


def test_product_path_refuses_cpu_tensors():
    import torch
    from openess_amd import hip
    with pytest.raises(RuntimeError):
        hip.masked_normalize(torch.zeros(4, 4))
