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
    header = open(os.path.join(ROOT, "include", "oess.h")).read()
    declared_version = int(re.search(r"#define\s+OESS_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.oess_abi_version() == declared_version == _lib.ABI_VERSION
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


def test_round3_entry_points_validate_arguments_on_the_host():
    """Size queries and argument checks of the entry points added in round 3 (no launches)."""
    from openess_amd import _lib
    lib = _lib.load()
    # split-K scratch: wanted by the ASPP geometry at the BASELINE size, not by an ordinary layer or a tiny map
    assert lib.oess_conv2d_fwd_workspace_bytes(8, 28, 40, 2048, 256, 3, 3, 1, 12, 12, 1, 0) >= 2 * 8960 * 256 * 4
    assert lib.oess_conv2d_fwd_workspace_bytes(8, 110, 160, 64, 256, 1, 1, 1, 0, 1, 1, 0) == 0
    assert lib.oess_conv2d_fwd_workspace_bytes(0, 28, 40, 2048, 256, 3, 3, 1, 12, 12, 1, 0) == 0
    assert lib.oess_norm_partials_bytes(1, 140800, 256, 0) > 0 and lib.oess_norm_partials_bytes(0, 10, 8, 0) == 0
    assert lib.oess_png_decode_scratch_bytes(24 * 7000, 24, 440, 640) >= 24 * 440 * 641
    assert lib.oess_png_decode_gray8_batch(None, None, 100, 1, 8, 8, None, None, None, 0, None, None, None) == -22
    assert lib.oess_norm_tile_stats_apply_nhwc_bf16(None, 70, 256, 8960.0, 1e-5, None, None, None, None, 0.1, None, None, None, 256,
                                                    None, 0, 1, 8960, None, 256, None) == -22
    assert lib.oess_norm_stats_finalize_nhwc_bf16(None, 64, 1, 100, 64, 1e-5, None, None, None, None, 0.1, None, None, None, None,
                                                  None, 0, None) == -22


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """A library built for another ABI version must not load (the version, not a missing symbol, is the check)."""
    import subprocess
    from openess_amd import _lib
    src = tmp_path / "fake.c"
    lines = ["int oess_abi_version(void) { return %d; }" % (_lib.ABI_VERSION - 1)]
    for name in _lib.SIGNATURES:
        if name != "oess_abi_version":
            lines.append(f"void* {name}(void) {{ return 0; }}")
    src.write_text("\n".join(lines))
    so = tmp_path / "libfake.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.LibraryMissing, match="C-ABI version"):
        _lib.load()


def test_round4_entry_points_validate_arguments_on_the_host():
    from openess_amd import _lib
    lib = _lib.load()
    assert lib.oess_segment_mean_fwd_workspace_bytes(800, 256) >= 800 * 256 * 16 + 800 * 4 + 4
    assert lib.oess_segment_mean_fwd_workspace_bytes(0, 256) == 0
    assert lib.oess_segment_mean_fwd(None, 0, None, 100, 100, 100, 256, 100, None, None, None, 0, None) == -22
    assert lib.oess_linear_probe_partials_bytes(11) == 1024 * (121 + 11) * 8 and lib.oess_linear_probe_partials_bytes(33) == 0
    assert lib.oess_linear_probe_fwd_f32(None, None, None, 100, 11, None, None) == -22
    assert lib.oess_linear_probe_bwd_f32(None, None, None, 100, 11, None, None, None, None, 0, None) == -22
    assert lib.oess_maxpool3x3s2_fwd_nhwc_bf16(None, 64, 1, 8, 8, 64, None, 64, None, None) == -22
    assert lib.oess_maxpool3x3s2_bwd_nhwc_bf16(None, 64, None, 1, 8, 8, 64, None, 64, None) == -22
    assert lib.oess_dropout_nhwc_bf16(None, 8, None, 8, 10, 8, 0.1, 1, 2, None) == -22
    assert lib.oess_aspp_pool_fwd_f32(None, 1.0, None, None, None, None, None, 0.1, 1e-5, 8, 2048, 256, None, None, None, None, None) == -22
    assert lib.oess_aspp_pool_bwd_f32(None, None, 1.0, None, None, None, None, None, 8, 2048, 256, None, None, None, None, None, None) == -22


def test_collate_keeps_undecoded_png_maps_as_one_byte_stream():
    """device_png: collate concatenates the per-sample file bytes of a slot and keeps lengths / flips / size;
    every other slot stacks as before and the batch keeps the 7-slot layout."""
    import torch
    from openess_amd.datasets.synthetic_events import collate
    def sample(i):
        png = {'png': torch.arange(5 + i, dtype=torch.uint8), 'flip': bool(i & 1), 'hw': (4, 6)}
        return (torch.zeros(3, 4, 6), png, torch.ones(3, 4, 6), png, torch.zeros(4, 6, dtype=torch.int64), torch.ones(2), f"p{i}")
    b = collate([sample(0), sample(1), sample(2)])
    assert len(b) == 7 and b[0].shape == (3, 3, 4, 6) and b[6] == ["p0", "p1", "p2"]
    assert b[1]['png_lengths'] == [5, 6, 7] and b[1]['flip'] == [False, True, False] and b[1]['hw'] == (4, 6)
    assert b[1]['png_bytes'].numel() == 18 and torch.equal(b[3]['png_bytes'], b[1]['png_bytes'])
    assert b[4].shape == (3, 4, 6)


def test_product_path_refuses_cpu_tensors():
    import torch
    from openess_amd import hip
    with pytest.raises(RuntimeError):
        hip.masked_normalize(torch.zeros(4, 4))
