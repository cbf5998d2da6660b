"""The C-ABI library loads and exports every symbol include/st3d.h declares (no compute calls:
this runs without a GPU).  Also: the Python binding table covers exactly the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "st3d.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(st3d_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    g.build()
    from st3d import _lib
    assert os.path.exists(_lib.SO_PATH)
    return _lib.SO_PATH


def test_header_declares_the_survey_symbol_set():
    syms = _declared_symbols()
    for need in ("st3d_project_verts", "st3d_raster_fwd", "st3d_shade_fwd", "st3d_shade_bwd", "st3d_conv3x3_fwd",
                 "st3d_conv3x3_dgrad", "st3d_gram_fwd", "st3d_gram_bwd", "st3d_masked_mse", "st3d_adam_step",
                 "st3d_vgg_create", "st3d_plan_loss"):
        assert need in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"libst3d.so lacks {s}"
    lib.st3d_version.restype = ctypes.c_int
    assert lib.st3d_version() >= 100
    lib.st3d_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.st3d_last_error(), bytes)


def test_binding_table_matches_header(lib_path):
    from st3d import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    _lib.load()


def test_invalid_arguments_are_reported_not_crashed(lib_path):
    """Argument validation happens on the host before any launch (works without a GPU)."""
    from st3d import _lib
    lib = _lib.load()
    rc = lib.st3d_project_verts(None, 0, None, None, 0, 1.0, None, None)
    assert rc == -1 and b"invalid argument" in lib.st3d_last_error()
    rc = lib.st3d_conv3x3_fwd(None, None, None, None, 1, 3, 64, 8, 8, 1, None)
    assert rc == -1
    assert lib.st3d_raster_workspace_bytes(2, 100) == 2 * 100 * 48 + 2 * 100 * 4       # records + packed tile ranges
    assert lib.st3d_raster_workspace_bytes(1, 7) == 7 * 48 + 8 * 4                     # range words padded to 2 faces
    assert lib.st3d_conv3x3_packed_floats(64, 3) == max(9 * 4 * 128, 9 * 64 * 128)
    assert lib.st3d_reduce_partials() == 1024


def test_product_path_refuses_cpu_tensors(lib_path):
    """No CPU fallback: device tensors or an error."""
    import torch
    from st3d import _lib, ops
    with pytest.raises(_lib.St3dError):
        ops.gram_fwd(torch.rand(1, 4, 2, 2))
    import style_transfer as ST
    with pytest.raises(RuntimeError):
        ST.gram_matrix(torch.rand(1, 4, 2, 2))


def test_winograd_shape_guards_are_pure_host_logic(lib_path):
    """st3d_wino_supported / st3d_wino43_supported / st3d_wino43_packed_floats touch no device: the tiling rules of
    include/st3d.h hold on a box without a GPU (F(4x4,3x3): Cin >= 64 in steps of 16, Cout in steps of 64, rows of a multiple
    of 64 pixels with H % 4 == 0 or of 32 pixels with H % 8 == 0, each tensor below 2^31 bytes per image)."""
    lib = ctypes.CDLL(lib_path)
    ok43 = lambda *a: lib.st3d_wino43_supported(*a)
    # the twelve Winograd layers of VGG-19 at 512^2: all on F(4x4,3x3)
    for cin, cout, d in ((64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 512, 8), (512, 512, 8),
                         (512, 512, 16)):
        assert ok43(cin, cout, 512 // d, 512 // d) == 1, (cin, cout, d)
    # the reference's default 768^2: conv4_x (96 wide) yes, conv5_1 (48 wide) no
    assert ok43(256, 512, 96, 96) == 1 and ok43(512, 512, 48, 48) == 0
    assert ok43(3, 64, 512, 512) == 0 and ok43(48, 64, 64, 64) == 0 and ok43(64, 32, 64, 64) == 0 and ok43(72, 64, 64, 64) == 0
    assert ok43(64, 64, 12, 96) == 0 and ok43(64, 64, 16, 96) == 1 and ok43(64, 64, 6, 64) == 0 and ok43(64, 64, 4, 64) == 1
    assert ok43(64, 64, 2880, 2880) == 1 and ok43(64, 64, 2944, 2944) == 0 and ok43(128, 128, 2048, 2048) == 0
    lib.st3d_wino43_packed_floats.restype = ctypes.c_size_t
    assert lib.st3d_wino43_packed_floats(512, 256) == 36 * 512 * 256
    assert lib.st3d_wino_supported(512, 512, 48, 48) == 1          # what the plan falls back to there


def test_roctx_ranges_bind_only_on_request(lib_path):
    """st3d_trace_push / st3d_trace_pop are no-ops unless ST3D_ROCTX=1, and then bind libroctx64 at run time;
    balanced pushes and pops around nothing must not disturb anything (no GPU needed)."""
    import subprocess
    import sys
    code = ("import ctypes, sys; l = ctypes.CDLL(%r); l.st3d_trace_push.argtypes = [ctypes.c_char_p];"
            "l.st3d_trace_push(b'render'); l.st3d_trace_push(b'vgg_forward'); l.st3d_trace_pop(); l.st3d_trace_pop();"
            "print(l.st3d_trace_enabled())") % lib_path
    env = dict(os.environ)
    env.pop("ST3D_ROCTX", None)
    off = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert off.returncode == 0 and off.stdout.strip() == "0", off.stderr
    env["ST3D_ROCTX"] = "1"
    on = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert on.returncode == 0 and on.stdout.strip() == "1", on.stderr
