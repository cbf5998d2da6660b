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
