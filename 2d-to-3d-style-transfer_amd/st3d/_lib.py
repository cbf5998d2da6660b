"""ctypes binding of libst3d.so (include/st3d.h).  The product path fails loudly when the
HIP library is missing or a call returns an error -- there is NO CPU fallback."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO_PATH = os.path.join(_PKG, "lib", "libst3d.so")

c_f32p = ctypes.c_void_p      # device pointers travel as integers
c_i32p = ctypes.c_void_p
c_u8p = ctypes.c_void_p
c_stream = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/st3d.h declares
SIGNATURES = {
    "st3d_version": (c_int, []),
    "st3d_last_error": (ctypes.c_char_p, []),
    "st3d_device_info": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_size), ctypes.c_char_p, c_int]),
    "st3d_project_verts": (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_int, c_float, c_f32p, c_stream]),
    "st3d_raster_workspace_bytes": (c_size, [c_int, c_int]),
    "st3d_raster_workspace_bytes_binned": (c_size, [c_int, c_int, c_int]),
    "st3d_raster_fwd": (c_int, [c_f32p, c_i32p, c_int, c_int, c_int, c_int, ctypes.c_void_p, c_size, c_i32p, c_f32p,
                                c_f32p, c_f32p, c_float, c_i32p, c_stream]),
    "st3d_shade_fwd": (c_int, [c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                               c_f32p, c_f32p, c_stream]),
    "st3d_shade_bwd": (c_int, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int,
                               c_int, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_shade_bwd_det_workspace_bytes": (c_size, [c_int]),
    "st3d_shade_bwd_det": (c_int, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int,
                                   c_int, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_size, c_stream]),
    "st3d_raster_bwd_det_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "st3d_raster_bwd_det": (c_int, [c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_f32p, ctypes.c_void_p, c_size,
                                    c_stream]),
    "st3d_raster_bwd": (c_int, [c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_f32p, c_stream]),
    "st3d_project_verts_bwd": (c_int, [c_f32p, c_int, c_f32p, c_f32p, c_int, c_float, c_f32p, c_int, c_f32p, c_stream]),
    "st3d_mesh_reg_scratch_floats": (c_size, [c_int, c_int]),
    "st3d_mesh_reg": (c_int, [c_f32p, c_f32p, c_int, c_i32p, c_int, c_i32p, c_i32p, c_i32p, c_int, c_i32p, c_i32p,
                              ctypes.POINTER(c_float), c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_face_setup": (c_int, [c_f32p, c_i32p, c_int, c_int, c_int, ctypes.c_void_p, c_size, c_stream]),
    "st3d_clip_records_bytes": (c_size, [c_int, c_int]),
    "st3d_face_setup_clip": (c_int, [c_f32p, c_i32p, c_int, c_int, c_int, c_float, c_int, ctypes.c_void_p, c_size, c_stream]),
    "st3d_raster_soft_fwd": (c_int, [c_f32p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_i32p, c_i32p,
                                     c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_shade_soft_fwd": (c_int, [c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int, c_float,
                                    c_float, ctypes.POINTER(c_float), c_f32p, c_f32p, c_stream]),
    "st3d_shade_soft_bwd": (c_int, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int,
                                    c_float, c_float, ctypes.POINTER(c_float), c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_raster_soft_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_i32p, c_float, c_f32p, c_stream]),
    "st3d_shade_soft_bwd_det_workspace_bytes": (c_size, [c_int]),
    "st3d_shade_soft_bwd_det": (c_int, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_int, c_int, c_int, c_int,
                                        c_float, c_float, ctypes.POINTER(c_float), c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p,
                                        c_size, c_stream]),
    "st3d_raster_soft_bwd_det_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "st3d_raster_soft_bwd_det": (c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_i32p, c_float, c_f32p, ctypes.c_void_p, c_size, c_stream]),
    "st3d_apply_background": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_f32p, c_stream]),
    "st3d_conv3x3_packed_floats": (c_size, [c_int, c_int]),
    "st3d_conv3x3_pack": (c_int, [c_f32p, c_int, c_int, c_f32p, c_f32p, c_stream]),
    "st3d_conv3x3_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "st3d_conv3x3_dgrad": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "st3d_conv3x3_dgrad_unpool": (c_int, [c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                                          c_stream]),
    "st3d_conv1_bwd_supported": (c_int, [c_int, c_int]),
    "st3d_conv1_bwd_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "st3d_conv1_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_float, c_f32p, ctypes.c_void_p, c_size, c_f32p, c_int, c_int, c_int,
                               c_stream]),
    "st3d_wino_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "st3d_wino_packed_floats": (c_size, [c_int, c_int]),
    "st3d_wino_pack": (c_int, [c_f32p, c_int, c_int, c_f32p, c_f32p, c_stream]),
    "st3d_wino_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int,
                              c_stream]),
    "st3d_wino_dgrad": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int, c_stream]),
    "st3d_wino_dgrad_unpool": (c_int, [c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_int,
                                       c_stream]),
    "st3d_wino_dgrad_chain": (c_int, [c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int,
                                      c_int, c_int, c_stream]),
    "st3d_wino43_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "st3d_wino43_packed_floats": (c_size, [c_int, c_int]),
    "st3d_wino43_pack": (c_int, [c_f32p, c_int, c_int, c_f32p, c_f32p, c_stream]),
    "st3d_wino43_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_stream]),
    "st3d_wino43_dgrad_chain": (c_int, [c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_float, c_f32p, c_int, c_int, c_int, c_int,
                                        c_int, c_stream]),
    "st3d_maxpool2x2_fwd": (c_int, [c_f32p, c_f32p, c_u8p, c_int, c_int, c_int, c_int, c_stream]),
    "st3d_gram_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "st3d_gram_fwd": (c_int, [c_f32p, c_int, c_int, c_int, ctypes.c_void_p, c_size, c_f32p, c_stream]),
    "st3d_gram_multi_workspace_bytes": (c_size, [ctypes.c_void_p, c_int]),
    "st3d_gram_fwd_multi": (c_int, [ctypes.c_void_p, c_int, ctypes.c_void_p, c_size, c_stream]),
    "st3d_gram_bwd": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_float, c_int, c_f32p, c_stream]),
    "st3d_gram_bwd_gated": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_float, c_int, c_f32p, c_stream]),
    "st3d_reduce_partials": (c_int, []),
    "st3d_sqdiff_sum": (c_int, [c_f32p, c_f32p, c_size, c_size, c_float, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_sqdiff_sum_multi": (c_int, [ctypes.c_void_p, c_int, c_f32p, c_f32p, c_int, c_int, c_float, c_float, c_stream]),
    "st3d_axpy_diff": (c_int, [c_f32p, c_f32p, c_size, c_float, c_int, c_f32p, c_stream]),
    "st3d_axpy_diff_gated": (c_int, [c_f32p, c_f32p, c_size, c_float, c_int, c_f32p, c_stream]),
    "st3d_masked_mse": (c_int, [c_f32p, c_f32p, c_f32p, c_int, c_int, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_tv_loss": (c_int, [c_f32p, c_f32p, c_int, c_int, c_int, c_int, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_range_loss": (c_int, [c_f32p, c_size, c_f32p, c_f32p, c_f32p, c_stream]),
    "st3d_adam_step": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_size, c_int, c_float, c_float, c_float, c_float, c_stream]),
    "st3d_vgg_create": (c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "st3d_vgg_set_conv": (c_int, [ctypes.c_void_p, c_int, c_f32p, c_f32p, c_stream]),
    "st3d_vgg_destroy": (c_int, [ctypes.c_void_p]),
    "st3d_plan_create": (c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, c_int, c_int]),
    "st3d_plan_destroy": (c_int, [ctypes.c_void_p]),
    "st3d_plan_bytes": (c_size, [ctypes.c_void_p]),
    "st3d_plan_forward": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_int, c_stream]),
    "st3d_plan_activation": (c_int, [ctypes.c_void_p, c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_int),
                                     ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "st3d_plan_set_content": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_stream]),
    "st3d_plan_get_content_features": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_stream]),
    "st3d_plan_set_content_features": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_stream]),
    "st3d_plan_set_style": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_int, c_stream]),
    "st3d_plan_loss": (c_int, [ctypes.c_void_p, c_f32p, c_int, c_int, c_float, c_float, c_f32p, c_f32p, c_stream]),
    "st3d_plan_graph": (c_int, [ctypes.c_void_p, c_int]),
    "st3d_plan_backward": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.POINTER(ctypes.c_void_p), c_f32p, c_stream]),
    "st3d_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "st3d_comm_init": (c_int, [ctypes.POINTER(ctypes.c_void_p), c_int, c_int, ctypes.c_char_p]),
    "st3d_allreduce_sum_f32": (c_int, [ctypes.c_void_p, c_f32p, c_size, c_stream]),
    "st3d_comm_destroy": (c_int, [ctypes.c_void_p]),
    "st3d_trace_push": (c_int, [ctypes.c_char_p]),
    "st3d_trace_pop": (c_int, []),
    "st3d_trace_enabled": (c_int, []),
    "st3d_plan_profile": (c_int, [ctypes.c_void_p, c_int]),
    "st3d_plan_profile_read": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_int)]),
    "st3d_plan_profile_launches": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_float), c_int,
                                           ctypes.POINTER(c_int)]),
}

_lib = None


class St3dError(RuntimeError):
    pass


def load():
    """dlopen libst3d.so and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise St3dError(f"{SO_PATH} not found: build it with `python 2d-to-3d-style-transfer_amd/build.py` "
                            "(there is no CPU fallback)")
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().st3d_last_error().decode("utf-8", "replace")
        raise St3dError(f"{what or 'libst3d'} failed (code {rc}): {msg}")


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)


def stream_ptr():
    """The torch current HIP stream as the void* the C ABI takes."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=None):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise St3dError("libst3d takes device tensors; got a CPU tensor (no CPU fallback)")
    if not t.is_contiguous():
        raise St3dError("libst3d takes contiguous tensors")
    if t.device.index != torch.cuda.current_device():
        # kernels launch on torch's CURRENT device and stream; a pointer of another GPU would fault there
        raise St3dError(f"tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                        "call torch.cuda.set_device(...) (or use `with torch.cuda.device(...)`) first")
    if dtype is not None and t.dtype != dtype:
        raise St3dError(f"expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())
