"""ctypes binding of libscpose_hip.so (C ABI declared in include/scpose.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a symbol is absent, importing callers get a RuntimeError.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libscpose_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
DEV_LIB_PATH = os.path.join(_HERE, "libscpose_hip_dev.so")   # same sources, -DSCPOSE_DEV_BUILD (csrc/Makefile: make dev)
if os.environ.get("SCPOSE_DEV") == "1":
    # developer scripts only (tools_dev/README.md): an explicit build for A/B runs, else the development build -- the
    # only one whose kernels contain the phase stamps and ablation switches -- when it has been built
    if os.environ.get("SCPOSE_LIB"):
        LIB_PATH = os.path.abspath(os.environ["SCPOSE_LIB"])
    elif os.path.exists(DEV_LIB_PATH):
        LIB_PATH = DEV_LIB_PATH

DT_BF16, DT_F16 = 0, 1
IN_F32_NCHW, IN_U8_NHWC = 0, 1
ABI_VERSION = 7


class HrnetDesc(ctypes.Structure):
    _fields_ = [
        ("num_joints", c_int32),
        ("final_conv_kernel", c_int32),
        ("num_stages", c_int32),
        ("num_modules", c_int32 * 3),
        ("num_branches", c_int32 * 3),
        ("num_blocks", (c_int32 * 4) * 3),
        ("num_channels", (c_int32 * 4) * 3),
        ("dtype", c_int32),
        ("mean", c_float * 3),
        ("std", c_float * 3),
        ("head", c_int32),
        ("block", c_int32 * 3),
    ]


# name -> (restype, argtypes); every symbol include/scpose.h declares
SYMBOLS = {
    "scpose_abi_version": (c_int32, []),
    "scpose_last_error": (c_char_p, []),
    "scpose_is_dev_build": (c_int32, []),
    "scpose_hrnet_create": (c_int32, [POINTER(HrnetDesc), POINTER(c_char_p), POINTER(c_void_p),
                                      POINTER(c_int64), c_int32, c_int32, POINTER(c_void_p)]),
    "scpose_hrnet_destroy": (c_int32, [c_void_p]),
    "scpose_hrnet_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "scpose_hrnet_heatmap_size": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "scpose_hrnet_stats": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_double),
                                     POINTER(c_double)]),
    "scpose_hrnet_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "scpose_hrnet_tail_fused": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32)]),
    "scpose_hrnet_forward_decode": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "scpose_hrnet_graph_create_decode": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                                   c_void_p, c_void_p, c_void_p, c_size_t, c_int32, POINTER(c_void_p)]),
    "scpose_hrnet_graph_workspace_bytes": (c_int32, [c_void_p, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "scpose_hrnet_graph_create": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_size_t,
                                            c_int32, POINTER(c_void_p)]),
    "scpose_hrnet_graph_launch": (c_int32, [c_void_p, c_void_p]),
    "scpose_hrnet_graph_nodes": (c_int32, [c_void_p, POINTER(c_int32)]),
    "scpose_hrnet_graph_destroy": (c_int32, [c_void_p]),
    "scpose_hrnet_tap_names": (c_int32, [c_void_p, c_char_p, c_int32]),
    "scpose_hrnet_forward_tap": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_char_p, c_void_p,
                                           POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), c_void_p, c_size_t, c_void_p]),
    "scpose_hrnet_forward_profiled": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                                c_void_p, c_size_t, c_void_p]),
    "scpose_hrnet_forward_decode_profiled": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "scpose_hrnet_profile_read": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                            c_void_p, POINTER(c_int32)]),
    "scpose_decode": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                c_void_p, c_void_p]),
    "scpose_max_preds": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p]),
    "scpose_crop_warp": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                   c_void_p]),
    "scpose_crop_warp_roi": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                       c_void_p]),
    "scpose_heatmap_accumulate": (c_int32, [c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "scpose_flip_merge": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                    c_void_p]),
    "scpose_pnp_epnp_ransac": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                         c_int32, c_double, c_int32, c_int32, c_double, c_double, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "scpose_pnp_epnp_ransac_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_double,
                                              c_int32, c_double, c_int32, c_int32, c_double, c_double, c_void_p, c_void_p]),
    "scpose_conv_create": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                     POINTER(c_void_p)]),
    "scpose_conv_destroy": (c_int32, [c_void_p]),
    "scpose_conv_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                      c_int32, c_void_p, c_void_p]),
    "scpose_basic_block_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "scpose_fuse_sum": (c_int32, [POINTER(c_void_p), POINTER(c_int32), c_int32, c_int32, c_int32, c_int32,
                                  c_int32, c_int32, c_void_p, c_void_p]),
    "scpose_nchw_f32_to_blocked": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                             c_void_p, c_void_p]),
    "scpose_blocked_to_nchw_f32": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                             c_void_p, c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "%s not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (or `make -C %s`). There is no CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    # PyTorch-ROCm ships its own libamdhip64; load it FIRST so that this library binds to the
    # same HIP runtime (two runtimes in one process do not see each other's device state).
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise NativeError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if handle.scpose_abi_version() != ABI_VERSION:
        raise NativeError("ABI version mismatch: library %d, binding %d" % (handle.scpose_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().scpose_last_error()
        raise NativeError("%s failed (%d): %s" % (what or "scpose call", rc, msg.decode() if msg else "?"))
