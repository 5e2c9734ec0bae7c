"""ctypes binding of libctd_hip.so (C ABI in include/ctd_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or cannot
be loaded, importing a symbol from here raises.  `build()` compiles it in-tree.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctd_hip.so")
SELFTEST_PATH = os.path.join(_HERE, "ctd_selftest")

# ---- constants mirrored from include/ctd_hip.h -------------------------------
ABI_VERSION = 6
OK = 0
PREC_F32, PREC_F16, PREC_F32S = 0, 1, 2
ACT = {"none": 0, "silu": 1, "leaky": 2, "relu": 3, "sigmoid": 4}
IN_NCHW_F32, IN_NHWC_U8 = 0, 1
(OP_INPUT, OP_CONV, OP_CONVT, OP_MAXPOOL, OP_AVGPOOL2, OP_DETECT, OP_EXPORT, OP_STEM, OP_SEG_FINAL,
 OP_DB_UP) = range(1, 11)
OUT_MASK, OUT_LINES = 0, 1


class CtdTensor(C.Structure):
    _fields_ = [("channels", C.c_int32), ("log2_down", C.c_int32), ("dtype", C.c_int32)]


class CtdOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("src0", C.c_int32), ("src0_coff", C.c_int32), ("src0_c", C.c_int32), ("src0_up", C.c_int32),
        ("src1", C.c_int32), ("src1_coff", C.c_int32), ("src1_c", C.c_int32), ("src1_up", C.c_int32),
        ("res", C.c_int32), ("res_coff", C.c_int32),
        ("dst", C.c_int32), ("dst_coff", C.c_int32),
        ("cout", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("act", C.c_int32),
        ("w_off", C.c_int64), ("b_off", C.c_int64),
        ("aux", C.c_int32 * 8),
        ("faux", C.c_float * 8),
    ]


class CtdTailPage(C.Structure):
    _fields_ = [("img_dev", C.c_void_p), ("im_h", C.c_int32), ("im_w", C.c_int32), ("dw", C.c_int32), ("dh", C.c_int32)]


class CtdTailParams(C.Structure):
    _fields_ = [("conf_thresh", C.c_float), ("nms_thresh", C.c_float), ("box_thresh", C.c_float),
                ("max_candidates", C.c_int32), ("unclip_ratio", C.c_double), ("refine", C.c_int32),
                ("refine_mode", C.c_int32), ("keep_undetected_mask", C.c_int32), ("pad_", C.c_int32)]


class CtdBlk(C.Structure):
    _fields_ = [("xyxy", C.c_int32 * 4), ("language", C.c_int32), ("vertical", C.c_int32), ("angle", C.c_int32),
                ("font_is_float", C.c_int32), ("font_size", C.c_double), ("vec", C.c_double * 2), ("norm", C.c_double),
                ("weight", C.c_double), ("merged", C.c_int32), ("line_off", C.c_int32), ("n_lines", C.c_int32),
                ("dist_off", C.c_int32), ("n_dist", C.c_int32), ("pad_", C.c_int32)]


# every symbol include/ctd_hip.h declares: (restype, argtypes)
_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "ctd_engine_create": (_i32, [C.POINTER(_vp), C.POINTER(CtdTensor), _i32, C.POINTER(CtdOp), _i32,
                                 C.POINTER(C.c_float), _i64, _i32, _i32]),
    "ctd_engine_destroy": (None, [_vp]),
    "ctd_engine_blks_shape": (_i32, [_vp, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "ctd_engine_forward": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctd_engine_n_ops": (_i32, [_vp]),
    "ctd_engine_op_work": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i32)]),
    "ctd_engine_op_kernel": (_i32, [_vp, _i32, C.c_char_p, _i32]),
    "ctd_engine_profile": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                  C.POINTER(C.c_float)]),
    "ctd_engine_read_tensor": (_i32, [_vp, _i32, C.POINTER(C.c_float), _i64]),
    "ctd_engine_workspace_bytes": (_i64, [_vp]),
    "ctd_engine_arena_generation": (_i32, [_vp]),
    "ctd_tuning_set": (_i32, [C.c_char_p, C.c_int64]),
    "ctd_nms_workspace_bytes": (C.c_size_t, [_i32, _i32]),
    "ctd_nms": (_i32, [_vp, _i32, _i32, _i32, _f, _f, _i32, _i32, _f, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ctd_db_step": (_i32, [_vp, _i32, _i32, _i32, _f, _vp, _vp, _f, _vp]),
    "ctd_ccl_workspace_bytes": (C.c_size_t, [_i32, _i32, _i32]),
    "ctd_ccl": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, C.c_size_t, _vp]),
    "ctd_ccl_dual": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, C.c_size_t, _vp]),
    "ctd_resize_linear_u8": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ctd_db_boxes": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, C.c_double, _vp, _vp, C.POINTER(_i32)]),
    "ctd_tail_create": (_i32, [C.POINTER(_vp), _i32]),
    "ctd_tail_destroy": (None, [_vp]),
    "ctd_tail_stream": (_vp, [_vp]),
    "ctd_tail_run": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i64, _vp, C.POINTER(CtdTailPage),
                            C.POINTER(CtdTailParams), _vp, _vp, _vp]),
    "ctd_tail_timings": (_i32, [_vp, C.POINTER(C.c_double)]),
    "ctd_tail_refine_paths": (_i32, [_vp, C.POINTER(_i32)]),
    "ctd_tail_db_boxes": (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _i32, C.c_double]),
    "ctd_tail_refine": (_i32, [_vp, _i32, C.POINTER(CtdTailPage), _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "ctd_tail_page_counts": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                    C.POINTER(_i32)]),
    "ctd_tail_page_fetch": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctd_tail_batch_counts": (_i32, [_vp, _vp]),
    "ctd_tail_batch_fetch": (_i32, [_vp, _vp, _vp, _vp]),
    "ctd_tail_pack_records": (_i32, [_vp, _i32, _i32, _vp]),
    "ctd_tail_set_threads": (_i32, [_vp, _i32]),
    "ctd_db_boxes_compact": (_i32, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _i32, C.c_double, _vp, _vp, C.POINTER(_i32)]),
    "ctd_topk_colors": (_i32, [_vp, _vp]),
    "ctd_otsu_from_hist": (_i32, [_vp]),
    "ctd_inrange_bounds": (None, [C.c_double, C.c_double, C.POINTER(_i32), C.POINTER(_i32)]),
    "ctd_group_output": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32,
                                C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "ctd_host_gather": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "ctd_last_error": (C.c_char_p, []),
    "ctd_abi_version": (_i32, []),
    "ctd_device_info": (_i32, [_i32, C.c_char_p, C.POINTER(_i32), C.POINTER(_i64)]),
}


class CtdError(RuntimeError):
    pass


def build(verbose: bool = False) -> None:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    import sys
    r = subprocess.run(["make", "-C", src, "-j8", f"PYTHON={sys.executable}"], capture_output=True, text=True)
    if r.returncode != 0:
        raise CtdError("building libctd_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])


_lib = None


def lib() -> C.CDLL:
    """Loads libctd_hip.so (loudly failing if absent) and types every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise CtdError(f"{LIB_PATH} is missing: the HIP extension was not built "
                       "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                       "There is no CPU fallback for the product path.")
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Import it
    # first so the process has ONE HIP runtime: streams and device pointers made by
    # torch are then valid in this library.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)       # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.ctd_abi_version() != ABI_VERSION:
        raise CtdError("libctd_hip.so ABI version mismatch; rebuild")
    _lib = L
    _apply_env_tuning(L)
    return L


# The library reads no environment variables; dispatch knobs go through `ctd_tuning_set`.  For A/B runs of unchanged
# scripts the host side applies CTD_TUNING="key=value,key=value" once at load time (and the older per-knob names).
_LEGACY_ENV = {"CTD_FUSE": "fuse", "CTD_NO_REUSE": "no_reuse", "CTD_F32_MFMA": "f32_mfma", "CTD_HALO_PAIR": "halo_pair",
               "CTD_HALO_MIN_PATCHES": "halo_min_patches", "CTD_DBUP_MFMA": "db_up_mfma", "CTD_SEGFINAL_MFMA": "seg_final_mfma"}


def _apply_env_tuning(L) -> None:
    items = []
    for env, key in _LEGACY_ENV.items():
        if env in os.environ:
            v = os.environ[env]
            items.append((key, 1 if (env == "CTD_NO_REUSE" and not v.lstrip("-").isdigit()) else int(v)))
    for part in os.environ.get("CTD_TUNING", "").split(","):
        if "=" in part:
            k, v = part.split("=", 1)
            items.append((k.strip(), int(v)))
    for k, v in items:
        if L.ctd_tuning_set(k.encode(), v) != OK:
            raise CtdError(f"unknown tuning key {k!r} (CTD_TUNING / legacy environment knob)")


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = lib().ctd_last_error().decode("utf8", "replace")
        raise CtdError(f"{what} failed (rc={rc}): {msg}")
