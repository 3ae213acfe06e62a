"""ctypes binding of libcaesar_yolo_hip.so (the C-ABI in include/caesar_yolo_hip.h).

There is no CPU fallback: if the shared library is missing, or an entry point fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcaesar_yolo_hip.so")

CY_MAX_DET = 300
CY_MAX_STAGES = 8
OP_BKG, OP_SHIFT, OP_CLIP, OP_ZSCALE, OP_HISTEQ, OP_MINMAX = 1, 2, 3, 4, 5, 6
F16, F32, F16X3 = 0, 1, 2
PRECISIONS = {"fp16": F16, "f16": F16, "half": F16, "fp32": F32, "f32": F32, "float": F32, "fp16x3": F16X3, "f16x3": F16X3, "split": F16X3}

EXPORTS = [
    "cy_create", "cy_destroy", "cy_last_error", "cy_load_weights", "cy_load_weights_mem", "cy_num_classes",
    "cy_weight_passes", "cy_class_name", "cy_plan_num_convs", "cy_plan_conv_desc", "cy_letterbox_geometry", "cy_num_anchors",
    "cy_pred_elems", "cy_profile_enable", "cy_profile_summary", "cy_profile_summary_lane", "cy_profile_layers", "cy_mosaic_prepare", "cy_letterbox_pack", "cy_preproc", "cy_preproc_planes", "cy_preproc_params", "cy_forward", "cy_debug_read_conv",
    "cy_decode_nms", "cy_debug_stamps", "cy_debug_fastdiv", "cy_debug_cand_counts", "cy_iou_merge", "cy_detect_tiles", "cy_detect_flush", "cy_detect_fence", "cy_compact_records", "cy_compact_records_ctx", "cy_detect_counters", "cy_conv_bn_silu", "cy_bottleneck64", "cy_make_tile_records",
    "cy_merge_edge_sources",
]


class CyError(RuntimeError):
    pass


class cy_config(C.Structure):
    _fields_ = [("precision", C.c_int), ("max_batch", C.c_int), ("max_h", C.c_int), ("max_w", C.c_int),
                ("max_cand", C.c_int)]


class cy_pre_stage(C.Structure):
    _fields_ = [("op", C.c_int), ("p0", C.c_double), ("p1", C.c_double), ("p2", C.c_double), ("flag", C.c_int)]


class cy_pre_program(C.Structure):
    _fields_ = [("n", C.c_int), ("st", cy_pre_stage * CY_MAX_STAGES)]


class cy_preproc_cfg(C.Structure):
    _fields_ = [("nprog", C.c_int), ("prog", cy_pre_program * 3)]


class cy_conv_desc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("cin", C.c_int), ("cout", C.c_int), ("k", C.c_int), ("s", C.c_int),
                ("act", C.c_int)]


class cy_prof_entry(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("ms", C.c_double), ("flops", C.c_double), ("launches", C.c_long)]


class cy_letterbox(C.Structure):
    _fields_ = [("new_h", C.c_int), ("new_w", C.c_int), ("top", C.c_int), ("left", C.c_int), ("H", C.c_int),
                ("W", C.c_int)]


_lib = None


def load():
    """Load the shared library (once).  Raises CyError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch (device memory / streams for this package) bundles its own HIP runtime.  It must be the one already mapped
    # when libcaesar_yolo_hip.so resolves libamdhip64: two HIP runtimes in one process do not both see the GPU.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CyError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    sig = {
        "cy_create": (C.c_int, [C.c_int, C.POINTER(cy_config), C.POINTER(vp)]),
        "cy_destroy": (C.c_int, [vp]),
        "cy_last_error": (C.c_char_p, [vp]),
        "cy_load_weights": (C.c_int, [vp, C.c_char_p]),
        "cy_load_weights_mem": (C.c_int, [vp, vp, C.c_size_t]),
        "cy_num_classes": (C.c_int, [vp]),
        "cy_weight_passes": (C.c_int, [vp, ip]),
        "cy_class_name": (C.c_char_p, [vp, C.c_int]),
        "cy_plan_num_convs": (C.c_int, [C.c_char, C.c_int]),
        "cy_plan_conv_desc": (C.c_int, [C.c_char, C.c_int, C.c_int, C.POINTER(cy_conv_desc)]),
        "cy_letterbox_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(cy_letterbox)]),
        "cy_num_anchors": (C.c_int, [C.c_int, C.c_int]),
        "cy_pred_elems": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
        "cy_mosaic_prepare": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp]),
        "cy_preproc": (C.c_int, [vp, vp, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(cy_preproc_cfg), vp, vp, vp]),
        "cy_preproc_planes": (C.c_int, [vp, vp, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.POINTER(cy_preproc_cfg), vp, vp, vp]),
        "cy_preproc_params": (C.c_int, [vp, dp, C.c_int]),
        "cy_letterbox_pack": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
        "cy_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
        "cy_profile_enable": (C.c_int, [vp, C.c_int]),
        "cy_profile_summary": (C.c_int, [vp, C.POINTER(cy_prof_entry), C.c_int]),
        "cy_profile_summary_lane": (C.c_int, [vp, C.POINTER(cy_prof_entry), C.c_int, C.c_int]),
        "cy_profile_layers": (C.c_int, [vp, C.POINTER(cy_prof_entry), C.c_int]),
        "cy_debug_read_conv": (C.c_int, [vp, C.c_char_p, fp, C.c_size_t, ip]),
        "cy_decode_nms": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                    vp, vp, vp, vp]),
        "cy_debug_cand_counts": (C.c_int, [vp, ip, C.c_int]),
        "cy_debug_stamps": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
        "cy_debug_fastdiv": (C.c_int, [vp, vp, vp, vp, C.c_int]),
        "cy_iou_merge": (C.c_int, [vp, vp, vp, C.c_int, C.c_float, C.c_double, C.c_double, vp, vp, vp, vp]),
        "cy_detect_tiles": (C.c_int, [vp, vp, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(cy_preproc_cfg), C.c_float, C.c_float, C.c_double, C.c_double,
                                      vp, vp, vp, vp]),
        "cy_detect_flush": (C.c_int, [vp, vp]),
        "cy_detect_fence": (C.c_int, [vp, vp]),
        "cy_compact_records": (C.c_int, [vp, C.c_longlong, vp, C.c_int, C.c_int, vp, vp, vp]),
        "cy_compact_records_ctx": (C.c_int, [vp, vp, C.c_longlong, vp, C.c_int, C.c_int, vp, vp, vp]),
        "cy_detect_counters": (C.c_int, [vp, C.POINTER(C.c_longlong), C.c_int]),
        "cy_conv_bn_silu": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, vp, vp, vp]),
        "cy_bottleneck64": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, C.c_int, vp, vp]),
        "cy_make_tile_records": (C.c_int, [fp, ip, C.c_int, ip, C.c_int, dp]),
        "cy_merge_edge_sources": (C.c_int, [dp, C.c_int, ip, C.c_int, dp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)          # AttributeError here = header/library mismatch
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc < 0:
        msg = load().cy_last_error(ctx)
        raise CyError("libcaesar_yolo_hip: %s (status %d)" % (msg.decode() if msg else "error", rc))
    return rc


def letterbox(h0, w0, imgsz):
    lb = cy_letterbox()
    check(load().cy_letterbox_geometry(h0, w0, imgsz, C.byref(lb)))
    return lb


def plan_convs(scale, nc):
    lib = load()
    n = check(lib.cy_plan_num_convs(scale.encode(), nc))
    out = []
    for i in range(n):
        d = cy_conv_desc()
        check(lib.cy_plan_conv_desc(scale.encode(), nc, i, C.byref(d)))
        out.append((d.name.decode(), d.cin, d.cout, d.k, d.s, d.act))
    return out
