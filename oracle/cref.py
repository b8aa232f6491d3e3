"""ctypes access to oracle/liboracle_c.so (roi_nms.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_c.so")
_lib = None


def build():
    src = os.path.join(_HERE, "roi_nms.c")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_c.so"], stdout=subprocess.DEVNULL)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.oracle_nms.restype = ctypes.c_int
        _lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                    ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        _lib.oracle_roi_align_avg.restype = None
        _lib.oracle_roi_align_avg.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.oracle_roi_align_max.restype = None
        _lib.oracle_roi_align_max.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
    return _lib


def nms(boxes_xyxy, scores, iou_threshold, offset=0, score_threshold=0.0, max_num=-1):
    """numpy in, numpy int64 indices out (mmcv.ops.nms semantics; see roi_nms.c)"""
    b = np.ascontiguousarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
    s = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    keep = np.empty((max(len(s), 1),), dtype=np.int64)
    k = _load().oracle_nms(b.ctypes.data, s.ctypes.data, len(s), float(iou_threshold), int(offset),
                           float(score_threshold), int(max_num), keep.ctypes.data)
    return keep[:k].copy()


def roi_align_avg(inp_nchw, rois, pooled, spatial_scale, sampling_ratio, aligned=True):
    x = np.ascontiguousarray(inp_nchw, dtype=np.float32)
    r = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, C, H, W = x.shape
    ph, pw = pooled
    out = np.zeros((r.shape[0], C, ph, pw), dtype=np.float32)
    if r.shape[0]:
        _load().oracle_roi_align_avg(x.ctypes.data, r.ctypes.data, out.ctypes.data, r.shape[0], C, H, W, ph, pw,
                                     float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
    return out


def roi_align_max(inp_nchw, rois, pooled, spatial_scale, sampling_ratio, aligned=True):
    """-> (output, argmax_y, argmax_x), the max-pooling branch (pool_mode 0)"""
    x = np.ascontiguousarray(inp_nchw, dtype=np.float32)
    r = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, C, H, W = x.shape
    ph, pw = pooled
    out, ay, ax = (np.zeros((r.shape[0], C, ph, pw), dtype=np.float32) for _ in range(3))
    if r.shape[0]:
        _load().oracle_roi_align_max(x.ctypes.data, r.ctypes.data, out.ctypes.data, ay.ctypes.data, ax.ctypes.data,
                                     r.shape[0], C, H, W, ph, pw, float(spatial_scale), int(sampling_ratio),
                                     int(bool(aligned)))
    return out, ay, ax


# ---- the reference's own CPU ops (oracle/_ref/libmmcv_ref.so, built by oracle/build_ref.py from /root/reference) ----
_REF = os.path.join(_HERE, "_ref", "libmmcv_ref.so")
_ref = None


def ref_available():
    return os.path.exists(_REF)


def _load_ref():
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (libtorch must be loaded first)
        _ref = ctypes.CDLL(_REF)
        _ref.ref_nms.restype = ctypes.c_int
        _ref.ref_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        _ref.ref_roi_align_avg.restype = ctypes.c_int
        _ref.ref_roi_align_avg.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
    return _ref


def ref_nms(boxes_xyxy, scores, iou_threshold, offset=0):
    """mmcv `_ext.nms` (CPU) itself -- note: its sort is unstable, compare on tie-free scores"""
    b = np.ascontiguousarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
    s = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    keep = np.empty((max(len(s), 1),), dtype=np.int64)
    k = _load_ref().ref_nms(b.ctypes.data, s.ctypes.data, len(s), float(iou_threshold), int(offset), keep.ctypes.data)
    if k < 0:
        raise RuntimeError("reference nms raised")
    return keep[:k].copy()


def ref_roi_align_avg(inp_nchw, rois, pooled, spatial_scale, sampling_ratio, aligned=True):
    """mmcv `_ext.roi_align_forward` (CPU) itself; returns None when the reference asserts (negative-width ROI)"""
    x = np.ascontiguousarray(inp_nchw, dtype=np.float32)
    r = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, C, H, W = x.shape
    out = np.zeros((r.shape[0], C, pooled[0], pooled[1]), dtype=np.float32)
    rc = _load_ref().ref_roi_align_avg(x.ctypes.data, r.ctypes.data, out.ctypes.data, N, C, H, W, r.shape[0], pooled[0],
                                       pooled[1], float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
    return out if rc == 0 else None
