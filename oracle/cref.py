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
