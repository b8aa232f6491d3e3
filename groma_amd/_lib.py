"""ctypes binding of libgroma_hip.so / libgroma_hip_f16.so / libgroma_hip_ref.so (include/groma_hip.h: one ABI, three operand
storage types -- bfloat16, IEEE half, and (hi, lo) pairs of halves for the reference-precision path).

The product path has NO fallback: if the HIP library is missing or an op returns non-zero we raise.
(The reference raises RuntimeError from TORCH_CHECK inside mmcv `_ext`; same error class here.)
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one library the product loads.  (Measurement scripts under tests/diag that A/B another build assign
# `groma_amd._lib.LIB_PATH = ...` before the first load -- tests/diag/_variant.py; the product reads no environment switch.)
LIB_PATH = os.path.join(_HERE, "csrc", "libgroma_hip.so")
LIB_PATH_F16 = os.path.join(_HERE, "csrc", "libgroma_hip_f16.so")
LIB_PATH_REF = os.path.join(_HERE, "csrc", "libgroma_hip_ref.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_long = ctypes.c_long
c_float = ctypes.c_float


class GemmDesc(ctypes.Structure):
    """mirror of `gr_gemm_desc` (include/groma_hip.h)"""
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("scale", c_void_p), ("resid", c_void_p), ("ws", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_long), ("ldw", c_long), ("ldc", c_long), ("ldr", c_long),
        ("act", c_int), ("out_f32", c_int), ("splits", c_int),
        ("conv_H", c_int), ("conv_W", c_int), ("conv_C", c_int),
        ("conv_seg_stride", c_long),
        ("resid_mod", c_int),
        ("c_group", c_int), ("c_group_stride", c_int), ("c_row_off", c_int),
        ("tile", c_int),
        ("fp8", c_int), ("a_scale", c_void_p), ("w_scale", c_void_p),
        ("a_parts", c_void_p), ("a_nsplit", c_int), ("a_hd", c_int),
    ]


class GemvDesc(ctypes.Structure):
    """mirror of `gr_gemv_desc` (include/groma_hip.h): the fused decode-step weight-streaming kernel"""
    _fields_ = [
        ("W", c_void_p), ("ldw", c_long), ("M", c_int), ("N", c_int), ("K", c_int),
        ("x_mode", c_int), ("A", c_void_p), ("lda", c_long), ("h", c_void_p), ("ldh", c_long), ("gamma", c_void_p), ("eps", c_float),
        ("a_parts", c_void_p), ("a_nsplit", c_int), ("a_hd", c_int),
        ("epi", c_int), ("C", c_void_p), ("ldc", c_long), ("resid", c_void_p), ("ldr", c_long),
        ("q", c_void_p), ("kc", c_void_p), ("vt", c_void_p), ("cosT", c_void_p), ("sinT", c_void_p),
        ("H", c_int), ("HD", c_int), ("pos0", c_int), ("kv_stride", c_int), ("pos_dev", c_void_p), ("pos_stride", c_int),
        ("w8", c_int), ("w_scale", c_void_p),
    ]


# name -> argtypes ; every function returns int
_P, _I, _L, _F = c_void_p, c_int, c_long, c_float
SIGNATURES = {
    "gr_abi_version": [],
    "gr_operand_type": [],
    "gr_prof_enable": [_I],
    "gr_prof_read": [_P, _P, _P],
    "gr_prof_read_launches": [_L, _P, _P, _P],
    "gr_gemm_yield": [_I],
    "gr_gemm_bf16": [ctypes.POINTER(GemmDesc), _P],
    "gr_gemv_fused": [ctypes.POINTER(GemvDesc), _P],
    "gr_gemm_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P],
    "gr_quant_rows_fp8": [_P, _I, _P, _P, _I, _I, _L, _P],
    "gr_norm_fp8": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "gr_layernorm": [_P, _P, _P, _P, _P, _I, _I, _L, _L, _F, _I, _I, _P],
    "gr_rmsnorm": [_P, _P, _P, _I, _I, _L, _L, _F, _I, _P],
    "gr_attention_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _L, _P, _P, _P],
    "gr_qkv_split": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "gr_decode_attention": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P, _P],
    "gr_resize_h_u8": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gr_resize_v_norm": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gr_cv2_resize_norm": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gr_patchify": [_P, _P, _I, _I, _I, _I, _P],
    "gr_fill_rows_f32": [_P, _P, _I, _I, _L, _P],
    "gr_mean4_tokens": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "gr_s2d_pack": [_P, _P, _I, _I, _I, _P],
    "gr_upsample_coord_pack": [_P, _P, _I, _I, _I, _I, _I, _P],
    "gr_gn_stats_blocks": [_I],
    "gr_gn_stats": [_P, _P, _I, _I, _I, _P],
    "gr_gn_finalize": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "gr_fuse_shuffle": [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "gr_fuse_shuffle_fp8": [_P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P],
    "gr_cast_f32_bf16": [_P, _P, _P, _L, _P],
    "gr_add_rows_f32": [_P, _P, _P, _L, _I, _I, _P],
    "gr_embed_gather": [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    "gr_scatter_rows_f32": [_P, _P, _P, _L, _I, _P],
    "gr_argmax_rows": [_P, _P, _I, _I, _L, _P],
    "gr_sample_rows": [_P, _P, _I, _I, _L, _P, _P, _P, _I, _I, _P],
    "gr_greedy_advance": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _I, _I, _I, _P],
    "gr_msda_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "gr_mha32_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "gr_ddetr_topk_gather": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "gr_box_refine": [_P, _P, _P, _L, _P],
    "gr_score_fuse": [_P, _P, _P, _L, _L, _P],
    "gr_topk_desc": [_P, _P, _I, _I, _I, _L, _P],
    "gr_nms_workspace_bytes": [_I, _I],
    "gr_nms_f32": [_P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P],
    "gr_nms": [_P, _P, _I, _F, _I, _P, _P, _P, _P],
    "gr_roi_align_forward": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "gr_roi_align_pack": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P],
    "gr_roi_align_pack_fp8": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _F, _P],
}

_lib = None       # the bf16 build (kept under this name: tests monkeypatch it)
_lib_f16 = None
_lib_ref = None


class ThreadSlot:
    """A one-element list whose element is PER THREAD (slot[0] reads / writes the calling thread's value; a thread that never
    wrote sees the default).  The reference's model worker serves from threads (R: groma/serve/model_worker.py): the active operand
    type and GEMM plan are switched by context managers around a model's entry points, and two models of different precision
    used from two threads must not see each other's switch."""

    def __init__(self, default):
        self._default, self._tl = default, threading.local()

    def __getitem__(self, i):
        return getattr(self._tl, "v", self._default)

    def __setitem__(self, i, v):
        self._tl.v = v


# the active operand storage type: "bf16" | "fp16" | "ref" (ops.precision() switches it around a model's calls)
PRECISION = ThreadSlot("bf16")


def _open(path, operand):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(groma_amd has no CPU / eager fallback by design)")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_long if name == "gr_nms_workspace_bytes" else c_int
    if lib.gr_abi_version() != 9:
        raise RuntimeError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.gr_operand_type() != operand:
        raise RuntimeError(f"{os.path.basename(path)} was built for another 16-bit operand type")
    # The float64-accumulating proposer GEMM (csrc/gemm_f32.hip) asks the hardware once which accumulator element holds which output
    # row of v_mfma_f64_16x16x4_f64.  Ask NOW, at load time, so that the probe (a tiny launch + a stream sync) can never fall into a
    # hipGraph capture; an unexpected answer fails here, loudly, instead of at the first proposer launch.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        probe = getattr(lib, "gr_diag_mfma64_rowmap", None)
        if probe is not None:
            probe.restype = c_int
            if probe() not in (0, 1):
                raise RuntimeError(f"{os.path.basename(path)}: the f64 MFMA accumulator layout probe failed (csrc/gemm_f32.hip mfma64_rowmap)")
    return lib


def load(precision=None):
    """The library of the active (or given) operand type, loaded and typed on first use.  Raises if anything is missing."""
    global _lib, _lib_f16, _lib_ref
    precision = precision or PRECISION[0]
    if precision == "bf16":
        if _lib is None:
            _lib = _open(LIB_PATH, 0)
        return _lib
    if precision == "fp16":
        if _lib_f16 is None:
            _lib_f16 = _open(LIB_PATH_F16, 1)
        return _lib_f16
    if precision == "ref":
        if _lib_ref is None:
            _lib_ref = _open(LIB_PATH_REF, 2)
        return _lib_ref
    raise ValueError(f"unknown precision {precision!r}")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (invalid argument)" if rc == 22 else " (HIP error)"))
