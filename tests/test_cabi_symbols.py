"""The C-ABI library loads and exports every symbol include/groma_hip.h declares (no compute calls: CPU suite)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "groma_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(gr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_typed():
    import __graft_entry__ as g
    g.build()
    from groma_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for path in (_lib.LIB_PATH, _lib.LIB_PATH_F16, _lib.LIB_PATH_REF):   # the three builds of the one ABI (bf16 / fp16 / split operands)
        lib = ctypes.CDLL(path)
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/groma_hip.h but not exported by {path}"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert _lib.load().gr_abi_version() == 9 and _lib.load().gr_operand_type() == 0
    assert _lib.load("fp16").gr_abi_version() == 9 and _lib.load("fp16").gr_operand_type() == 1
    assert _lib.load("ref").gr_abi_version() == 9 and _lib.load("ref").gr_operand_type() == 2


def test_split_build_rejects_what_it_has_no_form_of():
    """libgroma_hip_ref.so: the streaming decode kernels / e4m3 path return EINVAL before any launch (include/groma_hip.h)"""
    from groma_amd import _lib
    lib = _lib.load("ref")
    d = _lib.GemmDesc()
    d.A = d.W = d.C = d.ws = 1  # non-null; validation only
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = 4, 4096, 4096, 4096, 4096, 4096
    d.tile, d.splits = 1, 8
    assert lib.gr_gemm_bf16(ctypes.byref(d), None) == 22
    d.tile, d.splits, d.lda = 0, 1, 4100  # a 16-bit row stride that is not whole (hi, lo) blocks
    assert lib.gr_gemm_bf16(ctypes.byref(d), None) == 22
    assert lib.gr_decode_attention(ctypes.c_void_p(1), ctypes.c_void_p(1), ctypes.c_void_p(1), ctypes.c_void_p(1), None, 1, 1, 1, 64, 64,
                                   0, 1.0, None, 0, 1, None, None) == 22


def test_bad_arguments_are_rejected_without_a_gpu():
    """argument validation happens before any launch: EINVAL (22), never a crash"""
    from groma_amd import _lib
    lib = _lib.load()
    assert lib.gr_gemm_bf16(None, None) == 22
    d = _lib.GemmDesc()
    assert lib.gr_gemm_bf16(ctypes.byref(d), None) == 22
    assert lib.gr_topk_desc(None, None, 1, 10, 5, 10, None) == 22
    assert lib.gr_nms_f32(None, None, 1, 10, 0.5, 0.0, 10, None, None, None, None, None) == 22
    assert lib.gr_nms_workspace_bytes(2, 300) == 0 and lib.gr_nms_workspace_bytes(2, 513) > 2 * 4096 * 64 * 8
    assert lib.gr_nms(None, None, 10, 0.5, 0, None, None, None, None) == 22
    assert lib.gr_nms(None, None, 10, 0.5, 2, None, None, None, None) == 22  # offset must be 0 or 1
    assert lib.gr_roi_align_forward(None, None, None, None, None, 0, 8, 4, 4, 2, 2, 1.0, 2, 1, 1, None) == 0  # K = 0
    assert lib.gr_roi_align_forward(None, None, None, None, None, 3, 8, 4, 4, 2, 2, 1.0, 2, 1, 1, None) == 22
    assert lib.gr_roi_align_forward(None, None, None, None, None, 3, 8, 4, 4, 2, 2, 1.0, 2, 7, 1, None) == 22  # pool_mode
    assert lib.gr_roi_align_pack(None, None, None, 0, 8, 4, 4, 14, 14, 1.0, 2, 1, 1, 0, None) == 0  # empty ROI set is fine
    assert lib.gr_roi_align_pack(None, None, None, 3, 8, 4, 4, 14, 14, 1.0, 2, 1, 1, 0, None) == 22
    assert lib.gr_attention_bf16(None, None, None, None, None, 1, 1, 1, 1, 64, 64, 0, 0, 1.0, None, 0, 0, None, None, None) == 22
    assert lib.gr_decode_attention(None, None, None, None, None, 1, 1, 1, 64, 64, 0, 1.0, None, 0, 1, None, None) == 22
    assert lib.gr_greedy_advance(None, None, None, None, None, None, None, 4, -1, 0, 8, 1, 1, None) == 22


def test_missing_library_fails_loudly(monkeypatch):
    from groma_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgroma_hip.so")
    import pytest
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.load()
