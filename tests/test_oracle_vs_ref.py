"""Oracle restatement (oracle/roi_nms.c) against the reference's OWN CPU ops compiled from /root/reference
(oracle/_ref/libmmcv_ref.so, recipe oracle/build_ref.py).  Skipped where the prebuilt library is absent."""
import numpy as np
import pytest

from oracle import cref

pytestmark = pytest.mark.skipif(not cref.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("seed", range(4))
def test_nms_equals_reference_cpu(seed):
    rng = np.random.default_rng(seed)
    n = 300
    xy = rng.random((n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + 0.03 + 0.25 * rng.random((n, 2)).astype(np.float32)], 1)
    scores = rng.permutation(n).astype(np.float32) / n  # tie-free: the reference's sort is unstable
    for thr in (0.3, 0.6):
        assert np.array_equal(cref.nms(boxes, scores, thr), cref.ref_nms(boxes, scores, thr))


def test_roi_align_equals_reference_cpu_on_legal_rois():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 32, 32)).astype(np.float32)
    R = 30
    x1y1 = rng.random((R, 2)).astype(np.float32) * 300
    wh = rng.random((R, 2)).astype(np.float32) * 200 + 5
    rois = np.concatenate([(np.arange(R) % 2)[:, None].astype(np.float32), x1y1, x1y1 + wh], 1)
    for scale in (1 / 7.0, 1 / 14.0):
        ref = cref.ref_roi_align_avg(x, rois, (14, 14), scale, 2, True)
        got = cref.roi_align_avg(x, rois, (14, 14), scale, 2, True)
        assert ref is not None
        # the CPU implementation pre-computes weights in a different association order: fp32-roundoff agreement
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)


def test_reference_cpu_rejects_groma_rois_but_oracle_follows_cuda():
    """the reference CPU op asserts on (cx,cy,w,h)*448 ROIs; the CUDA arithmetic (our oracle) does not."""
    x = np.arange(64, dtype=np.float32).reshape(1, 1, 8, 8)
    rois = [[0, 5.0, 5.0, 1.0, 1.0]]
    assert cref.ref_roi_align_avg(x, rois, (2, 2), 1.0, 2, True) is None
    assert np.isfinite(cref.roi_align_avg(x, rois, (2, 2), 1.0, 2, True)).all()
