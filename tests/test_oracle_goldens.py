"""Pins the oracle (test infrastructure) to the reference's own golden vectors -- CPU only.
  mmcv/tests/test_ops/test_nms.py:13-29          nms, 4 boxes, iou 0.3 -> [1,0,3] + dets
  mmcv/mmcv/ops/nms.py:139-150                   docstring example, 7 boxes, iou 0.6 -> 3 kept
  mmcv/tests/test_ops/test_roi_align.py:14-32    3 RoIAlign cases, pool 2x2, scale 1, sampling 2, aligned
  mmcv/tests/test_ops/test_ms_deformable_attn.py:54-70  MSDA recipe (self-consistency in double precision)"""
import numpy as np
import torch

from oracle import cref
from oracle import groma_oracle as O


def test_nms_golden_mmcv_test():
    boxes = np.array([[6.0, 3.0, 8.0, 7.0], [3.0, 6.0, 9.0, 11.0], [3.0, 7.0, 10.0, 12.0], [1.0, 4.0, 13.0, 7.0]],
                     dtype=np.float32)
    scores = np.array([0.6, 0.9, 0.7, 0.2], dtype=np.float32)
    inds = cref.nms(boxes, scores, 0.3, 0)
    assert inds.tolist() == [1, 0, 3]
    dets = np.concatenate([boxes[inds], scores[inds, None]], 1)
    assert np.allclose(dets, [[3.0, 6.0, 9.0, 11.0, 0.9], [6.0, 3.0, 8.0, 7.0, 0.6], [1.0, 4.0, 13.0, 7.0, 0.2]])


def test_nms_docstring_example():
    boxes = np.array([[49.1, 32.4, 51.0, 35.9], [49.3, 32.9, 51.0, 35.3], [49.2, 31.8, 51.0, 35.4],
                      [35.1, 11.5, 39.1, 15.7], [35.6, 11.8, 39.3, 14.2], [35.3, 11.5, 39.9, 14.5],
                      [35.2, 11.7, 39.7, 15.7]], dtype=np.float32)
    scores = np.array([0.9, 0.9, 0.5, 0.5, 0.5, 0.4, 0.3], dtype=np.float32)
    assert len(cref.nms(boxes, scores, 0.6)) == 3


def test_nms_score_threshold_and_max_num():
    rng = np.random.default_rng(0)
    xy = rng.random((200, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + 0.05 + 0.2 * rng.random((200, 2)).astype(np.float32)], 1)
    scores = rng.random(200).astype(np.float32)
    all_k = cref.nms(boxes, scores, 0.6)
    assert np.all(np.diff(scores[all_k]) <= 0)                       # descending score order
    k = cref.nms(boxes, scores, 0.6, 0, 0.5, 10)
    assert len(k) <= 10 and np.all(scores[k] > 0.5)
    assert np.array_equal(k, [i for i in all_k if scores[i] > 0.5][:10])  # filter commutes with greedy order
    assert len(cref.nms(boxes, np.zeros(200, np.float32), 0.6, 0, 0.15, 100)) == 0  # nothing passes (T5)
    assert len(cref.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.6)) == 0


GOLD_IN = [([[[[1., 2.], [3., 4.]]]], [[0., 0., 0., 1., 1.]]),
           ([[[[1., 2.], [3., 4.]], [[4., 3.], [2., 1.]]]], [[0., 0., 0., 1., 1.]]),
           ([[[[1., 2., 5., 6.], [3., 4., 7., 8.], [9., 10., 13., 14.], [11., 12., 15., 16.]]]], [[0., 0., 0., 3., 3.]])]
GOLD_OUT = [[[[[1.0, 1.25], [1.5, 1.75]]]],
            [[[[1.0, 1.25], [1.5, 1.75]], [[4.0, 3.75], [3.5, 3.25]]]],
            [[[[1.9375, 4.75], [7.5625, 10.375]]]]]


def test_roi_align_goldens():
    for (x, r), exp in zip(GOLD_IN, GOLD_OUT):
        out = cref.roi_align_avg(np.array(x), np.array(r), (2, 2), 1.0, 2, True)
        assert np.allclose(out, np.array(exp), atol=1e-3)  # the reference's own tolerance (test_roi_align.py:92-95)
        assert np.array_equal(out, np.array(exp, dtype=np.float32))


def test_roi_align_negative_width_is_cuda_semantics():
    """Groma's (cx,cy,w,h)*448 ROIs: the CUDA kernel mirrors the sampling grid instead of asserting."""
    x = np.arange(64, dtype=np.float32).reshape(1, 1, 8, 8)
    fwd = cref.roi_align_avg(x, [[0, 1.0, 1.0, 5.0, 5.0]], (2, 2), 1.0, 2, True)
    rev = cref.roi_align_avg(x, [[0, 5.0, 5.0, 1.0, 1.0]], (2, 2), 1.0, 2, True)
    assert np.allclose(rev, fwd[:, :, ::-1, ::-1])
    far = cref.roi_align_avg(x, [[0, 400.0, 300.0, 50.0, 60.0]], (2, 2), 1.0, 2, True)  # samples beyond H -> 0 (T2)
    assert far.shape == (1, 1, 2, 2) and np.isfinite(far).all()


def test_msda_reference_recipe_double_vs_float():
    """mmcv/tests/test_ops/test_ms_deformable_attn.py:54-70 recipe: seeded inputs, float vs double agreement."""
    torch.manual_seed(3)
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    a = O.msda_pytorch(value.double(), shapes, loc.double(), aw.double())
    b = O.msda_pytorch(value, shapes, loc, aw)
    assert (a - b.double()).abs().max() < 1e-8
