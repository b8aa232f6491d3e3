"""The inner boundary under the reference's own signatures (groma_amd/mmcv_ext.py -> gr_nms / gr_roi_align_forward),
driven through ctypes with mmcv's golden vectors and, bit for bit, against the C oracle:
  mmcv/tests/test_ops/test_nms.py:13-29, mmcv/mmcv/ops/nms.py:139-150 (docstring), mmcv/tests/test_ops/test_roi_align.py:14-32.
Also the general (n > 512) NMS path, offset = 1, max pooling, and top-k beyond a 32x32 grid -- the former EINVAL limits."""
import numpy as np
import pytest
import torch

from oracle import cref
from tests.test_oracle_goldens import GOLD_IN, GOLD_OUT

pytestmark = pytest.mark.gpu


def test_nms_goldens_through_reference_signature(dev):
    from groma_amd import mmcv_ext
    boxes = torch.tensor([[6.0, 3.0, 8.0, 7.0], [3.0, 6.0, 9.0, 11.0], [3.0, 7.0, 10.0, 12.0], [1.0, 4.0, 13.0, 7.0]], device=dev)
    scores = torch.tensor([0.6, 0.9, 0.7, 0.2], device=dev)
    inds = mmcv_ext.ext_module.nms(boxes, scores, iou_threshold=0.3, offset=0)
    assert inds.dtype == torch.int64 and inds.tolist() == [1, 0, 3]
    dets, inds2 = mmcv_ext.nms(boxes, scores, iou_threshold=0.3)  # mmcv.ops.nms level: (dets, inds)
    assert inds2.tolist() == [1, 0, 3]
    assert np.allclose(dets.cpu().numpy(), [[3.0, 6.0, 9.0, 11.0, 0.9], [6.0, 3.0, 8.0, 7.0, 0.6], [1.0, 4.0, 13.0, 7.0, 0.2]])
    b7 = torch.tensor([[49.1, 32.4, 51.0, 35.9], [49.3, 32.9, 51.0, 35.3], [49.2, 31.8, 51.0, 35.4], [35.1, 11.5, 39.1, 15.7],
                       [35.6, 11.8, 39.3, 14.2], [35.3, 11.5, 39.9, 14.5], [35.2, 11.7, 39.7, 15.7]], device=dev)
    s7 = torch.tensor([0.9, 0.9, 0.5, 0.5, 0.5, 0.4, 0.3], device=dev)
    dets, inds = mmcv_ext.nms(b7, s7, iou_threshold=0.6)
    assert len(inds) == 3 and inds.tolist() == cref.nms(b7.cpu().numpy(), s7.cpu().numpy(), 0.6).tolist()
    assert mmcv_ext.ext_module.nms(boxes[:0], scores[:0], 0.5, 0).numel() == 0  # cpu/nms.cpp:6-8


@pytest.mark.parametrize("n,offset,thr", [(300, 0, 0.6), (300, 1, 0.5), (512, 0, 0.3), (513, 0, 0.6), (1500, 0, 0.5),
                                          (4096, 1, 0.7), (4096, 0, 0.05)])
def test_nms_bit_exact_vs_c_oracle_any_size(dev, n, offset, thr):
    """n <= 512 runs in one workgroup; 513..4096 takes the three-launch general path: same indices either way"""
    from groma_amd import mmcv_ext
    rng = np.random.default_rng(n + offset)
    scale = 100.0 if offset else 1.0
    xy = rng.random((n, 2)).astype(np.float32) * scale
    wh = (0.03 + 0.3 * rng.random((n, 2)).astype(np.float32)) * scale
    boxes = np.concatenate([xy, xy + wh], 1)
    boxes[n // 2] = boxes[n // 3]  # an exact duplicate box
    scores = rng.random(n).astype(np.float32)
    scores[5:9] = scores[4]  # exact score ties: order must be (score desc, index asc)
    want = cref.nms(boxes, scores, thr, offset)
    got = mmcv_ext.ext_module.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thr, offset)
    assert got.cpu().numpy().tolist() == want.tolist()
    # mmcv.ops.nms glue: score filter + max_num
    dets, inds = mmcv_ext.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thr, offset,
                              score_threshold=0.4, max_num=50)
    assert inds.cpu().numpy().tolist() == cref.nms(boxes, scores, thr, offset, 0.4, 50).tolist()


def test_groma_batched_nms_beyond_512_candidates(dev):
    """300 proposals + 400 refer/ground boxes per image (formerly EINVAL above 512) through the batched cxcywh entry"""
    from groma_amd import ops
    rng = np.random.default_rng(3)
    B, n = 3, 700
    c = rng.random((B, n, 2)).astype(np.float32)
    wh = (0.03 + 0.2 * rng.random((B, n, 2))).astype(np.float32)
    boxes = np.concatenate([c, wh], -1)
    scores = rng.random((B, n)).astype(np.float32)
    n_valid = np.array([700, 650, 513], dtype=np.int32)
    keep, n_keep = ops.nms(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.6, 0.15, 100,
                           n_valid=torch.from_numpy(n_valid).to(dev))
    for b in range(B):
        nv = n_valid[b]
        bx = boxes[b, :nv]
        xyxy = np.stack([bx[:, 0] - 0.5 * bx[:, 2], bx[:, 1] - 0.5 * bx[:, 3], bx[:, 0] + 0.5 * bx[:, 2], bx[:, 1] + 0.5 * bx[:, 3]], 1)
        want = cref.nms(xyxy, scores[b, :nv], 0.6, 0, 0.15, 100)
        k = int(n_keep[b])
        assert keep[b, :k].cpu().numpy().tolist() == want.tolist()
        assert (keep[b, k:] == -1).all()


def test_roi_align_goldens_through_reference_signature(dev):
    from groma_amd import mmcv_ext
    for (x, r), exp in zip(GOLD_IN, GOLD_OUT):
        xt = torch.tensor(x, dtype=torch.float32, device=dev)
        rt = torch.tensor(r, dtype=torch.float32, device=dev)
        out = torch.zeros((rt.shape[0], xt.shape[1], 2, 2), device=dev)
        mmcv_ext.ext_module.roi_align_forward(xt, rt, out, xt.new_zeros(0), xt.new_zeros(0), pooled_height=2, pooled_width=2,
                                              spatial_scale=1.0, sampling_ratio=2, pool_mode=1, aligned=True)
        assert np.array_equal(out.cpu().numpy(), np.array(exp, dtype=np.float32))
        layer = mmcv_ext.RoIAlign((2, 2), 1.0, 2)  # mmcv.ops.RoIAlign level
        assert np.allclose(layer(xt, rt).cpu().numpy(), np.array(exp), atol=1e-3)  # test_roi_align.py:92-95 tolerance


@pytest.mark.parametrize("aligned,sampling_ratio", [(True, 2), (False, 2), (True, 0)])
def test_roi_align_forward_bit_exact_avg_and_max(dev, aligned, sampling_ratio):
    from groma_amd import ops
    rng = np.random.default_rng(11)
    N, C, H, W, K = 2, 24, 32, 40, 37
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.zeros((K, 5), dtype=np.float32)
    rois[:, 0] = rng.integers(0, N, K)
    rois[:, 1:] = rng.random((K, 4)) * 448  # Groma-style: (cx,cy,w,h)*448 read as corners -> negative widths too (T1)
    if sampling_ratio == 0:  # adaptive grid = ceil(roi/pooled): keep ROIs positive and modest
        rois[:, 3:] = rois[:, 1:3] + 20 + rng.random((K, 2)) * 200
    xt, rt = torch.from_numpy(x).to(dev), torch.from_numpy(rois).to(dev)
    out = torch.empty((K, C, 7, 5), device=dev)
    ops.roi_align_forward(xt, rt, out, None, None, 7, 5, 1 / 14.0, sampling_ratio, 1, aligned)
    want = cref.roi_align_avg(x, rois, (7, 5), 1 / 14.0, sampling_ratio, aligned)
    assert np.array_equal(out.cpu().numpy(), want)
    ay, ax = torch.empty_like(out), torch.empty_like(out)
    ops.roi_align_forward(xt, rt, out, ay, ax, 7, 5, 1 / 14.0, sampling_ratio, 0, aligned)
    w_out, w_ay, w_ax = cref.roi_align_max(x, rois, (7, 5), 1 / 14.0, sampling_ratio, aligned)
    assert np.array_equal(out.cpu().numpy(), w_out)
    assert np.array_equal(ay.cpu().numpy(), w_ay) and np.array_equal(ax.cpu().numpy(), w_ax)


@pytest.mark.parametrize("case", ["table_in_passes", "taps_on_the_fly", "degenerate_grid", "many_rois"])
def test_roi_align_forward_table_passes_and_large_grids(dev, case):
    """the compat kernel's three regimes (csrc/roi_align.hip roi_align_planes_kernel): the ROI's sample table built in several passes
    over bin ranges (more samples than the 1024-entry LDS table), taps evaluated on the fly (one bin alone exceeds the table), bins
    without samples (adaptive grid <= 0 on a negative-extent ROI), and the 32-planes-per-workgroup split -- all bit-exact against the
    C oracle, avg and max; a non-finite pixel never leaks through a sample that lies outside the map"""
    from groma_amd import ops
    rng = np.random.default_rng(23)
    N, C, H, W = 2, 40, 48, 64
    K = {"many_rois": 40}.get(case, 9)
    if case == "many_rois":
        C = 44           # (a channel count that is not a multiple of the 8-plane chunk)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.zeros((K, 5), dtype=np.float32)
    rois[:, 0] = rng.integers(0, N, K)
    if case == "table_in_passes":      # grid ~ 8 x 12 = 96 samples x 35 bins > 1024 entries
        sr, scale = 0, 1.0
        rois[:, 1:3] = rng.random((K, 2)) * 8
        rois[:, 3] = rois[:, 1] + 50 + rng.random(K) * 9     # width / 5 bins -> grid_w 10..12
        rois[:, 4] = rois[:, 2] + 44 + rng.random(K) * 4     # height / 7 bins -> grid_h 7
    elif case == "taps_on_the_fly":    # one bin: ceil(300 / 7) x ceil(300 / 5) = 43 x 60 samples > 1024
        sr, scale = 0, 1.0
        rois[:, 1:3] = -100 + rng.random((K, 2)) * 50
        rois[:, 3:] = rois[:, 1:3] + 280 + rng.random((K, 2)) * 20
    elif case == "degenerate_grid":    # negative extents with an adaptive grid: ceil(negative) <= 0 -> no samples at all
        sr, scale = 0, 1.0
        rois[:, 1:3] = 20 + rng.random((K, 2)) * 20
        rois[:, 3:] = rois[:, 1:3] - 1 - rng.random((K, 2)) * 15
    else:
        sr, scale = 2, 1 / 7.0
        rois[:, 1:] = rng.random((K, 4)) * 448
    x[0, :, 0, :] = np.inf                # row 0 of image 0 is non-finite: samples clamped ONTO it see it, samples outside the map must not
    x[1, :, :, W - 1] = np.nan
    xt, rt = torch.from_numpy(x).to(dev), torch.from_numpy(rois).to(dev)
    out = torch.empty((K, C, 7, 5), device=dev)
    ops.roi_align_forward(xt, rt, out, None, None, 7, 5, scale, sr, 1, True)
    want = cref.roi_align_avg(x, rois, (7, 5), scale, sr, True)
    assert np.array_equal(out.cpu().numpy(), want, equal_nan=True)
    ay, ax = torch.empty_like(out), torch.empty_like(out)
    ops.roi_align_forward(xt, rt, out, ay, ax, 7, 5, scale, sr, 0, True)
    w_out, w_ay, w_ax = cref.roi_align_max(x, rois, (7, 5), scale, sr, True)
    assert np.array_equal(out.cpu().numpy(), w_out, equal_nan=True)
    assert np.array_equal(ay.cpu().numpy(), w_ay) and np.array_equal(ax.cpu().numpy(), w_ax)


def test_roi_align_forward_wide_channel_split(dev):
    """K * ceil(C / 32) >= 1024 workgroups: 32 channel planes per workgroup (the other tests run the 8-plane split)"""
    from groma_amd import ops
    rng = np.random.default_rng(29)
    N, C, H, W, K = 1, 256, 16, 16, 130
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.zeros((K, 5), dtype=np.float32)
    rois[:, 1:] = rng.random((K, 4)) * 448
    out = torch.empty((K, C, 14, 14), device=dev)
    ops.roi_align_forward(torch.from_numpy(x).to(dev), torch.from_numpy(rois).to(dev), out, None, None, 14, 14, 1 / 28.0, 2, 1, True)
    assert np.array_equal(out.cpu().numpy(), cref.roi_align_avg(x, rois, (14, 14), 1 / 28.0, 2, True))


def test_nchw_entry_equals_packed_hot_path_entry(dev):
    """gr_roi_align_forward (NCHW f32, reference layout) and gr_roi_align_pack (NHWC bf16 -> f32, the hot-path entry)
    are the same operator on bf16-representable inputs"""
    from groma_amd import ops
    rng = np.random.default_rng(5)
    C, H, W, K = 64, 32, 32, 50
    x = torch.from_numpy(rng.standard_normal((1, C, H, W)).astype(np.float32)).to(dev).bfloat16()
    rois = torch.zeros((K, 5), device=dev)
    rois[:, 1:] = torch.from_numpy(rng.random((K, 4)).astype(np.float32)).to(dev) * 448
    a = torch.empty((K, C, 14, 14), device=dev)
    ops.roi_align_forward(x.float().contiguous(), rois, a, None, None, 14, 14, 1 / 7.0, 2, 1, True)
    b = torch.zeros((K, 14, 14, C), device=dev)
    ops.roi_align_pack(x.permute(0, 2, 3, 1).contiguous(), rois, b, C=C, H=H, W=W, ph=14, pw=14, spatial_scale=1 / 7.0,
                       sampling_ratio=2, aligned=True, pad=0, out_f32=True)
    assert torch.equal(a, b.permute(0, 3, 1, 2))


@pytest.mark.parametrize("S,K", [(1024, 300), (1369, 300), (4096, 900), (37, 5)])
def test_topk_any_grid(dev, S, K):
    from groma_amd import ops
    g = torch.Generator().manual_seed(S)
    x = torch.randn((3, S), generator=g)
    x[0, 7] = x[0, 3]  # tie -> lower index first
    idx = ops.topk_desc(x.to(dev), K)
    want = torch.sort(x, dim=1, descending=True, stable=True)[1][:, :K]
    assert torch.equal(idx.cpu().long(), want)
