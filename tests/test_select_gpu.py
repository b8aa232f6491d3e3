"""Index / sample-exact parity of the native ops against the C oracle (oracle/roi_nms.c), through the C ABI.
Golden vectors: mmcv/tests/test_ops/test_nms.py:13-29, mmcv/mmcv/ops/nms.py:139-150,
mmcv/tests/test_ops/test_roi_align.py:14-32."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def c2c(b):  # HF center_to_corners_format
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def corners_to_center_exact(xyxy):
    # goldens are corner boxes with small integers: (cx,cy,w,h) reproduces them exactly in fp32
    x1, y1, x2, y2 = xyxy.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)


def test_nms_goldens(dev):
    from groma_amd import ops
    boxes = torch.tensor([[6.0, 3.0, 8.0, 7.0], [3.0, 6.0, 9.0, 11.0], [3.0, 7.0, 10.0, 12.0], [1.0, 4.0, 13.0, 7.0]])
    scores = torch.tensor([0.6, 0.9, 0.7, 0.2])
    keep, nk = ops.nms(corners_to_center_exact(boxes)[None].to(dev), scores[None].to(dev), 0.3, 0.0, 100)
    assert keep[0, :nk.item()].tolist() == [1, 0, 3]
    assert (keep[0, nk.item():] == -1).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("score_thr", [0.0, 0.15])
def test_nms_random_exact(dev, seed, score_thr):
    from groma_amd import ops
    g = torch.Generator().manual_seed(seed)
    B, n = 3, 317
    ctr = torch.rand((B, n, 2), generator=g)
    wh = torch.rand((B, n, 2), generator=g) * 0.3 + 0.01
    boxes = torch.cat([ctr, wh], -1)
    boxes[:, 10] = boxes[:, 3]          # exact duplicates
    scores = torch.rand((B, n), generator=g)
    scores[:, 20] = scores[:, 7]        # score tie -> index order
    n_valid = torch.tensor([n, n - 17, 300], dtype=torch.int32)
    keep, nk = ops.nms(boxes.to(dev), scores.to(dev), 0.6, score_thr, 100, n_valid=n_valid.to(dev))
    for b in range(B):
        nv = n_valid[b].item()
        exp = cref.nms(c2c(boxes[b, :nv]).numpy(), scores[b, :nv].numpy(), 0.6, 0, score_thr, 100)
        got = keep[b, :nk[b].item()].cpu().numpy()
        assert np.array_equal(got, exp), (b, got[:10], exp[:10])


def test_nms_nothing_survives_threshold(dev):
    from groma_amd import ops
    boxes = torch.rand((1, 50, 4)).to(dev)
    scores = torch.full((1, 50), 0.01).to(dev)
    keep, nk = ops.nms(boxes, scores, 0.6, 0.15, 100)
    assert nk.item() == 0 and (keep == -1).all()


GOLD_IN = [([[[[1., 2.], [3., 4.]]]], [[0., 0., 0., 1., 1.]]),
           ([[[[1., 2.], [3., 4.]], [[4., 3.], [2., 1.]]]], [[0., 0., 0., 1., 1.]]),
           ([[[[1., 2., 5., 6.], [3., 4., 7., 8.], [9., 10., 13., 14.], [11., 12., 15., 16.]]]], [[0., 0., 0., 3., 3.]])]
GOLD_OUT = [[[[[1.0, 1.25], [1.5, 1.75]]]],
            [[[[1.0, 1.25], [1.5, 1.75]], [[4.0, 3.75], [3.5, 3.25]]]],
            [[[[1.9375, 4.75], [7.5625, 10.375]]]]]


def _run_roi(dev, x_nchw, rois, ph, scale, sr, out_f32, pad=0):
    from groma_amd import ops
    N, C, H, W = x_nchw.shape
    Cp = (C + 7) // 8 * 8
    feat = torch.zeros((N, H, W, Cp), dtype=torch.bfloat16)
    feat[..., :C] = x_nchw.permute(0, 2, 3, 1).bfloat16()
    R = rois.shape[0]
    out = torch.zeros((R, ph + 2 * pad, ph + 2 * pad, Cp), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    ops.roi_align_pack(feat.to(dev), rois.to(dev), out, C=Cp, H=H, W=W, ph=ph, pw=ph, spatial_scale=scale,
                       sampling_ratio=sr, aligned=True, pad=pad, out_f32=out_f32)
    return out.cpu()


def test_roi_align_goldens(dev):
    for (x, r), exp in zip(GOLD_IN, GOLD_OUT):
        x, r, exp = torch.tensor(x), torch.tensor(r), torch.tensor(exp)
        out = _run_roi(dev, x, r, 2, 1.0, 2, True)
        C = x.shape[1]
        assert torch.equal(out[..., :C].permute(0, 3, 1, 2), exp)


@pytest.mark.parametrize("scale", [1 / 1.75, 1 / 3.5, 1 / 7.0])
def test_roi_align_groma_rois_exact(dev, scale):
    """Groma's ROIs: (cx,cy,w,h)*448 read as x1y1x2y2 -> negative widths, out-of-map samples (T1/T2)."""
    g = torch.Generator().manual_seed(5)
    N, C, H = 2, 16, {1 / 1.75: 128, 1 / 3.5: 64, 1 / 7.0: 32}[scale]
    x = torch.randn((N, C, H, H), generator=g).bfloat16().float()
    R = 40
    cxcywh = torch.cat([torch.rand((R, 2), generator=g), torch.rand((R, 2), generator=g) * 0.6 + 0.02], -1)
    rois = torch.cat([(torch.arange(R) % N).float()[:, None], cxcywh * 448], 1)
    assert (rois[:, 3] < rois[:, 1]).any()  # negative widths present
    exp = torch.from_numpy(cref.roi_align_avg(x.numpy(), rois.numpy(), (14, 14), scale, 2, True))
    out = _run_roi(dev, x, rois, 14, scale, 2, True)
    assert torch.equal(out.permute(0, 3, 1, 2), exp)          # fp32: bit exact
    out16 = _run_roi(dev, x, rois, 14, scale, 2, False, pad=1)
    assert torch.equal(out16[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float(), exp.bfloat16().float())
    assert out16[:, 0].abs().max() == 0 and out16[:, :, -1].abs().max() == 0


def test_roi_align_empty(dev):
    from groma_amd import ops
    feat = torch.zeros((1, 4, 4, 8), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((0, 16, 16, 8), dtype=torch.bfloat16, device=dev)
    ops.roi_align_pack(feat, torch.zeros((0, 5), device=dev), out, C=8, H=4, W=4, ph=14, pw=14, spatial_scale=1.0,
                       sampling_ratio=2)
