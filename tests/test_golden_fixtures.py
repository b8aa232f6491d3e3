"""Committed golden fixtures (tests/golden/, generator tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP ops reproduce the op fixtures bit-exactly through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import groma_oracle as O
from tests import util

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_ops_golden():
    d = np.load(os.path.join(G, "ops_golden.npz"))
    xyxy = O.center_to_corners_format(torch.from_numpy(d["nms_boxes_cxcywh"])).numpy()
    assert np.array_equal(cref.nms(xyxy, d["nms_scores"], 0.6, 0, 0.0, 100), d["nms_keep_iou06"])
    assert np.array_equal(cref.nms(xyxy, d["nms_scores"], 0.6, 0, 0.15, 100), d["nms_keep_iou06_thr015"])
    assert np.array_equal(cref.roi_align_avg(d["roi_feat_nchw"], d["roi_rois"], (14, 14), 1.0 / 7.0, 2, True), d["roi_out"])


def test_oracle_reproduces_e2e_golden():
    d = np.load(os.path.join(G, "tiny_e2e_golden.npz"))
    cfg, sd, tk = util.tiny_setup(seed=0)
    from groma_amd import synth
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=1234)
    torch.manual_seed(77)
    out = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images)
    assert np.array_equal(out["det"]["topk_idx"].numpy(), d["topk_idx"])
    assert np.array_equal(np.stack([i.numpy() for i in out["nms_inds"]]), d["nms_inds"])
    assert np.array_equal(np.stack([p.numpy() for p in out["perms"]]), d["perms"])
    assert np.array_equal(out["input_ids"].numpy(), d["input_ids"])
    assert np.allclose(np.stack([b.numpy() for b in out["pred_boxes"]]), d["pred_boxes"], atol=1e-6)
    assert np.allclose(out["logits"][:, -1, 32014:32114].numpy(), d["last_logits_region"], atol=2e-4)
    assert abs(out["logits"].abs().mean().item() - float(d["logits_mean_abs"])) < 1e-5


@pytest.mark.gpu
def test_hip_ops_reproduce_ops_golden(dev):
    from groma_amd import ops
    d = np.load(os.path.join(G, "ops_golden.npz"))
    boxes = torch.from_numpy(d["nms_boxes_cxcywh"])[None].to(dev)
    scores = torch.from_numpy(d["nms_scores"])[None].to(dev)
    for thr, key in ((0.0, "nms_keep_iou06"), (0.15, "nms_keep_iou06_thr015")):
        keep, nk = ops.nms(boxes, scores, 0.6, thr, 100)
        assert np.array_equal(keep[0, : nk.item()].cpu().numpy(), d[key])
    feat = torch.from_numpy(d["roi_feat_nchw"])
    N, C, H, W = feat.shape
    nhwc = feat.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)
    rois = torch.from_numpy(d["roi_rois"]).to(dev)
    out = torch.zeros((rois.shape[0], 14, 14, C), dtype=torch.float32, device=dev)
    ops.roi_align_pack(nhwc, rois, out, C=C, H=H, W=W, ph=14, pw=14, spatial_scale=1.0 / 7.0, sampling_ratio=2, pad=0,
                       out_f32=True)
    assert torch.equal(out.permute(0, 3, 1, 2).cpu(), torch.from_numpy(d["roi_out"]))
