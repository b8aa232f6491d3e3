"""Generates the committed golden fixtures (run in the build container: `python tests/golden/make_golden.py`).
  ops_golden.npz       native-op cases computed by the C oracle (itself pinned to the mmcv golden vectors and to the
                       reference's compiled CPU ops): Groma-style NMS and RoIAlign inputs/outputs
  tiny_e2e_golden.npz  index-valued results + logits digest of the CPU oracle's end-to-end forward (tiny architecture,
                       seeds fixed) -- guards the oracle itself against drift
The reference cannot be imported here (mmcv/mmdet/torchvision absent, transformers 5.15 != 4.32: SURVEY.md §8c), so the
vectors come from the oracle, not from the reference package."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cref  # noqa: E402
from oracle import groma_oracle as O  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ops_golden():
    g = torch.Generator().manual_seed(20240924)
    n = 310
    boxes = torch.cat([torch.rand((n, 2), generator=g), torch.rand((n, 2), generator=g) * 0.3 + 0.01], -1)
    scores = torch.rand((n,), generator=g)
    xyxy = O.center_to_corners_format(boxes).numpy()
    keep06 = cref.nms(xyxy, scores.numpy(), 0.6, 0, 0.0, 100)
    keep_thr = cref.nms(xyxy, scores.numpy(), 0.6, 0, 0.15, 100)
    feat = torch.randn((2, 8, 32, 32), generator=g).bfloat16().float()
    R = 24
    cxcywh = torch.cat([torch.rand((R, 2), generator=g), torch.rand((R, 2), generator=g) * 0.6 + 0.02], -1)
    rois = torch.cat([(torch.arange(R) % 2).float()[:, None], cxcywh * 448], 1)
    roi_out = cref.roi_align_avg(feat.numpy(), rois.numpy(), (14, 14), 1.0 / 7.0, 2, True)
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), nms_boxes_cxcywh=boxes.numpy(), nms_scores=scores.numpy(),
                        nms_keep_iou06=keep06, nms_keep_iou06_thr015=keep_thr, roi_feat_nchw=feat.numpy(),
                        roi_rois=rois.numpy(), roi_out=roi_out)


def e2e_golden():
    cfg, sd, tk = util.tiny_setup(seed=0)
    from groma_amd import synth
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=1234)
    torch.manual_seed(77)
    out = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images)
    last = out["logits"][:, -1]
    np.savez_compressed(os.path.join(HERE, "tiny_e2e_golden.npz"),
                        topk_idx=out["det"]["topk_idx"].numpy().astype(np.int32),
                        nms_inds=np.stack([i.numpy() for i in out["nms_inds"]]),
                        perms=np.stack([p.numpy() for p in out["perms"]]),
                        input_ids=out["input_ids"].numpy(),
                        pred_boxes=np.stack([b.numpy() for b in out["pred_boxes"]]),
                        last_logits_top5=last.topk(5, dim=-1).indices.numpy(),
                        last_logits_region=last[:, 32014:32114].numpy(),
                        logits_mean_abs=np.float64(out["logits"].abs().mean().item()))


if __name__ == "__main__":
    ops_golden()
    e2e_golden()
    print("wrote", os.listdir(HERE))
