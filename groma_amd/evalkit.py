"""REC / LVIS evaluation post-processing for batched, image-sharded runs.

The reference's eval scripts (groma/eval/eval_rec.py:89-131, eval_lvis.py:134-168) call `model.generate` one image at
a time, pick the `<r_k>` ids out of the new tokens, look the k-th selected box up in
`outputs.hidden_states[0][-1]['pred_boxes'][0]`, and score it (REC: IoU of the FIRST grounded box against the best-matching
ground-truth box, accuracy at a threshold, mean IoU, 'missing' count, three scalar `reduce`s at the end; LVIS: xywh boxes
scaled to the image size).  Those scripts run unchanged on `groma_amd.groma.GromaModel` (same attributes and outputs).
This module is the same bookkeeping for a *batch* of images per GPU -- what the MI355X path is efficient at -- with one
all-reduce of the three counters over RCCL.  Box maths is a few dozen floats per image: host side, as in the reference.
"""
import torch


def cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def pairwise_iou(a, b):
    """torchvision.ops.box_iou semantics on xyxy boxes"""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def grounded_boxes(new_ids, pred_boxes, box_idx_token_ids):
    """new_ids: generated ids of ONE image; pred_boxes [N,4] cxcywh in the order <r_j> indexes.  Returns the boxes the
    answer points at, in generation order; ids that are not <r_k>, or k >= N, are dropped (eval_rec.py:106-108)."""
    first = box_idx_token_ids[0]
    n_tok = len(box_idx_token_ids)
    ks = [int(t) - first for t in new_ids.tolist()]
    if box_idx_token_ids != list(range(first, first + n_tok)):  # non-contiguous vocabulary: generic lookup
        table = {t: i for i, t in enumerate(box_idx_token_ids)}
        ks = [table.get(int(t), -1) for t in new_ids.tolist()]
    ks = [k for k in ks if 0 <= k < n_tok and k < pred_boxes.shape[0]]
    return pred_boxes[ks] if ks else pred_boxes[:0]


class RecMeter:
    """Running REC counters of eval_rec.py:85-87,110-124; `summary` all-reduces them when a process group is up."""

    def __init__(self, threshold=0.5):
        self.threshold = threshold
        self.m_iou = 0.0
        self.hits = 0.0
        self.invalid = 0.0
        self.count = 0.0

    def update(self, sequences, prompt_len, pred_boxes_list, gt_boxes_list, box_idx_token_ids):
        """sequences [bs, P+new] (GenerateOutput.sequences); pred_boxes_list / gt_boxes_list: per-image cxcywh boxes"""
        for i in range(sequences.shape[0]):
            self.count += 1
            sel = grounded_boxes(sequences[i, prompt_len:].cpu(), pred_boxes_list[i].float().cpu(), box_idx_token_ids)
            if sel.shape[0] == 0:
                self.invalid += 1
                continue
            ious = pairwise_iou(cxcywh_to_xyxy(sel), cxcywh_to_xyxy(gt_boxes_list[i].float().cpu())).max(dim=-1).values
            self.m_iou += float(ious[0])
            self.hits += 1.0 if float(ious[0]) > self.threshold else 0.0

    def summary(self, device=None):
        t = torch.tensor([self.hits, self.m_iou, self.invalid, self.count], dtype=torch.float64)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t)
            t = t.cpu()
        hits, miou, invalid, n = t.tolist()
        n = max(n, 1.0)
        return {f"iou@{self.threshold} accu": hits / n, "m_iou": miou / n, "missing percentage": invalid / n, "count": int(n)}


def lvis_results(sequences, prompt_len, pred_boxes_list, img_ids, labels, img_shapes, box_idx_token_ids, label2cat=None):
    """eval_lvis.py:147-166: every grounded box as an LVIS detection dict, xywh scaled to (h, w) of the image."""
    out = []
    for i in range(sequences.shape[0]):
        sel = grounded_boxes(sequences[i, prompt_len:].cpu(), pred_boxes_list[i].float().cpu(), box_idx_token_ids)
        if sel.shape[0] == 0:
            continue
        h, w = img_shapes[i]
        xywh = torch.stack([(sel[:, 0] - 0.5 * sel[:, 2]) * w, (sel[:, 1] - 0.5 * sel[:, 3]) * h, sel[:, 2] * w, sel[:, 3] * h], -1)
        cat = label2cat[labels[i]] if label2cat is not None else labels[i]
        out += [{"image_id": img_ids[i], "category_id": cat, "bbox": b, "score": 1.0} for b in xywh.tolist()]
    return out
