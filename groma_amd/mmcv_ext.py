"""The inner drop-in boundary: the two `mmcv._ext` ops Groma's forward reaches, under the reference's own names,
argument lists and error behaviour, backed by libgroma_hip.so.

    mmcv/ops/csrc/pytorch/pybind.cpp:175   Tensor nms(Tensor boxes, Tensor scores, float iou_threshold, int offset)
    mmcv/ops/csrc/pytorch/pybind.cpp:596   void roi_align_forward(Tensor input, Tensor rois, Tensor output,
                                                Tensor argmax_y, Tensor argmax_x, int aligned_height, int aligned_width,
                                                float spatial_scale, int sampling_ratio, int pool_mode, bool aligned)

`ext_module` below is what `mmcv.utils.ext_loader.load_ext('_ext', ['nms', 'roi_align_forward'])` would return
(mmcv/ops/nms.py:11, mmcv/ops/roi_align.py:12); `nms()` / `RoIAlign` restate the thin Python layer mmcv puts on top
(mmcv/ops/nms.py:14-33,118-178; mmcv/ops/roi_align.py:65-107,193-214) so a caller written against `mmcv.ops` runs unchanged:

    from groma_amd.mmcv_ext import nms, RoIAlign           # instead of: from mmcv.ops import nms, RoIAlign

Arithmetic is that of the reference's CPU nms (csrc/pytorch/cpu/nms.cpp:5-54) and CUDA RoIAlign kernel
(csrc/common/cuda/roi_align_cuda_kernel.cuh:17-108), bit for bit (tests/test_mmcv_ext_gpu.py drives both through this
module with mmcv's own golden vectors).  The Groma hot path itself uses the batched / packed entries
(ops.nms, ops.roi_align_pack) that skip the NCHW fp32 round trip; this module is the compatibility surface.
"""
import torch

from . import ops


class _ExtModule:
    @staticmethod
    def nms(boxes, scores, iou_threshold=0.5, offset=0):
        """-> int64[k] indices into `boxes`, by descending score (one device->host read of k, as the reference's CUDA
        op also synchronises: csrc/pytorch/cuda/nms_cuda.cu:27-50)"""
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.int64, device=boxes.device)
        keep, n_keep = ops.nms_xyxy(boxes.contiguous(), scores.contiguous(), iou_threshold, offset)
        return keep[: int(n_keep.item())]

    @staticmethod
    def roi_align_forward(input, rois, output, argmax_y, argmax_x, aligned_height=None, aligned_width=None,
                          spatial_scale=1.0, sampling_ratio=0, pool_mode=1, aligned=True, pooled_height=None,
                          pooled_width=None):
        # mmcv passes pooled_height / pooled_width as keywords (roi_align.py:96-97); pybind names them aligned_*
        ph = aligned_height if aligned_height is not None else pooled_height
        pw = aligned_width if aligned_width is not None else pooled_width
        ops.roi_align_forward(input.contiguous(), rois.contiguous(), output, argmax_y, argmax_x, ph, pw, spatial_scale,
                              sampling_ratio, pool_mode, aligned)


ext_module = _ExtModule()


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    """mmcv.ops.nms (mmcv/ops/nms.py:118-178 around NMSop.forward :14-33) -> (dets [k,5], inds [k])"""
    assert boxes.size(1) == 4
    assert boxes.size(0) == scores.size(0)
    assert offset in (0, 1)
    is_filtering_by_score = score_threshold > 0
    if is_filtering_by_score:
        valid_mask = scores > score_threshold
        boxes, scores = boxes[valid_mask], scores[valid_mask]
        valid_inds = torch.nonzero(valid_mask, as_tuple=False).squeeze(dim=1)
    inds = ext_module.nms(boxes.float(), scores.float(), iou_threshold=float(iou_threshold), offset=offset)
    if max_num > 0:
        inds = inds[:max_num]
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    if is_filtering_by_score:
        inds = valid_inds[inds]
    return dets, inds


class RoIAlign:
    """mmcv.ops.RoIAlign forward (mmcv/ops/roi_align.py:130-214); inference only (no backward on this path)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True):
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio = float(spatial_scale), int(sampling_ratio)
        assert pool_mode in ('max', 'avg')
        self.pool_mode, self.aligned = pool_mode, aligned

    def __call__(self, input, rois):
        assert rois.size(1) == 5, 'RoI must be (idx, x1, y1, x2, y2)!'
        ph, pw = self.output_size
        output = input.new_zeros((rois.size(0), input.size(1), ph, pw), dtype=torch.float32)
        mode = 0 if self.pool_mode == 'max' else 1
        amy = input.new_zeros(output.shape, dtype=torch.float32) if mode == 0 else input.new_zeros(0)
        amx = input.new_zeros(output.shape, dtype=torch.float32) if mode == 0 else input.new_zeros(0)
        ext_module.roi_align_forward(input.float(), rois.float(), output, amy, amx, pooled_height=ph, pooled_width=pw,
                                     spatial_scale=self.spatial_scale, sampling_ratio=self.sampling_ratio, pool_mode=mode,
                                     aligned=self.aligned)
        return output

    forward = __call__
