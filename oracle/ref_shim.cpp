// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// C-ABI doorway into the REFERENCE's own CPU ops, compiled from the sources where they lie under
// /root/reference (mmcv/mmcv/ops/csrc/pytorch/{nms,roi_align}.cpp + cpu/{nms,roi_align}.cpp); nothing is copied.
// The dispatcher entry points `nms(...)` / `roi_align_forward(...)` (pytorch/nms.cpp:21, pytorch/roi_align.cpp:25)
// are exactly what mmcv's pybind module binds (pybind.cpp:175,596).
#include <torch/types.h>

#include <cstdint>
#include <cstring>

using at::Tensor;

Tensor nms(Tensor boxes, Tensor scores, float iou_threshold, int offset);
void roi_align_forward(Tensor input, Tensor rois, Tensor output, Tensor argmax_y, Tensor argmax_x, int aligned_height,
                       int aligned_width, float spatial_scale, int sampling_ratio, int pool_mode, bool aligned);

extern "C" int ref_nms(const float* boxes, const float* scores, int n, float iou_threshold, int offset, int64_t* keep) {
  try {
    auto b = at::from_blob(const_cast<float*>(boxes), {n, 4}, at::kFloat).clone();
    auto s = at::from_blob(const_cast<float*>(scores), {n}, at::kFloat).clone();
    Tensor k = nms(b, s, iou_threshold, offset).contiguous();
    std::memcpy(keep, k.data_ptr<int64_t>(), sizeof(int64_t) * k.numel());
    return (int)k.numel();
  } catch (const std::exception&) {
    return -1;
  }
}

// returns 0, or -1 when the reference throws (it asserts on negative-width ROIs: cpu/roi_align.cpp:137-139)
extern "C" int ref_roi_align_avg(const float* input, const float* rois, float* output, int N, int C, int H, int W, int R,
                                 int ph, int pw, float spatial_scale, int sampling_ratio, int aligned) {
  try {
    auto x = at::from_blob(const_cast<float*>(input), {N, C, H, W}, at::kFloat).clone();
    auto r = at::from_blob(const_cast<float*>(rois), {R, 5}, at::kFloat).clone();
    auto out = at::zeros({R, C, ph, pw}, at::kFloat);
    auto ay = at::zeros({0}, at::kFloat), ax = at::zeros({0}, at::kFloat);
    roi_align_forward(x, r, out, ay, ax, ph, pw, spatial_scale, sampling_ratio, /*avg*/ 1, aligned != 0);
    std::memcpy(output, out.data_ptr<float>(), sizeof(float) * out.numel());
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
