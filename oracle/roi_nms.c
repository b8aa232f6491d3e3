/*
 * oracle/roi_nms.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 * Plain-C restatement of the two native mmcv ops on Groma's hot path.  Only tests/, bench.py's cpu_baseline
 * leg and __graft_entry__.smoke() may load this; the product (groma_amd/) never does.
 *
 *  oracle_nms            follows  mmcv/mmcv/ops/nms.py:14-33 (NMSop.forward: score filter, max_num, index map)
 *                                 mmcv/mmcv/ops/csrc/pytorch/cpu/nms.cpp:5-54 (nms_cpu: areas, descending sort,
 *                                 greedy suppression with  inter/(a+b-inter) > thr )
 *                        sort tie rule: the reference calls an unstable sort; we fix (score desc, index asc).
 *  oracle_roi_align_avg  follows  mmcv/mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh:17-108 (avg branch)
 *                                 mmcv/mmcv/ops/csrc/common/cuda/common_cuda_helper.hpp:28-70 (bilinear_interpolate)
 *                        i.e. the CUDA arithmetic, because the CPU implementation asserts on the negative-width
 *                        ROIs Groma produces (mmcv/mmcv/ops/csrc/pytorch/cpu/roi_align.cpp:137-139).
 *
 * Pinned against the reference's own golden vectors: mmcv/tests/test_ops/test_nms.py:13-29,
 * mmcv/mmcv/ops/nms.py:139-150 (docstring example), mmcv/tests/test_ops/test_roi_align.py:14-32
 * -- see tests/test_oracle_goldens.py.  Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static const float* g_scores;
static int cmp_desc(const void* a, const void* b) {
  const int64_t ia = *(const int64_t*)a, ib = *(const int64_t*)b;
  const float sa = g_scores[ia], sb = g_scores[ib];
  if (sa > sb) return -1;
  if (sa < sb) return 1;
  return ia < ib ? -1 : (ia > ib ? 1 : 0);
}

/* boxes: [n,4] x1,y1,x2,y2; returns number kept, indices (into the n inputs) in keep[] */
int oracle_nms(const float* boxes, const float* scores, int n, float iou_threshold, int offset, float score_threshold,
               int max_num, int64_t* keep) {
  if (n <= 0) return 0;
  int64_t* valid = (int64_t*)malloc(sizeof(int64_t) * n);
  int nv = 0;
  /* mmcv/ops/nms.py:21-26 */
  for (int i = 0; i < n; ++i) {
    if (score_threshold > 0) {
      if (scores[i] > score_threshold) valid[nv++] = i;
    } else {
      valid[nv++] = i;
    }
  }
  /* cpu/nms.cpp:16: order = scores.sort(descending) */
  g_scores = scores;
  qsort(valid, nv, sizeof(int64_t), cmp_desc);
  float* area = (float*)malloc(sizeof(float) * (nv > 0 ? nv : 1));
  unsigned char* select = (unsigned char*)malloc(nv > 0 ? nv : 1);
  for (int i = 0; i < nv; ++i) {
    const float* b = boxes + valid[i] * 4;
    area[i] = (b[2] - b[0] + offset) * (b[3] - b[1] + offset); /* cpu/nms.cpp:14 */
    select[i] = 1;
  }
  for (int _i = 0; _i < nv; ++_i) { /* cpu/nms.cpp:28-53 */
    if (!select[_i]) continue;
    const float* bi = boxes + valid[_i] * 4;
    const float iarea = area[_i];
    for (int _j = _i + 1; _j < nv; ++_j) {
      if (!select[_j]) continue;
      const float* bj = boxes + valid[_j] * 4;
      const float xx1 = fmaxf(bi[0], bj[0]);
      const float yy1 = fmaxf(bi[1], bj[1]);
      const float xx2 = fminf(bi[2], bj[2]);
      const float yy2 = fminf(bi[3], bj[3]);
      const float w = fmaxf(0.f, xx2 - xx1 + offset);
      const float h = fmaxf(0.f, yy2 - yy1 + offset);
      const float inter = w * h;
      const float ovr = inter / (iarea + area[_j] - inter);
      if (ovr > iou_threshold) select[_j] = 0;
    }
  }
  int k = 0;
  for (int i = 0; i < nv; ++i) {
    if (!select[i]) continue;
    if (max_num > 0 && k >= max_num) break; /* nms.py:29-30 */
    keep[k++] = valid[i];                   /* nms.py:31-32: valid_inds[inds] */
  }
  free(valid);
  free(area);
  free(select);
  return k;
}

static float bilinear_interpolate(const float* input, int height, int width, float y, float x) {
  if (y < -1.0f || y > height || x < -1.0f || x > width) return 0; /* helper.hpp:33 */
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v1 = input[y_low * width + x_low], v2 = input[y_low * width + x_high];
  const float v3 = input[y_high * width + x_low], v4 = input[y_high * width + x_high];
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* input NCHW f32 [N,C,H,W]; rois [R,5] (batch, x1,y1,x2,y2); output [R,C,PH,PW] */
void oracle_roi_align_avg(const float* input, const float* rois, float* output, int R, int channels, int height,
                          int width, int pooled_height, int pooled_width, float spatial_scale, int sampling_ratio,
                          int aligned) {
  const long nthreads = (long)R * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; ++index) {
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int c = (index / pooled_width / pooled_height) % channels;
    const int n = index / pooled_width / pooled_height / channels;
    const float* offset_rois = rois + n * 5;
    const int roi_batch_ind = (int)offset_rois[0];
    const float offset = aligned ? 0.5f : 0.0f;
    const float roi_start_w = offset_rois[1] * spatial_scale - offset;
    const float roi_start_h = offset_rois[2] * spatial_scale - offset;
    const float roi_end_w = offset_rois[3] * spatial_scale - offset;
    const float roi_end_h = offset_rois[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
      roi_width = fmaxf(roi_width, 1.f);
      roi_height = fmaxf(roi_height, 1.f);
    }
    const float bin_size_h = roi_height / (float)pooled_height;
    const float bin_size_w = roi_width / (float)pooled_width;
    const float* offset_input = input + ((long)roi_batch_ind * channels + c) * height * width;
    const int roi_bin_grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_height);
    const int roi_bin_grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_width);
    const int cnt = roi_bin_grid_h * roi_bin_grid_w;
    const float count = (float)(cnt > 1 ? cnt : 1);
    float output_val = 0.f;
    for (int iy = 0; iy < roi_bin_grid_h; iy++) {
      const float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)roi_bin_grid_h;
      for (int ix = 0; ix < roi_bin_grid_w; ix++) {
        const float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)roi_bin_grid_w;
        output_val += bilinear_interpolate(offset_input, height, width, y, x);
      }
    }
    output[index] = output_val / count;
  }
}

/* max pooling branch of the same kernel (roi_align_cuda_kernel.cuh:66-87): output = max over the bin's samples,
 * argmax_y / argmax_x = the sampling coordinates of the first maximum (-1 when the bin has no sample above -FLT_MAX) */
void oracle_roi_align_max(const float* input, const float* rois, float* output, float* argmax_y, float* argmax_x, int R,
                          int channels, int height, int width, int pooled_height, int pooled_width, float spatial_scale,
                          int sampling_ratio, int aligned) {
  const long nthreads = (long)R * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; ++index) {
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int c = (index / pooled_width / pooled_height) % channels;
    const int n = index / pooled_width / pooled_height / channels;
    const float* offset_rois = rois + n * 5;
    const int roi_batch_ind = (int)offset_rois[0];
    const float offset = aligned ? 0.5f : 0.0f;
    const float roi_start_w = offset_rois[1] * spatial_scale - offset;
    const float roi_start_h = offset_rois[2] * spatial_scale - offset;
    const float roi_end_w = offset_rois[3] * spatial_scale - offset;
    const float roi_end_h = offset_rois[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
      roi_width = fmaxf(roi_width, 1.f);
      roi_height = fmaxf(roi_height, 1.f);
    }
    const float bin_size_h = roi_height / (float)pooled_height;
    const float bin_size_w = roi_width / (float)pooled_width;
    const float* offset_input = input + ((long)roi_batch_ind * channels + c) * height * width;
    const int roi_bin_grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_height);
    const int roi_bin_grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_width);
    float maxval = -3.402823466e+38f, maxidx_y = -1.f, maxidx_x = -1.f;
    for (int iy = 0; iy < roi_bin_grid_h; iy++) {
      const float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)roi_bin_grid_h;
      for (int ix = 0; ix < roi_bin_grid_w; ix++) {
        const float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)roi_bin_grid_w;
        const float val = bilinear_interpolate(offset_input, height, width, y, x);
        if (val > maxval) { maxval = val; maxidx_y = y; maxidx_x = x; }
      }
    }
    output[index] = maxval;
    argmax_y[index] = maxidx_y;
    argmax_x[index] = maxidx_x;
  }
}
