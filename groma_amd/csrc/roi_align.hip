// Fused RoIAlign + token-pack for gfx950 (north_star "fused ROIAlign+token-pack").
//
// Semantics: mmcv RoIAlign forward, avg pooling, `aligned` flag, in the arithmetic of the CUDA kernel
// (mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh:17-108 + bilinear_interpolate,
// common_cuda_helper.hpp:28-70): no ROI-size clamp when aligned, samples with y<-1 or y>H contribute 0,
// negative-width ROIs are legal (Groma feeds (cx,cy,w,h)*448 as if it were x1y1x2y2:
// groma/model/roi_align.py:288-293 -- trap T1; strides are 2x off -- T2; both reproduced, not fixed).
// Compiled with -ffp-contract=off: the fp32 sequence equals oracle/roi_nms.c bit for bit.
//
// MI355X design: the feature map is NHWC bf16 (channels contiguous), one workgroup per (roi, bin-row);
// a lane owns 8 channels (16-B loads) of one bin: every bilinear tap is a fully coalesced 16-B/lane read
// of a C*2-byte channel row (the 44 MB/image pyramid stays L2/MALL resident across the ROIs of an image),
// and the result is written straight into the zero-bordered [R, PH+2, PW+2, C] bf16 tile that the
// following per-ROI 3x3 conv consumes as an implicit-GEMM A operand ("pack") -- the reference's
// [3, N, 1024, 14, 14] fp32 roi_feats tensor (241 MB/image) is never materialised.
#include "gr_common.h"
#include "../../include/groma_hip.h"

// fmap = base of the image's map, c0 = first of this lane's 8 channels: element indices stay logical (gr_common.h)
__device__ __forceinline__ void bilinear8(const bf16_t* __restrict__ fmap, long c0, int H, int W, int C, float y, float x,
                                          float* val) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
#pragma unroll
    for (int i = 0; i < 8; ++i) val[i] = 0.f;
    return;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  float v1[8], v2[8], v3[8], v4[8];
  ld8f(fmap, ((long)y_low * W + x_low) * C + c0, v1);
  ld8f(fmap, ((long)y_low * W + x_high) * C + c0, v2);
  ld8f(fmap, ((long)y_high * W + x_low) * C + c0, v3);
  ld8f(fmap, ((long)y_high * W + x_high) * C + c0, v4);
#pragma unroll
  for (int i = 0; i < 8; ++i) val[i] = w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i];
}

__global__ __launch_bounds__(256) void roi_align_pack_kernel(const bf16_t* __restrict__ feat, const float* __restrict__ rois,
                                                             void* __restrict__ out, int R, int C, int H, int W, int PH,
                                                             int PW, float spatial_scale, int sampling_ratio, int aligned,
                                                             int pad, int out_f32, float q_inv) {  // out_f32 == 2: e4m3 bytes of value * q_inv
  const int n = blockIdx.x / PH;
  const int ph = blockIdx.x - n * PH;
  const float* r = rois + (long)n * 5;
  const int batch = (int)r[0];
  const float offset = aligned ? 0.5f : 0.0f;
  const float roi_start_w = r[1] * spatial_scale - offset;
  const float roi_start_h = r[2] * spatial_scale - offset;
  const float roi_end_w = r[3] * spatial_scale - offset;
  const float roi_end_h = r[4] * spatial_scale - offset;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  if (!aligned) {
    roi_width = fmaxf(roi_width, 1.f);
    roi_height = fmaxf(roi_height, 1.f);
  }
  const float bin_size_h = roi_height / (float)PH;
  const float bin_size_w = roi_width / (float)PW;
  const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)PH);
  const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)PW);
  const float count = (float)max(grid_h * grid_w, 1);
  const bf16_t* fmap = feat + (long)batch * H * W * C * GR_SPW;
  const int c8n = C >> 3;
  const int OPH = PH + 2 * pad, OPW = PW + 2 * pad;
  for (int item = threadIdx.x; item < PW * c8n; item += 256) {
    const int pw = item / c8n;
    const int c = (item - pw * c8n) << 3;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
        float val[8];
        bilinear8(fmap, c, H, W, C, y, x, val);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += val[i];
      }
    }
    const long o = (((long)n * OPH + ph + pad) * OPW + pw + pad) * C + c;
    if (out_f32 == 2) {
      float r8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) r8[i] = acc[i] / count;
      st8q((uint8_t*)out, o, r8, q_inv);
    } else if (out_f32) {
      float* dst = (float*)out + o;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = acc[i] / count;
    } else {
      float r8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) r8[i] = acc[i] / count;
      st8f((bf16_t*)out, o, r8);
    }
  }
}

extern "C" int gr_roi_align_pack(const void* feat_nhwc, const float* rois, void* out, int R, int C, int H, int W,
                                 int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                 int pad, int out_f32, hipStream_t stream) {
  if (R < 0 || C <= 0 || C % 8 != 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0) return GR_EINVAL;
  if (GR_SP && C % 32 != 0) return GR_EINVAL;
  if (pad < 0 || pad > 1) return GR_EINVAL;
  if (R == 0) return GR_OK;  // empty ROI set: nothing to do (reference: roi_align.py:300)
  if (!feat_nhwc || !rois || !out) return GR_EINVAL;
  hipLaunchKernelGGL(roi_align_pack_kernel, dim3(R * pooled_h), dim3(256), 0, stream, (const bf16_t*)feat_nhwc, rois, out,
                     R, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, pad, out_f32 ? 1 : 0, 0.f);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
// the same tiles as OCP e4m3 bytes of value * inv_scale (clamped to +-448): the A operand of the e4m3 per-ROI conv
extern "C" int gr_roi_align_pack_fp8(const void* feat_nhwc, const float* rois, void* out, int R, int C, int H, int W,
                                     int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                     int pad, float inv_scale, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;
  if (R < 0 || C <= 0 || C % 8 != 0 || H <= 0 || W <= 0 || pooled_h <= 0 || pooled_w <= 0) return GR_EINVAL;
  if (pad < 0 || pad > 1 || !(inv_scale > 0.f)) return GR_EINVAL;
  if (R == 0) return GR_OK;
  if (!feat_nhwc || !rois || !out) return GR_EINVAL;
  hipLaunchKernelGGL(roi_align_pack_kernel, dim3(R * pooled_h), dim3(256), 0, stream, (const bf16_t*)feat_nhwc, rois, out,
                     R, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, pad, 2, inv_scale);
  GR_CHECK_LAUNCH();
  return GR_OK;
}


// ---- the reference op's own layout: NCHW fp32 in, [K,C,PH,PW] fp32 out, avg or max pooling -------------------------
// (mmcv `_ext.roi_align_forward`, pybind.cpp:596).  One lane per output element with pw fastest, so the 64 lanes of a
// wave read neighbouring taps of one channel plane; same fp32 operation order as roi_align_cuda_kernel.cuh:17-108.
__device__ __forceinline__ float bilinear1(const float* __restrict__ plane, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v1 = plane[y_low * W + x_low], v2 = plane[y_low * W + x_high];
  const float v3 = plane[y_high * W + x_low], v4 = plane[y_high * W + x_high];
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}
__global__ __launch_bounds__(256) void roi_align_nchw_kernel(long total, const float* __restrict__ input,
                                                             const float* __restrict__ rois, float* __restrict__ output,
                                                             float* __restrict__ argmax_y, float* __restrict__ argmax_x,
                                                             int C, int H, int W, int PH, int PW, float spatial_scale,
                                                             int sampling_ratio, int pool_mode, int aligned) {
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < total; index += (long)gridDim.x * 256) {
    const int pw = (int)(index % PW);
    const int ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C);
    const long n = index / PW / PH / C;
    const float* r = rois + n * 5;
    const int batch = (int)r[0];
    const float offset = aligned ? 0.5f : 0.0f;
    const float roi_start_w = r[1] * spatial_scale - offset;
    const float roi_start_h = r[2] * spatial_scale - offset;
    const float roi_end_w = r[3] * spatial_scale - offset;
    const float roi_end_h = r[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
      roi_width = fmaxf(roi_width, 1.f);
      roi_height = fmaxf(roi_height, 1.f);
    }
    const float bin_size_h = roi_height / (float)PH;
    const float bin_size_w = roi_width / (float)PW;
    const float* plane = input + ((long)batch * C + c) * H * W;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)PH);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)PW);
    if (pool_mode == 0) {
      float maxval = -3.402823466e+38f, my = -1.f, mx = -1.f;
      for (int iy = 0; iy < grid_h; ++iy) {
        const float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ++ix) {
          const float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
          const float val = bilinear1(plane, H, W, y, x);
          if (val > maxval) { maxval = val; my = y; mx = x; }
        }
      }
      output[index] = maxval;
      argmax_y[index] = my;
      argmax_x[index] = mx;
    } else {
      const float count = (float)max(grid_h * grid_w, 1);
      float acc = 0.f;
      for (int iy = 0; iy < grid_h; ++iy) {
        const float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ++ix) {
          const float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
          acc += bilinear1(plane, H, W, y, x);
        }
      }
      output[index] = acc / count;
    }
  }
}
extern "C" int gr_roi_align_forward(const float* input, const float* rois, float* output, float* argmax_y, float* argmax_x,
                                    int K, int C, int H, int W, int aligned_height, int aligned_width, float spatial_scale,
                                    int sampling_ratio, int pool_mode, int aligned, hipStream_t stream) {
  if (K < 0 || C <= 0 || H <= 0 || W <= 0 || aligned_height <= 0 || aligned_width <= 0) return GR_EINVAL;
  if (pool_mode != 0 && pool_mode != 1) return GR_EINVAL;
  if (K == 0) return GR_OK;
  if (!input || !rois || !output || (pool_mode == 0 && (!argmax_y || !argmax_x))) return GR_EINVAL;
  const long total = (long)K * C * aligned_height * aligned_width;
  const int blocks = (int)((total + 255) / 256 < 65536L * 16 ? (total + 255) / 256 : 65536L * 16);
  hipLaunchKernelGGL(roi_align_nchw_kernel, dim3(blocks), dim3(256), 0, stream, total, input, rois, output, argmax_y,
                     argmax_x, C, H, W, aligned_height, aligned_width, spatial_scale, sampling_ratio, pool_mode, aligned);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
