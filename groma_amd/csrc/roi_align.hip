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

// ---- the operator's geometry, shared by both kernels below ----------------------------------------------------------
// The VALUES are forced by bit-exactness with the reference (`roi_start = x1 * scale - 0.5`, no size clamp when aligned, the sample
// position `start + p * bin + (i + .5) * bin / grid`, the border rule of the bilinear taps); how they are evaluated is this file's:
// once per ROI (RoiGeom), once per sample (Tap), and never per channel.
struct RoiGeom {
  float start_w, start_h, bin_w, bin_h, count;
  int grid_h, grid_w, batch;
};
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ r, float spatial_scale, int PH, int PW, int sampling_ratio,
                                            int aligned) {
  RoiGeom g;
  g.batch = (int)r[0];
  const float off = aligned ? 0.5f : 0.0f;
  g.start_w = r[1] * spatial_scale - off;
  g.start_h = r[2] * spatial_scale - off;
  float w = (r[3] * spatial_scale - off) - g.start_w, h = (r[4] * spatial_scale - off) - g.start_h;
  if (!aligned) { w = fmaxf(w, 1.f); h = fmaxf(h, 1.f); }   // (aligned: a negative extent stays negative -- trap T1)
  g.bin_h = h / (float)PH;
  g.bin_w = w / (float)PW;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(h / (float)PH);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(w / (float)PW);
  g.count = (float)max(g.grid_h * g.grid_w, 1);
  return g;
}
// position of sample i (of `grid`) of bin p along one axis
__device__ __forceinline__ float sample_pos(float start, int p, float bin, int i, int grid) {
  return start + p * bin + ((float)i + .5f) * bin / (float)grid;
}
// One bilinear sample as four (element offset inside an H x W plane, weight) pairs; o[0] < 0: the sample lies outside the map and
// contributes exactly 0 (the map is then not read: a non-finite pixel must not leak through 0 * inf).
struct Tap {
  int o[4];
  float w[4];
};
__device__ __forceinline__ Tap make_tap(int H, int W, float y, float x) {
  Tap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.o[0] = t.o[1] = t.o[2] = t.o[3] = -1;
    t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
    return t;
  }
  y = y <= 0.f ? 0.f : y;
  x = x <= 0.f ? 0.f : x;
  int y0 = (int)y, x0 = (int)x, y1 = y0 + 1, x1 = x0 + 1;
  if (y0 >= H - 1) { y0 = y1 = H - 1; y = (float)y0; }
  if (x0 >= W - 1) { x0 = x1 = W - 1; x = (float)x0; }
  const float ly = y - y0, lx = x - x0, hy = 1.f - ly, hx = 1.f - lx;
  t.o[0] = y0 * W + x0; t.o[1] = y0 * W + x1; t.o[2] = y1 * W + x0; t.o[3] = y1 * W + x1;
  t.w[0] = hy * hx; t.w[1] = hy * lx; t.w[2] = ly * hx; t.w[3] = ly * lx;
  return t;
}

__global__ __launch_bounds__(256) void roi_align_pack_kernel(const bf16_t* __restrict__ feat, const float* __restrict__ rois,
                                                             void* __restrict__ out, int R, int C, int H, int W, int PH,
                                                             int PW, float spatial_scale, int sampling_ratio, int aligned,
                                                             int pad, int out_f32, float q_inv) {  // out_f32 == 2: e4m3 bytes of value * q_inv
  const int n = blockIdx.x / PH;
  const int ph = blockIdx.x - n * PH;
  const RoiGeom g = roi_geom(rois + (long)n * 5, spatial_scale, PH, PW, sampling_ratio, aligned);
  const bf16_t* fmap = feat + (long)g.batch * H * W * C * GR_SPW;   // element indices stay logical (gr_common.h)
  const int c8n = C >> 3;
  const int OPH = PH + 2 * pad, OPW = PW + 2 * pad;
  for (int item = threadIdx.x; item < PW * c8n; item += 256) {   // a lane = 8 channels of one bin: 16-B taps, coalesced NHWC rows
    const int pw = item / c8n;
    const long c = (item - pw * c8n) << 3;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int iy = 0; iy < g.grid_h; ++iy) {
      const float y = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
      for (int ix = 0; ix < g.grid_w; ++ix) {
        const Tap t = make_tap(H, W, y, sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w));
        if (t.o[0] < 0) continue;   // (+0 is the additive identity of an accumulator that starts at +0 and can never reach -0)
        float v0[8], v1[8], v2[8], v3[8];
        ld8f(fmap, (long)t.o[0] * C + c, v0);
        ld8f(fmap, (long)t.o[1] * C + c, v1);
        ld8f(fmap, (long)t.o[2] * C + c, v2);
        ld8f(fmap, (long)t.o[3] * C + c, v3);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += t.w[0] * v0[i] + t.w[1] * v1[i] + t.w[2] * v2[i] + t.w[3] * v3[i];
      }
    }
    float r8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = acc[i] / g.count;
    const long o = (((long)n * OPH + ph + pad) * OPW + pw + pad) * C + c;
    if (out_f32 == 2) {
      st8q((uint8_t*)out, o, r8, q_inv);
    } else if (out_f32) {
      float* dst = (float*)out + o;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = r8[i];
    } else {
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
// (mmcv `_ext.roi_align_forward`, pybind.cpp:596; the boundary's compat entry, not on Groma's hot path.)
// Sample table + channel sweep: a workgroup owns (one ROI, a chunk of channel planes).  The ROI's samples -- PH*PW bins x grid_h*grid_w
// samples, each four (offset, weight) taps -- depend on the ROI alone, so the workgroup evaluates them ONCE into an LDS table
// (sample-major: the lanes of a wave read consecutive entries) and then sweeps its channel planes with lanes along the bin index:
// per (bin, channel) a lane walks the bin's table column and gathers from one L2-resident plane; the [K,C,PH,PW] stores of a wave
// are contiguous.  The coordinate arithmetic runs once per ROI-sample instead of once per (ROI-sample, channel).  Tables larger than
// the LDS budget are built in passes over bin ranges; a single bin with more samples than the table holds (adaptive grids on huge
// ROIs) evaluates its taps on the fly with the same make_tap.
#define RA_TABLE 1024   // table entries (40 B each: int4 offsets, float4 weights, float2 sample position for the arg-max outputs)
template <bool MAXPOOL>
__device__ __forceinline__ void ra_fold(const Tap& t, const float* __restrict__ plane, float y, float x, float& acc, float& my, float& mx) {
  const float val = t.o[0] < 0 ? 0.f : t.w[0] * plane[t.o[0]] + t.w[1] * plane[t.o[1]] + t.w[2] * plane[t.o[2]] + t.w[3] * plane[t.o[3]];
  if (MAXPOOL) {
    if (val > acc) { acc = val; my = y; mx = x; }
  } else {
    acc += val;
  }
}
template <bool MAXPOOL>
__global__ __launch_bounds__(256) void roi_align_planes_kernel(const float* __restrict__ input, const float* __restrict__ rois,
                                                               float* __restrict__ output, float* __restrict__ argmax_y,
                                                               float* __restrict__ argmax_x, int C, int H, int W, int PH, int PW,
                                                               float spatial_scale, int sampling_ratio, int aligned, int CH, int chunks) {
  __shared__ int4 t_off[RA_TABLE];
  __shared__ float4 t_w[RA_TABLE];
  __shared__ float2 t_yx[MAXPOOL ? RA_TABLE : 1];
  const int n = blockIdx.x / chunks, c0 = (blockIdx.x - n * chunks) * CH, nc = min(CH, C - c0);
  const RoiGeom g = roi_geom(rois + (long)n * 5, spatial_scale, PH, PW, sampling_ratio, aligned);
  const int NB = PH * PW;
  const int S = (g.grid_h > 0 && g.grid_w > 0) ? g.grid_h * g.grid_w : 0;   // 0: bins without samples (avg: 0 / count; max: the -FLT_MAX / -1 sentinels)
  const float* img = input + ((long)g.batch * C + c0) * H * W;
  float* out_n = output + ((long)n * C + c0) * NB;
  const bool tabled = S > 0 && S <= RA_TABLE;
  const int nbp = tabled ? min(NB, RA_TABLE / S) : NB;   // bins per pass
  for (int b0 = 0; b0 < NB; b0 += nbp) {
    const int nb = min(nbp, NB - b0);
    if (tabled) {
      __syncthreads();   // (the previous pass has been swept)
      for (int e = threadIdx.x; e < nb * S; e += 256) {   // entry e = s * nb + bin: sample-major
        const int s = e / nb, bin = b0 + e - s * nb;
        const int ph = bin / PW, pw = bin - ph * PW, iy = s / g.grid_w, ix = s - iy * g.grid_w;
        const float y = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h), x = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
        const Tap t = make_tap(H, W, y, x);
        t_off[e] = make_int4(t.o[0], t.o[1], t.o[2], t.o[3]);
        t_w[e] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
        if (MAXPOOL) t_yx[e] = make_float2(y, x);
      }
      __syncthreads();
    }
    for (int o = threadIdx.x; o < nc * nb; o += 256) {   // lanes along the bins of a plane, then the next plane
      const int c = o / nb, bl = o - c * nb, bin = b0 + bl;
      const float* plane = img + (long)c * H * W;
      float acc = MAXPOOL ? -3.402823466e+38f : 0.f, my = -1.f, mx = -1.f;
      if (tabled) {
        for (int s = 0; s < S; ++s) {
          const int4 of = t_off[s * nb + bl];
          const float4 wt = t_w[s * nb + bl];
          Tap t;
          t.o[0] = of.x; t.o[1] = of.y; t.o[2] = of.z; t.o[3] = of.w;
          t.w[0] = wt.x; t.w[1] = wt.y; t.w[2] = wt.z; t.w[3] = wt.w;
          const float2 yx = MAXPOOL ? t_yx[s * nb + bl] : make_float2(0.f, 0.f);
          ra_fold<MAXPOOL>(t, plane, yx.x, yx.y, acc, my, mx);
        }
      } else {
        const int ph = bin / PW, pw = bin - ph * PW;
        for (int iy = 0; iy < g.grid_h; ++iy) {
          const float y = sample_pos(g.start_h, ph, g.bin_h, iy, g.grid_h);
          for (int ix = 0; ix < g.grid_w; ++ix) {
            const float x = sample_pos(g.start_w, pw, g.bin_w, ix, g.grid_w);
            ra_fold<MAXPOOL>(make_tap(H, W, y, x), plane, y, x, acc, my, mx);
          }
        }
      }
      const long oi = (long)c * NB + bin;
      if (MAXPOOL) {
        out_n[oi] = acc;
        argmax_y[((long)n * C + c0) * NB + oi] = my;
        argmax_x[((long)n * C + c0) * NB + oi] = mx;
      } else {
        out_n[oi] = acc / g.count;
      }
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
  // channel planes per workgroup: 32 (the table is built once per 32 planes) unless that leaves the chip short of workgroups
  const int CH = (long)K * ((C + 31) / 32) >= 1024 ? 32 : 8;
  const int chunks = (C + CH - 1) / CH;
  if ((long)K * chunks > 0x7fffffffL) return GR_EINVAL;
  if (pool_mode == 0)
    hipLaunchKernelGGL(roi_align_planes_kernel<true>, dim3(K * chunks), dim3(256), 0, stream, input, rois, output, argmax_y, argmax_x,
                       C, H, W, aligned_height, aligned_width, spatial_scale, sampling_ratio, aligned, CH, chunks);
  else
    hipLaunchKernelGGL(roi_align_planes_kernel<false>, dim3(K * chunks), dim3(256), 0, stream, input, rois, output, argmax_y, argmax_x,
                       C, H, W, aligned_height, aligned_width, spatial_scale, sampling_ratio, aligned, CH, chunks);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
