// On-device image preprocessing: the step in front of the path.
//   reference: PIL `Image.resize((448, 448))` (default BICUBIC, antialiased, 8-bit fixed point; groma/eval/run_groma.py:79,
//   groma/data/datasets/groma.py:95) followed by the HF image processor's rescale (1/255) + ImageNet normalise.
// Pillow's 8-bit resampler is integer arithmetic: per output pixel a window [xmin, xmin+n) of the input and n
// coefficients in 22-bit fixed point (computed in double on the host, preprocess.py, exactly as Pillow does), a
// horizontal pass that ROUNDS TO uint8, then a vertical pass.  Both passes are restated here with the same integer
// arithmetic, so the resized image is bit-identical to PIL's; rescale+normalise is a 256-entry fp32 table per channel
// built on the host with the processor's own numpy expression, fused into the vertical pass (uint8 HWC -> f32 NCHW).
// HBM-bound and tiny: one read of the source, one 448-wide uint8 intermediate, one f32 write.
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define PIL_PRECISION_BITS 22  // 32 - 8 - 2 (Pillow Resample.c)

__device__ __forceinline__ uint8_t pil_clip8(int v) {
  v >>= PIL_PRECISION_BITS;  // arithmetic shift, as the C code's lookup index
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in u8 [H, Win, 3] -> out u8 [H, Wout, 3];  bounds i32 [Wout, 2] = (xmin, n);  coef i32 [Wout, ksize]
__global__ void resize_h_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int* __restrict__ bounds,
                                   const int* __restrict__ coef, int H, int Win, int Wout, int ksize) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * Wout) return;
  const int xx = (int)(idx % Wout);
  const long y = idx / Wout;
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = coef + (long)xx * ksize;
  const uint8_t* row = in + (y * Win + xmin) * 3;
  int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kk = k[x];
    s0 += row[3 * x] * kk;
    s1 += row[3 * x + 1] * kk;
    s2 += row[3 * x + 2] * kk;
  }
  uint8_t* o = out + idx * 3;
  o[0] = pil_clip8(s0);
  o[1] = pil_clip8(s1);
  o[2] = pil_clip8(s2);
}

// in u8 [Hin, W, 3] -> out_u8 [Hout, W, 3] (optional) and out_f32 [3, Hout, W] = lut[c][value] (optional)
__global__ void resize_v_norm_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out_u8, float* __restrict__ out_f32,
                                     const int* __restrict__ bounds, const int* __restrict__ coef,
                                     const float* __restrict__ lut, int Hin, int Hout, int W, int ksize) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Hout * W) return;
  const int xx = (int)(idx % W);
  const int yy = (int)(idx / W);
  const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = coef + (long)yy * ksize;
  const uint8_t* col = in + ((long)ymin * W + xx) * 3;
  int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int kk = k[y];
    const uint8_t* p = col + (long)y * W * 3;
    s0 += p[0] * kk;
    s1 += p[1] * kk;
    s2 += p[2] * kk;
  }
  const uint8_t v0 = pil_clip8(s0), v1 = pil_clip8(s1), v2 = pil_clip8(s2);
  if (out_u8) {
    uint8_t* o = out_u8 + idx * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
  }
  if (out_f32) {
    const long plane = (long)Hout * W;
    out_f32[idx] = lut[v0];
    out_f32[plane + idx] = lut[256 + v1];
    out_f32[2 * plane + idx] = lut[512 + v2];
  }
}

extern "C" int gr_resize_h_u8(const void* in, void* out, const int* bounds, const int* coef, int H, int Win, int Wout,
                              int ksize, hipStream_t stream) {
  if (!in || !out || !bounds || !coef || H <= 0 || Win <= 0 || Wout <= 0 || ksize <= 0) return GR_EINVAL;
  const long tot = (long)H * Wout;
  hipLaunchKernelGGL(resize_h_u8_kernel, dim3(gr_cdiv(tot, 256)), dim3(256), 0, stream, (const uint8_t*)in, (uint8_t*)out,
                     bounds, coef, H, Win, Wout, ksize);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

extern "C" int gr_resize_v_norm(const void* in, void* out_u8, float* out_f32, const int* bounds, const int* coef,
                                const float* lut, int Hin, int Hout, int W, int ksize, hipStream_t stream) {
  if (!in || (!out_u8 && !out_f32) || (out_f32 && !lut) || !bounds || !coef || Hin <= 0 || Hout <= 0 || W <= 0 || ksize <= 0)
    return GR_EINVAL;
  const long tot = (long)Hout * W;
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3(gr_cdiv(tot, 256)), dim3(256), 0, stream, (const uint8_t*)in,
                     (uint8_t*)out_u8, out_f32, bounds, coef, lut, Hin, Hout, W, ksize);
  GR_CHECK_LAUNCH();
  return GR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// The eval datasets' mmdet pipeline (groma/data/datasets/refcoco_rec.py:38-65): Resize((448,448), keep_ratio=False) =
// cv2.resize INTER_LINEAR on the 8-bit BGR image (mmcv/image/geometric.py:51-101), then Normalize(mean, std, to_rgb) =
// mmcv.imnormalize (mmcv/image/photometric.py:9-45), fused: one thread per output pixel.
// OpenCV's 8-bit bilinear arithmetic (resize.cpp; cv2 itself is not in this image -- see oracle/cv2_pipeline.py):
//   horizontal  D = S[sx] * a0 + S[sx + 1] * a1 (int32, 11-bit coefficients from the host tables),
//   vertical    dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2,
//   exact 2x down-scale in both directions = the INTER_AREA fast path (s00 + s01 + s10 + s11 + 2) >> 2.
// imnormalize with non-integer scalars runs in double and stores float32: y = f32(f64(f32(f64(v) - mean)) * stdinv).
__global__ __launch_bounds__(256) void cv2_resize_norm_kernel(const uint8_t* __restrict__ in, int Hin, int Win,
                                                              const int* __restrict__ xofs, const short* __restrict__ xa,
                                                              const int* __restrict__ yofs, const short* __restrict__ yb,
                                                              uint8_t* __restrict__ out_u8, float* __restrict__ out_f32,
                                                              const double* __restrict__ mean,
                                                              const double* __restrict__ stdinv, int to_rgb, int Hout,
                                                              int Wout, int area2) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Hout * Wout) return;
  const int y = idx / Wout, x = idx - y * Wout;
  int v[3];
  if (area2) {
    const uint8_t* r0 = in + ((long)(2 * y) * Win + 2 * x) * 3;
    const uint8_t* r1 = r0 + (long)Win * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ((int)r0[c] + (int)r0[3 + c] + (int)r1[c] + (int)r1[3 + c] + 2) >> 2;
  } else {
    const int sx = xofs[x], sx1 = min(sx + 1, Win - 1);
    const int a0 = xa[2 * x], a1 = xa[2 * x + 1];
    const int sy = yofs[y];
    const int y0 = min(max(sy, 0), Hin - 1), y1 = min(max(sy + 1, 0), Hin - 1);
    const int b0 = yb[2 * y], b1 = yb[2 * y + 1];
    const uint8_t* p0 = in + (long)y0 * Win * 3;
    const uint8_t* p1 = in + (long)y1 * Win * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int d0 = (int)p0[sx * 3 + c] * a0 + (int)p0[sx1 * 3 + c] * a1;
      const int d1 = (int)p1[sx * 3 + c] * a0 + (int)p1[sx1 * 3 + c] * a1;
      v[c] = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
    }
  }
  if (out_u8) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out_u8[(long)idx * 3 + c] = (uint8_t)v[c];
  }
  if (out_f32) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // c = output channel; to_rgb: cvtColor(BGR2RGB) before the arithmetic
      const int src = to_rgb ? 2 - c : c;
      const float t = (float)((double)v[src] - mean[c]);
      out_f32[(long)c * Hout * Wout + idx] = (float)((double)t * stdinv[c]);
    }
  }
}
extern "C" int gr_cv2_resize_norm(const void* in, int Hin, int Win, const int* xofs, const short* xalpha, const int* yofs,
                                  const short* ybeta, void* out_u8, float* out_f32, const double* mean, const double* stdinv,
                                  int to_rgb, int Hout, int Wout, hipStream_t stream) {
  if (!in || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || (!out_u8 && !out_f32)) return GR_EINVAL;
  if (out_f32 && (!mean || !stdinv)) return GR_EINVAL;
  const int area2 = Hin == 2 * Hout && Win == 2 * Wout;
  if (!area2 && (!xofs || !xalpha || !yofs || !ybeta)) return GR_EINVAL;
  hipLaunchKernelGGL(cv2_resize_norm_kernel, dim3(gr_cdiv((long)Hout * Wout, 256)), dim3(256), 0, stream, (const uint8_t*)in,
                     Hin, Win, xofs, xalpha, yofs, ybeta, (uint8_t*)out_u8, out_f32, mean, stdinv, to_rgb, Hout, Wout, area2);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
