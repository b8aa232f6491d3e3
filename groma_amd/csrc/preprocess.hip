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
