// Row normalisations for the fp32 residual streams (gfx950).
//  * LayerNorm  (DINOv2 pre-LN eps 1e-6; DDETR post-LN eps 1e-5; region pos-MLP LN;
//                ConvNeXt-style channel LN of input_proj, groma/model/ddetr.py:25-45)
//  * RMSNorm    (LLaMA, HF LlamaRMSNorm: w * x * rsqrt(mean(x^2) + eps), fp32 math)
// One wave per row, float4 accesses, statistics two-pass in registers (matches
// torch's mean / biased variance definition).  HBM-bound: read C*4 B, write C*2|4 B per row.
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define MAXV 16  // up to 16 float4 per lane -> C <= 4096

// NV = float4 per lane (compile-time: keeps the row in exactly NV*4 registers so 6-8 waves/SIMD stay resident and
// the HBM latency of one row is hidden behind the others)
template <bool RMS, int NV>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ out, float* __restrict__ sum_out, int rows, int C,
                                                        long ldx, long ldo, float eps, int out_bf16, int relu_in) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const float* ar = add ? add + (long)row * ldx : nullptr;
  constexpr int nv = NV;  // float4 per lane (C == NV*256)
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nv) {
      v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
      if (ar) v[i] += *(const f32x4*)(ar + i * 256 + lane * 4);
      if (relu_in) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
      }
      if (RMS) s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
      else s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  s = wave_sum(s);
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(s / (float)C + eps);
  } else {
    mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[i][e] - mean;
          q += d * d;
        }
      }
    }
    q = wave_sum(q);
    rstd = 1.0f / sqrtf(q / (float)C + eps);
  }
  if (sum_out) {  // optionally keep the (added) pre-norm row: residual stream update for post-LN blocks
    float* so = sum_out + (long)row * ldx;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i < nv) *(f32x4*)(so + i * 256 + lane * 4) = v[i];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nv) {
      const int c = i * 256 + lane * 4;
      const f32x4 g = *(const f32x4*)(gamma + c);
      f32x4 o;
      if (RMS) {
        o = g * (v[i] * rstd);
      } else {
        o = (v[i] - mean) * rstd * g;
        if (beta) o += *(const f32x4*)(beta + c);
      }
      if (out_bf16) {
        st4f((bf16_t*)out + (long)row * ldo * GR_SPW, c, o);
      } else {
        *(f32x4*)((float*)out + (long)row * ldo + c) = o;
      }
    }
  }
}

// Any C % 4 == 0 that the register-resident kernel does not cover (not a power-of-two multiple of 256, or > 4096: ViT-B's 768,
// LLaMA-13B's 5120 ...): same arithmetic (mean, then biased variance about the mean; fp32), the row is re-read from L2 for
// each pass instead of being held in registers.  One wave per row.
template <bool RMS>
__global__ __launch_bounds__(256) void norm_rows_generic_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                void* __restrict__ out, int rows, int C, long ldx, long ldo,
                                                                float eps, int out_bf16, int relu_in) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  const float* ar = add ? add + (long)row * ldx : nullptr;
  auto load = [&](int c) {
    f32x4 v = *(const f32x4*)(xr + c);
    if (ar) v += *(const f32x4*)(ar + c);
    if (relu_in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    return v;
  };
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = load(c);
    if (RMS) s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    else s += v[0] + v[1] + v[2] + v[3];
  }
  s = wave_sum(s);
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(s / (float)C + eps);
  } else {
    mean = s / (float)C;
    float q = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 v = load(c);
#pragma unroll
      for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
    }
    q = wave_sum(q);
    rstd = 1.0f / sqrtf(q / (float)C + eps);
  }
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = load(c);
    const f32x4 g = *(const f32x4*)(gamma + c);
    f32x4 o;
    if (RMS) {
      o = g * (v * rstd);
    } else {
      o = (v - mean) * rstd * g;
      if (beta) o += *(const f32x4*)(beta + c);
    }
    if (out_bf16) {
      st4f((bf16_t*)out + (long)row * ldo * GR_SPW, c, o);
    } else {
      *(f32x4*)((float*)out + (long)row * ldo + c) = o;
    }
  }
}
static inline bool norm_fast_shape(int C) { return C % 256 == 0 && C <= 256 * MAXV && (((C >> 8) & ((C >> 8) - 1)) == 0); }

extern "C" int gr_layernorm(const float* x, const float* add, const float* gamma, const float* beta, void* out,
                            int rows, int C, long ldx, long ldo, float eps, int out_bf16, int relu_in,
                            hipStream_t stream) {
  if (!x || !gamma || !out || rows <= 0 || C <= 0 || C % 4 != 0 || (ldx & 3) || (ldo & 3)) return GR_EINVAL;
  if (GR_SP && out_bf16 && (ldo % 32 != 0 || C % 32 != 0)) return GR_EINVAL;  // split rows are whole hi / lo block pairs
  if (!norm_fast_shape(C)) {
    if (out == (const void*)x) return GR_EINVAL;  // the generic kernel re-reads the row: no in-place normalisation
    hipLaunchKernelGGL(norm_rows_generic_kernel<false>, dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, add, gamma, beta, out,
                       rows, C, ldx, ldo, eps, out_bf16, relu_in);
    GR_CHECK_LAUNCH();
    return GR_OK;
  }
#define LAUNCH_LN(NVAL)                                                                                            \
  hipLaunchKernelGGL((norm_rows_kernel<false, NVAL>), dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, add, gamma, \
                     beta, out, (float*)nullptr, rows, C, ldx, ldo, eps, out_bf16, relu_in)
  switch (C >> 8) {
    case 1: LAUNCH_LN(1); break;
    case 2: LAUNCH_LN(2); break;
    case 4: LAUNCH_LN(4); break;
    case 8: LAUNCH_LN(8); break;
    case 16: LAUNCH_LN(16); break;
    default: return GR_EINVAL;
  }
  GR_CHECK_LAUNCH();
  return GR_OK;
}

extern "C" int gr_rmsnorm(const float* x, const float* gamma, void* out, int rows, int C, long ldx, long ldo, float eps,
                          int out_bf16, hipStream_t stream) {
  if (!x || !gamma || !out || rows <= 0 || C <= 0 || C % 4 != 0 || (ldx & 3) || (ldo & 3)) return GR_EINVAL;
  if (GR_SP && out_bf16 && (ldo % 32 != 0 || C % 32 != 0)) return GR_EINVAL;  // split rows are whole hi / lo block pairs
  if (!norm_fast_shape(C)) {
    if (out == (const void*)x) return GR_EINVAL;
    hipLaunchKernelGGL(norm_rows_generic_kernel<true>, dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, (const float*)nullptr,
                       gamma, (const float*)nullptr, out, rows, C, ldx, ldo, eps, out_bf16, 0);
    GR_CHECK_LAUNCH();
    return GR_OK;
  }
#define LAUNCH_RMS(NVAL)                                                                                          \
  hipLaunchKernelGGL((norm_rows_kernel<true, NVAL>), dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x,             \
                     (const float*)nullptr, gamma, (const float*)nullptr, out, (float*)nullptr, rows, C, ldx, ldo, eps, \
                     out_bf16, 0)
  switch (C >> 8) {
    case 1: LAUNCH_RMS(1); break;
    case 2: LAUNCH_RMS(2); break;
    case 4: LAUNCH_RMS(4); break;
    case 8: LAUNCH_RMS(8); break;
    case 16: LAUNCH_RMS(16); break;
    default: return GR_EINVAL;
  }
  GR_CHECK_LAUNCH();
  return GR_OK;
}
