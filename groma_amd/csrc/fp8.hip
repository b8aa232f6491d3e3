// OCP fp8 (e4m3fn) row quantisation for the fp8 GEMM path (BASELINE.json configs[4]).
//   q[m,k] = e4m3(x[m,k] / s[m]),  s[m] = max_k |x[m,k]| / 448      (per-row dynamic scale, dequantised in the GEMM epilogue)
// Conversion = v_cvt_pk_fp8_f32 (round-to-nearest-even, OCP encoding on gfx950).  One wave per row, two passes over the
// row (the second is L2-resident).  Fused variant: RMSNorm / LayerNorm -> fp8 in norm_fp8_rows_kernel.
#include "gr_common.h"
#include "../../include/groma_hip.h"

// Single-pass form for 16-bit rows of K <= 12 288 (round 5): the row lives in registers -- every load of the row is issued before the
// first use, the maximum is taken, and the SAME registers are quantised -- instead of a second read (the generic kernel below walks
// the row twice, chunk by chunk).  Same arithmetic, same bytes out.
template <int NCH>  // 512-element chunks per row held per wave (K <= 512 * NCH)
__global__ __launch_bounds__(256) void quant_rows_fp8_reg_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                                 float* __restrict__ scale, int rows, int K, long ldx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (long)row * ldx;
  bf16x8 v[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int k = i * 512 + lane * 8;
    v[i] = k < K ? *(const bf16x8*)(xr + k) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(bf2f((bf16_t)v[i][e])));
  amax = wave_max(amax);
  const float s = fmaxf(amax, 1e-20f) / 448.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int k = i * 512 + lane * 8;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)v[i][e]);
    uint2 o;
    o.x = pack4_fp8(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
    o.y = pack4_fp8(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
    if (k < K) *(uint2*)(q + (long)row * K + k) = o;
  }
}

template <bool F32>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const void* __restrict__ x, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int rows, int K, long ldx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float amax = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    float f[8];
    if (F32) {
      const f32x4 a = *(const f32x4*)((const float*)x + row * ldx + k), b = *(const f32x4*)((const float*)x + row * ldx + k + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    } else {
      const bf16x8 v = *(const bf16x8*)((const bf16_t*)x + row * ldx + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
  }
  amax = wave_max(amax);
  const float s = fmaxf(amax, 1e-20f) / 448.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
  for (int k = lane * 8; k < K; k += 512) {
    float f[8];
    if (F32) {
      const f32x4 a = *(const f32x4*)((const float*)x + row * ldx + k), b = *(const f32x4*)((const float*)x + row * ldx + k + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    } else {
      const bf16x8 v = *(const bf16x8*)((const bf16_t*)x + row * ldx + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)v[e]);
    }
    uint2 o;
    o.x = pack4_fp8(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
    o.y = pack4_fp8(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
    *(uint2*)(q + (long)row * K + k) = o;
  }
}

extern "C" int gr_quant_rows_fp8(const void* x, int x_is_f32, void* q, float* scale, int rows, int K, long ldx,
                                 hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;  // the streaming decode / e4m3 kernels do not exist in the split-operand build (gr_common.h)
  if (!x || !q || !scale || rows <= 0 || K <= 0 || K % 8 != 0 || ldx % 8 != 0) return GR_EINVAL;
#ifndef FP8_QUANT_REG
#define FP8_QUANT_REG 1
#endif
  if (FP8_QUANT_REG && !x_is_f32 && K <= 12288) {
    const dim3 g(gr_cdiv(rows, 4)), b(256);
    const bf16_t* xb = (const bf16_t*)x;
    if (K <= 1024) hipLaunchKernelGGL(quant_rows_fp8_reg_kernel<2>, g, b, 0, stream, xb, (uint8_t*)q, scale, rows, K, ldx);
    else if (K <= 4096) hipLaunchKernelGGL(quant_rows_fp8_reg_kernel<8>, g, b, 0, stream, xb, (uint8_t*)q, scale, rows, K, ldx);
    else if (K <= 11264) hipLaunchKernelGGL(quant_rows_fp8_reg_kernel<22>, g, b, 0, stream, xb, (uint8_t*)q, scale, rows, K, ldx);
    else hipLaunchKernelGGL(quant_rows_fp8_reg_kernel<24>, g, b, 0, stream, xb, (uint8_t*)q, scale, rows, K, ldx);
  } else if (x_is_f32)
    hipLaunchKernelGGL(quant_rows_fp8_kernel<true>, dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, (uint8_t*)q, scale, rows,
                       K, ldx);
  else
    hipLaunchKernelGGL(quant_rows_fp8_kernel<false>, dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, (uint8_t*)q, scale, rows,
                       K, ldx);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// RMSNorm (RMS=true) / LayerNorm of an fp32 row, written as fp8 + per-row scale (the A operand of the next fp8 GEMM)
template <bool RMS, int NV>
__global__ __launch_bounds__(256) void norm_fp8_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, uint8_t* __restrict__ q,
                                                            float* __restrict__ scale, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
    if (RMS) s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    else s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  s = wave_sum(s);
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(s / (float)C + eps);
  } else {
    mean = s / (float)C;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        qq += d * d;
      }
    qq = wave_sum(qq);
    rstd = 1.0f / sqrtf(qq / (float)C + eps);
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    const f32x4 g = *(const f32x4*)(gamma + c);
    if (RMS) v[i] = g * (v[i] * rstd);
    else {
      v[i] = (v[i] - mean) * rstd * g;
      if (beta) v[i] += *(const f32x4*)(beta + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
  }
  amax = wave_max(amax);
  const float sc = fmaxf(amax, 1e-20f) / 448.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    *(uint32_t*)(q + (long)row * C + i * 256 + lane * 4) = pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
}

extern "C" int gr_norm_fp8(const float* x, const float* gamma, const float* beta, void* q, float* scale, int rows, int C,
                           float eps, int rms, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;  // the streaming decode / e4m3 kernels do not exist in the split-operand build (gr_common.h)
  if (!x || !gamma || !q || !scale || rows <= 0 || C % 256 != 0 || C > 4096 || ((C >> 8) & ((C >> 8) - 1))) return GR_EINVAL;
#define LAUNCH_NF(R, NVAL)                                                                                          \
  hipLaunchKernelGGL((norm_fp8_rows_kernel<R, NVAL>), dim3(gr_cdiv(rows, 4)), dim3(256), 0, stream, x, gamma, beta, \
                     (uint8_t*)q, scale, rows, C, eps)
#define SWITCH_NF(R)                      \
  switch (C >> 8) {                       \
    case 1: LAUNCH_NF(R, 1); break;       \
    case 2: LAUNCH_NF(R, 2); break;       \
    case 4: LAUNCH_NF(R, 4); break;       \
    case 8: LAUNCH_NF(R, 8); break;       \
    case 16: LAUNCH_NF(R, 16); break;     \
    default: return GR_EINVAL;            \
  }
  if (rms) { SWITCH_NF(true) } else { SWITCH_NF(false) }
  GR_CHECK_LAUNCH();
  return GR_OK;
}
