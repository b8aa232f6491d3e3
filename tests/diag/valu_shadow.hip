// Diagnostic (not part of the product library): how much VALU work fits in the shadow of a 16x16x32 MFMA stream INSIDE one wave?
// Stream "MFMA, n x filler, MFMA, n x filler, ..." (16 independent accumulators; fillers on registers of their own, no dependence
// on the MFMAs), against the bare MFMA stream and the bare filler stream.  The prefill attention's soft-max is ~290 VALU slots
// (34 v_exp_f32) per 64 MFMAs: if the two pipes overlapped inside a wave, an interleaved loop would cost max(MFMA, VALU), not the sum.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC valu_shadow.hip -o libvalu_shadow.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define SB __builtin_amdgcn_sched_barrier(0);

enum { K_FMA, K_EXP, K_EXP_FMA2, K_MAX3, K_CVT, K_PKMUL, K_COUNT };

template <int KIND, int N>
__device__ __forceinline__ void fill(float (&v)[8], float c) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(c));
    else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
    else if (KIND == K_EXP_FMA2) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[(3 * i) & 7]));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(3 * i + 1) & 7]) : "v"(c));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(3 * i + 2) & 7]) : "v"(c));
    } else if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(c));
    else if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(c));
    else if (KIND == K_PKMUL) {
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      f32x2* pv = (f32x2*)&v[(2 * i) & 6];
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*pv) : "v"(*pv));
    }
  }
}

// MODE 0: MFMA + fillers, 1: MFMA only, 2: fillers only
template <int KIND, int N, int MODE>
__global__ __launch_bounds__(256, 1) void shadow_loop(int iters, float* sink, unsigned long long* clk) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (threadIdx.x * 8 + i) * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    a[i] = (short)((h & 0x807f) | 0x3e00);
    h *= 0x9e3779b1u; h ^= h >> 16;
    b[i] = (short)((h & 0x807f) | 0x3e00);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const float c = 0.5f;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  unsigned long long c0 = 0, w0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE != 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      SB
      if (MODE != 1) fill<KIND, N>(v, c);
      SB
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  float res = 0.f;
  for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) res += v[i];
  if (res == 123.456f) sink[0] = res;
}

template <int KIND, int N>
static void launch3(int mode, int blocks, int iters, float* sink, unsigned long long* clk, hipStream_t s) {
  if (mode == 0) hipLaunchKernelGGL((shadow_loop<KIND, N, 0>), dim3(blocks), dim3(256), 0, s, iters, sink, clk);
  else if (mode == 1) hipLaunchKernelGGL((shadow_loop<KIND, N, 1>), dim3(blocks), dim3(256), 0, s, iters, sink, clk);
  else hipLaunchKernelGGL((shadow_loop<KIND, N, 2>), dim3(blocks), dim3(256), 0, s, iters, sink, clk);
}
template <int KIND>
static void launch2(int n, int mode, int blocks, int iters, float* sink, unsigned long long* clk, hipStream_t s) {
  switch (n) {
    case 1: launch3<KIND, 1>(mode, blocks, iters, sink, clk, s); break;
    case 2: launch3<KIND, 2>(mode, blocks, iters, sink, clk, s); break;
    case 3: launch3<KIND, 3>(mode, blocks, iters, sink, clk, s); break;
    case 4: launch3<KIND, 4>(mode, blocks, iters, sink, clk, s); break;
    case 6: launch3<KIND, 6>(mode, blocks, iters, sink, clk, s); break;
    default: break;
  }
}
// slots (one MFMA + n fillers each) per wave per launch
extern "C" long diag_valu_shadow(int kind, int n, int mode, int blocks, int iters, float* sink, unsigned long long* clk, hipStream_t s) {
  switch (kind) {
    case K_FMA: launch2<K_FMA>(n, mode, blocks, iters, sink, clk, s); break;
    case K_EXP: launch2<K_EXP>(n, mode, blocks, iters, sink, clk, s); break;
    case K_EXP_FMA2: launch2<K_EXP_FMA2>(n, mode, blocks, iters, sink, clk, s); break;
    case K_MAX3: launch2<K_MAX3>(n, mode, blocks, iters, sink, clk, s); break;
    case K_CVT: launch2<K_CVT>(n, mode, blocks, iters, sink, clk, s); break;
    case K_PKMUL: launch2<K_PKMUL>(n, mode, blocks, iters, sink, clk, s); break;
    default: break;
  }
  return (long)iters * 16;
}
