// Argument block shared by the decode-step weight-streaming kernels (gemv_fused.hip: 16-bit weights; gemv_fp8.hip: e4m3 weights).
#pragma once
#include "gr_common.h"

struct GemvFArgs {
  const bf16_t* W;
  long ldw;
  int M, N, K;
  int x_mode;  // 0: A bf16 [M,K] ; 1: RMSNorm(h) * gamma ; 2: merged attention slices (a_parts)
  const bf16_t* A;
  long lda;
  const float* h;
  long ldh;
  const float* gamma;
  float eps;
  const float* a_parts;
  int a_nsplit, a_hd;
  int epi;  // 0: f32 out ; 1: resid += ; 2: SwiGLU -> bf16 [M, N/2] ; 3: QKV RoPE + cache write
  void* C;
  long ldc;
  float* resid;
  long ldr;
  bf16_t* q;
  bf16_t* kc;
  bf16_t* vt;
  const float* cosT;
  const float* sinT;
  int H, HD, pos0, kv_stride;
  const int* pos_dev;
  int pos_stride;
  const float* w_scale;  // e4m3 stream: per-output-row scale of the weight [N]
};

// the e4m3 stream (gemv_fp8.hip); MB = 4 | 8 batch rows staged
int gr_launch_gemv_fp8(const GemvFArgs& p, int MB, int n_cu, hipStream_t stream);

