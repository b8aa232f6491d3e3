// Skinny bf16 GEMM for the decode steps of generate(): out[M<=8, N] = x[M,K] . W[N,K]^T.
// HBM-bound: every step streams the 13.2 GB of bf16 LLaMA weights once, shared by all rows of the per-GPU batch
// (algorithmic bytes = N*K*2 per launch).  No MFMA, no LDS: a weight element is used M times and never re-read, so
//   * each wave streams ROWS rows of W for a 512-wide K slice with direct 16-B/lane loads (1 KB per instruction,
//     all ROWS loads issued before the first use: 16 KB in flight per wave, 8 waves per CU);
//   * the matching slice of x (M x 8 values per lane) is loaded once per wave into registers and reused for all rows;
//   * fp32 partial dot products are wave-reduced and written to the split-K workspace [K/512, M, N]; the existing
//     deterministic split-K reduce kernel (fixed summation order -> bit-reproducible) applies the epilogue.
#include "gemm_common.h"

#define GV_ROWS 16   // W rows per wave
#define GV_KS 512    // K slice per block (64 lanes x 8 bf16)

template <int MB>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemmArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // K-slice fastest in the block id: concurrently running blocks sweep whole rows of W (contiguous DRAM pages)
  const int z = blockIdx.x % p.splits;
  const int nb = blockIdx.x / p.splits;
  const int k0 = z * GV_KS + lane * 8;
  const bool kin = k0 < p.K;  // K % 64 == 0, so a lane's 8 values are all in or all out
  const int n_base = (nb * 4 + wave) * GV_ROWS;
  if (n_base >= p.N) return;
  // x slice: MB rows x 8 values
  float xv[MB][8];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p.a_parts) {  // x = merged single-query attention slices (decode.hip), rounded to bf16 like the unsplit kernel
      if (kin && m < p.M) {
        const int hd = p.a_hd, hh = k0 / hd, dd = k0 - hh * hd;
        const float* base = p.a_parts + ((long)(m * (p.K / hd) + hh) * p.a_nsplit) * (hd + 2);
        float mx = -1e30f;
        for (int i = 0; i < p.a_nsplit; ++i) mx = fmaxf(mx, base[i * (hd + 2) + hd]);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, l = 0.f;
        for (int i = 0; i < p.a_nsplit; ++i) {
          const float* bi = base + i * (hd + 2);
          const float f = __expf(bi[hd] - mx);
          l += f * bi[hd + 1];
          const f32x4 o0 = *(const f32x4*)(bi + dd), o1 = *(const f32x4*)(bi + dd + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] += f * o0[e]; o[4 + e] += f * o1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[m][e] = bf2f(f2bf(o[e] / l));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[m][e] = 0.f;
      }
      continue;
    }
    if (kin && m < p.M) t = *(const bf16x8*)(p.A + (long)m * p.lda + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[m][e] = bf2f((bf16_t)t[e]);
  }
  bf16x8 w[GV_ROWS];
#pragma unroll
  for (int r = 0; r < GV_ROWS; ++r) {
    int n = n_base + r;
    if (n > p.N - 1) n = p.N - 1;
    // weights are read exactly once per step by exactly one wave: non-temporal, do not displace x / KV in L2/MALL
    w[r] = kin ? __builtin_nontemporal_load((const bf16x8*)(p.W + (long)n * p.ldw + k0)) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  if constexpr (MB >= 4) {
    // 64 per-lane partials (4 rows x... = 64/MB rows x MB) are summed across the wave with a reduce-scatter butterfly:
    // 63 shuffles per 64 results instead of 6 per result; lane l ends up owning result l.
    constexpr int RPC = 64 / MB;  // rows per 64-value chunk
#pragma unroll
    for (int c = 0; c < GV_ROWS / RPC; ++c) {
      float v[64];
#pragma unroll
      for (int rr = 0; rr < RPC; ++rr) {
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = bf2f((bf16_t)w[c * RPC + rr][e]);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          float a = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) a += wf[e] * xv[m][e];
          v[rr * MB + m] = a;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < off; ++i) {
          const float keep = hi ? v[i + off] : v[i];
          const float send = hi ? v[i] : v[i + off];
          v[i] = keep + __shfl_xor(send, off, 64);
        }
      }
      const int rr = lane / MB, m = lane % MB;
      const int n = n_base + c * RPC + rr;
      if (n < p.N && m < p.M) p.ws[((long)z * p.M + m) * p.N + n] = v[0];
    }
  } else {
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) {
      float wf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[e] = bf2f((bf16_t)w[r][e]);
      float acc[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += wf[e] * xv[m][e];
        acc[m] = wave_sum(a);
      }
      const int n = n_base + r;
      if (lane == 0 && n < p.N) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (m < p.M) p.ws[((long)z * p.M + m) * p.N + n] = acc[m];
      }
    }
  }
}

int gr_launch_gemv(const GemmArgs& p, hipStream_t stream) {
  dim3 grid(gr_cdiv(p.N, 4 * GV_ROWS) * p.splits);
  if (p.M <= 1) hipLaunchKernelGGL(gemv_bf16_kernel<1>, grid, dim3(256), 0, stream, p);
  else if (p.M <= 2) hipLaunchKernelGGL(gemv_bf16_kernel<2>, grid, dim3(256), 0, stream, p);
  else if (p.M <= 4) hipLaunchKernelGGL(gemv_bf16_kernel<4>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gemv_bf16_kernel<8>, grid, dim3(256), 0, stream, p);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
