// Exact-fp32 GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32) for the Deformable-DETR
// proposer (kept fp32 so top-k / NMS indices are reproducible) and the
// small region-encoder MLPs (groma/model/roi_align.py:254-261).
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias)      A, W, C row-major fp32
// The MFMA result is a k-ordered fmaf chain (cdna_hip_programming.md §3), i.e. plain fp32
// arithmetic.  Tile 64x64x16, 256 threads = 4 waves (2x2), each wave 32x32 = 2x2 MFMA tiles.
// M = 300..4096, N <= 1024, K <= 1024 here: launch-latency class, not roofline class.
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define FBM 64
#define FBN 64
#define FBK 16
#define LDP 17  // padded k-stride (floats) -> conflict-free column reads

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       float* __restrict__ C, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, int M, int N, int K, long lda,
                                                       long ldw, long ldc, int act) {
  __shared__ float as[2][FBM * LDP];
  __shared__ float ws[2][FBN * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * FBM, n0 = blockIdx.x * FBN;
  // staging: thread -> (row = tid/4, 4 floats at k = (tid%4)*4)
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  int am = m0 + srow; if (am > M - 1) am = M - 1;
  int wnr = n0 + srow; if (wnr > N - 1) wnr = N - 1;
  const float* ap = A + (long)am * lda + sk;
  const float* wp = W + (long)wnr * ldw + sk;
  const int nt = K / FBK;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 ra = *(const f32x4*)ap, rw = *(const f32x4*)wp;
  const int fr = lane & 15, fk = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      as[buf][srow * LDP + sk + e] = ra[e];
      ws[buf][srow * LDP + sk + e] = rw[e];
    }
    __syncthreads();
    if (t + 1 < nt) {
      ra = *(const f32x4*)(ap + (long)(t + 1) * FBK);
      rw = *(const f32x4*)(wp + (long)(t + 1) * FBK);
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      float af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = as[buf][(wm * 32 + i * 16 + fr) * LDP + k4 * 4 + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = ws[buf][(wn * 32 + j * 16 + fr) * LDP + k4 * 4 + fk];
      // swapped operands: rows of the MFMA result = n, columns = m  (lane gets 4 consecutive n)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    // next iteration writes the other buffer; the barrier at its top orders those writes after these reads
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + fk * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < N) {
          float v = acc[i][j][e];
          if (bias) v += bias[n + e];
          if (act == 2) v = fmaxf(v, 0.f);
          if (resid) v += resid[(long)m * ldc + n + e];
          C[(long)m * ldc + n + e] = v;
        }
      }
    }
  }
}

extern "C" int gr_gemm_f32(const float* A, const float* W, float* C, const float* bias, const float* resid, int M, int N,
                           int K, long lda, long ldw, long ldc, int act, hipStream_t stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K % FBK != 0) return GR_EINVAL;
  if (lda % 4 != 0 || ldw % 4 != 0) return GR_EINVAL;
  dim3 grid(gr_cdiv(N, FBN), gr_cdiv(M, FBM));
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, A, W, C, bias, resid, M, N, K, lda, ldw, ldc, act);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
