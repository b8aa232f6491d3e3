// Exact-fp32 GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32) for the Deformable-DETR
// proposer (kept fp32 so top-k / NMS indices are reproducible) and the
// small region-encoder MLPs (groma/model/roi_align.py:254-261).
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ resid)      A, W, C row-major fp32
// fp32 operands, products and sums in float64 (round 6, below), one rounding to fp32 per output element.
//
// Round-2 kernel.  Round 1's 64x64x16 tile read its operands with scalar ds_read_b32 (4 LDS reads per 4 MFMAs) and
// measured 17 TF/s = 11 % of the 157 TF/s f32-MFMA peak -- 11.8 ms of a 138 ms step at 14 images, and NOT hidden: the
// proposer's kernels fill the chip, so they time-share it with the region pyramid on the other stream.  Now:
//  * tile BM x BN x 16 with BM = BN = 128 (4 waves = 2x2, each 64x64 = 4x4 MFMA tiles) when that still yields >= 128
//    blocks, else 64 x 64 (waves 32x32): 64 / 16 MFMAs (32 clk each) per wave and K-step against 8 / 4 LDS reads;
//  * the K index inside a 16-deep step is PERMUTED so that a lane's four operands (one per 16x16x4 MFMA) are contiguous:
//    lane (row fr, k-group fk) owns k = 4*fk .. 4*fk+3 and fetches them with ONE ds_read_b128 (any permutation is legal
//    as long as A and W use the same one -- it only reorders the fp32 sum);
//  * LDS rows are 64 B (16 floats); the 16-B chunk c of row r is stored at c ^ ((r >> 2) & 3): the 16 lanes of a b128
//    read (16 consecutive rows, same chunk) then cover all 64 banks exactly once -- conflict-free without padding;
//  * global -> register prefetch of step t+1 is issued before the MFMAs of step t; double-buffered LDS, one barrier per step;
//  * 16-B epilogue stores (bias, ReLU, residual fused).
#include "gr_common.h"
#include "../../include/groma_hip.h"

//
// Round 6: float64 ACCUMULATION (v_mfma_f64_16x16x4_f64 on the fp32 operands widened in registers, the bias / residual added in
// float64, ONE rounding to fp32 at the store).  Why: the proposer's class logits decide the top-300 ranking against gaps of
// ~1e-5 (profiles/r06_index_survival.txt), and two fp32 evaluations of the same proposer -- this kernel's sequential fmaf chains
// over K and the reference's host / cuBLAS blocking -- differ by about that much; measured against the oracle re-evaluated in
// float64 the device's logits were 1.5x noisier than the oracle's own fp32 (rms 3.8e-6 vs 2.5e-6).  With the dot products exact
// to fp32 rounding the device stops adding its own summation noise: what is left between the two is the reference's.  The fp64 matrix
// pipe runs at half the fp32 one's rate (78 vs 157 TF/s); the proposer is 14.5 GFLOP per image and runs beside the region pyramid.
#define FBK 16
typedef __attribute__((ext_vector_type(4))) double f64x4;

// Which accumulator element of which lane holds which output row of v_mfma_f64_16x16x4_f64 is a property of the instruction; this file
// does not assume it, it ASKS: the first call multiplies A[i][k] = i (k = 0) by B[k][j] = (k == 0) and reads the row index every
// (lane, element) ends up with -- 4 * (lane / 16) + v (the fp32 16x16x4 form) or 4 * v + lane / 16 -- and the epilogue takes the
// answer as `rowmap`.  Anything else fails the call (GR_EINVAL) instead of storing rows in the wrong place.
__global__ void mfma64_probe_kernel(int* out) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
  f64x4 r = {0., 0., 0., 0.}, c = {0., 0., 0., 0.}, kk = {0., 0., 0., 0.};
  r = __builtin_amdgcn_mfma_f64_16x16x4f64(fk == 0 ? (double)fr : 0., fk == 0 ? 1. : 0., r, 0, 0, 0);   // D[i][j] = i
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(fk == 0 ? 1. : 0., fk == 0 ? (double)fr : 0., c, 0, 0, 0);   // D[i][j] = j
  // both operands index k by lane / 16: A[i][k] = 10^k, B[k][j] = k + 1  ->  D = 1 + 20 + 300 + 4000 everywhere
  kk = __builtin_amdgcn_mfma_f64_16x16x4f64(fk == 0 ? 1. : fk == 1 ? 10. : fk == 2 ? 100. : 1000., (double)(fk + 1), kk, 0, 0, 0);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    out[lane * 8 + v] = (int)r[v];
    out[lane * 8 + 4 + v] = kk[v] == 4321. ? (int)c[v] : -1;
  }
}
// -> 0: row = 4 * (lane / 16) + v;  1: row = 4 * v + lane / 16;  -1: neither (or the column is not lane % 16)
static int mfma64_rowmap(hipStream_t stream) {
  static int cached = -2;
  if (cached != -2) return cached;
  int* d = nullptr;
  int h[64 * 8];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return -1;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cs);
  if (cs != hipStreamCaptureStatusNone) { (void)hipFree(d); return -1; }   // (the first call of a process is never inside a capture: warm-ups run eagerly)
  hipLaunchKernelGGL(mfma64_probe_kernel, dim3(1), dim3(64), 0, stream, d);
  const bool ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return -1;
  bool m0 = true, m1 = true, col = true;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      m0 = m0 && h[l * 8 + v] == 4 * (l >> 4) + v;
      m1 = m1 && h[l * 8 + v] == 4 * v + (l >> 4);
      col = col && h[l * 8 + 4 + v] == (l & 15);
    }
  cached = !col ? -1 : m0 ? 0 : m1 ? 1 : -1;
  return cached;
}
extern "C" int gr_diag_mfma64_rowmap(void) { return mfma64_rowmap(nullptr); }

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       float* __restrict__ C, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, int M, int N, int K, long lda,
                                                       long ldw, long ldc, int act, int rowmap) {
  constexpr int TM = BM / 32, TN = BN / 32;         // 16x16 MFMA tiles per wave in m / n
  constexpr int NA = BM * 4 / 256, NW = BN * 4 / 256;  // 16-B chunks staged per thread and step
  __shared__ __attribute__((aligned(16))) float as[2][BM * FBK];
  __shared__ __attribute__((aligned(16))) float ws[2][BN * FBK];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int fr = lane & 15, fk = lane >> 4;

  // staging: chunk q = i*256 + tid -> row q>>2, chunk q&3 (4 consecutive k); rows past the edge re-read the last row
  const float* ap[NA];
  const float* wp[NW];
  int aoff[NA], woff[NW];  // swizzled LDS float offsets
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = i * 256 + tid, r = q >> 2, c = q & 3;
    int m = m0 + r;
    if (m > M - 1) m = M - 1;
    ap[i] = A + (long)m * lda + c * 4;
    aoff[i] = r * FBK + ((c ^ ((r >> 2) & 3)) << 2);
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int q = i * 256 + tid, r = q >> 2, c = q & 3;
    int n = n0 + r;
    if (n > N - 1) n = N - 1;
    wp[i] = W + (long)n * ldw + c * 4;
    woff[i] = r * FBK + ((c ^ ((r >> 2) & 3)) << 2);
  }
  f32x4 ra[NA], rw[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) ra[i] = *(const f32x4*)ap[i];
#pragma unroll
  for (int i = 0; i < NW; ++i) rw[i] = *(const f32x4*)wp[i];

  f64x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f64x4){0., 0., 0., 0.};

  // fragment reads: row R = base + 16*i + fr -> (R >> 2) & 3 == (fr >> 2) & 3 (bases are multiples of 16)
  const int fsw = ((fk ^ ((fr >> 2) & 3)) << 2);
  const int a_lane = (wm * (BM / 2) + fr) * FBK + fsw;
  const int w_lane = (wn * (BN / 2) + fr) * FBK + fsw;

  const int nt = K / FBK;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
#pragma unroll
    for (int i = 0; i < NA; ++i) *(f32x4*)(&as[buf][aoff[i]]) = ra[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) *(f32x4*)(&ws[buf][woff[i]]) = rw[i];
    __syncthreads();
    if (t + 1 < nt) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *(const f32x4*)(ap[i] + (long)(t + 1) * FBK);
#pragma unroll
      for (int i = 0; i < NW; ++i) rw[i] = *(const f32x4*)(wp[i] + (long)(t + 1) * FBK);
    }
    f32x4 af[TM], wf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(&as[buf][a_lane + i * 16 * FBK]);
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[j] = *(const f32x4*)(&ws[buf][w_lane + j * 16 * FBK]);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      double ad[TM], wd[TN];   // fp32 -> fp64 is exact
#pragma unroll
      for (int i = 0; i < TM; ++i) ad[i] = (double)af[i][k4];
#pragma unroll
      for (int j = 0; j < TN; ++j) wd[j] = (double)wf[j][k4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)  // MFMA rows = m, columns = n: the 16 lanes of a row group store 16 consecutive n
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[i], wd[j], acc[i][j], 0, 0, 0);
    }
    // the next iteration writes the other buffer; the barrier at its top orders those writes after these reads
  }
  // epilogue: element v of lane (fr, fk) is output row rowmap(v, fk) of the 16 x 16 tile, column fr; bias, activation and residual
  // in float64 too -- one rounding, at the store
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + fr;
      if (n >= N) continue;
      const double b = bias ? (double)bias[n] : 0.;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = m0 + wm * (BM / 2) + i * 16 + (rowmap ? 4 * v + fk : 4 * fk + v);
        if (m >= M) continue;
        double x = acc[i][j][v] + b;
        if (act == 2) x = x > 0. ? x : 0.;
        if (resid) x += (double)resid[(long)m * ldc + n];
        C[(long)m * ldc + n] = (float)x;
      }
    }
}

static int g_f32_tile = 0;   // 0: chosen per launch; 64 / 128: forced (tests/diag/f32_bench.py)
extern "C" int gr_diag_gemm_f32_tile(int t) {
  if (t != 0 && t != 64 && t != 128) return GR_EINVAL;
  g_f32_tile = t;
  return GR_OK;
}

extern "C" int gr_gemm_f32(const float* A, const float* W, float* C, const float* bias, const float* resid, int M, int N,
                           int K, long lda, long ldw, long ldc, int act, hipStream_t stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K % FBK != 0) return GR_EINVAL;
  if (lda % 4 != 0 || ldw % 4 != 0) return GR_EINVAL;
  if ((((uintptr_t)A) & 15) != 0 || (((uintptr_t)W) & 15) != 0) return GR_EINVAL;
  const int rowmap = mfma64_rowmap(stream);
  if (rowmap < 0) return GR_EINVAL;
  const long big = (long)gr_cdiv(M, 128) * gr_cdiv(N, 128);
  // Round 6: with float64 accumulation the 64 x 64 tile (100 registers, several waves per SIMD) beats the 128 x 128 one (316 registers, one
  // wave per SIMD) on EVERY proposer shape -- 4.70 against 5.69 ms of GEMM per 14-image step, 14336 x 1024 x 256: 140 vs 203 us
  // (profiles/r06_f64_gemm_tiles.txt); the large tile stays reachable for measurements only.
  (void)big;
  if (g_f32_tile == 128) {
    dim3 grid(gr_cdiv(N, 128), gr_cdiv(M, 128));
    hipLaunchKernelGGL((gemm_f32_kernel<128, 128>), grid, dim3(256), 0, stream, A, W, C, bias, resid, M, N, K, lda, ldw, ldc, act, rowmap);
  } else {
    dim3 grid(gr_cdiv(N, 64), gr_cdiv(M, 64));
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), grid, dim3(256), 0, stream, A, W, C, bias, resid, M, N, K, lda, ldw, ldc, act, rowmap);
  }
  GR_CHECK_LAUNCH();
  return GR_OK;
}
