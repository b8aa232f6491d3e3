// Skinny GEMM for decode steps of 9..64 rows (round 6; SURVEY 8f rank 1: continuous batching past the 8-row weight streams).
//
//   y[M <= 64, N] = epilogue(x[M, K] . W[N, K]^T)        16-bit operands, fp32 accumulate        (gr_gemm_desc.tile == 3)
//
// A decode step reads every weight once (13.2 GB per token at Groma-7B) whatever the number of rows, so rows are nearly free until
// the matrix unit or the L2 saturates: one MFMA 16x16x32 retires 1 KB of W against 16 batch rows, i.e. the 256 CUs can retire
// ~130 TB/s of weights at 16 rows and ~33 TB/s at 64 -- HBM (8 TB/s) stays the bound.  The 8-row streams (gemv_fused.hip) do their
// products on the VALU (v_dot2c), which is what caps them at 8 rows; the prefill kernels tile 128 / 256 rows of x and would stream
// W through a handful of workgroups.  This kernel is the weight stream with the WEIGHTS as the MFMA A operand:
//   * a workgroup owns 64 rows of W (wave w: rows 16 w .. 16 w + 15) and walks K in slices of 256 (K % 32 == 0; a partial last slice multiplies zeros); a lane (fr = lane & 15,
//     fg = lane >> 4) holds, per 32-deep k-block, the 16-B chunk fg of row fr -- the A fragment as it comes out of the load, three
//     slices (24 KB per wave) in flight, non-temporal;
//   * the x slice [16 NB rows x 256 k] is shared by the four waves: staged global -> registers -> LDS two slices ahead, into an
//     XOR-swizzled image (chunk position ^ (row & 15)) whose ds_read_b128 B-fragment reads are conflict-free; rows >= M re-read
//     row M - 1 (their results are never stored);
//   * accumulators: lane (fr, fg) ends with y[batch row 16 nb + fr][W rows 4 fg .. 4 fg + 3] -- four consecutive output columns, so
//     the epilogue is one 16-B (fp32) / 8-B (16-bit) access per lane and column block, and SwiGLU's (gate, up) pairs sit in one lane;
//   * every output element's K summation order is fixed (slice by slice, one accumulator) and independent of M: a row's result does
//     not depend on its batch company (tests/test_serving_gpu.py).
// Epilogues (the prefill GEMM's, gemm_common.h): bias, f32 / 16-bit out, fp32 residual (in place allowed), SwiGLU over interleaved rows.
// Waits are the compiler's: every operand goes through registers (no LDS-DMA), so hipcc counts vmcnt exactly; the loop is unrolled by 6
// (the W ring has 3 slots, the x ring 2) and its steady state carries no guard.
#include "gemm_common.h"

#define SK_RW 64    // rows of W per workgroup
#define SK_KS 256   // k-values per slice (512 B per row)

template <int NB>  // 16-row blocks of x: M <= 16 NB
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  constexpr int XI = NB * 2;            // 16-B chunks of an x slice per thread (16 NB rows x 32 chunks / 256 threads)
  constexpr int XBYTES = NB * 16 * 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x-slice images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * SK_RW;
  const int ns = (p.K + SK_KS - 1) / SK_KS;   // K % 32 == 0; a last partial slice multiplies zeros (its W loads re-read k = 0)

  int wr = n0 + 16 * wave + fr;
  if (wr > p.N - 1) wr = p.N - 1;
  const bf16_t* wrow = p.W + (long)wr * p.ldw + fg * 8;
  const bf16_t* xsrc[XI];
  int xdst[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int s = i * 256 + tid, row = s >> 5, cpos = s & 31;
    int m = row;
    if (m > p.M - 1) m = p.M - 1;
    xsrc[i] = p.A + (long)m * p.lda + ((cpos ^ (row & 15)) << 3);   // slot (row, cpos) holds logical chunk cpos ^ (row & 15)
    xdst[i] = s << 4;
  }
  bf16x8 w[3][8], xr[2][XI];
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // EDGE: the slice may be the last, partial one -- k-blocks / chunks at or beyond K read k = 0 of the row (finite) against x = 0
  auto load_w = [&](int s, bf16x8* dst, bool edge) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      long k = (long)s * SK_KS + kb * 32;
      if (edge && k >= p.K) k = 0;
      dst[kb] = __builtin_nontemporal_load((const bf16x8*)(wrow + k));
    }
  };
  auto load_x = [&](int s, bf16x8* dst, bool edge) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int cpos = (i * 256 + tid) & 31, row = (i * 256 + tid) >> 5;
      const bool in = !edge || s * SK_KS + ((cpos ^ (row & 15)) << 3) < p.K;
      dst[i] = in ? *(const bf16x8*)(xsrc[i] + (long)s * SK_KS) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto put_x = [&](const bf16x8* src, int buf) {
#pragma unroll
    for (int i = 0; i < XI; ++i) *(bf16x8*)(smem + buf * XBYTES + xdst[i]) = src[i];
  };
  auto compute = [&](const bf16x8* wv, int buf) {
    const char* xb = smem + buf * XBYTES;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int row = nb * 16 + fr;
        const bf16x8 xv = *(const bf16x8*)(xb + row * 512 + (((kb * 4 + fg) ^ fr) << 4));
        acc[nb] = GR_MFMA_16x16x32(wv[kb], xv, acc[nb]);
      }
  };
  // slice s: W in w[s % 3], x in LDS image s % 2; x of s + 1, s + 2 in xr[(s + 1) % 2], xr[s % 2]
  load_w(0, w[0], true);
  if (ns > 1) load_w(1, w[1], true);
  if (ns > 2) load_w(2, w[2], true);
  load_x(0, xr[0], true);
  if (ns > 1) load_x(1, xr[1], true);
  put_x(xr[0], 0);
  if (ns > 2) load_x(2, xr[0], true);
  __syncthreads();
  // one iteration: multiply slice s, refill its W slot with slice s + 3, publish x of s + 1 and refill that register slot with s + 3
#define SK_ITER(S, D, GUARD)                                            \
  {                                                                     \
    const int s_ = (S);                                                 \
    compute(w[(D) % 3], (D) & 1);                                       \
    if (!(GUARD) || s_ + 3 < ns) load_w(s_ + 3, w[(D) % 3], GUARD);     \
    if (!(GUARD) || s_ + 1 < ns) put_x(xr[((D) + 1) & 1], ((D) + 1) & 1); \
    if (!(GUARD) || s_ + 3 < ns) load_x(s_ + 3, xr[((D) + 1) & 1], GUARD); \
    __syncthreads();                                                    \
  }
  int s = 0;
  for (; s + 6 + 3 < ns; s += 6) {    // steady state: every slice touched below exists and is whole
    SK_ITER(s, 0, false) SK_ITER(s + 1, 1, false) SK_ITER(s + 2, 2, false)
    SK_ITER(s + 3, 3, false) SK_ITER(s + 4, 4, false) SK_ITER(s + 5, 5, false)
  }
  for (; s < ns; s += 6) {            // the last rounds, guarded
    SK_ITER(s, 0, true)
    if (s + 1 < ns) SK_ITER(s + 1, 1, true)
    if (s + 2 < ns) SK_ITER(s + 2, 2, true)
    if (s + 3 < ns) SK_ITER(s + 3, 3, true)
    if (s + 4 < ns) SK_ITER(s + 4, 4, true)
    if (s + 5 < ns) SK_ITER(s + 5, 5, true)
  }
#undef SK_ITER

  // ---- epilogue: lane (fr, fg) holds y[m = 16 nb + fr][n = n0 + 16 wave + 4 fg .. + 3]
  const int n = n0 + 16 * wave + 4 * fg;
  if (n >= p.N) return;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + n);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int m = nb * 16 + fr;
    if (m >= p.M) continue;
    f32x4 v = acc[nb] + bias;
    if (p.act == 3) {   // interleaved rows: even = gate_j, odd = up_j (weights.pack_llm) -> 2 outputs
      uint32_t hi, lo;
      split2(silu_f(v[0]) * v[1], silu_f(v[2]) * v[3], hi, lo);
      *(uint32_t*)((bf16_t*)p.C + (long)m * p.ldc + (n >> 1)) = hi;
      continue;
    }
    if (p.act == 1 || p.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
    }
    if (p.resid) v += *(const f32x4*)(p.resid + (long)m * p.ldr + n);
    if (p.out_f32) *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
    else st4f((bf16_t*)p.C, (long)m * p.ldc + n, v);
  }
}

int gr_launch_gemm_skinny(const GemmArgs& p, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;   // (operand pairs: the general kernels)
  if (p.M < 1 || p.M > 64 || p.K % 32 != 0 || p.N % 4 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0) return GR_EINVAL;
  if (p.conv_C > 0 || p.splits > 1 || p.scale || p.a_parts || p.resid_mod > 0 || p.c_group > 0) return GR_EINVAL;
  if (p.act == 3 && (p.resid || p.out_f32 || (p.N & 7))) return GR_EINVAL;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.W)) & 15) return GR_EINVAL;
  const int nb = (p.M + 15) / 16;
  const dim3 grid(gr_cdiv(p.N, SK_RW));
  if (nb <= 1) hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 2 * 1 * 16 * 512, stream, p);
  else if (nb == 2) hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 2 * 2 * 16 * 512, stream, p);
  else hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, dim3(256), 2 * 4 * 16 * 512, stream, p);
  return GR_OK;
}
