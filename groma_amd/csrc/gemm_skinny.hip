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
//
// The same source builds the OCP e4m3 variant (gemm_skinny_fp8.hip defines SK_FP8 = 1; round 6: e4m3 models past 8 rows).  A slice is 512
// BYTES of a row in both builds and a lane's eight 16-B chunks of it sit at the same byte offsets (chunk i at 64 i + 16 fg), so loads, the
// x image and its swizzle are identical; what differs is the product -- four v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: chunks
// 2 j, 2 j + 1 are exactly the 32 k-values per lane the instruction takes) instead of eight 16x16x32 -- and the epilogue's dequantisation
// acc * w_scale[n] * a_scale[m] (per-row activation scales from gr_norm_fp8 / gr_quant_rows_fp8, per-output-channel weight scales), in the
// e4m3 GEMM's order.
#include "gemm_common.h"

#ifndef SK_FP8
#define SK_FP8 0
#endif
#define SK_SB 512                          // bytes of a row per slice
#define SK_EB (SK_FP8 ? 1 : 2)             // bytes per operand element
#if SK_FP8
#define gemm_skinny_kernel gemm_skinny_fp8_kernel
#define gr_launch_gemm_skinny gr_launch_gemm_skinny_fp8
#define gr_diag_skinny_kw gr_diag_skinny_fp8_kw
typedef __attribute__((ext_vector_type(8))) int sk_i32x8;
typedef __attribute__((ext_vector_type(4))) int sk_i32x4;
#endif

// NB: 16-row blocks of x (M <= 16 NB).  KW: K-ways -- the four waves of a workgroup are 4 / KW row blocks x KW slices of the same
// iteration, so a workgroup owns 64 / KW rows of W and a launch has N KW / 64 workgroups: the N = 4096 matrices (o-proj, down-proj)
// would otherwise stream through 64 of the 256 CUs.  The KW partial sums of a row block meet in LDS in a fixed order.
template <int NB, int KW>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  // Ring depths.  Three W slices (24 KB) in flight per wave; six were measured and are SLOWER on every shape (QKV at 16 rows 27.5 -> 31.2
  // us, gate/up 49.7 -> 54.1: profiles/r06_skinny_bench.txt -- the stream is not waiting for its own loads; 248 instead of 160 registers
  // only cost occupancy).  The x register ring shrinks to one stage when a stage is large (x comes from L2, one iteration of lead is enough).
  constexpr int WD = 3;                      // W slices in flight per wave
  constexpr int XD = NB * KW > 4 ? 1 : 2;    // x stages held in registers
  constexpr int XA = XD + 1;                 // ... so x loads run XA iterations ahead
  constexpr int AHEAD = WD > XA ? WD : XA;
  constexpr int RBW = 4 / KW;                // row blocks (waves along N) per workgroup
  constexpr int XI = NB * 2 * KW;            // 16-B chunks of an x stage per thread (16 NB rows x 32 KW chunks / 256 threads)
  constexpr int XROW = SK_SB * KW;           // bytes per staged x row
  constexpr int XBYTES = NB * 16 * XROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x stages (re-used by the KW reduction)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wave % RBW, kh = wave / RBW;
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * RBW);
  const long KB = (long)p.K * SK_EB;                          // bytes of a row
  const int nit = (int)((KB + SK_SB * KW - 1) / (SK_SB * KW));  // (KB % 64 == 0; chunks past the row multiply zeros: their W loads re-read byte 0)

  int wr = n0 + 16 * rb + fr;
  if (wr > p.N - 1) wr = p.N - 1;
  const char* wrow = (const char*)p.W + ((long)wr * p.ldw) * SK_EB + fg * 16 + kh * SK_SB;
  const char* xsrc[XI];
  int xdst[XI], xk[XI];   // xk: byte offset of the chunk inside an iteration's x rows
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int s = i * 256 + tid, row = s / (32 * KW), cpos = s % (32 * KW);
    int m = row;
    if (m > p.M - 1) m = p.M - 1;
    xk[i] = (cpos ^ (row & 15)) << 4;                              // slot (row, cpos) holds logical chunk cpos ^ (row & 15)
    xsrc[i] = (const char*)p.A + ((long)m * p.lda) * SK_EB + xk[i];
    xdst[i] = s << 4;
  }
  bf16x8 w[WD][8], xr[XD][XI];
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // EDGE: the iteration may reach past K -- k-blocks / chunks at or beyond K read k = 0 of the row (finite) against x = 0
  auto load_w = [&](int t, bf16x8* dst, bool edge) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {   // chunk kb of the slice: bytes 64 kb + 16 fg ..
      long k = (long)t * (SK_SB * KW) + kb * 64;
      if (edge && k + kh * SK_SB >= KB) k = -(long)kh * SK_SB;
      dst[kb] = __builtin_nontemporal_load((const bf16x8*)(wrow + k));
    }
  };
  auto load_x = [&](int t, bf16x8* dst, bool edge) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const bool in = !edge || (long)t * (SK_SB * KW) + xk[i] < KB;
      dst[i] = in ? *(const bf16x8*)(xsrc[i] + (long)t * (SK_SB * KW)) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto put_x = [&](const bf16x8* src, int buf) {
#pragma unroll
    for (int i = 0; i < XI; ++i) *(bf16x8*)(smem + buf * XBYTES + xdst[i]) = src[i];
  };
  auto compute = [&](const bf16x8* wv, int buf) {
    const char* xb = smem + buf * XBYTES;
#if SK_FP8
#pragma unroll
    for (int j = 0; j < 4; ++j)   // one 128-byte k-block: chunks 2 j, 2 j + 1 of the slice for A and B alike
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int row = nb * 16 + fr;
        union { struct { sk_i32x4 a, b; } h; sk_i32x8 v; } ua, ub;
        ua.h.a = __builtin_bit_cast(sk_i32x4, wv[2 * j]);
        ua.h.b = __builtin_bit_cast(sk_i32x4, wv[2 * j + 1]);
        ub.h.a = *(const sk_i32x4*)(xb + row * XROW + (((kh * 32 + (2 * j) * 4 + fg) ^ fr) << 4));
        ub.h.b = *(const sk_i32x4*)(xb + row * XROW + (((kh * 32 + (2 * j + 1) * 4 + fg) ^ fr) << 4));
        // formats e4m3 x e4m3 (cbsz = blgp = 0), both block scales the e8m0 code for 2^0: a plain product at the MX rate
        acc[nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ua.v, ub.v, acc[nb], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
#else
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int row = nb * 16 + fr;
        const bf16x8 xv = *(const bf16x8*)(xb + row * XROW + (((kh * 32 + kb * 4 + fg) ^ fr) << 4));
        acc[nb] = GR_MFMA_16x16x32(wv[kb], xv, acc[nb]);
      }
#endif
  };
  // iteration t: W in w[t % WD], x in LDS stage t % 2; x of t + 1 (.. t + XD) in the register ring
#pragma unroll
  for (int d = 0; d < WD; ++d)
    if (d < nit) load_w(d, w[d], true);
  load_x(0, xr[0], true);
  if (XD == 2 && nit > 1) load_x(1, xr[XD - 1], true);
  put_x(xr[0], 0);
  if (nit > XD) load_x(XD, xr[0], true);
  __syncthreads();
  // one iteration: multiply, refill the W slot with iteration t + WD, publish x of t + 1 and refill that register stage with t + XA
#define SK_XS(D) (XD == 2 ? (((D) + 1) & 1) : 0)
#define SK_ITER(S, D, GUARD)                                                \
  {                                                                         \
    const int s_ = (S);                                                     \
    compute(w[(D) % WD], (D) & 1);                                          \
    if (!(GUARD) || s_ + WD < nit) load_w(s_ + WD, w[(D) % WD], GUARD);     \
    if (!(GUARD) || s_ + 1 < nit) put_x(xr[SK_XS(D)], ((D) + 1) & 1);       \
    if (!(GUARD) || s_ + XA < nit) load_x(s_ + XA, xr[SK_XS(D)], GUARD);    \
    __syncthreads();                                                        \
  }
  int s = 0;
  for (; s + 6 + AHEAD < nit; s += 6) {   // steady state: every iteration touched below exists and is whole
    SK_ITER(s, 0, false) SK_ITER(s + 1, 1, false) SK_ITER(s + 2, 2, false)
    SK_ITER(s + 3, 3, false) SK_ITER(s + 4, 4, false) SK_ITER(s + 5, 5, false)
  }
  for (; s < nit; s += 6) {               // the last rounds, guarded
    SK_ITER(s, 0, true)
    if (s + 1 < nit) SK_ITER(s + 1, 1, true)
    if (s + 2 < nit) SK_ITER(s + 2, 2, true)
    if (s + 3 < nit) SK_ITER(s + 3, 3, true)
    if (s + 4 < nit) SK_ITER(s + 4, 4, true)
    if (s + 5 < nit) SK_ITER(s + 5, 5, true)
  }
#undef SK_XS
#undef SK_ITER

  if (KW > 1) {   // the K-ways of a row block: partial sums through LDS (the x stages are free), added in the fixed order kh = 1, 2, 3
    f32x4* red = (f32x4*)smem;   // [KW - 1][RBW][NB][64]
    if (kh > 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) red[(((kh - 1) * RBW + rb) * NB + nb) * 64 + lane] = acc[nb];
    }
    __syncthreads();
    if (kh > 0) return;
#pragma unroll
    for (int j = 1; j < KW; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] += red[(((j - 1) * RBW + rb) * NB + nb) * 64 + lane];
  }

  // ---- epilogue: lane (fr, fg) holds y[m = 16 nb + fr][n = n0 + 16 rb + 4 fg .. + 3]
  const int n = n0 + 16 * rb + 4 * fg;
  if (n >= p.N) return;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f32x4*)(p.bias + n);
#if SK_FP8
  const f32x4 wsc = *(const f32x4*)(p.w_scale + n);
#endif
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int m = nb * 16 + fr;
    if (m >= p.M) continue;
#if SK_FP8
    f32x4 v = acc[nb] * wsc * (p.a_scale ? p.a_scale[m] : 1.f) + bias;   // dequantise: acc * w_scale[n] * a_scale[m], the e4m3 GEMM epilogue's order
#else
    f32x4 v = acc[nb] + bias;
#endif
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

static int g_skinny_kw = 0;   // 0: chosen per launch; 1 / 2 / 4: forced (tests/diag/skinny_bench.py)
extern "C" int gr_diag_skinny_kw(int kw) {
  if (kw != 0 && kw != 1 && kw != 2 && kw != 4) return GR_EINVAL;
  g_skinny_kw = kw;
  return GR_OK;
}

template <int NB, int KW>
static int launch_skinny(const GemmArgs& p, hipStream_t stream) {
  const size_t lds = 2 * (size_t)NB * 16 * SK_SB * KW;
  static bool attr_set = false;
  if (!attr_set && lds > 65536) {
    if (hipFuncSetAttribute((const void*)gemm_skinny_kernel<NB, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GR_EINVAL;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_skinny_kernel<NB, KW>), dim3(gr_cdiv(p.N, 64 / KW)), dim3(256), lds, stream, p);
  return GR_OK;
}

int gr_launch_gemm_skinny(const GemmArgs& p, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;   // (operand pairs: the general kernels)
  if (p.M < 1 || p.M > 64 || ((long)p.K * SK_EB) % 64 != 0 || p.N % 4 != 0 || ((long)p.lda * SK_EB) % 16 != 0 || ((long)p.ldw * SK_EB) % 16 != 0) return GR_EINVAL;
  if (SK_FP8 && (!p.w_scale || p.K % 128 != 0)) return GR_EINVAL;
  if (p.conv_C > 0 || p.splits > 1 || p.scale || p.a_parts || p.resid_mod > 0 || p.c_group > 0) return GR_EINVAL;
  if (p.act == 3 && (p.resid || p.out_f32 || (p.N & 7))) return GR_EINVAL;
  if ((((uintptr_t)p.A) | ((uintptr_t)p.W)) & 15) return GR_EINVAL;
  const int nb = (p.M + 15) / 16;
  // K-ways: the N = 4096 matrices (o-proj, down-proj) are 64 workgroups at KW = 1 and measured 1.2-1.4 TB/s there against 2.7-3.3 at
  // KW = 4; from N = 8192 up KW = 1 already fills the chip and the extra x traffic of KW > 1 costs (profiles/r06_skinny_bench.txt).
  // A function of the layer shape and of the row-block count only -- i.e. of the batcher's row CAPACITY (a decode step always runs all
  // its rows), never of which rows are occupied: a request's sums do not depend on its company.
  const int nbt = nb <= 1 ? 1 : nb == 2 ? 2 : 4;
  int kw = g_skinny_kw ? g_skinny_kw : (p.N >= 8192 ? 1 : 4);
  if (nbt == 4 && kw > 2) kw = 2;   // (64 rows x 4 K-ways: the x stage alone would need 128 registers)
  while (kw > 1 && (long)p.K * SK_EB < SK_SB * kw * 2) kw >>= 1;
  if (nbt == 1) return kw == 1 ? launch_skinny<1, 1>(p, stream) : kw == 2 ? launch_skinny<1, 2>(p, stream) : launch_skinny<1, 4>(p, stream);
  if (nbt == 2) return kw == 1 ? launch_skinny<2, 1>(p, stream) : kw == 2 ? launch_skinny<2, 2>(p, stream) : launch_skinny<2, 4>(p, stream);
  return kw == 1 ? launch_skinny<4, 1>(p, stream) : launch_skinny<4, 2>(p, stream);
}
