// Decode-step weight streaming with OCP e4m3 WEIGHTS (round 5; BASELINE configs[4] at L = 1: the decode step of an fp8 model streams
// 6.6 GB per token instead of 13.2; R: the decode branch of groma/model/groma.py:376-379 + HF LlamaDecoderLayer at L = 1).
//
// Same contract as gemv_fused.hip (gr_gemv_desc with w8 = 1): y[M <= 8, N] = x[M, K] . W[N, K]^T with the producer fused into the
// prologue and the consumer into the epilogue, a workgroup owning whole rows of W, nothing leaving the kernel but the result.
// What differs is the arithmetic, because the 16-bit stream's does not survive halving the bytes:
//   * the first e4m3 build converted a lane's 8 weight bytes to 16-bit pairs (v_cvt_scalef32_pk_bf16_fp8) and kept the v_dot2c dot
//     products.  Per weight BYTE that is twice the VALU work of the 16-bit stream plus the conversions, and it measured VALU-bound:
//     QKV 25.0 -> 20.8 us, gate/up 36.8 -> 32.3, o-proj 11.1 -> 13.5, down 21.0 -> 24.9 for half the bytes
//     (profiles/r05_gemv_w8_dot2.txt).  gfx950 has no fp8 dot instruction for the VALU (v_dot4_f32_fp8_fp8 needs dot11-insts);
//   * so the products run on the matrix unit: v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales: a plain e4m3 x e4m3 -> f32
//     product, the instruction the e4m3 GEMM uses) with the WEIGHTS as the A operand -- 16 rows of W x 128 k-values straight from
//     the registers the loads filled, no conversion -- and the quantised activation rows as B (columns = batch rows, the unused
//     ones zero).  One MFMA retires 2 KB of weights; the VALU is left with addressing.
//   * the operand x is quantised by the prologue exactly as the e4m3 prefill forms it (fp8.hip): per row s = max|x| / 448,
//     q = e4m3_rne(x * (1 / s)) -- a normalisation output straight from fp32 (x_mode 1), a stored activation from its 16-bit value
//     (x_mode 0 / 2) -- and kept as e4m3 BYTES in LDS (M x K bytes, rows padded by 16 B against bank conflicts);
//   * a lane (fr = lane & 15, fg = lane >> 4) holds row fr of the group and, per 128-byte k-block, the 16-byte chunks fg and fg + 4
//     (the operand layout of the e4m3 GEMM's fragments, for A and B alike); its accumulator is y[rows 4 fg .. 4 fg + 3][batch row fr].
//     The four waves take the 512-byte K slices round-robin, two slices (16 x 16 B per lane) in flight each; their partial
//     accumulators meet in LDS in a fixed order (bit-reproducible, independent of the other batch rows);
//   * 16 rows per group, THREE persistent workgroups per CU (<= 168 VGPRs -- the prologue lives beside the two slices already in flight --, 16-45 KB of LDS): a launch is then close to
//     bytes / bandwidth + one ramp instead of rounds x latency.
// The epilogue dequantises acc * w_scale[n] * s[m] (the e4m3 GEMM epilogue's order) and then is the 16-bit stream's: f32 out, in-place
// residual update, SwiGLU over interleaved (gate, up) rows (both in one lane), or RoPE + q / K row / V^T column (partner row 32 lanes away).
#include <type_traits>

#include "gemv_args.h"
#include "../../include/groma_hip.h"

#define G8_KS 512    // K bytes per slice and row (4 MFMA k-blocks of 128)
#define G8_ROWS 16   // rows of W per group = the MFMA's 16 A rows
#ifndef G8_WG_PER_CU
#define G8_WG_PER_CU 2   // (round 6: the prologue holds a whole batch of operand rows in registers beside the two slices in flight)
#endif

typedef __attribute__((ext_vector_type(8))) int g8_i32x8;
typedef __attribute__((ext_vector_type(4))) int g8_i32x4;

template <int MB>
__global__ __launch_bounds__(256, G8_WG_PER_CU) void gemv_fp8_kernel(GemvFArgs p) {
  constexpr int ROWS = G8_ROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int XLD = p.K + 16;                       // bytes per staged activation row (16-B pad: rows land on different banks)
  uint8_t* xs = (uint8_t*)smem;                   // [MB][XLD] e4m3
  bf16_t* tmp16 = (bf16_t*)(smem + MB * XLD);     // x_mode 2 only: the merged context as 16-bit values before it is quantised
  __shared__ f32x4 red[4][64];
  __shared__ float stat[4][4];
  __shared__ float xsc[8];
  __shared__ unsigned amax_u[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ngroups = p.N / ROWS;
  const uint8_t* W = (const uint8_t*)p.W;

  // ---- a group's 16 rows: two runs of 8 (epi 3: 8 dims d and their rotate_half partners d + HD/2 of one head; else consecutive)
  int r_lo, r_hi;
  const uint8_t* row_ptr;  // this lane's row fr of the current group
  auto set_group = [&](int g) {
    r_lo = g * ROWS;
    r_hi = g * ROWS + ROWS / 2;
    if (p.epi == 3 && (long)g * ROWS < 2L * p.H * p.HD) {
      const int bph = p.HD / ROWS;  // groups per head
      const int hh = g / bph, j = g - hh * bph;
      r_lo = hh * p.HD + j * (ROWS / 2);
      r_hi = r_lo + p.HD / 2;
    }
    const int n = fr < ROWS / 2 ? r_lo + fr : r_hi + (fr - ROWS / 2);
    row_ptr = W + (long)n * p.ldw + fg * 16;
  };
  const int ns = (p.K + G8_KS - 1) / G8_KS;  // K slices; wave w takes w, w + 4, ...
  const int cnt = wave < ns ? (ns - wave + 3) / 4 : 0;

  g8_i32x4 wA[8], wB[8];  // a slice: 4 k-blocks x (chunk fg, chunk fg + 4)
  auto load = [&](int i, g8_i32x4* w) {  // every load of the slice is issued before anything is consumed
    const int k0 = (wave + 4 * i) * G8_KS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // K % 128 == 0: a k-block is all in or all out; blocks beyond K (last slice of K = 11008) re-read the row's last block
      // instead of branching around the load -- their x is zero
      const int kb = min(k0 + j * 128, p.K - 128);
      w[2 * j] = __builtin_nontemporal_load((const g8_i32x4*)(row_ptr + kb));
      w[2 * j + 1] = __builtin_nontemporal_load((const g8_i32x4*)(row_ptr + kb + 64));
    }
  };
  auto start_group = [&](int g) {
    set_group(g);
    if (cnt > 0) load(0, wA);
    if (cnt > 1) load(1, wB);
  };
  int grp = blockIdx.x;
  set_group(grp);
  // The first group's two slices per wave: issued by the prologue AFTER its own first loads (below).  vmcnt retires in order, so a
  // prologue load queued BEHIND the weight slices is not usable before every slice has landed -- in the QKV launch, where a workgroup
  // owns exactly one group, that is the whole 50 MB burst: the round-5 kernel, which started the slices first, quantised its operand
  // AFTER the weights had arrived instead of in their shadow (tests/diag/gemv_w8_bench.py with -DG8_DIAG_NOPRO: QKV 20.4 -> 14.5 us,
  // down 20.3 -> 15.4, all streams of a token 2.66 -> 2.16 ms without a prologue; profiles/r06_gemv_w8_decomp.txt).  Unconditional
  // (waves without a slice re-read the row's last block: `load` clamps) so that the compiler can COUNT the loads it may leave in flight.
  auto first_slices = [&]() {
#ifdef G8_FIRST_BARRIER  // (tests/diag A/B, measured flat -- `bar` rows of profiles/r06_gemv_w8_decomp.txt: the order inside a wave is what matters)
    __syncthreads();  // every wave of the workgroup has issued its operand loads before any weight slice is requested
#endif
    load(0, wA);
    load(1, wB);
  };

  // ---- prologue: the quantised operand, once per workgroup, in the shadow of the first weight loads
  if (tid < 8) amax_u[tid] = 0u;
  __syncthreads();
  // rows given as 16-bit values through `src(m, c)` (8 values of chunk c of row m): max, then quantise -> bytes.  Two rows at a time,
  // every load of a batch issued before the first use and the rows kept in registers between the max and the quantising pass: the
  // operand is L2-resident, so a batch costs ONE L2 round trip (the first version walked row by row, chunk by chunk: the o-proj /
  // down-proj launches, whose operand comes this way, were SLOWER than their 16-bit twins -- 12.6 vs 11.2 us and 25.2 vs 21.5 us,
  // profiles/r05_gemv_w8_mfma.txt; round 5 re-read the rows for the second pass).  `hook` runs once, behind the first batch's loads.
  auto quantise_rows = [&](auto src, auto hook) {
    constexpr int CH = 6, RQ = 2;  // chunks of 8 per thread and row: K <= 12288; rows per batch (4 spill beside the weight slices)
    const int c8 = p.K >> 3;
    auto batch = [&](int m0, auto first) {
      bf16x8 v[RQ][CH];
#pragma unroll
      for (int mi = 0; mi < RQ; ++mi)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = tid + i * 256;
          v[mi][i] = (m0 + mi < p.M && c < c8) ? src(m0 + mi, c) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      if (decltype(first)::value) hook();
#pragma unroll
      for (int mi = 0; mi < RQ; ++mi) {
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(bf2f((bf16_t)v[mi][i][e])));
        am = wave_max(am);
        if (lane == 0) atomicMax(&amax_u[m0 + mi], __builtin_bit_cast(unsigned, am));
      }
      __syncthreads();
      if (tid < RQ) xsc[m0 + tid] = fmaxf(__builtin_bit_cast(float, amax_u[m0 + tid]), 1e-20f) / 448.0f;
#pragma unroll
      for (int mi = 0; mi < RQ; ++mi) {
        const float inv = 1.0f / (fmaxf(__builtin_bit_cast(float, amax_u[m0 + mi]), 1e-20f) / 448.0f);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int c = tid + i * 256;
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = bf2f((bf16_t)v[mi][i][e]);
          uint2 o;
          o.x = pack4_fp8(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
          o.y = pack4_fp8(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
          if (c < c8) *(uint2*)(xs + (long)(m0 + mi) * XLD + (c << 3)) = o;
        }
      }
    };
    batch(0, std::true_type{});
#pragma unroll 1
    for (int m0 = RQ; m0 < MB; m0 += RQ) batch(m0, std::false_type{});
    __syncthreads();
  };
#ifdef G8_DIAG_NOPRO  // (tests/diag only: what the launches cost WITHOUT the quantising prologue -- a constant operand)
  first_slices();
  for (int i = tid; i < MB * XLD / 4; i += 256) ((uint32_t*)xs)[i] = 0x38383838u;
  if (tid < 8) xsc[tid] = 1.0f;
  __syncthreads();
  if (true) {
#else
  if (p.x_mode == 0) {  // stored 16-bit activation rows (attention context, SwiGLU output), L2-resident
    quantise_rows([&](int m, int c) { return *(const bf16x8*)(p.A + (long)m * p.lda + (c << 3)); }, first_slices);
#endif
  } else if (p.x_mode == 1) {  // x = gamma * (h * rsqrt(mean(h^2) + eps)) (HF LlamaRMSNorm), quantised straight from fp32; K <= 4096
    constexpr int KJ = 4, RB = 4;
    f32x4 g[KJ];
    int cc[KJ];
    bool cin[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int c = tid * 4 + j * 1024;
      cin[j] = c < p.K;
      cc[j] = cin[j] ? c : 0;
      g[j] = *(const f32x4*)(p.gamma + cc[j]);
    }
    // RB rows at a time (16 x RB registers of h per thread): one batch at 4 batch rows, two at 8 -- the second batch's loads queue
    // behind the weight slices, which have landed by the time the first batch is done
    auto batch = [&](int m0, auto first) {
      f32x4 hv[RB][KJ];
#pragma unroll
      for (int mi = 0; mi < RB; ++mi)
#pragma unroll
        for (int j = 0; j < KJ; ++j) hv[mi][j] = *(const f32x4*)(p.h + (long)min(m0 + mi, p.M - 1) * p.ldh + cc[j]);
      if (decltype(first)::value) first_slices();
      float ss[RB];
#pragma unroll
      for (int mi = 0; mi < RB; ++mi) {
        ss[mi] = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          const f32x4 v = hv[mi][j];
          ss[mi] += cin[j] ? v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3] : 0.f;
        }
        ss[mi] = wave_sum(ss[mi]);
      }
      __syncthreads();  // (every thread has read the previous batch's maxima in stat[])
      if (lane == 0) {
#pragma unroll
        for (int mi = 0; mi < RB; ++mi) stat[wave][mi] = ss[mi];
      }
      __syncthreads();
      float am[RB];
#pragma unroll
      for (int mi = 0; mi < RB; ++mi) {
        const int m = m0 + mi;
        const float rstd = m < p.M ? rsqrtf((stat[0][mi] + stat[1][mi] + stat[2][mi] + stat[3][mi]) / (float)p.K + p.eps) : 0.f;
        am[mi] = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
          hv[mi][j] = g[j] * (hv[mi][j] * rstd);  // (padding rows m >= M: rstd = 0 -> x = 0)
          if (cin[j]) am[mi] = fmaxf(fmaxf(am[mi], fmaxf(fabsf(hv[mi][j][0]), fabsf(hv[mi][j][1]))), fmaxf(fabsf(hv[mi][j][2]), fabsf(hv[mi][j][3])));
        }
        am[mi] = wave_max(am[mi]);
      }
      __syncthreads();  // (every thread has read the sums in stat[])
      if (lane == 0) {
#pragma unroll
        for (int mi = 0; mi < RB; ++mi) stat[wave][mi] = am[mi];
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < RB; ++mi) {
        const int m = m0 + mi;
        const float sc = fmaxf(fmaxf(fmaxf(stat[0][mi], stat[1][mi]), fmaxf(stat[2][mi], stat[3][mi])), 1e-20f) / 448.0f;
        const float inv = 1.0f / sc;
        if (tid == 0) xsc[m] = sc;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
          if (cin[j]) *(uint32_t*)(xs + (long)m * XLD + cc[j]) = pack4_fp8(hv[mi][j][0] * inv, hv[mi][j][1] * inv, hv[mi][j][2] * inv, hv[mi][j][3] * inv);
      }
    };
    batch(0, std::true_type{});
    if constexpr (MB > RB) batch(RB, std::false_type{});
    __syncthreads();
  } else {  // x_mode 2: merge the key slices of decode_attention (slice order; rounded like its nsplit = 1 output), then quantise
    const int hd = p.a_hd, c8 = p.K >> 3;
    for (int idx = tid; idx < MB * c8; idx += 256) {
      const int m = idx / c8, k0 = (idx - m * c8) << 3;
      union { bf16x8 v; uint32_t u[4]; } pk;
      pk.v = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (m < p.M) {
        const int hh = k0 / hd, dd = k0 - hh * hd;
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
        for (int e = 0; e < 4; ++e) pk.u[e] = pack2bf(o[2 * e] / l, o[2 * e + 1] / l);
      }
      *(bf16x8*)(tmp16 + (long)m * p.K + k0) = pk.v;
    }
    first_slices();  // (behind the merge's own global reads; the quantising passes below read LDS only)
    __syncthreads();
    quantise_rows([&](int m, int c) { return *(const bf16x8*)(tmp16 + (long)m * p.K + (c << 3)); }, [] {});
  }

  const uint8_t* x_lane = xs + (long)fr * XLD + fg * 16;  // this lane's batch row (column of the product); rows >= MB read as zero
  f32x4 acc;
  auto consume = [&](int i, const g8_i32x4* w) {
    const int k0 = (wave + 4 * i) * G8_KS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kb = k0 + j * 128;
      g8_i32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
      if (fr < MB && kb < p.K) {
        x0 = *(const g8_i32x4*)(x_lane + kb);
        x1 = *(const g8_i32x4*)(x_lane + kb + 64);
      }
      union { struct { g8_i32x4 a, b; } h; g8_i32x8 v; } ua, ub;
      ua.h.a = w[2 * j]; ua.h.b = w[2 * j + 1];
      ub.h.a = x0; ub.h.b = x1;
      // formats e4m3 x e4m3 (cbsz = blgp = 0), both block scales the e8m0 code for 2^0: a plain product at the MX rate
      acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ua.v, ub.v, acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
  };

#pragma unroll 1
  while (true) {
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef G8_DIAG_NOSTREAM  // (tests/diag only: prologue + reduction + epilogue, one slice per wave)
    for (int i = 0; i < (cnt < 1 ? cnt : 1); i += 2) {
#else
    for (int i = 0; i < cnt; i += 2) {
#endif
      consume(i, wA);
      if (i + 2 < cnt) load(i + 2, wA);
      if (i + 1 < cnt) {
        consume(i + 1, wB);
        if (i + 3 < cnt) load(i + 3, wB);
      }
    }
    const int e_lo = r_lo, e_hi = r_hi;  // the finished group's rows (the epilogue's)
    const int nxt = grp + (int)gridDim.x;
    red[wave][lane] = acc;
    if (nxt < ngroups) start_group(nxt);  // the next group's first slices fly while this one is reduced and stored
    __syncthreads();
    if (wave == 0) {
      // lane (fr, fg): rows 4 fg .. 4 fg + 3 of the group, batch row fr; waves summed in a fixed order
      const f32x4 v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
      const int m = fr;
      const bool ok = m < p.M;
      const float sx = xsc[m < MB ? m : 0];
      float y[4];
      int n[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * fg + r;
        n[r] = rr < ROWS / 2 ? e_lo + rr : e_hi + (rr - ROWS / 2);
        y[r] = v[r] * p.w_scale[n[r]] * sx;  // dequantise: acc * w_scale[n] * a_scale[m], the e4m3 GEMM epilogue's order
      }
      if (p.epi == 0) {
        if (ok)
#pragma unroll
          for (int r = 0; r < 4; ++r) ((float*)p.C)[(long)m * p.ldc + n[r]] = y[r];
      } else if (p.epi == 1) {
        if (ok)
#pragma unroll
          for (int r = 0; r < 4; ++r) p.resid[(long)m * p.ldr + n[r]] += y[r];
      } else if (p.epi == 2) {  // interleaved rows: even = gate_j, odd = up_j -- both in this lane
        if (ok) {
          ((bf16_t*)p.C)[(long)m * p.ldc + (n[0] >> 1)] = f2bf(silu_f(y[0]) * y[1]);
          ((bf16_t*)p.C)[(long)m * p.ldc + (n[2] >> 1)] = f2bf(silu_f(y[2]) * y[3]);
        }
      } else {  // epi 3: the projection is rounded to 16 bits first (what the prefill GEMM stores), then rotate_half in f32
        const int HHD = p.H * p.HD;
        const int sect = e_lo / HHD, nn = e_lo - sect * HHD;
        const int hh = nn / p.HD;
        const int HALF = p.HD / 2;
        int pos = 0;
        long bh = 0;
        if (ok) {
          pos = p.pos_dev ? p.pos_dev[m * p.pos_stride] : p.pos0;
          bh = (long)m * p.H + hh;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = n[r] - sect * HHD - hh * p.HD;
          const float a = bf2f(f2bf(y[r]));
          const float partner = __shfl_xor(a, 32, 64);  // row rr <-> rr +- 8 = d <-> d +- HD/2: two fg steps = 32 lanes away
          if (ok) {
            if (sect == 2) {
              p.vt[(bh * p.HD + d) * p.kv_stride + pos] = f2bf(y[r]);
            } else {
              float o = a;
              if (p.cosT) {
                const int dc = d < HALF ? d : d - HALF;
                const float sgn = d < HALF ? -1.f : 1.f;
                o = a * p.cosT[(long)pos * HALF + dc] + sgn * partner * p.sinT[(long)pos * HALF + dc];
              }
              if (sect == 0) p.q[bh * p.HD + d] = f2bf(o);
              else p.kc[(bh * p.kv_stride + pos) * p.HD + d] = f2bf(o);
            }
          }
        }
      }
    }
    if (nxt >= ngroups) break;
    grp = nxt;
    __syncthreads();  // wave 0 has read red[] before the next group's partials overwrite it
  }
}

int gr_launch_gemv_fp8(const GemvFArgs& p, int MB, int n_cu, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;
  size_t lds = (size_t)MB * (p.K + 16);
  if (p.x_mode == 2) lds += (size_t)MB * p.K * sizeof(bf16_t);  // the merged context before it is quantised
  if (lds > 128 * 1024 || (p.x_mode != 1 && p.K > 12288)) return GR_EINVAL;  // (the row quantiser holds 6 chunks of 8 per thread and row)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemv_fp8_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemv_fp8_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return GR_EINVAL;
    attr_set = true;
  }
  const int groups = p.N / G8_ROWS;
  // persistent over row groups, every workgroup the same number of them (+-1): 768 groups on 512 slots would leave half the
  // workgroups a second group to run alone (round 6: grid = groups / rounds instead of the slot count; gate/up 29.0 -> 26.2 us, head
  // 35.8 -> 33.6 us at 4 rows, all streams of a token 2.51 -> 2.42 ms, unchanged at 8 rows: profiles/r06_gemv_grid.txt)
#ifdef G8_GRID_SLOTS
  const dim3 grid(groups < G8_WG_PER_CU * n_cu ? groups : G8_WG_PER_CU * n_cu);
#else
  const int cap = G8_WG_PER_CU * n_cu, rounds = (groups + cap - 1) / cap;
  const dim3 grid((groups + rounds - 1) / rounds);
#endif
  if (MB == 4) hipLaunchKernelGGL(gemv_fp8_kernel<4>, grid, dim3(256), lds, stream, p);
  else hipLaunchKernelGGL(gemv_fp8_kernel<8>, grid, dim3(256), lds, stream, p);
  return GR_OK;
}
