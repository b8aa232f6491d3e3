// Decode-step weight streaming, round 4: one kernel per weight matrix, nothing between them (R: the decode
// branch of groma/model/groma.py:376-379 + HF LlamaDecoderLayer at L = 1).
//
// A decode step streams every 16-bit weight once (13.2 GB), shared by the <= 8 rows of the per-GPU batch: HBM-bound.  Round 1-3
// ran each GEMV as K/512 independent slices whose fp32 partials a SECOND kernel summed (fused with the RMSNorm / RoPE /
// SwiGLU that followed): 9 launches per layer, 294 per token, and the small kernels between the GEMVs cost 1.3 ms of a 4.1 ms
// token (65 x 7.2 us of reduce + norm on ONE workgroup per row, 32 x 5 us of RoPE, 32 x 4.7 us of SwiGLU) next to 2.8 ms of
// streaming (profiles/r03: 0.40 of the HBM roofline end to end).  Here a workgroup owns ROWS whole rows of W:
//   * its 4 waves take the 512-wide K slices round-robin and keep TWO slices of loads in flight (all ROWS x 16 B of a slice are
//     issued before the previous slice is consumed: 2 x 8 KB per wave, 64 KB per workgroup, and the register budget -- < 256
//     VGPRs -- keeps TWO workgroups on a CU, so one streams while the other runs its prologue / reduction / epilogue) -- no
//     partial sums leave the kernel, a cross-lane butterfly + one LDS exchange finish the dot products in a fixed order
//     (bit-reproducible);
//   * the operand x is built by a PROLOGUE that runs while the first weight loads fly: x = f16(RMSNorm(h) * gamma) from the fp32
//     residual stream (every workgroup recomputes the row statistics: 64 KB of L2 reads against 128+ KB of weights), or the
//     merge of decode_attention's key slices, staged once in LDS; or plain 16-bit rows read from L2 one slice ahead;
//   * the EPILOGUE is the consumer the old reduce kernels were: fp32 residual update in place (o-proj, down-proj), SwiGLU over
//     the interleaved (gate, up) rows, HF rotate_half RoPE + q / K-cache row / V^T-cache column (the workgroup's rows are the
//     pairs d, d + hd/2 of one head, so the partner sits in the same wave), or fp32 logits.
// 5 launches per layer (QKV, attention, o-proj, gate/up, down) instead of 9.
#include <type_traits>

#include "gr_common.h"
#include "../../include/groma_hip.h"

#define GF_KS 512  // K slice: 64 lanes x 8 elements
#ifndef GF_ROWS
#define GF_ROWS 8  // rows of W per row group at either batch width (A/B on one box, tests/diag/gemv_bench.py -> profiles/r04_gemv_ab.txt:
#endif             //  4 rows x 8 batch rows 3.3 TB/s, 8 x 8 3.6; 16 x 4 2.8, 8 x 4 4.5)
#ifndef GF_WG_PER_CU
#define GF_WG_PER_CU 2  // persistent workgroups per CU
#endif

#include "gemv_args.h"

// XG: the operand comes from global memory (x_mode 0) -- else it is staged in LDS by the prologue (compile-time, so neither
// instantiation carries the other's registers: both stay well under 256 VGPRs = two or three workgroups per CU).
//
// A workgroup is PERSISTENT over row groups g = blockIdx.x, + gridDim.x, ... (grid = min(groups, 2 per CU)): the prologue --
// whose L2 reads (the fp32 rows of h once + gamma: 80 KB at 4 rows x 4096) are larger than one group's 64 KB of weights -- is
// paid once per workgroup, not once per group (first version: 2.8 TB/s on the QKV / gate-up / head shapes, because every
// 8-row group redid it), and the first two slices of the NEXT group are requested before the butterfly / LDS exchange /
// epilogue of the current one, so the weight stream does not drain between groups.
template <int MB, bool XG>
__global__ __launch_bounds__(256, (XG && MB == 8) ? 1 : 2) void gemv_fused_kernel(GemvFArgs p) {
  constexpr int ROWS = GF_ROWS;  // rows of W per group
  constexpr int NV = ROWS * MB;  // dot products per row group (32 at <= 4 batch rows, 64 at 8): lanes 0..NV-1 own one each after the butterfly
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;                         // x_mode 1 / 2: [MB][K]
  __shared__ float red[4][NV];
  __shared__ float stat[4][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngroups = (p.N + ROWS - 1) / ROWS;

  // ---- a group's rows.  epi 3 (fused QKV): inside the q and k sections a group takes ROWS/2 dims d and their rotate_half
  // partners d + HD/2 of one head; the v section and every other mode take ROWS consecutive rows.
  // (row numbers and row bases are wave-uniform scalars: the loads below must issue back to back)
  // The two halves are runs of ROWS/2 consecutive rows: two scalar bases + r * ldw in the per-lane offset (N % ROWS == 0 is
  // checked by the entry point, so no row is ever clamped).
  int r_lo, r_hi;                // first row of the lower / upper half of the current group's rows
  const char *base_lo, *base_hi; // their addresses (SGPR pairs)
  const unsigned ldwB = (unsigned)p.ldw * 2u;
  auto set_group = [&](int g) {
    r_lo = g * ROWS;
    r_hi = g * ROWS + ROWS / 2;
    if (p.epi == 3 && (long)g * ROWS < 2L * p.H * p.HD) {
      const int bph = p.HD / ROWS;  // groups per head
      const int hh = g / bph, j = g - hh * bph;
      r_lo = hh * p.HD + j * (ROWS / 2);
      r_hi = r_lo + p.HD / 2;
    }
    base_lo = (const char*)(p.W + (long)r_lo * p.ldw);
    base_hi = (const char*)(p.W + (long)r_hi * p.ldw);
  };
  const int ns = (p.K + GF_KS - 1) / GF_KS;          // K slices; wave w takes w, w + 4, ...
  const int cnt = wave < ns ? (ns - wave + 3) / 4 : 0;

  bf16x8 wA[ROWS], wB[ROWS];
  bf16x8 xn[XG ? MB : 1];  // XG: raw x of the NEXT slice to be consumed (one buffer: requested at the start of the previous consume)
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_x = [&](int i) {
    if constexpr (XG) {
      const int k0 = (wave + 4 * i) * GF_KS + lane * 8;
      const unsigned voff = (unsigned)min(k0, p.K - 8) * 2u;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        xn[m] = *(const bf16x8*)((const char*)(p.A + (long)min(m, p.M - 1) * p.lda) + voff);
        if (k0 >= p.K || m >= p.M) xn[m] = zero8;
      }
    }
  };
  auto load = [&](int i, bf16x8* w) {  // slice i of this wave: every load issued before anything is consumed
    const int k0 = (wave + 4 * i) * GF_KS + lane * 8;
    // K % 64 == 0: a lane's 8 values are all in or all out.  Lanes beyond K (last slice of K = 11008) re-read the row's last
    // 16 B instead of branching around the load -- their x is zero, so the product is 0 whatever the (finite) weight is.
    const unsigned voff = (unsigned)min(k0, p.K - 8) * 2u;
    // read exactly once per step by exactly one wave: non-temporal (do not displace the KV cache / x in L2 / MALL)
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      w[r] = __builtin_nontemporal_load((const bf16x8*)((r < ROWS / 2 ? base_lo : base_hi) + (voff + (unsigned)(r % (ROWS / 2)) * ldwB)));
  };
  auto start_group = [&](int g) {  // request the first two slices of group g (and the first x slice)
    set_group(g);
    if (cnt > 0) { load(0, wA); load_x(0); }
    if (cnt > 1) load(1, wB);
  };
  int grp = blockIdx.x;
  set_group(grp);
  // The first group's two slices: requested by the prologue BEHIND its own first loads.  vmcnt retires in order, so an operand load
  // queued behind the weight slices cannot be used before every slice has landed -- rounds 4-5 started the slices first and the
  // RMSNorm prologue then ran AFTER the weights had arrived instead of in their shadow (found on the e4m3 stream, csrc/gemv_fp8.hip;
  // profiles/r06_gemv_w8_decomp.txt).  Unconditional (`load` clamps: a wave without a slice re-reads the row's last 16 B) so that the
  // compiler can COUNT the loads it may leave in flight.
  auto first_slices = [&]() {
    load(0, wA);
    load_x(0);
    load(1, wB);
  };

  // ---- prologue (x_mode 1 / 2), once per workgroup, in the shadow of the first weight loads
#ifdef GF_DIAG_NOPRO  // (tests/diag only: what the launches cost WITHOUT the normalising prologue -- a constant operand)
  if (!XG && p.x_mode == 1) {
    first_slices();
    for (int i = tid; i < MB * p.K / 2; i += 256) ((uint32_t*)xs)[i] = 0x38003800u;
    __syncthreads();
  } else
#endif
  if (!XG && p.x_mode == 1) {  // x = f16(gamma * (h * rsqrt(mean(h^2) + eps)))  (HF LlamaRMSNorm), 4 batch rows at a time
    constexpr int KJ = 4;      // float4 per thread and row held in registers (K <= 4096); wider rows take the two-pass form below
    if (p.K <= KJ * 1024) {
      // straight-line: every load of a 4-row chunk (16 x h, 4 x gamma per thread) is issued before the first use -- ONE exposed
      // L2 latency per chunk (a per-row "load gamma, wait, store" form measured 2.8 TB/s on the shapes with this prologue)
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
      auto chunk = [&](int m0, auto first) {
        f32x4 hv[4][KJ];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int j = 0; j < KJ; ++j) hv[mi][j] = *(const f32x4*)(p.h + (long)min(m0 + mi, p.M - 1) * p.ldh + cc[j]);
        if (decltype(first)::value) first_slices();
        float ss[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          ss[mi] = 0.f;
#pragma unroll
          for (int j = 0; j < KJ; ++j) {
            const f32x4 v = hv[mi][j];
            ss[mi] += cin[j] ? v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3] : 0.f;
          }
          ss[mi] = wave_sum(ss[mi]);
        }
        __syncthreads();  // (stat[] of the previous chunk has been read)
        if (lane == 0) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) stat[wave][mi] = ss[mi];
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int m = m0 + mi;
          const float rstd = m < p.M ? rsqrtf((stat[0][mi] + stat[1][mi] + stat[2][mi] + stat[3][mi]) / (float)p.K + p.eps) : 0.f;
#pragma unroll
          for (int j = 0; j < KJ; ++j) {
            const f32x4 o = g[j] * (hv[mi][j] * rstd);  // (padding rows m >= M: rstd = 0 -> x = 0)
            uint2 pk;
            pk.x = pack2bf(o[0], o[1]);
            pk.y = pack2bf(o[2], o[3]);
            if (cin[j]) *(uint2*)(xs + (long)m * p.K + cc[j]) = pk;
          }
        }
      };
      chunk(0, std::true_type{});
#pragma unroll 1
      for (int m0 = 4; m0 < MB; m0 += 4) chunk(m0, std::false_type{});
    } else {
      first_slices();
      for (int m = 0; m < MB; ++m) {  // wide rows (K > 4096): statistics pass, then a second read of the row
        float ss = 0.f;
        if (m < p.M)
          for (int c = tid * 4; c < p.K; c += 1024) {
            const f32x4 v = *(const f32x4*)(p.h + (long)m * p.ldh + c);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
          }
        ss = wave_sum(ss);
        __syncthreads();
        if (lane == 0) stat[wave][0] = ss;
        __syncthreads();
        const float rstd = m < p.M ? rsqrtf((stat[0][0] + stat[1][0] + stat[2][0] + stat[3][0]) / (float)p.K + p.eps) : 0.f;
        for (int c = tid * 4; c < p.K; c += 1024) {
          const f32x4 v = *(const f32x4*)(p.h + (long)min(m, p.M - 1) * p.ldh + c);
          const f32x4 o = *(const f32x4*)(p.gamma + c) * (v * rstd);
          uint2 pk;
          pk.x = pack2bf(o[0], o[1]);
          pk.y = pack2bf(o[2], o[3]);
          *(uint2*)(xs + (long)m * p.K + c) = pk;
        }
      }
    }
    __syncthreads();
  } else if (!XG && p.x_mode == 2) {  // merge the key slices of decode_attention (slice order; rounded like its nsplit = 1 output)
    first_slices();
    const int hd = p.a_hd, c8 = p.K >> 3;
    for (int idx = tid; idx < MB * c8; idx += 256) {
      const int m = idx / c8, k0 = (idx - m * c8) << 3;
      union { bf16x8 v; uint32_t u[4]; } pk;
      pk.v = zero8;
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
      *(bf16x8*)(xs + (long)m * p.K + k0) = pk.v;
    }
    __syncthreads();
  } else {
    first_slices();
  }

  float acc[ROWS][MB];
  auto consume = [&](int i, const bf16x8* w) {
    // raw 16-bit pairs straight into v_dot2c_f32_(bf16|f16): 4 instructions per (row, batch row) and slice, no conversions
    union X8 { bf16x8 v; uint32_t u[4]; };
    X8 xv[MB];
    const int k0 = (wave + 4 * i) * GF_KS + lane * 8;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if constexpr (XG) xv[m].v = xn[m];
      else xv[m].v = k0 < p.K ? *(const bf16x8*)(xs + (long)m * p.K + k0) : zero8;
    }
    if (i + 1 < cnt) load_x(i + 1);  // (L2-resident rows: one consume of lead time)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      X8 wv;
      wv.v = w[r];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        float a = acc[r][m];
#pragma unroll
        for (int e = 0; e < 4; ++e) a = GR_DOT2(wv.u[e], xv[m].u[e], a);
        acc[r][m] = a;
      }
    }
  };

#pragma unroll 1
  while (true) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
    // ---- this group's slices: two in flight per wave
    for (int i = 0; i < cnt; i += 2) {
      consume(i, wA);
      if (i + 2 < cnt) load(i + 2, wA);
      if (i + 1 < cnt) {
        consume(i + 1, wB);
        if (i + 3 < cnt) load(i + 3, wB);
      }
    }
    const int e_lo = r_lo, e_hi = r_hi;  // the finished group's rows (the epilogue's)
    const int nxt = grp + (int)gridDim.x;
    // (the next group's first slices are requested AFTER the reduction: requesting them before it, and doing the 63 lane exchanges of
    //  the butterfly with DPP / permlane swaps instead of ds_bpermute, were both built and measured 1.5 % SLOWER end to end --
    //  profiles/r04_decode_ab_rejected.txt; the co-resident workgroup covers the gap)

    // ---- NV per-lane partials x 64 lanes -> lane l (mod NV) owns dot product (row l / MB, batch row l % MB): all-lanes adds
    // across the lane bits >= NV, then a reduce-scatter butterfly; fixed order
    float v[NV];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) v[r * MB + m] = acc[r][m];
    if constexpr (NV <= 32) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], 32, 64);
    }
    constexpr int OFF0 = NV / 2 < 32 ? NV / 2 : 32;
    auto stage = [&](auto offc) {
      constexpr int off = decltype(offc)::value;
      if constexpr (off >= 1 && off <= OFF0) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int i = 0; i < off; ++i) {
          const float keep = hi ? v[i + off] : v[i];
          const float send = hi ? v[i] : v[i + off];
          v[i] = keep + __shfl_xor(send, off, 64);
        }
      }
    };
    stage(std::integral_constant<int, 32>{});
    stage(std::integral_constant<int, 16>{});
    stage(std::integral_constant<int, 8>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 1>{});
    if (lane < NV) red[wave][lane] = v[0];
    __syncthreads();
    if (wave == 0) {
      const int lv = lane & (NV - 1);
      const float val = ((red[0][lv] + red[1][lv]) + red[2][lv]) + red[3][lv];  // waves in a fixed order
      // ---- epilogue (wave 0; all 64 lanes stay active for the shuffles, lanes >= NV mirror the others and do not write)
      const int rr = lv / MB, m = lv % MB;
      const int n = rr < ROWS / 2 ? e_lo + rr : e_hi + (rr - ROWS / 2);
      const bool ok = lane < NV && m < p.M;
      if (p.epi == 0) {
        if (ok) ((float*)p.C)[(long)m * p.ldc + n] = val;
      } else if (p.epi == 1) {
        if (ok) p.resid[(long)m * p.ldr + n] += val;
      } else if (p.epi == 2) {  // interleaved rows: even = gate_j, odd = up_j ; partner row = lane ^ MB
        const float other = __shfl_xor(val, MB, 64);
        if (ok && (rr & 1) == 0) ((bf16_t*)p.C)[(long)m * p.ldc + (n >> 1)] = f2bf(silu_f(val) * other);
      } else {  // epi 3: the summed projection is rounded to 16 bits first (what the prefill GEMM stores), then rotate_half in f32
        // all rows of a group lie in ONE head (HD % ROWS == 0): section, head and the first dim come from the group's first row --
        // wave-uniform scalars, no per-lane division
        const int HHD = p.H * p.HD;
        const int sect = e_lo / HHD, nn = e_lo - sect * HHD;
        const int hh = nn / p.HD;
        const int d = (rr < ROWS / 2 ? e_lo + rr : e_hi + (rr - ROWS / 2)) - sect * HHD - hh * p.HD;
        const int HALF = p.HD / 2;
        const float a = bf2f(f2bf(val));
        const float partner = __shfl_xor(a, (ROWS / 2) * MB, 64);  // row r <-> r + ROWS/2 = d <-> d + HD/2 (q / k sections)
        if (ok) {
          const int pos = p.pos_dev ? p.pos_dev[m * p.pos_stride] : p.pos0;
          const long bh = (long)m * p.H + hh;
          if (sect == 2) {
            p.vt[(bh * p.HD + d) * p.kv_stride + pos] = f2bf(val);
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
    if (nxt >= ngroups) break;
    grp = nxt;
    start_group(nxt);
    __syncthreads();  // wave 0 has read red[] before the next group's partials overwrite it
  }
}

// bench.py's HIP-event hook (gemm_bf16.hip): tag 8 = decode-step weight stream, 32 = the fused kernel
int gr_prof_begin(hipStream_t stream, int M, int N, int K, int tag);
void gr_prof_end(hipStream_t stream, int idx);

extern "C" int gr_gemv_fused(const gr_gemv_desc* d, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;  // no split-operand form: a decode step of the reference-precision build runs the general kernels
  if (!d || !d->W || d->M <= 0 || d->M > 8 || d->N <= 0 || d->K <= 0 || d->K % 64 != 0 || d->ldw < d->K) return GR_EINVAL;
  if (d->x_mode < 0 || d->x_mode > 2 || d->epi < 0 || d->epi > 3) return GR_EINVAL;
  if (d->N % GF_ROWS != 0 || (long)d->ldw * 2 * GF_ROWS >= (1L << 31)) return GR_EINVAL;  // whole row groups; 32-bit row offsets
  if (d->x_mode == 0 && (!d->A || d->lda < d->K)) return GR_EINVAL;
  if (d->x_mode == 1 && (!d->h || !d->gamma || d->K % 4 != 0 || d->ldh < d->K)) return GR_EINVAL;
  if (d->x_mode == 2 && (!d->a_parts || d->a_nsplit < 1 || d->a_hd < 8 || d->a_hd % 8 != 0 || d->K % d->a_hd != 0)) return GR_EINVAL;
  if (d->epi == 0 && (!d->C || d->ldc < d->N)) return GR_EINVAL;
  if (d->epi == 1 && (!d->resid || d->ldr < d->N)) return GR_EINVAL;
  if (d->epi == 2 && (!d->C || d->N % 2 != 0 || d->ldc < d->N / 2)) return GR_EINVAL;
  const int MB = d->M <= 4 ? 4 : 8, ROWS = GF_ROWS;
  if (d->epi == 3) {
    if (!d->q || !d->kc || !d->vt || d->H <= 0 || d->HD <= 0 || d->HD % (2 * ROWS) != 0 || d->N != 3 * d->H * d->HD) return GR_EINVAL;
    if ((d->cosT == nullptr) != (d->sinT == nullptr) || d->kv_stride <= 0 || (!d->pos_dev && (d->pos0 < 0 || d->pos0 >= d->kv_stride)))
      return GR_EINVAL;
  }
  const bool w8 = d->w8 != 0;
  if (w8 && (!d->w_scale || d->K % 128 != 0 || d->N % 16 != 0 || (d->x_mode == 1 && d->K > 4096))) return GR_EINVAL;
  const size_t lds = (d->x_mode == 0 || w8) ? 0 : (size_t)MB * d->K * sizeof(bf16_t);
  if (lds > 128 * 1024) return GR_EINVAL;
  GemvFArgs p;
  p.W = (const bf16_t*)d->W; p.ldw = d->ldw; p.M = d->M; p.N = d->N; p.K = d->K;
  p.x_mode = d->x_mode; p.A = (const bf16_t*)d->A; p.lda = d->lda; p.h = d->h; p.ldh = d->ldh; p.gamma = d->gamma; p.eps = d->eps;
  p.a_parts = d->a_parts; p.a_nsplit = d->a_nsplit; p.a_hd = d->a_hd;
  p.epi = d->epi; p.C = d->C; p.ldc = d->ldc; p.resid = d->resid; p.ldr = d->ldr;
  p.q = (bf16_t*)d->q; p.kc = (bf16_t*)d->kc; p.vt = (bf16_t*)d->vt; p.cosT = d->cosT; p.sinT = d->sinT;
  p.H = d->H; p.HD = d->HD; p.pos0 = d->pos0; p.kv_stride = d->kv_stride; p.pos_dev = d->pos_dev; p.pos_stride = d->pos_stride;
  p.w_scale = d->w_scale;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemv_fused_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemv_fused_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
      return GR_EINVAL;
    attr_set = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GR_EINVAL;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int groups = gr_cdiv(d->N, ROWS);
  // persistent over row groups.  (An even split -- grid = groups / rounds -- was measured in round 6: the e4m3 stream gains from it,
  // gemv_fp8.hip; here gate/up loses 35.5 -> 38.1 us at 4 rows, profiles/r06_gemv_grid.txt.)
  const dim3 grid(groups < GF_WG_PER_CU * n_cu ? groups : GF_WG_PER_CU * n_cu);
  if (w8) {  // e4m3 weights: the MFMA stream (gemv_fp8.hip)
    if (d->epi == 3 && d->HD % 32 != 0) return GR_EINVAL;
    const int prof8 = gr_prof_begin(stream, d->M, d->N, d->K, 8 | 16 | 32);
    const int rc = gr_launch_gemv_fp8(p, MB, n_cu, stream);
    gr_prof_end(stream, prof8);
    if (rc != GR_OK) return rc;
    GR_CHECK_LAUNCH();
    return GR_OK;
  }
  const int prof = gr_prof_begin(stream, d->M, d->N, d->K, 8 | 32);
  if (d->x_mode == 0) {
    if (MB == 4) hipLaunchKernelGGL((gemv_fused_kernel<4, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemv_fused_kernel<8, true>), grid, dim3(256), 0, stream, p);
  } else {
    if (MB == 4) hipLaunchKernelGGL((gemv_fused_kernel<4, false>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((gemv_fused_kernel<8, false>), grid, dim3(256), lds, stream, p);
  }
  gr_prof_end(stream, prof);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
