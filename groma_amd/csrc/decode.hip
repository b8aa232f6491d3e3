// Decode-step attention (one new token per sequence row, M = rows <= 8): SURVEY 8a row a22.
// A decode step streams 13.2 GB of 16-bit weights (HBM-bound: gemv_fused.hip, one kernel per weight matrix with its producer and
// consumer fused in); what is left between the weight streams is this kernel: the 128-query-row MFMA attention tile (1 useful
// row) replaced by a one-block-per-(row, head) dot-product kernel, optionally over key slices that the o-proj stream merges.
//   gr_decode_attention   : softmax(q K^T * scale) V over the cache, one block per (row, head[, key slice])
// Positions come from device memory when pos_dev is given (hipGraph replay / ragged continuous batching).
#include "gr_common.h"
#include "../../include/groma_hip.h"

// ------------------------------------------------------------------------------------------------
// Single-query attention over the cache: one 1024-thread block per (row b, head h).
//   scores : 16 lanes per key (16 B = 8 dims each; a wave instruction reads 4 consecutive K rows = 1 KiB contiguous)
//   softmax: fp32, scores kept in LDS (S <= DEC_SMAX)
//   P.V    : V is cached TRANSPOSED [hd, kv_stride], so out[d] is a dot along contiguous keys: 1024/HD threads per d,
//            16-B loads, reduced with cross-lane adds
// Key s is visible iff s < S_b, S_b = (pos_dev ? pos_dev[b*stride] : q_pos0) + 1, and s < kv_len[b] if given.
#define DEC_SMAX 8192
#define DEC_U 11  // chunks in flight per thread and round (K and V^T): 11 x 64 = 704 keys per round at hd 128; 12 spills

// DS = blocks per (row, head[, key slice]) along the OUTPUT dims (round 4): every block computes all scores and the soft-max of its
// keys, but only HD / DS rows of V^T.  At 4 rows x 32 heads that is 256 blocks on 256 CUs instead of 128, each streaming 224 KB
// instead of 300 KB, and -- unlike key slices -- nothing to merge: a block writes its dims of the context directly.
template <int HD, int DS>
__global__ __launch_bounds__(1024) void decode_attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                               const bf16_t* __restrict__ vt, bf16_t* __restrict__ out,
                                                               const int* __restrict__ kv_len, int H, int kv_stride,
                                                               int q_pos0, float scale, const int* __restrict__ pos_dev,
                                                               int pos_stride, float* __restrict__ ws) {
  extern __shared__ float sc[];  // [round_up(S, 64)] scores -> probabilities
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / H;
  const int nsplit = gridDim.y, z = blockIdx.y;
  int Sall = (pos_dev ? pos_dev[b * pos_stride] : q_pos0) + 1;
  if (kv_len) Sall = min(Sall, kv_len[b]);
  // this block's slice of the keys (64-aligned start so the 16-B V^T loads stay aligned)
  const int chunk = (((Sall + nsplit - 1) / nsplit) + 63) & ~63;
  const int sb = z * chunk;
  const int S = max(0, min(Sall, sb + chunk) - sb);
  const int Spad = (S + 63) & ~63;
  const bf16_t* Kp = k + ((long)bh * kv_stride + sb) * HD;
  const bf16_t* Vp = vt + (long)bh * HD * kv_stride + sb;

  // All global loads of a round are issued before anything consumes them (one exposed memory latency per round of
  // DEC_U x 64 keys, for K and for V^T together): a decode block is a pure latency chain otherwise.
  constexpr int LPK = HD / 8;          // lanes per key (16 for hd 128, 8 for hd 64)
  constexpr int KPI = 1024 / LPK;      // keys per pass of the block
  constexpr int HDB = HD / DS;         // output dims of this block
  constexpr int TPD = 1024 / HDB;      // threads per output dim (8 / 16; twice that with DS = 2)
  const int j = tid % LPK, g = tid / LPK;
  const int d = (int)blockIdx.z * HDB + tid / TPD, part = tid % TPD;
  const bf16_t* vrow = Vp + (long)d * kv_stride;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  float qf[8];
  {
    const bf16x8 qv = *(const bf16x8*)(q + (long)bh * HD + j * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = bf2f((bf16_t)qv[e]) * scale;
  }
  bf16x8 vv[DEC_U];  // V^T chunks of round 0 (keys part*8 + u*TPD*8 ..+8), loaded alongside the K rows
#pragma unroll
  for (int u = 0; u < DEC_U; ++u) {
    const int s0 = part * 8 + u * TPD * 8;
    vv[u] = s0 < Spad ? *(const bf16x8*)(vrow + s0) : zero8;
  }
  float mloc = -1e30f;
  for (int r0 = 0; r0 < Spad; r0 += KPI * DEC_U) {
    bf16x8 kv[DEC_U];
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int s = r0 + u * KPI + g;
      kv[u] = s < S ? *(const bf16x8*)(Kp + (long)s * HD + j * 8) : zero8;
    }
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int s = r0 + u * KPI + g;
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += qf[e] * bf2f((bf16_t)kv[u][e]);
#pragma unroll
      for (int o = LPK / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
      if (j == 0 && s < Spad) {
        sc[s] = s < S ? dot : -1e30f;
        if (s < S) mloc = fmaxf(mloc, dot);
      }
    }
  }
  // ---- block max
  mloc = wave_max(mloc);
  if (lane == 0) red[wave] = mloc;
  __syncthreads();
  float mx = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
  // ---- probabilities + sum
  float lsum = 0.f;
  for (int s = tid; s < Spad; s += 1024) {
    const float p = s < S ? __expf(sc[s] - mx) : 0.f;
    sc[s] = p;
    lsum += p;
  }
  const float denom = block_sum(lsum, red);  // (syncs: sc[] complete)
  // ---- P.V along contiguous keys of V^T
  float acc = 0.f;
  auto pv = [&](const bf16x8 v8, int s0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float p = sc[s0 + e];
      acc += p > 0.f ? p * bf2f((bf16_t)v8[e]) : 0.f;  // p == 0 beyond S: never multiply stale cache bytes (NaN-safe)
    }
  };
#pragma unroll
  for (int u = 0; u < DEC_U; ++u) {
    const int s0 = part * 8 + u * TPD * 8;
    if (s0 < Spad) pv(vv[u], s0);
  }
  for (int r0 = DEC_U * TPD * 8; r0 < Spad; r0 += DEC_U * TPD * 8) {  // long caches: further rounds
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int s0 = r0 + part * 8 + u * TPD * 8;
      vv[u] = s0 < Spad ? *(const bf16x8*)(vrow + s0) : zero8;
    }
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int s0 = r0 + part * 8 + u * TPD * 8;
      if (s0 < Spad) pv(vv[u], s0);
    }
  }
#pragma unroll
  for (int o = TPD / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (nsplit == 1) {
    if (part == 0) out[(long)bh * HD + d] = f2bf(acc / denom);
    return;
  }
  // ---- split keys: publish (un-normalised o, max, sum) of this slice; the o-proj GEMV merges the slices in slice
  // order while loading its operand (gemv_bf16.hip) -- kernel boundaries provide the visibility, no fences here
  float* wo = ws + ((long)bh * nsplit + z) * (HD + 2);
  if (part == 0) wo[d] = acc;
  if (tid == 0 && blockIdx.z == 0) { wo[HD] = mx; wo[HD + 1] = denom; }
}

extern "C" int gr_decode_attention(const void* q, const void* k, const void* vt, void* out, const int* kv_len, int B, int H,
                                   int Smax, int kv_stride, int head_dim, int q_pos0, float scale, const int* pos_dev,
                                   int pos_stride, int nsplit, float* parts, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;  // the streaming decode / e4m3 kernels do not exist in the split-operand build (gr_common.h)
  if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Smax <= 0 || Smax > DEC_SMAX) return GR_EINVAL;
  if (nsplit < 1 || nsplit > 16 || (nsplit > 1 && !parts)) return GR_EINVAL;
  if (kv_stride % 64 != 0 || kv_stride < Smax || (!pos_dev && q_pos0 + 1 > Smax)) return GR_EINVAL;
  const size_t lds = (size_t)((Smax + 63) & ~63) * sizeof(float);
#ifndef DA_DSPLIT_MAX_BLOCKS
#define DA_DSPLIT_MAX_BLOCKS 128
#endif
  const int ds = B * H * nsplit <= DA_DSPLIT_MAX_BLOCKS ? 2 : 1;  // fewer blocks than half the CUs: two per (row, head) along the output dims
#define LAUNCH_DA(HDV, DSV)                                                                                                    \
  hipLaunchKernelGGL((decode_attention_kernel<HDV, DSV>), dim3(B * H, nsplit, DSV), dim3(1024), lds, stream, (const bf16_t*)q, \
                     (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, kv_len, H, kv_stride, q_pos0, scale, pos_dev,         \
                     pos_stride, parts)
  if (head_dim == 128) { if (ds == 2) LAUNCH_DA(128, 2); else LAUNCH_DA(128, 1); }
  else if (head_dim == 64) { if (ds == 2) LAUNCH_DA(64, 2); else LAUNCH_DA(64, 1); }
  else return GR_EINVAL;
#undef LAUNCH_DA
  GR_CHECK_LAUNCH();
  return GR_OK;
}
