// Layout / packing kernels around the GEMMs (all HBM-bound, 16-B vector accesses).
// Each cites the reference statement whose data movement it performs.
#include "gr_common.h"
#include "../../include/groma_hip.h"

// ---------------------------------------------------------------------------------------
// QKV split (+ RoPE) : qkv bf16 [B*L, 3*H*hd] (q|k|v blocks, HF q_proj/k_proj/v_proj order)
//   -> q  [B,H,L,hd]
//   -> k  [B,H,kv_stride,hd]   rows pos0..pos0+L-1        (KV cache, post-RoPE)
//   -> vt [B,H,hd,kv_stride]   columns pos0..pos0+L-1     (V cache, transposed)
// RoPE as HF LLaMA rotate_half (transformers 4.32 modeling_llama.apply_rotary_pos_emb):
//   x'[d] = x[d]*cos[d] - x[d+hd/2]*sin[d]        (d <  hd/2)
//   x'[d] = x[d]*cos[d-hd/2] + x[d-hd/2]*sin[d-hd/2] (d >= hd/2)
// One block = 64 tokens x one head; V goes through an LDS transpose so Vt rows are written
// as 128-B segments.
#if GR_SP
// Split-operand build (gr_common.h): same work items, values carried as fp32 (hi + lo) between the helper loads and stores;
// RoPE is evaluated on the reconstructed 22-bit values.  (Separate body so the tuned 16-bit kernel below stays as it is.)
template <int HD>
__global__ __launch_bounds__(256) void qkv_split_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ q,
                                                        bf16_t* __restrict__ k, bf16_t* __restrict__ vt,
                                                        const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                        int B, int H, int L, int pos0_arg, int kv_stride,
                                                        const int* __restrict__ pos_dev, int pos_stride) {
  __shared__ float vs[64][HD + 1];
  const int t0 = blockIdx.x * 64;
  const int h = blockIdx.y, b = blockIdx.z;
  const int pos0 = pos_dev ? pos_dev[b * pos_stride] : pos0_arg;
  const int tid = threadIdx.x;
  const long row_stride = 3L * H * HD;
  constexpr int HALF = HD / 2;
  constexpr int CH = HD / 8;
  for (int it = tid; it < 64 * CH; it += 256) {
    const int tt = it / CH, d = (it - tt * CH) * 8;
    const int t = t0 + tt;
    if (t >= L) {
#pragma unroll
      for (int i = 0; i < 8; ++i) vs[tt][d + i] = 0.f;
      continue;
    }
    const long src = ((long)b * L + t) * row_stride + h * HD;  // logical flat index of (token, head) in the q block
    const int pos = pos0 + t;
    float v8[8];
    ld8f(qkv, src + 2L * H * HD + d, v8);
#pragma unroll
    for (int i = 0; i < 8; ++i) vs[tt][d + i] = v8[i];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      if (which == 0 && !q) continue;
      const long sp = src + (long)which * H * HD;
      bf16_t* dst = which == 0 ? q : k;
      const long di = which == 0 ? (((long)b * H + h) * L + t) * HD : (((long)b * H + h) * kv_stride + pos) * HD;
      float x[8];
      ld8f(qkv, sp + d, x);
      if (cosT) {
        const int dp = d < HALF ? d + HALF : d - HALF;
        float y[8];
        ld8f(qkv, sp + dp, y);
        const int dc = d < HALF ? d : d - HALF;
        const float sgn = d < HALF ? -1.f : 1.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          x[i] = x[i] * cosT[(long)pos * HALF + dc + i] + sgn * y[i] * sinT[(long)pos * HALF + dc + i];
      }
      st8f(dst, di + d, x);
    }
  }
  __syncthreads();
  const bool aligned = ((pos0 + t0) & 7) == 0;
  for (int it = tid; it < HD * 8; it += 256) {
    const int d = it >> 3, seg = it & 7;
    const int tb = t0 + seg * 8;
    if (tb >= L) continue;
    const long di = (((long)b * H + h) * HD + d) * kv_stride + pos0 + tb;
    if (aligned && tb + 8 <= L) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = vs[seg * 8 + i][d];
      st8f(vt, di, o);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (tb + i < L) st1f(vt, di + i, vs[seg * 8 + i][d]);
    }
  }
}
#else
template <int HD>
__global__ __launch_bounds__(256) void qkv_split_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ q,
                                                        bf16_t* __restrict__ k, bf16_t* __restrict__ vt,
                                                        const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                        int B, int H, int L, int pos0_arg, int kv_stride,
                                                        const int* __restrict__ pos_dev, int pos_stride) {
  __shared__ __attribute__((aligned(16))) bf16_t vs[64][HD + 8];
  const int t0 = blockIdx.x * 64;
  const int h = blockIdx.y, b = blockIdx.z;
  // device-resident write position (graph-captured / ragged decode): row b appends at pos_dev[b * pos_stride]
  const int pos0 = pos_dev ? pos_dev[b * pos_stride] : pos0_arg;
  const int tid = threadIdx.x;
  const long row_stride = 3L * H * HD;
  constexpr int HALF = HD / 2;
  constexpr int CH = HD / 8;  // 16-B chunks per head row
  // ---- q, k (+ RoPE) and v -> LDS: one 16-B chunk per work item ----
  for (int it = tid; it < 64 * CH; it += 256) {
    const int tt = it / CH, d = (it - tt * CH) * 8;
    const int t = t0 + tt;
    if (t >= L) {
      *(bf16x8*)&vs[tt][d] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      continue;
    }
    const bf16_t* src = qkv + ((long)b * L + t) * row_stride + h * HD;
    const int pos = pos0 + t;
    *(bf16x8*)&vs[tt][d] = *(const bf16x8*)(src + 2L * H * HD + d);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const bf16_t* sp = src + (long)which * H * HD;
      if (which == 0 && !q) continue;  // q stays in the fused buffer: gr_attention_bf16 reads (and rotates) it there
      bf16_t* dst = which == 0 ? q + (((long)b * H + h) * L + t) * HD : k + (((long)b * H + h) * kv_stride + pos) * HD;
      const bf16x8 x = *(const bf16x8*)(sp + d);
      if (cosT) {
        const int dp = d < HALF ? d + HALF : d - HALF;  // rotate_half partner
        const bf16x8 y = *(const bf16x8*)(sp + dp);
        const int dc = d < HALF ? d : d - HALF;
        const float sgn = d < HALF ? -1.f : 1.f;
        const f32x4 c0 = *(const f32x4*)(cosT + (long)pos * HALF + dc), c1 = *(const f32x4*)(cosT + (long)pos * HALF + dc + 4);
        const f32x4 s0 = *(const f32x4*)(sinT + (long)pos * HALF + dc), s1 = *(const f32x4*)(sinT + (long)pos * HALF + dc + 4);
        union { bf16x8 v; uint32_t u[4]; } o;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          const float ca = i < 4 ? c0[i] : c1[i - 4], cb = i < 4 ? c0[i + 1] : c1[i - 3];
          const float sa = i < 4 ? s0[i] : s1[i - 4], sb = i < 4 ? s0[i + 1] : s1[i - 3];
          const float r0 = bf2f((bf16_t)x[i]) * ca + sgn * bf2f((bf16_t)y[i]) * sa;
          const float r1 = bf2f((bf16_t)x[i + 1]) * cb + sgn * bf2f((bf16_t)y[i + 1]) * sb;
          o.u[i >> 1] = pack2bf(r0, r1);
        }
        *(bf16x8*)(dst + d) = o.v;
      } else {
        *(bf16x8*)(dst + d) = x;
      }
    }
  }
  __syncthreads();
  // ---- transposed write of V: work item = (d, 8-token group) -> one 16-B store when aligned ----
  const bool aligned = ((pos0 + t0) & 7) == 0;
  for (int it = tid; it < HD * 8; it += 256) {
    const int d = it >> 3, seg = it & 7;
    const int tb = t0 + seg * 8;
    if (tb >= L) continue;
    bf16_t* dst = vt + (((long)b * H + h) * HD + d) * kv_stride + pos0 + tb;
    if (aligned && tb + 8 <= L) {
      union { bf16x8 v; bf16_t e[8]; } o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o.e[i] = vs[seg * 8 + i][d];
      *(bf16x8*)dst = o.v;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (tb + i < L) dst[i] = vs[seg * 8 + i][d];
    }
  }
}

#endif  // GR_SP

extern "C" int gr_qkv_split(const void* qkv, void* q, void* k, void* vt, const float* cosT, const float* sinT, int B,
                            int H, int L, int head_dim, int pos0, int kv_stride, const int* pos_dev, int pos_stride,
                            hipStream_t stream) {
  if (!qkv || !k || !vt || B <= 0 || H <= 0 || L <= 0) return GR_EINVAL;  // q may be NULL (left in qkv)
  if ((cosT == nullptr) != (sinT == nullptr)) return GR_EINVAL;
  if (GR_SP && kv_stride % 32 != 0) return GR_EINVAL;
  dim3 grid(gr_cdiv(L, 64), H, B);
  if (head_dim == 128)
    hipLaunchKernelGGL(qkv_split_kernel<128>, grid, dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)q, (bf16_t*)k,
                       (bf16_t*)vt, cosT, sinT, B, H, L, pos0, kv_stride, pos_dev, pos_stride);
  else if (head_dim == 64)
    hipLaunchKernelGGL(qkv_split_kernel<64>, grid, dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)q, (bf16_t*)k,
                       (bf16_t*)vt, cosT, sinT, B, H, L, pos0, kv_stride, pos_dev, pos_stride);
  else return GR_EINVAL;
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// Patchify for the DINOv2 patch-embedding conv (kernel = stride = P): images f32 NCHW
// [B,3,S,S] -> A bf16 [B*G*G, Kpad], k = c*P*P + ky*P + kx (the Conv2d weight flattening),
// zero padded to Kpad.   (HF Dinov2PatchEmbeddings.projection)
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int S, int P, int G,
                                int Kpad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * G * G * Kpad;
  if (idx >= total) return;
  const int kk = (int)(idx % Kpad);
  const long m = idx / Kpad;
  const int K = 3 * P * P;
  float v = 0.f;
  if (kk < K) {
    const int c = kk / (P * P), r = kk % (P * P);
    const int ky = r / P, kx = r % P;
    const int gx = (int)(m % G), gy = (int)((m / G) % G), b = (int)(m / ((long)G * G));
    v = img[(((long)b * 3 + c) * S + gy * P + ky) * S + gx * P + kx];
  }
  st1f(out, idx, v);
}

extern "C" int gr_patchify(const float* images, void* out, int B, int S, int P, int Kpad, hipStream_t stream) {
  if (!images || !out || B <= 0 || S % P != 0 || Kpad < 3 * P * P || (GR_SP && Kpad % 32 != 0)) return GR_EINVAL;
  const int G = S / P;
  const long total = (long)B * G * G * Kpad;
  hipLaunchKernelGGL(patchify_kernel, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, images, (bf16_t*)out, B, S, P, G,
                     Kpad);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// dst[r*ld_dst .. +C] = src[0..C) for r in [0,rows)  (CLS+pos[0] row of every image)
__global__ void fill_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int C, long ld_dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * C) return;
  const int c = (int)(idx % C);
  const long r = idx / C;
  dst[r * ld_dst + c] = src[c];
}
extern "C" int gr_fill_rows_f32(const float* src, float* dst, int rows, int C, long ld_dst, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || C <= 0) return GR_EINVAL;
  hipLaunchKernelGGL(fill_rows_kernel, dim3(gr_cdiv((long)rows * C, 256)), dim3(256), 0, stream, src, dst, rows, C,
                     ld_dst);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// mean of the last 4 ViT hidden states, CLS dropped (groma/model/groma.py:240-241):
// h[i]: f32 [B, T, C] -> out f32 [B*(T-1), C]
__global__ void mean4_kernel(const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ h2,
                             const float* __restrict__ h3, float* __restrict__ out, int B, int T, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  const int c4 = C >> 2;
  const long total = (long)B * (T - 1) * c4;
  if (idx >= total) return;
  const int c = (int)(idx % c4) << 2;
  const long m = idx / c4;
  const int b = (int)(m / (T - 1)), tkn = (int)(m % (T - 1));
  const long off = ((long)b * T + 1 + tkn) * C + c;
  f32x4 s = *(const f32x4*)(h0 + off);
  s += *(const f32x4*)(h1 + off);
  s += *(const f32x4*)(h2 + off);
  s += *(const f32x4*)(h3 + off);
  *(f32x4*)(out + m * C + c) = s * 0.25f;
}
extern "C" int gr_mean4_tokens(const float* h0, const float* h1, const float* h2, const float* h3, float* out, int B,
                               int T, int C, hipStream_t stream) {
  if (!h0 || !h1 || !h2 || !h3 || !out || C % 4 != 0) return GR_EINVAL;
  const long total = (long)B * (T - 1) * (C >> 2);
  hipLaunchKernelGGL(mean4_kernel, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, h0, h1, h2, h3, out, B, T, C);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// 2x2 space-to-depth of the last hidden state (groma/model/groma.py:227-237):
// h f32 [B, 1+G*G, C] -> out bf16 [B*(G/2)^2, 4C], channel blocks (0::2,0::2),(1::2,0::2),(0::2,1::2),(1::2,1::2)
__global__ void s2d_kernel(const float* __restrict__ h, bf16_t* __restrict__ out, int B, int G, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 4-element index
  const int c4 = C;  // 4C/4
  const int G2 = G >> 1;
  const long total = (long)B * G2 * G2 * c4;
  if (idx >= total) return;
  const int cc = (int)(idx % c4) << 2;  // channel in [0,4C)
  const long m = idx / c4;
  const int j = (int)(m % G2), i = (int)((m / G2) % G2), b = (int)(m / ((long)G2 * G2));
  const int blk = cc / C, c = cc % C;
  const int dy = blk & 1, dx = blk >> 1;  // blocks: (0,0),(1,0),(0,1),(1,1) as (row offset, col offset)
  const int y = 2 * i + dy, x = 2 * j + dx;
  const f32x4 v = *(const f32x4*)(h + ((long)b * (1 + G * G) + 1 + y * G + x) * C + c);
  st4f(out, m * 4 * C + cc, v);
}
extern "C" int gr_s2d_pack(const float* h, void* out, int B, int G, int C, hipStream_t stream) {
  if (!h || !out || G % 2 != 0 || C % 4 != 0 || (GR_SP && C % 32 != 0)) return GR_EINVAL;
  const long total = (long)B * (G / 2) * (G / 2) * C;
  hipLaunchKernelGGL(s2d_kernel, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, h, (bf16_t*)out, B, G, C);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// Region-encoder input: bilinear (align_corners=True) upsample of one ViT hidden state from
// GxG to HoxHo (groma/model/roi_align.py:220-227) + the two coordinate channels x,y in [-1,1]
// (roi_align.py:118-126,181-187), packed as the A operand of the 1x1 input conv:
// h f32 [B, 1+G*G, C] -> out bf16 [B*Ho*Ho, Cpad]   (channels C..C+1 = x,y ; rest zero)
__global__ void upsample_coord_kernel(const float* __restrict__ h, bf16_t* __restrict__ out, int B, int G, int Ho, int C,
                                      int Cpad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 4-channel index
  const int c4 = Cpad >> 2;
  const long total = (long)B * Ho * Ho * c4;
  if (idx >= total) return;
  const int c = (int)(idx % c4) << 2;
  const long m = idx / c4;
  const int x = (int)(m % Ho), y = (int)((m / Ho) % Ho), b = (int)(m / ((long)Ho * Ho));
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    // align_corners=True source coordinate (torch area_pixel_compute_source_index)
    const float sc = Ho > 1 ? (float)(G - 1) / (float)(Ho - 1) : 0.f;
    const float sy = sc * y, sx = sc * x;
    int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < G - 1 ? 1 : 0), x1 = x0 + (x0 < G - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* base = h + ((long)b * (1 + G * G) + 1) * C + c;
    const f32x4 v00 = *(const f32x4*)(base + (long)(y0 * G + x0) * C);
    const f32x4 v01 = *(const f32x4*)(base + (long)(y0 * G + x1) * C);
    const f32x4 v10 = *(const f32x4*)(base + (long)(y1 * G + x0) * C);
    const f32x4 v11 = *(const f32x4*)(base + (long)(y1 * G + x1) * C);
    v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
  } else if (c == C) {
    // torch.linspace(-1, 1, n): start + i*step for the first half, end - (n-1-i)*step for the second
    const float step = Ho > 1 ? 2.f / (float)(Ho - 1) : 0.f;
    v[0] = x < Ho / 2 ? -1.f + step * x : 1.f - step * (Ho - 1 - x);
    v[1] = y < Ho / 2 ? -1.f + step * y : 1.f - step * (Ho - 1 - y);
  }
  st4f(out, m * Cpad + c, v);
}
extern "C" int gr_upsample_coord_pack(const float* h, void* out, int B, int G, int Ho, int C, int Cpad,
                                      hipStream_t stream) {
  if (!h || !out || C % 4 != 0 || Cpad % 4 != 0 || Cpad < C + 2 || (GR_SP && Cpad % 32 != 0)) return GR_EINVAL;
  const long total = (long)B * Ho * Ho * (Cpad >> 2);
  hipLaunchKernelGGL(upsample_coord_kernel, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, h, (bf16_t*)out, B, G, Ho,
                     C, Cpad);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics of a conv output: x bf16 [imgs*HW, C] -> per-row-block partial sums f32 [imgs, nblk, C, 2]
// (sum, sumsq).  No atomics: the finalize kernel adds the nblk partials in a fixed order, so the statistics -- and
// everything downstream -- are bit-reproducible run to run and independent of the batch composition.
// (mmcv ConvModule norm = GN(64 groups), mmcv/cnn/bricks/conv_module.py:196-206)
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ sums, int HW,
                                                       int C, int rows_per_block) {
  const int img = blockIdx.z;
  const int cblk = blockIdx.y;  // 64-channel slab
  const int r0 = blockIdx.x * rows_per_block;
  const int tid = threadIdx.x;
  const int ch = (tid & 7) * 8;  // 8 channels per thread
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
  const int rend = min(HW, r0 + rows_per_block);
  for (int r = r0 + (tid >> 3); r < rend; r += 32) {
    float v[8];
    ld8f(x, ((long)img * HW + r) * C + cblk * 64 + ch, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s[i] += v[i];
      q[i] += v[i] * v[i];
    }
  }
  // reduce over the 32 row-lanes sharing (tid&7): shuffle within wave over bits 3..5, then LDS across 4 waves
  __shared__ float red[4][8][16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      s[i] += __shfl_xor(s[i], o, 64);
      q[i] += __shfl_xor(q[i], o, 64);
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[wave][lane][i] = s[i];
      red[wave][lane][8 + i] = q[i];
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int cc = tid >> 1, which = tid & 1;  // channel within slab 0..63
    const int l8 = cc >> 3, i = cc & 7;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += red[w][l8][which * 8 + i];
    sums[(((long)img * gridDim.x + blockIdx.x) * C + cblk * 64 + cc) * 2 + which] = t;
  }
}
extern "C" int gr_gn_stats_blocks(int HW) { return gr_cdiv(HW, 512); }
extern "C" int gr_gn_stats(const void* x, float* sums, int imgs, int HW, int C, hipStream_t stream) {
  if (!x || !sums || C % 64 != 0) return GR_EINVAL;
  const int rpb = 512;
  dim3 grid(gr_cdiv(HW, rpb), C / 64, imgs);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, stream, (const bf16_t*)x, sums, HW, C, rpb);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// sums [imgs, C, 2] -> coef [imgs, 2, C]: y = x*a[c] + b[c]  (a = rstd*gamma, b = beta - mean*rstd*gamma);
// torch.group_norm semantics: biased variance over (HW x C/groups) elements, eps inside the sqrt.
// One wave per (image, group): lane l adds the partials e = l, l + 64, ... (e = block * cpg + channel-in-group) in that fixed
// order, then the wave reduction combines the 64 lane sums in a fixed butterfly -- deterministic and batch-independent like
// before, but ~nblk * cpg / 64 dependent loads per lane instead of nblk * cpg per thread (and no 16-fold redundant re-summation
// by the channels of a group): 58 us -> a few us per launch at 128^2 x 1024 channels.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ coef, int C,
                                                         int cpg, int nblk, float n, float eps) {
  const int groups = C / cpg;
  const int img = blockIdx.x / groups, g0 = (blockIdx.x - img * groups) * cpg;
  const int lane = threadIdx.x;
  float sm = 0.f, sq = 0.f;
  for (int e = lane; e < nblk * cpg; e += 64) {
    const int bk = e / cpg, k = e - bk * cpg;
    const float2 v = *(const float2*)(sums + (((long)img * nblk + bk) * C + g0 + k) * 2);
    sm += v.x;
    sq += v.y;
  }
  sm = wave_sum(sm);
  sq = wave_sum(sq);
  const float mean = sm / n;
  const float var = fmaxf(sq / n - mean * mean, 0.f);
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int k = lane; k < cpg; k += 64) {
    const int ch = g0 + k;
    const float a = rstd * gamma[ch];
    coef[((long)img * 2 + 0) * C + ch] = a;
    coef[((long)img * 2 + 1) * C + ch] = beta[ch] - mean * a;
  }
}
extern "C" int gr_gn_finalize(const float* sums, const float* gamma, const float* beta, float* coef, int imgs, int HW,
                              int C, int groups, float eps, hipStream_t stream) {
  if (!sums || !gamma || !beta || !coef || groups <= 0 || C % groups != 0) return GR_EINVAL;
  const int cpg = C / groups;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(imgs * groups), dim3(64), 0, stream, sums, gamma, beta, coef, C, cpg,
                     gr_cdiv(HW, 512), (float)HW * (float)cpg, eps);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// GroupNorm + ReLU + cross-level channel shuffle (groma/model/roi_align.py:150-178), producing the
// zero-bordered NHWC bf16 input of the next 3x3 conv for ONE target level:
//   ch [0, C/2)        <- tar   [0, C/2)       same pixel
//   ch [C/2, 3C/4)     <- top   [3C/4, C)      bilinear align_corners=True resize to the target size
//   ch [3C/4, C)       <- down  [C/2, 3C/4)    bilinear align_corners=True resize to the target size
// Each source is relu(gn(conv_out)) when its coef pointer (gr_gn_finalize) is non-null, else the raw map (round 0).
// shuffle==0: plain relu(gn(.)) of tar for all channels (final round -> RoIAlign input, pad=0 layout allowed).
struct ShufSrc {
  const bf16_t* x;    // [imgs*S*S, C]
  const float* coef;  // [imgs, 2, C] from gr_gn_finalize, or null (raw map, round 0)
  int S;
};
__device__ __forceinline__ void load_coef(const ShufSrc& s, int img, int c, int C, float* a, float* bb) {
  const float* pa = s.coef + ((long)img * 2) * C + c;
  const f32x4 a0 = *(const f32x4*)pa, a1 = *(const f32x4*)(pa + 4);
  const f32x4 b0 = *(const f32x4*)(pa + C), b1 = *(const f32x4*)(pa + C + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = a0[i]; a[4 + i] = a1[i]; bb[i] = b0[i]; bb[4 + i] = b1[i]; }
}
__device__ __forceinline__ void load8(const ShufSrc& s, bool norm, const float* a, const float* bb, int img, int y, int x,
                                      int c, int C, float* o) {
  ld8f(s.x, (((long)img * s.S + y) * s.S + x) * C + c, o);
  if (norm) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(o[i] * a[i] + bb[i], 0.f);
  }
}
// Q8: the map is written as OCP e4m3 bytes of value * q_inv (clamped to +-448) -- the A operand of an e4m3 implicit-GEMM conv whose
// activation scale is a constant of the weights (groma_amd/weights.py: conv_act_bound), folded into that GEMM's w_scale.
template <bool Q8>
__global__ __launch_bounds__(256) void fuse_shuffle_kernel(ShufSrc tar, ShufSrc top, ShufSrc down, void* __restrict__ out,
                                                           int imgs, int C, int shuffle, int pad, float q_inv) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 8-channel chunk
  const int c8 = C >> 3;
  const int S = tar.S;
  const long total = (long)imgs * S * S * c8;
  if (idx >= total) return;
  const int c = (int)(idx % c8) << 3;
  const long m = idx / c8;
  const int x = (int)(m % S), y = (int)((m / S) % S), img = (int)(m / ((long)S * S));
  float o[8];
  float a[8], bb[8];
  if (!shuffle || c < C / 2) {
    const bool norm = tar.coef != nullptr;
    if (norm) load_coef(tar, img, c, C, a, bb);
    load8(tar, norm, a, bb, img, y, x, c, C, o);
  } else {
    const bool is_top = c < 3 * C / 4;
    const ShufSrc& src = is_top ? top : down;
    const int sc = is_top ? c + C / 4 : c - C / 4;  // source channel
    const bool norm = src.coef != nullptr;
    if (norm) load_coef(src, img, sc, C, a, bb);
    const int Ss = src.S;
    if (Ss == S) {
      load8(src, norm, a, bb, img, y, x, sc, C, o);
    } else {
      const float scl = S > 1 ? (float)(Ss - 1) / (float)(S - 1) : 0.f;
      const float sy = scl * y, sx = scl * x;
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < Ss - 1 ? 1 : 0), x1 = x0 + (x0 < Ss - 1 ? 1 : 0);
      const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
      float v00[8], v01[8], v10[8], v11[8];
      load8(src, norm, a, bb, img, y0, x0, sc, C, v00);
      load8(src, norm, a, bb, img, y0, x1, sc, C, v01);
      load8(src, norm, a, bb, img, y1, x0, sc, C, v10);
      load8(src, norm, a, bb, img, y1, x1, sc, C, v11);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = hy * (hx * v00[i] + lx * v01[i]) + ly * (hx * v10[i] + lx * v11[i]);
    }
  }
  const int Sp = S + 2 * pad;
  const long oi = (((long)img * Sp + y + pad) * Sp + x + pad) * C + c;
  if (Q8) st8q((uint8_t*)out, oi, o, q_inv);
  else st8f((bf16_t*)out, oi, o);
}
extern "C" int gr_fuse_shuffle(const void* tar, const float* tar_coef, int tarS, const void* top, const float* top_coef,
                               int topS, const void* down, const float* down_coef, int downS, void* out, int imgs, int C,
                               int shuffle, int pad, hipStream_t stream) {
  if (!tar || !out || C % 32 != 0) return GR_EINVAL;
  if (shuffle && (!top || !down)) return GR_EINVAL;
  ShufSrc a{(const bf16_t*)tar, tar_coef, tarS}, b{(const bf16_t*)top, top_coef, topS},
      c{(const bf16_t*)down, down_coef, downS};
  const long total = (long)imgs * tarS * tarS * (C >> 3);
  hipLaunchKernelGGL(fuse_shuffle_kernel<false>, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, a, b, c, out, imgs, C,
                     shuffle, pad, 0.f);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
extern "C" int gr_fuse_shuffle_fp8(const void* tar, const float* tar_coef, int tarS, const void* top, const float* top_coef,
                                   int topS, const void* down, const float* down_coef, int downS, void* out, int imgs, int C,
                                   int shuffle, int pad, float inv_scale, hipStream_t stream) {
  if (GR_SP) return GR_EINVAL;  // no e4m3 path in the split-operand build
  if (!tar || !out || C % 32 != 0 || !(inv_scale > 0.f)) return GR_EINVAL;
  if (shuffle && (!top || !down)) return GR_EINVAL;
  ShufSrc a{(const bf16_t*)tar, tar_coef, tarS}, b{(const bf16_t*)top, top_coef, topS},
      c{(const bf16_t*)down, down_coef, downS};
  const long total = (long)imgs * tarS * tarS * (C >> 3);
  hipLaunchKernelGGL(fuse_shuffle_kernel<true>, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, a, b, c, out, imgs, C,
                     shuffle, pad, inv_scale);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// out bf16 = a (+ b)   (f32 -> bf16 cast of a GEMM A operand)
__global__ void cast_add_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ out,
                                long n4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n4) return;
  f32x4 v = *(const f32x4*)(a + idx * 4);
  if (b) v += *(const f32x4*)(b + idx * 4);
  st4f(out, idx * 4, v);
}
extern "C" int gr_cast_f32_bf16(const float* a, const float* b, void* out, long n, hipStream_t stream) {
  if (!a || !out || n <= 0 || n % 4 != 0 || (GR_SP && n % 32 != 0)) return GR_EINVAL;
  hipLaunchKernelGGL(cast_add_kernel, dim3(gr_cdiv(n / 4, 256)), dim3(256), 0, stream, a, b, (bf16_t*)out, n / 4);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// out f32 [rows, C] = a[rows, C] + b[(row % b_mod), C]
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                long rows, int C, int b_mod) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (idx >= rows * c4) return;
  const int c = (int)(idx % c4) << 2;
  const long r = idx / c4;
  const long rb = b_mod > 0 ? r % b_mod : r;
  *(f32x4*)(out + r * C + c) = *(const f32x4*)(a + r * C + c) + *(const f32x4*)(b + rb * C + c);
}
extern "C" int gr_add_rows_f32(const float* a, const float* b, float* out, long rows, int C, int b_mod,
                               hipStream_t stream) {
  if (!a || !b || !out || rows <= 0 || C % 4 != 0) return GR_EINVAL;
  hipLaunchKernelGGL(add_rows_kernel, dim3(gr_cdiv(rows * (C >> 2), 256)), dim3(256), 0, stream, a, b, out, rows, C,
                     b_mod);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// Split-vocabulary embedding lookup (groma/model/groma.py:165-174): ids < V0 -> table0, else table1[id - V0]
// tables bf16, out f32 [n, C]
__global__ void embed_gather_kernel(const long* __restrict__ ids, const bf16_t* __restrict__ t0,
                                    const bf16_t* __restrict__ t1, float* __restrict__ out, long n, int C, int V0,
                                    int V1) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // 8-chunk
  const int c8 = C >> 3;
  if (idx >= n * c8) return;
  const int c = (int)(idx % c8) << 3;
  const long r = idx / c8;
  long id = ids[r];
  const bf16_t* tab = t0;
  if (id >= V0) {
    id -= V0;
    if (id >= V1) id = V1 - 1;
    tab = t1;
  } else if (id < 0) {
    id = 0;
  }
  float v[8];
  ld8f(tab, id * C + c, v);
  float* o = out + r * C + c;
  *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
  *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
extern "C" int gr_embed_gather(const long* ids, const void* table0, const void* table1, float* out, long n, int C,
                               int V0, int V1, hipStream_t stream) {
  if (!ids || !table0 || !table1 || !out || n <= 0 || C % 8 != 0 || (GR_SP && C % 32 != 0)) return GR_EINVAL;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(gr_cdiv(n * (C >> 3), 256)), dim3(256), 0, stream, ids,
                     (const bf16_t*)table0, (const bf16_t*)table1, out, n, C, V0, V1);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// dst[row_idx[r], :] = src[r, :]   (masked_scatter_ of image / region features, groma.py:364-369)
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_idx,
                                    float* __restrict__ dst, long n, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (idx >= n * c4) return;
  const int c = (int)(idx % c4) << 2;
  const long r = idx / c4;
  *(f32x4*)(dst + (long)row_idx[r] * C + c) = *(const f32x4*)(src + r * C + c);
}
extern "C" int gr_scatter_rows_f32(const float* src, const int* row_idx, float* dst, long n, int C, hipStream_t stream) {
  if (!src || !row_idx || !dst || C % 4 != 0) return GR_EINVAL;
  if (n <= 0) return GR_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(gr_cdiv(n * (C >> 2), 256)), dim3(256), 0, stream, src, row_idx, dst, n,
                     C);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// Greedy next-token: argmax over logits f32 [rows, ld] restricted to [0, V); first maximal index wins.
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, long* __restrict__ out, int V, long ld) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (long)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  };
  const bool vec = (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0;
  const int nv = vec ? V >> 2 : 0;
  for (int i = tid; i < nv; i += 1024) {
    const f32x4 v = *(const f32x4*)(xr + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) take(v[e], i * 4 + e);
  }
  for (int i = nv * 4 + tid; i < V; i += 1024) take(xr[i], i);
  // wave-level then block-level (value desc, index asc)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    take(ov, oi);
  }
  __shared__ float sv[16];
  __shared__ int si[16];
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) take(sv[w], si[w]);
    out[row] = bi;
  }
}
// HF 4.32 greedy_search bookkeeping for one step, entirely on the device (so the decode step is hipGraph-capturable):
//   n = unfinished ? argmax : pad ; sequences[:, step] = n ; next input token = n ; unfinished &= (n != eos) ;
//   step += 1 ; (inc_pos) every KV write position += 1 ; n_unfinished = sum(unfinished)
__global__ __launch_bounds__(256) void greedy_advance_kernel(const long* __restrict__ nxt, long* __restrict__ tok,
                                                             long* __restrict__ unfinished, long* __restrict__ seq,
                                                             int* __restrict__ pos, int* __restrict__ step,
                                                             int* __restrict__ n_unfinished, int rows, long eos, long pad,
                                                             int seq_ld, int pos_rows, int inc_pos) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int st = *step;
  const bool per_row = inc_pos == 2;  // ragged batch: every row owns its KV write position (pos_rows == rows)
  for (int b = threadIdx.x; b < rows; b += 256) {
    long n = nxt[b];
    const bool was = unfinished[b] != 0;
    if (eos >= 0 || per_row) {  // (a ragged batch also uses `unfinished` as the row-occupied mask)
      if (!was) n = pad;
      if (eos >= 0 && n == eos) unfinished[b] = 0;
      if (unfinished[b]) atomicAdd(&cnt, 1);
    } else {
      atomicAdd(&cnt, 1);
    }
    if (st < seq_ld) seq[(long)b * seq_ld + st] = n;
    tok[b] = n;
    if (inc_pos && per_row && was) pos[b] += 1;  // idle rows keep re-writing one scratch position
  }
  if (inc_pos && !per_row)
    for (int r = threadIdx.x; r < pos_rows; r += 256) pos[r] += 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    *step = st + 1;
    *n_unfinished = cnt;
  }
}
extern "C" int gr_greedy_advance(const long* nxt, long* tok, long* unfinished, long* seq, int* pos, int* step,
                                 int* n_unfinished, int rows, long eos, long pad, int seq_ld, int pos_rows, int inc_pos,
                                 hipStream_t stream) {
  if (!nxt || !tok || !unfinished || !seq || !pos || !step || !n_unfinished || rows <= 0 || seq_ld <= 0) return GR_EINVAL;
  if (inc_pos < 0 || inc_pos > 2 || (inc_pos == 2 && pos_rows != rows)) return GR_EINVAL;
  hipLaunchKernelGGL(greedy_advance_kernel, dim3(1), dim3(256), 0, stream, nxt, tok, unfinished, seq, pos, step,
                     n_unfinished, rows, eos, pad, seq_ld, pos_rows, inc_pos);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---------------------------------------------------------------------------------------
// Next-token sampling of the serving loop (groma/serve/model_worker.py:307-311): temperature < 1e-4 -> arg-max, else
// probs = softmax(logits / temperature); token ~ multinomial(probs).  The draw is an inverse-CDF lookup with a COUNTER-BASED
// uniform u = splitmix64(seed[row], position of the new token) -- a function of the request's seed and the token's absolute
// position only, so a row's samples do not depend on batching, admission time or graph replay, and a CPU checker can
// recompute u exactly.  One 1024-thread block per row; every thread owns a contiguous chunk of the vocabulary (fixed
// summation order: bit-reproducible run to run).  inv_temp[row] == 0 selects the greedy branch (same tie rule as
// argmax_kernel: first maximal index).
__device__ __forceinline__ float sample_uniform(long seed, int counter) {
  unsigned long long z = (unsigned long long)seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(counter + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
}
__global__ __launch_bounds__(1024) void sample_rows_kernel(const float* __restrict__ x, long* __restrict__ out, int V, long ld,
                                                           const float* __restrict__ inv_temp, const long* __restrict__ seed,
                                                           const int* __restrict__ pos, int pos_stride, int pos_off) {
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (long)row * ld;
  const float it = inv_temp ? inv_temp[row] : 0.f;
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int s_last, s_win;
  const int chunk = (V + 1023) / 1024;
  const int lo = min(V, tid * chunk), hi = min(V, lo + chunk);
  // ---- max (and its first index: the greedy answer)
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  };
  for (int i = lo; i < hi; ++i) take(xr[i], i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    take(ov, oi);
  }
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  if (tid == 0) { s_last = -1; s_win = 0x7fffffff; }
  __syncthreads();
  for (int w = 0; w < 16; ++w) take(sv[w], si[w]);  // every thread now holds the row max
  if (it == 0.f) {
    if (tid == 0) out[row] = bi;
    return;
  }
  const float m = best;
  __syncthreads();  // sv is reused below
  // ---- un-normalised probabilities: chunk sums, block exclusive scan (wave scan + sequential wave totals)
  float local = 0.f;
  for (int i = lo; i < hi; ++i) local += __expf((xr[i] - m) * it);
  float incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) sv[wave] = incl;
  if (local > 0.f) atomicMax(&s_last, tid);
  __syncthreads();
  float base = 0.f, total = 0.f;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base += sv[w];
    total += sv[w];
  }
  const float excl = base + incl - local;
  const int counter = (pos ? pos[(long)row * pos_stride] : 0) + pos_off;
  const float target = sample_uniform(seed ? seed[row] : 0, counter) * total;
  // the owning chunk: excl <= target < excl + local; fp slack at the top end falls to the last chunk with mass
  if (local > 0.f && excl <= target && (target < excl + local || tid == s_last)) {
    float acc = excl;
    int pick = -1;
    for (int i = lo; i < hi; ++i) {
      const float e = __expf((xr[i] - m) * it);
      acc += e;
      if (e > 0.f) {
        pick = i;
        if (target < acc) break;
      }
    }
    if (pick >= 0) atomicMin(&s_win, pick);
  }
  __syncthreads();
  if (tid == 0) out[row] = s_win != 0x7fffffff ? s_win : bi;
}
extern "C" int gr_sample_rows(const float* x, long* out, int rows, int V, long ld, const float* inv_temp, const long* seed,
                              const int* pos, int pos_stride, int pos_off, hipStream_t stream) {
  if (!x || !out || rows <= 0 || V <= 0 || pos_stride < 0) return GR_EINVAL;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(rows), dim3(1024), 0, stream, x, out, V, ld, inv_temp, seed, pos, pos_stride, pos_off);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

extern "C" int gr_argmax_rows(const float* x, long* out, int rows, int V, long ld, hipStream_t stream) {
  if (!x || !out || rows <= 0 || V <= 0) return GR_EINVAL;
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, stream, x, out, V, ld);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
