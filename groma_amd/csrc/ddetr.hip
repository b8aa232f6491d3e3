// Deformable-DETR proposer kernels (fp32), SURVEY §8a a6-a10.
//  * msda:   multi-scale deformable attention sampling, 1 level (mmcv
//            ops/csrc/common/cuda/ms_deform_attn_cuda_kernel.cuh:19-66,202-256 semantics ==
//            grid_sample(bilinear, zeros, align_corners=False) of
//            mmcv/ops/multi_scale_deform_attn.py:93-150), fused with the softmax over points and the
//            sampling-location arithmetic of HF DeformableDetrMultiscaleDeformableAttention.
//  * mha32:  decoder self-attention (8 heads x 32, <=320 queries) in fp32.
//  * small elementwise pieces of the two-stage / box-refine logic
//            (groma/model/ddetr_transformer.py:150-166, 432-446, 550-568, 696-728; groma.py:247-249).
#include "gr_common.h"
#include "../../include/groma_hip.h"

// ---- MSDA -----------------------------------------------------------------------------------
// value  f32 [B, S=Hs*Ws, heads, 32]
// offw   f32 [B, Q, ld]: [0, heads*P*2) sampling offsets (h,p,xy), [heads*P*2, heads*P*3) attention logits (h,p)
// ref    f32 [B*Q or Q, rdim] (rdim 2: loc = ref + off/(W,H); rdim 4: loc = ref_xy + off/P * ref_wh * 0.5)
// out    f32 [B, Q, heads*32]
// thread = (b, q, head, channel); 32 channels of a head = half a wave -> 128-B coalesced value reads.
// Every channel lane recomputes the head's soft-max (4 expf) and the four sampling locations: 32x redundant ALU work -- and it is the
// faster form.  Round 6 built the de-duplicated kernel VERDICT r05 asked for (one thread per (query, head, point) evaluates soft-max,
// location and corner taps once into LDS, the channel lanes only gather): 329 us per encoder launch with 8 groups per block, 343 us
// with 64 (all lanes busy in the tap phase), against 163 us for this form at 14 images beside the pyramid (profiles/r06_timeline_b14.txt,
// r05_timeline_b14.txt; alone on the chip 0.15 vs 0.06 ms per step).  The kernel is bound by the latency of its 16 dependent-free
// gathers per thread, which this form issues from 8x as many resident threads with nothing between the offset loads and the gathers
// but ALU; an LDS hand-over plus a block barrier in front of the gathers costs more than 30 redundant expf save.  Kept as it was.
template <int P>
__global__ __launch_bounds__(256) void msda_kernel(const float* __restrict__ value, const float* __restrict__ offw,
                                                   const float* __restrict__ ref, float* __restrict__ out, int B, int Q,
                                                   int heads, int Hs, int Ws, int ld, int rdim, int ref_batched) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Q * heads * 32;
  if (idx >= total) return;
  const int c = (int)(idx & 31);
  const int h = (int)((idx >> 5) % heads);
  const long bq = idx / (32L * heads);
  const int q = (int)(bq % Q), b = (int)(bq / Q);
  const float* ow = offw + bq * ld;
  const float* rp = ref + (ref_batched ? bq : (long)q) * rdim;
  // softmax over the P points of this head (single level)
  float lg[P], mx = -INFINITY;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    lg[p] = ow[heads * P * 2 + h * P + p];
    mx = fmaxf(mx, lg[p]);
  }
  float den = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    lg[p] = expf(lg[p] - mx);
    den += lg[p];
  }
  const float rx = rp[0], ry = rp[1];
  float acc = 0.f;
  const float* vb = value + (long)b * Hs * Ws * heads * 32 + h * 32 + c;
  const long vstride = (long)heads * 32;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float ox = ow[(h * P + p) * 2], oy = ow[(h * P + p) * 2 + 1];
    float lx, ly;
    if (rdim == 2) {
      lx = rx + ox / (float)Ws;
      ly = ry + oy / (float)Hs;
    } else {
      lx = rx + ox / (float)P * rp[2] * 0.5f;
      ly = ry + oy / (float)P * rp[3] * 0.5f;
    }
    const float w_im = lx * Ws - 0.5f, h_im = ly * Hs - 0.5f;
    float val = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)Hs && w_im < (float)Ws) {
      const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
      const int hh = hl + 1, wh = wl + 1;
      const float lh = h_im - hl, lw = w_im - wl;
      const float uh = 1.f - lh, uw = 1.f - lw;
      float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
      if (hl >= 0 && wl >= 0) v1 = vb[((long)hl * Ws + wl) * vstride];
      if (hl >= 0 && wh <= Ws - 1) v2 = vb[((long)hl * Ws + wh) * vstride];
      if (hh <= Hs - 1 && wl >= 0) v3 = vb[((long)hh * Ws + wl) * vstride];
      if (hh <= Hs - 1 && wh <= Ws - 1) v4 = vb[((long)hh * Ws + wh) * vstride];
      val = uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
    }
    acc += val * (lg[p] / den);
  }
  out[idx] = acc;
}

extern "C" int gr_msda_f32(const float* value, const float* offw, const float* ref, float* out, int B, int Q, int heads,
                           int n_points, int Hs, int Ws, int ld, int rdim, int ref_batched, hipStream_t stream) {
  if (!value || !offw || !ref || !out || n_points != 4 || (rdim != 2 && rdim != 4)) return GR_EINVAL;
  const long total = (long)B * Q * heads * 32;
  hipLaunchKernelGGL(msda_kernel<4>, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, value, offw, ref, out, B, Q, heads,
                     Hs, Ws, ld, rdim, ref_batched);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---- decoder self-attention, fp32, head_dim 32 ------------------------------------------------
// qk f32 [B, Q, ldqk]: q at [0, D), k at [D, 2D);  v f32 [B, Q, D];  out f32 [B, Q, D];  D = heads*32
// block = (b, head), one thread per query row; K/V of the head live in LDS (broadcast reads).
// softmax(q.k * scale) is the exact two-pass form (max, then exp/sum), as torch.softmax.
__global__ __launch_bounds__(320) void mha32_kernel(const float* __restrict__ qk, const float* __restrict__ v,
                                                    float* __restrict__ out, int Q, int heads, int ldqk, float scale) {
  extern __shared__ float sm[];
  float* ks = sm;            // [Q][32]
  float* vs = sm + Q * 32;   // [Q][32]
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int D = heads * 32;
  for (int i = threadIdx.x; i < Q * 32; i += blockDim.x) {
    const int r = i >> 5, c = i & 31;
    ks[i] = qk[((long)b * Q + r) * ldqk + D + h * 32 + c];
    vs[i] = v[((long)b * Q + r) * D + h * 32 + c];
  }
  __syncthreads();
  for (int r = threadIdx.x; r < Q; r += blockDim.x) {
    float qv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) qv[c] = qk[((long)b * Q + r) * ldqk + h * 32 + c] * scale;
    float mx = -INFINITY;
    for (int j = 0; j < Q; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) s += qv[c] * ks[j * 32 + c];
      mx = fmaxf(mx, s);
    }
    float o[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] = 0.f;
    float den = 0.f;
    for (int j = 0; j < Q; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) s += qv[c] * ks[j * 32 + c];
      const float e = expf(s - mx);
      den += e;
#pragma unroll
      for (int c = 0; c < 32; ++c) o[c] += e * vs[j * 32 + c];
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int c = 0; c < 32; ++c) out[((long)b * Q + r) * D + h * 32 + c] = o[c] * inv;
  }
}

// ---- the same operator on the f32-input MFMA (Q <= 320: the reference runs 300 queries) --------------------------------
// Round 1's kernel above walks 300 keys x 32 dims per query thread with one broadcast LDS read per FMA (240 us standalone at
// 1 image, 441 us at 14 while time-sharing the chip: 2.6 ms of a step).  Here a wave owns 16 queries:
//   S^T = K . (Q*scale)^T      20 key tiles x 8 v_mfma_f32_16x16x4_f32, all of S^T kept in registers (80 VGPRs / lane) --
//                              a lane holds 4 consecutive keys of ONE query, so the exact two-pass soft-max (max, then
//                              exp / sum: torch.softmax semantics) is in-lane + 2 permlane swaps;
//   O^T = V^T . P^T            P^T is the MFMA B operand straight from those registers (the key order inside a 16-key tile
//                              is permuted identically on the V^T side: key fg*4 + e is "k-slot fg" of step e);
// K rows are padded to 36 floats and V is transposed into [dim][key + 4] while staging, so every fragment is one
// conflict-free ds_read_b128.  Everything is fp32 (products, sums, exp): same tolerance class as gemm_f32.
template <int NKT>  // key tiles of 16
__global__ __launch_bounds__(640) void mha32_mfma_kernel(const float* __restrict__ qk, const float* __restrict__ v,
                                                         float* __restrict__ out, int Q, int heads, int ldqk, float scale) {
  constexpr int QP = NKT * 16, KS = 36, VS = QP + 4;
  extern __shared__ __attribute__((aligned(16))) float sm2[];
  float* ks = sm2;             // [QP][KS]   K rows (dims 0..31)
  float* vt = sm2 + QP * KS;   // [32][VS]   V transposed
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int D = heads * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < QP * 8; i += 640) {  // 16-B chunks of K and V
    const int r = i >> 3, c = (i & 7) * 4;
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (r < Q) {
      kv = *(const f32x4*)(qk + ((long)b * Q + r) * ldqk + D + h * 32 + c);
      vv = *(const f32x4*)(v + ((long)b * Q + r) * D + h * 32 + c);
    }
    *(f32x4*)(ks + r * KS + c) = kv;
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[(c + e) * VS + r] = vv[e];
  }
  __syncthreads();
  const int q0 = (blockIdx.x * 10 + wave) * 16;
  if (q0 >= Q) return;
  int qrow = q0 + fr;
  if (qrow > Q - 1) qrow = Q - 1;
  // B operand of S^T: lane (query fr, k-group fg) owns dims fg*8 .. fg*8+7 (the same dim permutation as the K side)
  const float* qp = qk + ((long)b * Q + qrow) * ldqk + h * 32 + fg * 8;
  f32x4 qa = *(const f32x4*)qp, qb = *(const f32x4*)(qp + 4);
  qa *= scale; qb *= scale;
  f32x4 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const float* kp = ks + (t * 16 + fr) * KS + fg * 8;
    const f32x4 ka = *(const f32x4*)kp, kb = *(const f32x4*)(kp + 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[e], qa[e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[e], qb[e], acc, 0, 0, 0);
    s[t] = acc;  // keys t*16 + fg*4 + {0..3} of query fr
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (t * 16 + fg * 4 + e >= Q) s[t][e] = -INFINITY;
      mx = fmaxf(mx, s[t][e]);
    }
  mx = rows_max(mx);
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = expf(s[t][e] - mx);  // exp(-inf) = 0 for the padded keys
      s[t][e] = p;
      den += p;
    }
  den = rows_sum(den);
  f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const f32x4 va = *(const f32x4*)(vt + (dt * 16 + fr) * VS + t * 16 + fg * 4);  // V^T[dim][keys t*16 + fg*4 + e]
#pragma unroll
      for (int e = 0; e < 4; ++e) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[e], s[t][e], o[dt], 0, 0, 0);
    }
  }
  if (q0 + fr < Q) {
    const float inv = 1.f / den;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      *(f32x4*)(out + ((long)b * Q + q0 + fr) * D + h * 32 + dt * 16 + fg * 4) = o[dt] * inv;
  }
}

extern "C" int gr_mha32_f32(const float* qk, const float* v, float* out, int B, int Q, int heads, int ldqk, float scale,
                            hipStream_t stream) {
  if (!qk || !v || !out || B <= 0 || Q <= 0 || Q > 1024) return GR_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mha32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mha32_mfma_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const bool aligned = (ldqk & 3) == 0 && ((((uintptr_t)qk) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0;
  if (Q <= 320 && aligned) {  // the reference's 300 queries: MFMA kernel, 160 queries per block
    constexpr int NKT = 20;
    const size_t smem = ((size_t)NKT * 16 * 36 + 32 * (NKT * 16 + 4)) * sizeof(float);
    hipLaunchKernelGGL(mha32_mfma_kernel<NKT>, dim3(gr_cdiv(Q, 160), B * heads), dim3(640), smem, stream, qk, v, out, Q, heads,
                       ldqk, scale);
    GR_CHECK_LAUNCH();
    return GR_OK;
  }
  const size_t smem = (size_t)Q * 64 * sizeof(float);
  if (smem > 160 * 1024) return GR_EINVAL;
  hipLaunchKernelGGL(mha32_kernel, dim3(B * heads), dim3(320), smem, stream, qk, v, out, Q, heads, ldqk, scale);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---- two-stage bookkeeping ---------------------------------------------------------------------
// gather the top-k proposals: coord logits = delta[idx] + proposals[idx]; reference = sigmoid; sine embedding
// (ddetr_transformer.py:432-446,552-565).  idx int32 [B, Kq]; delta f32 [B, S, 4]; prop f32 [S, 4]
// -> ref f32 [B,Kq,4], pos f32 [B,Kq,4*npf]  (npf = d_model/2; layout (coord, npf) with sin/cos interleaved)
__global__ void topk_gather_kernel(const int* __restrict__ idx, const float* __restrict__ delta,
                                   const float* __restrict__ prop, float* __restrict__ ref, float* __restrict__ pos,
                                   int B, int S, int Kq, int npf) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Kq * 4 * npf;
  if (t >= total) return;
  const int f = (int)(t % npf);
  const int cd = (int)((t / npf) % 4);
  const long bk = t / (4L * npf);
  const int b = (int)(bk / Kq);
  const int i = idx[bk];
  const float logit = delta[((long)b * S + i) * 4 + cd] + prop[(long)i * 4 + cd];
  const float sg = 1.0f / (1.0f + expf(-logit));
  if (f == 0) ref[bk * 4 + cd] = sg;
  // dim_t = 10000 ** (2*(f//2)/npf);  pos = sg*2pi / dim_t ; even f -> sin, odd f -> cos
  const float dim_t = powf(10000.0f, (float)(2 * (f / 2)) / (float)npf);
  const float a = sg * 6.283185307179586f / dim_t;
  pos[bk * 4 * npf + cd * npf + f] = (f & 1) ? cosf(a) : sinf(a);
}
extern "C" int gr_ddetr_topk_gather(const int* idx, const float* delta, const float* prop, float* ref, float* pos, int B,
                                    int S, int Kq, int npf, hipStream_t stream) {
  if (!idx || !delta || !prop || !ref || !pos) return GR_EINVAL;
  const long total = (long)B * Kq * 4 * npf;
  hipLaunchKernelGGL(topk_gather_kernel, dim3(gr_cdiv(total, 256)), dim3(256), 0, stream, idx, delta, prop, ref, pos, B,
                     S, Kq, npf);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// out = sigmoid(tmp + inverse_sigmoid(ref)), inverse_sigmoid eps 1e-5 (HF inverse_sigmoid)
__global__ void box_refine_kernel(const float* __restrict__ tmp, const float* __restrict__ ref, float* __restrict__ out,
                                  long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = fminf(fmaxf(ref[i], 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-5f), x2 = fmaxf(1.f - x, 1e-5f);
  const float z = tmp[i] + logf(x1 / x2);
  out[i] = 1.0f / (1.0f + expf(-z));
}
extern "C" int gr_box_refine(const float* tmp, const float* ref, float* out, long n, hipStream_t stream) {
  if (!tmp || !ref || !out || n <= 0) return GR_EINVAL;
  hipLaunchKernelGGL(box_refine_kernel, dim3(gr_cdiv(n, 256)), dim3(256), 0, stream, tmp, ref, out, n);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// fused objectness: sigmoid(coco)^0.4 * sigmoid(sa1b)^0.6 (groma/model/groma.py:247-249)
__global__ void score_fuse_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  long n, long lda) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sa = 1.0f / (1.0f + expf(-a[i * lda]));
  const float sb = 1.0f / (1.0f + expf(-b[i * lda]));
  out[i] = powf(sa, 0.4f) * powf(sb, 0.6f);
}
extern "C" int gr_score_fuse(const float* coco, const float* sa1b, float* out, long n, long ld, hipStream_t stream) {
  if (!coco || !sa1b || !out || n <= 0) return GR_EINVAL;
  hipLaunchKernelGGL(score_fuse_kernel, dim3(gr_cdiv(n, 256)), dim3(256), 0, stream, coco, sa1b, out, n, ld);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
