// Index-producing selection kernels (bit-exact vs oracle/): compiled with -ffp-contract=off so the
// fp32 IoU arithmetic is the plain IEEE sequence of mmcv's CPU NMS.
//  * topk_desc:  two-stage proposal selection, top-K of S class logits per image
//                (groma/model/ddetr_transformer.py:555-556), order = (value desc, index asc).
//  * nms:        mmcv.ops.nms as called at groma/model/groma.py:266-272 -- score threshold
//                (mmcv/ops/nms.py:21-26), stable descending sort, greedy IoU suppression in the CPU form
//                inter/(a+b-inter) > thr (mmcv/ops/csrc/pytorch/cpu/nms.cpp:28-53), first max_num kept.
//                Boxes arrive as (cx,cy,w,h) and are converted with HF center_to_corners_format.
// One workgroup per image; n <= 512 (NMS) / 1024 (top-k).  The O(n^2) IoU test is a 64-bit-mask matrix (wave64 = one
// word per 64 candidates) followed by a single-wave greedy scan: no host round trip (the reference's
// CUDA path copies the mask to the host, mmcv/ops/csrc/pytorch/cuda/nms_cuda.cu:27-50).
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define NMAX 1024     // sort capacity (top-k of S <= 1024)
#define NMS_MAX 512   // NMS candidates per image (300 proposals + refer + ground boxes)

// bitonic sort of (key desc, idx asc) pairs in LDS; n padded to NMAX with (-inf, big idx)
__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) {
  // true if (ka, ia) must come before (kb, ib)
  return ka > kb || (ka == kb && ia < ib);
}
__device__ void bitonic_sort_desc(float* key, int* idx, int tid, int nthreads) {
  for (int k = 2; k <= NMAX; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < NMAX; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = (i & k) == 0;  // ascending position order == "before" order
          const float ka = key[i], kb = key[ixj];
          const int ia = idx[i], ib = idx[ixj];
          const bool a_first = before(ka, ia, kb, ib);
          if (up ? !a_first : a_first) {
            key[i] = kb; key[ixj] = ka;
            idx[i] = ib; idx[ixj] = ia;
          }
        }
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void topk_desc_kernel(const float* __restrict__ x, int* __restrict__ out, int S, int K,
                                                        long ldx) {
  __shared__ float key[NMAX];
  __shared__ int idx[NMAX];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < NMAX; i += 512) {
    key[i] = i < S ? x[(long)b * ldx + i] : -INFINITY;
    idx[i] = i < S ? i : 0x7fffffff - (NMAX - i);
  }
  bitonic_sort_desc(key, idx, tid, 512);
  for (int i = tid; i < K; i += 512) out[(long)b * K + i] = idx[i];
}
extern "C" int gr_topk_desc(const float* x, int* out_idx, int B, int S, int K, long ldx, hipStream_t stream) {
  if (!x || !out_idx || B <= 0 || S <= 0 || S > NMAX || K <= 0 || K > S) return GR_EINVAL;
  hipLaunchKernelGGL(topk_desc_kernel, dim3(B), dim3(512), 0, stream, x, out_idx, S, K, ldx);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// boxes_cxcywh f32 [B, n, 4], scores f32 [B, n]  ->  keep int64 [B, max_num] (indices into the n inputs, in
// descending-score order, -1 padded), n_keep int32 [B].  n_valid (optional, [B]): per-image candidate count <= n.
__global__ __launch_bounds__(512) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                  long* __restrict__ keep, int* __restrict__ n_keep,
                                                  const int* __restrict__ n_valid, int n, float iou_thr, float score_thr, int max_num) {
  __shared__ float key[NMAX];
  __shared__ int idx[NMAX];
  __shared__ float x1[NMS_MAX], y1[NMS_MAX], x2[NMS_MAX], y2[NMS_MAX], area[NMS_MAX];
  __shared__ unsigned long long mask[NMS_MAX][NMS_MAX / 64];  // 32 KB
  __shared__ int nvalid_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bx = boxes + (long)b * n * 4;
  const float* sc = scores + (long)b * n;
  if (tid == 0) nvalid_s = 0;
  __syncthreads();
  // score filter: only when score_thr > 0 (mmcv/ops/nms.py:21); invalid entries sort last
  int cnt = 0;
  for (int i = tid; i < NMAX; i += 512) {
    bool valid = i < n && (!n_valid || i < n_valid[b]);
    if (valid && score_thr > 0.f) valid = sc[i] > score_thr;
    key[i] = valid ? sc[i] : -INFINITY;
    idx[i] = i < n ? (valid ? i : 0x40000000 + i) : 0x7fffffff - (NMAX - i);
    cnt += valid ? 1 : 0;
  }
  atomicAdd(&nvalid_s, cnt);
  bitonic_sort_desc(key, idx, tid, 512);
  const int nv = nvalid_s;  // valid entries occupy sorted positions [0, nv) unless a valid score is -inf
  for (int i = tid; i < nv; i += 512) {
    const int o = idx[i] & 0x3fffffff;
    const float cx = bx[o * 4 + 0], cy = bx[o * 4 + 1], w = bx[o * 4 + 2], h = bx[o * 4 + 3];
    const float a = cx - 0.5f * w, c = cy - 0.5f * h, d = cx + 0.5f * w, e = cy + 0.5f * h;
    x1[i] = a; y1[i] = c; x2[i] = d; y2[i] = e;
    area[i] = (d - a) * (e - c);
  }
  __syncthreads();
  const int nw = (nv + 63) >> 6;
  // mask[i][w] bit j: sorted box (w*64+j) is suppressed by sorted box i (only j > i matters)
  for (int t = tid; t < nv * nw; t += 512) {
    const int i = t / nw, w = t - i * nw;
    unsigned long long m = 0ull;
    const float ix1 = x1[i], iy1 = y1[i], ix2 = x2[i], iy2 = y2[i], ia = area[i];
    const int j0 = w * 64;
    for (int jj = 0; jj < 64; ++jj) {
      const int j = j0 + jj;
      if (j > i && j < nv) {
        const float xx1 = fmaxf(ix1, x1[j]), yy1 = fmaxf(iy1, y1[j]);
        const float xx2 = fminf(ix2, x2[j]), yy2 = fminf(iy2, y2[j]);
        const float ww = fmaxf(0.f, xx2 - xx1), hh = fmaxf(0.f, yy2 - yy1);
        const float inter = ww * hh;
        const float ovr = inter / (ia + area[j] - inter);
        if (ovr > iou_thr) m |= 1ull << jj;
      }
    }
    mask[i][w] = m;
  }
  __syncthreads();
  // greedy scan by wave 0: lane w owns removed-word w
  if (tid < 64) {
    unsigned long long removed = 0ull;
    int kept = 0;
    for (int i = 0; i < nv; ++i) {
      const unsigned long long wrd = __shfl(removed, i >> 6, 64);
      const bool dead = (wrd >> (i & 63)) & 1ull;
      if (!dead) {
        if (kept < max_num || max_num <= 0) {
          if (tid == 0) keep[(long)b * max_num + kept] = idx[i] & 0x3fffffff;
        }
        ++kept;
        if (tid < nw) removed |= mask[i][tid];
      }
    }
    const int kk = max_num > 0 ? min(kept, max_num) : kept;
    for (int i = kk + tid; i < max_num; i += 64) keep[(long)b * max_num + i] = -1;
    if (tid == 0) n_keep[b] = kk;
  }
}
extern "C" int gr_nms_f32(const float* boxes_cxcywh, const float* scores, int B, int n, float iou_thr, float score_thr,
                          int max_num, const int* n_valid, long* keep, int* n_keep, hipStream_t stream) {
  if (!boxes_cxcywh || !scores || !keep || !n_keep || B <= 0 || n <= 0 || n > NMS_MAX || max_num <= 0) return GR_EINVAL;
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(512), 0, stream, boxes_cxcywh, scores, keep, n_keep, n_valid, n, iou_thr,
                     score_thr, max_num);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
