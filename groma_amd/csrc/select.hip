// Index-producing selection kernels (bit-exact vs oracle/): compiled with -ffp-contract=off so the
// fp32 IoU arithmetic is the plain IEEE sequence of mmcv's CPU NMS.
//  * topk_desc:  two-stage proposal selection, top-K of S class logits per image
//                (groma/model/ddetr_transformer.py:555-556), order = (value desc, index asc).
//  * nms:        mmcv.ops.nms as called at groma/model/groma.py:266-272 -- score threshold
//                (mmcv/ops/nms.py:21-26), stable descending sort, greedy IoU suppression in the CPU form
//                inter/(a+b-inter) > thr (mmcv/ops/csrc/pytorch/cpu/nms.cpp:28-53), first max_num kept.
//                Boxes arrive as (cx,cy,w,h) and are converted with HF center_to_corners_format.
// One workgroup per image up to n <= 512 (NMS) / 1024 (top-k): the O(n^2) IoU test is a 64-bit-mask matrix (wave64 =
// one word per 64 candidates) followed by a single-wave greedy scan: no host round trip (the reference's CUDA path
// copies the mask to the host, mmcv/ops/csrc/pytorch/cuda/nms_cuda.cu:27-50).  Larger inputs (<= 4096) take the same
// three steps as three launches over a caller-owned workspace (sort | mask matrix | scan) instead of failing.
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define NMAX 1024     // sort capacity of the one-workgroup fast paths (top-k of S <= 1024, NMS of n <= 512)
#define NMS_MAX 512   // NMS candidates per image on the fast path (300 proposals + refer + ground boxes)
#define BIG_MAX 4096  // capacity of the general paths (larger grids / candidate sets): 32 KB of LDS for the sort

// bitonic sort of (key desc, idx asc) pairs in LDS; n padded to CAP with (-inf, big idx)
__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) {
  // true if (ka, ia) must come before (kb, ib)
  return ka > kb || (ka == kb && ia < ib);
}
template <int CAP>
__device__ void bitonic_sort_desc(float* key, int* idx, int tid, int nthreads) {
  for (int k = 2; k <= CAP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < CAP; i += nthreads) {
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

template <int CAP>
__global__ __launch_bounds__(512) void topk_desc_kernel(const float* __restrict__ x, int* __restrict__ out, int S, int K,
                                                        long ldx) {
  __shared__ float key[CAP];
  __shared__ int idx[CAP];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < CAP; i += 512) {
    key[i] = i < S ? x[(long)b * ldx + i] : -INFINITY;
    idx[i] = i < S ? i : 0x7fffffff - (CAP - i);
  }
  bitonic_sort_desc<CAP>(key, idx, tid, 512);
  for (int i = tid; i < K; i += 512) out[(long)b * K + i] = idx[i];
}
extern "C" int gr_topk_desc(const float* x, int* out_idx, int B, int S, int K, long ldx, hipStream_t stream) {
  if (!x || !out_idx || B <= 0 || S <= 0 || S > BIG_MAX || K <= 0 || K > S) return GR_EINVAL;
  if (S <= NMAX)
    hipLaunchKernelGGL(topk_desc_kernel<NMAX>, dim3(B), dim3(512), 0, stream, x, out_idx, S, K, ldx);
  else  // feature grids beyond 32x32 (e.g. 37x37 at 518 px)
    hipLaunchKernelGGL(topk_desc_kernel<BIG_MAX>, dim3(B), dim3(512), 0, stream, x, out_idx, S, K, ldx);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

// ---- NMS ---------------------------------------------------------------------------------------------------------
// fmt 0: boxes are (cx,cy,w,h), converted with HF center_to_corners_format (the call at groma/model/groma.py:268);
// fmt 1: boxes are (x1,y1,x2,y2) as mmcv's `nms(boxes, scores, iou_threshold, offset)` takes them (pybind.cpp:175).
// `off` is mmcv's offset (0 or 1) in areas and intersections (cpu/nms.cpp:15,44-45).
__device__ __forceinline__ void load_corners(const float* bx, int o, int fmt, float& a, float& c, float& d, float& e) {
  const float v0 = bx[o * 4 + 0], v1 = bx[o * 4 + 1], v2 = bx[o * 4 + 2], v3 = bx[o * 4 + 3];
  if (fmt == 0) {
    a = v0 - 0.5f * v2; c = v1 - 0.5f * v3; d = v0 + 0.5f * v2; e = v1 + 0.5f * v3;
  } else {
    a = v0; c = v1; d = v2; e = v3;
  }
}
// bit jj of the result: sorted box j0+jj (j > i) is suppressed by sorted box i
__device__ __forceinline__ unsigned long long suppress_word(const float* x1, const float* y1, const float* x2,
                                                            const float* y2, const float* area, int i, int j0, int nv,
                                                            float iou_thr, float off) {
  unsigned long long m = 0ull;
  const float ix1 = x1[i], iy1 = y1[i], ix2 = x2[i], iy2 = y2[i], ia = area[i];
  for (int jj = 0; jj < 64; ++jj) {
    const int j = j0 + jj;
    if (j > i && j < nv) {
      const float xx1 = fmaxf(ix1, x1[j]), yy1 = fmaxf(iy1, y1[j]);
      const float xx2 = fminf(ix2, x2[j]), yy2 = fminf(iy2, y2[j]);
      const float ww = fmaxf(0.f, xx2 - xx1 + off), hh = fmaxf(0.f, yy2 - yy1 + off);
      const float inter = ww * hh;
      const float ovr = inter / (ia + area[j] - inter);
      if (ovr > iou_thr) m |= 1ull << jj;
    }
  }
  return m;
}

// boxes f32 [B, n, 4], scores f32 [B, n]  ->  keep int64 [B, max_num] (indices into the n inputs, in
// descending-score order, -1 padded), n_keep int32 [B].  n_valid (optional, [B]): per-image candidate count <= n.
__global__ __launch_bounds__(512) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                  long* __restrict__ keep, int* __restrict__ n_keep,
                                                  const int* __restrict__ n_valid, int n, float iou_thr, float score_thr,
                                                  int max_num, int fmt, float off) {
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
  bitonic_sort_desc<NMAX>(key, idx, tid, 512);
  const int nv = nvalid_s;  // valid entries occupy sorted positions [0, nv) unless a valid score is -inf
  for (int i = tid; i < nv; i += 512) {
    float a, c, d, e;
    load_corners(bx, idx[i] & 0x3fffffff, fmt, a, c, d, e);
    x1[i] = a; y1[i] = c; x2[i] = d; y2[i] = e;
    area[i] = (d - a + off) * (e - c + off);
  }
  __syncthreads();
  const int nw = (nv + 63) >> 6;
  // mask[i][w] bit j: sorted box (w*64+j) is suppressed by sorted box i (only j > i matters)
  for (int t = tid; t < nv * nw; t += 512) {
    const int i = t / nw, w = t - i * nw;
    mask[i][w] = suppress_word(x1, y1, x2, y2, area, i, w * 64, nv, iou_thr, off);
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

// ---- general path, 512 < n <= 4096: the same three steps as three launches over a caller-owned workspace ----------
// per image: idx int[BIG_MAX] | x1,y1,x2,y2,area f32[BIG_MAX] each | nv int (padded to 64 B) | mask u64[BIG_MAX][64]
#define BIG_WORDS (BIG_MAX / 64)
struct NmsWs {
  int* idx; float *x1, *y1, *x2, *y2, *area; int* nv; unsigned long long* mask;
};
static inline size_t nms_ws_per_image() { return (size_t)BIG_MAX * 24 + 64 + (size_t)BIG_MAX * BIG_WORDS * 8; }
__host__ __device__ static inline NmsWs nms_ws_at(char* base, size_t per, int b) {
  char* p = base + per * (size_t)b;
  NmsWs w;
  w.idx = (int*)p; p += BIG_MAX * 4;
  w.x1 = (float*)p; p += BIG_MAX * 4;
  w.y1 = (float*)p; p += BIG_MAX * 4;
  w.x2 = (float*)p; p += BIG_MAX * 4;
  w.y2 = (float*)p; p += BIG_MAX * 4;
  w.area = (float*)p; p += BIG_MAX * 4;
  w.nv = (int*)p; p += 64;
  w.mask = (unsigned long long*)p;
  return w;
}
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        const int* __restrict__ n_valid, int n, float score_thr, int fmt,
                                                        float off, char* ws, size_t per) {
  __shared__ float key[BIG_MAX];
  __shared__ int idx[BIG_MAX];
  __shared__ int nvalid_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bx = boxes + (long)b * n * 4;
  const float* sc = scores + (long)b * n;
  if (tid == 0) nvalid_s = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < BIG_MAX; i += 1024) {
    bool valid = i < n && (!n_valid || i < n_valid[b]);
    if (valid && score_thr > 0.f) valid = sc[i] > score_thr;
    key[i] = valid ? sc[i] : -INFINITY;
    idx[i] = i < n ? (valid ? i : 0x40000000 + i) : 0x7fffffff - (BIG_MAX - i);
    cnt += valid ? 1 : 0;
  }
  atomicAdd(&nvalid_s, cnt);
  bitonic_sort_desc<BIG_MAX>(key, idx, tid, 1024);
  const int nv = nvalid_s;
  const NmsWs w = nms_ws_at(ws, per, b);
  if (tid == 0) *w.nv = nv;
  for (int i = tid; i < nv; i += 1024) {
    const int o = idx[i] & 0x3fffffff;
    float a, c, d, e;
    load_corners(bx, o, fmt, a, c, d, e);
    w.idx[i] = o;
    w.x1[i] = a; w.y1[i] = c; w.x2[i] = d; w.y2[i] = e;
    w.area[i] = (d - a + off) * (e - c + off);
  }
}
__global__ __launch_bounds__(256) void nms_mask_kernel(char* ws, size_t per, float iou_thr, float off) {
  const NmsWs w = nms_ws_at(ws, per, blockIdx.y);
  const int nv = *w.nv;
  const int nw = (nv + 63) >> 6;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < (long)nv * nw; t += (long)gridDim.x * 256) {
    const int i = (int)(t / nw), wd = (int)(t - (long)i * nw);
    // rows only suppress later boxes: words entirely before i are zero
    w.mask[(long)i * BIG_WORDS + wd] = (wd * 64 + 63 > i) ? suppress_word(w.x1, w.y1, w.x2, w.y2, w.area, i, wd * 64, nv, iou_thr, off) : 0ull;
  }
}
__global__ __launch_bounds__(64) void nms_scan_kernel(char* ws, size_t per, long* __restrict__ keep, int* __restrict__ n_keep,
                                                      int max_num) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const NmsWs w = nms_ws_at(ws, per, b);
  const int nv = *w.nv;
  const int nw = (nv + 63) >> 6;
  unsigned long long removed = 0ull;
  int kept = 0;
  unsigned long long row = (nv > 0 && tid < nw) ? w.mask[tid] : 0ull;  // row i's word for this lane, fetched one row ahead
  for (int i = 0; i < nv; ++i) {
    const unsigned long long cur = row;
    if (i + 1 < nv) row = tid < nw ? w.mask[(long)(i + 1) * BIG_WORDS + tid] : 0ull;
    const unsigned long long wrd = __shfl(removed, i >> 6, 64);
    const bool dead = (wrd >> (i & 63)) & 1ull;
    if (!dead) {
      if (kept < max_num && tid == 0) keep[(long)b * max_num + kept] = w.idx[i];
      ++kept;
      removed |= cur;
    }
  }
  const int kk = min(kept, max_num);
  for (int i = kk + tid; i < max_num; i += 64) keep[(long)b * max_num + i] = -1;
  if (tid == 0) n_keep[b] = kk;
}

extern "C" long gr_nms_workspace_bytes(int B, int n) {
  if (n <= NMS_MAX || B <= 0) return 0;
  return (long)(nms_ws_per_image() * (size_t)B);
}

static int nms_launch(const float* boxes, const float* scores, int B, int n, float iou_thr, float score_thr, int max_num,
                      const int* n_valid, long* keep, int* n_keep, int fmt, float off, void* ws, hipStream_t stream) {
  if (!boxes || !scores || !keep || !n_keep || B <= 0 || n <= 0 || n > BIG_MAX || max_num <= 0) return GR_EINVAL;
  if (n <= NMS_MAX) {
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(512), 0, stream, boxes, scores, keep, n_keep, n_valid, n, iou_thr,
                       score_thr, max_num, fmt, off);
    GR_CHECK_LAUNCH();
    return GR_OK;
  }
  if (!ws) return GR_EINVAL;  // the general path needs gr_nms_workspace_bytes(B, n) bytes from the caller
  const size_t per = nms_ws_per_image();
  hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), 0, stream, boxes, scores, n_valid, n, score_thr, fmt, off,
                     (char*)ws, per);
  GR_CHECK_LAUNCH();
  const int nw = (n + 63) >> 6;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(gr_cdiv((long)n * nw, 256), B), dim3(256), 0, stream, (char*)ws, per, iou_thr, off);
  GR_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, stream, (char*)ws, per, keep, n_keep, max_num);
  GR_CHECK_LAUNCH();
  return GR_OK;
}

extern "C" int gr_nms_f32(const float* boxes_cxcywh, const float* scores, int B, int n, float iou_thr, float score_thr,
                          int max_num, const int* n_valid, long* keep, int* n_keep, void* ws, hipStream_t stream) {
  return nms_launch(boxes_cxcywh, scores, B, n, iou_thr, score_thr, max_num, n_valid, keep, n_keep, 0, 0.f, ws, stream);
}

// The reference's own op signature: mmcv `_ext.nms(boxes[n,4] xyxy, scores[n], iou_threshold, offset)` -> int64[k]
// (pybind.cpp:175; NMSop.forward calls it at mmcv/ops/nms.py:26-27 after its own score filter and before its max_num cut).
extern "C" int gr_nms(const float* boxes_xyxy, const float* scores, int n, float iou_threshold, int offset, long* keep,
                      int* n_keep, void* ws, hipStream_t stream) {
  if (offset != 0 && offset != 1) return GR_EINVAL;
  if (n == 0) {  // cpu/nms.cpp:6-8: empty input -> empty result
    if (!n_keep) return GR_EINVAL;
    hipError_t e = hipMemsetAsync(n_keep, 0, sizeof(int), stream);
    return e == hipSuccess ? GR_OK : (int)e;
  }
  return nms_launch(boxes_xyxy, scores, 1, n, iou_threshold, 0.f, n, nullptr, keep, n_keep, 1, (float)offset, ws, stream);
}
