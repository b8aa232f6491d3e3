// bf16 MFMA GEMM family for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Covers every dense contraction on the Groma hot path (SURVEY.md §8a rows a1, a3,
// a15, a17, a20, a21): ViT/LLaMA projections and MLPs, the img_txt_bridge, the
// lm_head (+ extra head), and -- through the implicit-GEMM A gather -- the 3x3
// convolutions of the region encoder (reference: groma/model/roi_align.py:131-143,
// :251-253) over zero-bordered NHWC bf16 feature maps.
//
// Structure (cdna_hip_programming.md §5): 128x128x64 tile, 256 threads = 4 waves
// (2x2), each wave a 64x64 output sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Operands are staged HBM -> LDS with global_load_lds (16 B/lane), double buffered,
// one barrier per K-step; the LDS image is lane-linear and bank-conflict-free via a
// source-side XOR swizzle (chunk ^= row&7) mirrored on the ds_read_b128 side.
// The MFMA is issued "swapped" (W as the A operand) so each lane ends up with 4
// consecutive output columns of one row -> vector epilogue loads/stores.
#include "gr_common.h"
#include "../../include/groma_hip.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const float* bias;
  const float* scale;
  const float* resid;
  float* ws;
  int M, N, K;
  long lda, ldw, ldc, ldr;
  int act, out_f32, splits;
  int conv_H, conv_W, conv_C;
  long conv_seg_stride;
  int resid_mod;
  int c_group, c_group_stride, c_row_off;
  int tiles_m, tiles_n;
};

// XCD-aware, L2-friendly tile order: consecutive ids on one XCD (block b runs on XCD b%8),
// grouped so 8 row-tiles share each W panel.
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = pid / per_group;
  const int first_m = g * GROUP;
  const int gsize = min(tiles_m - first_m, GROUP);
  const int in_g = pid - g * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return gelu_erf(v);
  if (act == 2) return fmaxf(v, 0.f);
  return v;
}

__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: buf b in {0,1}: A tile at b*32768, W tile at b*32768 + 16384
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // K range of this split
  const int ksteps_total = p.K / BK;
  const int z = blockIdx.y;
  const int ks_per = (ksteps_total + p.splits - 1) / p.splits;
  const int ks_begin = z * ks_per;
  const int ks_end = min(ksteps_total, ks_begin + ks_per);
  const int nt = ks_end - ks_begin;

  // ---- per-thread staging sources: 4 A chunks + 4 W chunks per K-step ----
  // chunk q = i*256 + tid: row = q>>3, lds chunk position = q&7, logical k-chunk = pos ^ (row&7)
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid;
    const int row = q >> 3;
    const int kc = (q & 7) ^ (row & 7);
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    long abase;
    if (p.conv_C > 0) {
      const int hw = p.conv_H * p.conv_W;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int y = rem / p.conv_W;
      const int x = rem - y * p.conv_W;
      abase = ((long)(img * (p.conv_H + 2) + y) * (p.conv_W + 2) + x) * p.conv_C;
    } else {
      abase = (long)m * p.lda;
    }
    a_src[i] = p.A + abase + kc * 8;
    int n = n0 + row;
    if (n > p.N - 1) n = p.N - 1;
    w_src[i] = p.W + (long)n * p.ldw + kc * 8;
  }
  const int lds_wave_off = wave * 1024;  // this wave's 64 lanes * 16 B within each 4 KB instruction group

  auto stage = [&](int ks, int buf) {
    const long k0 = (long)ks * BK;
    long aoff;
    if (p.conv_C > 0) {
      const int tapc = (int)(k0 / p.conv_C);  // segment*9 + tap
      const int c0 = (int)(k0 - (long)tapc * p.conv_C);
      const int seg = tapc / 9;
      const int tap = tapc - seg * 9;
      const int ky = tap / 3, kx = tap - ky * 3;
      aoff = (long)seg * p.conv_seg_stride + (long)(ky * (p.conv_W + 2) + kx) * p.conv_C + c0;
    } else {
      aoff = k0;
    }
    char* abuf = smem + buf * 32768;
    char* wbuf = abuf + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + aoff, abuf + i * 4096 + lds_wave_off);
      glds16(w_src[i] + k0, wbuf + i * 4096 + lds_wave_off);
    }
  };

  f32x4 acc[4][4];  // [j: n-tile][i: m-tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15;  // fragment row
  const int fg = lane >> 4;  // k-group

  if (nt > 0) stage(ks_begin, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) stage(ks_begin + t + 1, (t + 1) & 1);
    const char* abuf = smem + (t & 1) * 32768;
    const char* wbuf = abuf + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], wf[4];
      const int c = kk * 4 + fg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        af[i] = *(const bf16x8*)(abuf + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        wf[j] = *(const bf16x8*)(wbuf + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds n = nb + fg*4 + {0..3}, m = mb + fr ----
  const bool partial = p.splits > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
    long orow = m;
    if (p.c_group > 0) orow = (long)(m / p.c_group) * p.c_group_stride + p.c_row_off + (m % p.c_group);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[j][i];
      if (partial) {
        float* dst = p.ws + ((long)z * p.M + m) * p.N + n;
        *(f32x4*)dst = v;
        continue;
      }
      if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
        v += b;
      }
      if (p.act == 3) {  // SwiGLU on interleaved (gate, up) pairs
        const float o0 = silu_f(v[0]) * v[1];
        const float o1 = silu_f(v[2]) * v[3];
        bf16_t* dst = (bf16_t*)p.C + orow * p.ldc + (n >> 1);
        *(uint32_t*)dst = pack2bf(o0, o1);
        continue;
      }
      if (p.act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
      }
      if (p.scale) {
        const f32x4 s = *(const f32x4*)(p.scale + n);
        v *= s;
      }
      if (p.resid) {
        const long rrow = p.resid_mod > 0 ? (long)(m % p.resid_mod) : orow;
        const f32x4 r = *(const f32x4*)(p.resid + rrow * p.ldr + n);
        v += r;
      }
      if (p.out_f32) {
        *(f32x4*)((float*)p.C + orow * p.ldc + n) = v;
      } else {
        uint2 pk;
        pk.x = pack2bf(v[0], v[1]);
        pk.y = pack2bf(v[2], v[3]);
        *(uint2*)((bf16_t*)p.C + orow * p.ldc + n) = pk;
      }
    }
  }
}

// split-K reduce + epilogue: one thread per 4 consecutive n
__global__ void gemm_splitk_reduce_kernel(GemmArgs p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = p.N >> 2;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long)m * n4) << 2;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.splits; ++z) v += *(const f32x4*)(p.ws + ((long)z * p.M + m) * p.N + n);
  long orow = m;
  if (p.c_group > 0) orow = (long)(m / p.c_group) * p.c_group_stride + p.c_row_off + (m % p.c_group);
  if (p.bias) v += *(const f32x4*)(p.bias + n);
  if (p.act == 3) {
    bf16_t* dst = (bf16_t*)p.C + orow * p.ldc + (n >> 1);
    *(uint32_t*)dst = pack2bf(silu_f(v[0]) * v[1], silu_f(v[2]) * v[3]);
    return;
  }
  if (p.act) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
  }
  if (p.scale) v *= *(const f32x4*)(p.scale + n);
  if (p.resid) {
    const long rrow = p.resid_mod > 0 ? (long)(m % p.resid_mod) : orow;
    v += *(const f32x4*)(p.resid + rrow * p.ldr + n);
  }
  if (p.out_f32) {
    *(f32x4*)((float*)p.C + orow * p.ldc + n) = v;
  } else {
    uint2 pk;
    pk.x = pack2bf(v[0], v[1]);
    pk.y = pack2bf(v[2], v[3]);
    *(uint2*)((bf16_t*)p.C + orow * p.ldc + n) = pk;
  }
}

// ---- timing hook (bench.py roofline leg): HIP events on the launch stream around every GEMM launch ----
#include <vector>
static bool g_prof_on = false;
struct ProfRec { hipEvent_t a, b; double flops; };
static std::vector<ProfRec> g_prof;

extern "C" int gr_abi_version(void) { return GROMA_HIP_ABI_VERSION; }
extern "C" int gr_prof_enable(int on) {
  g_prof_on = on != 0;
  return GR_OK;
}
extern "C" int gr_prof_read(double* total_ms, long* launches, double* flops) {
  double ms = 0.0, fl = 0.0;
  long n = 0;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return GR_EINVAL;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return GR_EINVAL;
    ms += t; fl += r.flops; ++n;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  return GR_OK;
}

extern "C" int gr_gemm_bf16(const gr_gemm_desc* d, hipStream_t stream) {
  if (!d || !d->A || !d->W || !d->C) return GR_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return GR_EINVAL;
  if (d->K % BK != 0 || d->N % 4 != 0) return GR_EINVAL;
  if (d->conv_C > 0 && (d->conv_C % BK != 0 || d->K % (9 * d->conv_C) != 0)) return GR_EINVAL;
  const int splits = d->splits > 1 ? d->splits : 1;
  if (splits > 1 && !d->ws) return GR_EINVAL;
  if (d->act == 3 && (d->resid || d->scale || d->out_f32)) return GR_EINVAL;
  GemmArgs p;
  p.A = (const bf16_t*)d->A;
  p.W = (const bf16_t*)d->W;
  p.C = d->C;
  p.bias = d->bias;
  p.scale = d->scale;
  p.resid = d->resid;
  p.ws = d->ws;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
  p.act = d->act; p.out_f32 = d->out_f32; p.splits = splits;
  p.conv_H = d->conv_H; p.conv_W = d->conv_W; p.conv_C = d->conv_C;
  p.conv_seg_stride = d->conv_seg_stride;
  p.resid_mod = d->resid_mod;
  p.c_group = d->c_group; p.c_group_stride = d->c_group_stride; p.c_row_off = d->c_row_off;
  p.tiles_m = gr_cdiv(p.M, BM);
  p.tiles_n = gr_cdiv(p.N, BN);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, splits);
  ProfRec rec;
  if (g_prof_on) {
    (void)hipEventCreate(&rec.a);
    (void)hipEventCreate(&rec.b);
    rec.flops = 2.0 * p.M * (double)p.N * p.K;
    (void)hipEventRecord(rec.a, stream);
  }
  hipLaunchKernelGGL(gemm_bf16_kernel, grid, dim3(NTHREADS), 65536, stream, p);
  if (g_prof_on) {
    (void)hipEventRecord(rec.b, stream);
    g_prof.push_back(rec);
  }
  GR_CHECK_LAUNCH();
  if (splits > 1) {
    const long tot = (long)p.M * (p.N >> 2);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gr_cdiv(tot, 256)), dim3(256), 0, stream, p);
    GR_CHECK_LAUNCH();
  }
  return GR_OK;
}
