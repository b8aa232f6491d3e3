// Shared pieces of the bf16 MFMA GEMM kernels (128x128 two-barrier kernel, 256x256 ping-pong kernel).
#pragma once
#include "gr_common.h"
#include "../../include/groma_hip.h"

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
  const float* a_scale;  // fp8 path: per-row dequantisation scale of A [M] (or null)
  const float* w_scale;  // fp8 path: per-output-channel scale of W [N] (or null)
  float* ws;
  const float* a_parts;  // decode GEMV only: A given as un-merged single-query attention slices (decode.hip), or null
  int a_nsplit, a_hd;
  int M, N, K;
  long lda, ldw, ldc, ldr;
  int act, out_f32, splits;
  int conv_H, conv_W, conv_C;
  long conv_seg_stride;
  int resid_mod;
  int c_group, c_group_stride, c_row_off;
  int tiles_m, tiles_n;
  int yield;  // ping-pong kernel: one workgroup per tile instead of the persistent grid (gr_gemm_yield)
};

// XCD-aware, L2-friendly tile order: consecutive ids on one XCD (block b runs on XCD b%8),
// grouped so 8 row-tiles share each W panel.
__device__ __forceinline__ void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
#ifndef GR_TILE_GROUP
#define GR_TILE_GROUP 8
#endif
  const int GROUP = GR_TILE_GROUP;
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

// A-operand row base (elements) of output row m: plain row-major, or the top-left tap of the 3x3 window in the
// zero-bordered NHWC map (implicit-GEMM convolution)
__device__ __forceinline__ long a_row_base(const GemmArgs& p, int m) {
  if (p.conv_C > 0) {
    const int hw = p.conv_H * p.conv_W;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int y = rem / p.conv_W;
    const int x = rem - y * p.conv_W;
    return ((long)(img * (p.conv_H + 2) + y) * (p.conv_W + 2) + x) * p.conv_C;
  }
  return (long)m * p.lda;
}
// A-operand K offset (elements) of K-step ks (kt elements wide: 64, or 128 in the e4m3 build of the ping-pong kernel): plain k,
// or (segment, tap, channel) of the conv gather (a K-step never straddles a tap: conv_C % kt == 0)
__device__ __forceinline__ long a_k_off(const GemmArgs& p, int ks, int kt = 64) {
  const long k0 = (long)ks * kt;
  if (p.conv_C > 0) {
    const int tapc = (int)(k0 / p.conv_C);  // segment*9 + tap
    const int c0 = (int)(k0 - (long)tapc * p.conv_C);
    const int seg = tapc / 9;
    const int tap = tapc - seg * 9;
    const int ky = tap / 3, kx = tap - ky * 3;
    return (long)seg * p.conv_seg_stride + (long)(ky * (p.conv_W + 2) + kx) * p.conv_C + c0;
  }
  return k0;
}

// ---- coalesced epilogue -------------------------------------------------------------------------------------
// The MFMA accumulator layout gives each lane 4 columns of ONE row, so a direct store touches 16 rows x 32 B per
// wave-instruction (store-issue bound: the bf16 output of a K=1024..4096 GEMM is its largest HBM stream).  Instead
// the tile goes through LDS once (free after the K loop): accumulators are written as float4 into an XOR-swizzled
// [rows][COLS] fp32 image (conflict-free ds_write_b128), then re-read row-contiguously so that every lane applies
// the epilogue to 4/8/16 consecutive columns and stores 16 B, 16-32 lanes covering one full row segment.
template <int COLS>  // fp32 columns per staged row (128 or 256)
__device__ __forceinline__ void stage_write4(char* base, int row, int n4, f32x4 v) {
  *(f32x4*)(base + ((long)row * COLS * 4) + (((n4 ^ (row & 7))) << 4)) = v;
}
template <int COLS>
__device__ __forceinline__ f32x4 stage_read4(const char* base, int row, int n4) {
  return *(const f32x4*)(base + ((long)row * COLS * 4) + (((n4 ^ (row & 7))) << 4));
}
__device__ __forceinline__ f32x4 epi_math4(const GemmArgs& p, f32x4 v, int m, long orow, int n, f32x4 bias, f32x4 scale) {
  v += bias;
  if (p.act == 1 || p.act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
  }
  v *= scale;
  if (p.resid) {
    const long rrow = p.resid_mod > 0 ? (long)(m % p.resid_mod) : orow;
    v += *(const f32x4*)(p.resid + rrow * p.ldr + n);
  }
  return v;
}
// Per-thread column constants: a thread's tile columns are the same for every row it finishes, so bias / LayerScale
// are loaded ONCE per thread (zero / one when absent).
template <int W4>
struct EpiCols {
  f32x4 bias[W4], scale[W4], wsc[W4];
  __device__ __forceinline__ void load(const GemmArgs& p, int n) {
#pragma unroll
    for (int w = 0; w < W4; ++w) {
      const bool in = n + 4 * w < p.N;
      wsc[w] = (p.w_scale && in) ? *(const f32x4*)(p.w_scale + n + 4 * w) : (f32x4){1.f, 1.f, 1.f, 1.f};
      bias[w] = (p.bias && in) ? *(const f32x4*)(p.bias + n + 4 * w) : (f32x4){0.f, 0.f, 0.f, 0.f};
      scale[w] = (p.scale && in) ? *(const f32x4*)(p.scale + n + 4 * w) : (f32x4){1.f, 1.f, 1.f, 1.f};
    }
  }
};
// ---- batched epilogue: NIT staged rows per thread in ONE straight-line sequence ---------------------------------
// vmcnt is an in-order counter shared by loads and stores: a row-at-a-time epilogue with a (possibly skipped) residual /
// scale load inside each row made the compiler put `s_waitcnt vmcnt(0)` on the common path, so every 16-B store had to
// COMPLETE (~800 clk round trip) before the next row started -- 3300 clk per 64-row pass, 8 us per 256x256 tile,
// measured with tests/diag/gemm_clk.py.  Here (1) every global LOAD of the pass (residual rows, per-row dequant
// scales) is issued first, (2) then LDS reads + math + stores run back to back with no load in between, and the
// variants without loads are separate instantiations with no wait at all.
// rm(it) -> staged row; mm(it) -> global row m.
#ifdef GR_EPI_NT
#define EPI_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define EPI_ST(ptr, val) (*(ptr) = (val))
#endif
// FULL = the tile lies entirely inside M x N and the launch uses neither the output-row remap (c_group) nor the residual-row
// broadcast (resid_mod) -- tile-uniform, decided once per tile: every row / column guard and the integer divisions of the
// remaps disappear at compile time, row addresses are base + constant * ld, and a pass is ONE straight-line block.  That is what lets the compiler count its waits: vmcnt
// is an in-order counter shared by loads and stores, and with a data-dependent `if (m >= M) continue;` between the loads
// at the top and their uses, hipcc's wait insertion merges the branch states conservatively and emits `s_waitcnt vmcnt(0)`
// in front of every row -- each 16-B store then had to COMPLETE (~850 clk round trip) before the next row's LDS read was
// consumed: 4 rows x ~850 clk = the "3300 clk per 64-row pass" of round 1/2, 8.5 us per 256x256 tile (found in the ISA in
// round 3; the round-2 comment that the waits were gone was wrong for every tile but none).  Edge tiles keep the guards.
template <int COLS, int W4, int NIT, bool F32OUT, bool RESID, bool DEQ, bool FULL, class SR, class MR>
__device__ __forceinline__ void epi_rows(const GemmArgs& p, const char* base, int c4, int n0, const EpiCols<W4>& ec, SR rm, MR mm) {
  const int n = n0 + c4 * 4;
  constexpr int NB = RESID ? (DEQ ? 1 : (NIT < 4 ? NIT : 4)) : (DEQ ? 1 : NIT);  // rows per load batch (register budget; the fp8 build is at the 256-VGPR limit)
  auto out_row = [&](int m) -> long {  // FULL also means: no output-row remap, no residual-row broadcast (plain row-major tile)
    return (!FULL && p.c_group > 0) ? (long)(m / p.c_group) * p.c_group_stride + p.c_row_off + (m % p.c_group) : (long)m;
  };
#pragma unroll
  for (int b0 = 0; b0 < NIT; b0 += NB) {
    f32x4 r[RESID ? NB : 1][W4];
    float as[DEQ ? NB : 1];
    if (DEQ) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int m = mm(b0 + i);
        as[DEQ ? i : 0] = (p.a_scale && (FULL || m < p.M)) ? p.a_scale[m] : 1.f;
      }
    }
    if (RESID) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int m = mm(b0 + i);
        const long rrow = (!FULL && p.resid_mod > 0) ? (long)(m % p.resid_mod) : out_row(m);
#pragma unroll
        for (int w = 0; w < W4; ++w)
          r[RESID ? i : 0][w] = (FULL || (m < p.M && n + 4 * w < p.N)) ? *(const f32x4*)(p.resid + rrow * p.ldr + n + 4 * w)
                                                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int m = mm(b0 + i);
      if (!FULL && m >= p.M) continue;
      const long orow = out_row(m);
      f32x4 v[W4];
#pragma unroll
      for (int w = 0; w < W4; ++w) {
        v[w] = stage_read4<COLS>(base, rm(b0 + i), c4 + w);
        if (DEQ) v[w] = v[w] * ec.wsc[w] * as[DEQ ? i : 0];
        v[w] += ec.bias[w];
        if (p.act == 1) {  // (one uniform branch per vector, not one per element)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[w][e] = gelu_erf(v[w][e]);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[w][e] = fmaxf(v[w][e], 0.f);
        }
        v[w] *= ec.scale[w];
        if (RESID) v[w] += r[RESID ? i : 0][w];
      }
      if (F32OUT) {
#pragma unroll
        for (int w = 0; w < W4; ++w)
          if (FULL || n + 4 * w < p.N) EPI_ST((f32x4*)((float*)p.C + orow * p.ldc + n + 4 * w), v[w]);
      } else {
        // (split build: p.ldc is the physical row stride; a thread's 8 columns lie inside one 32-column hi / lo block pair)
        bf16_t* dst = (bf16_t*)p.C + orow * p.ldc + sp_idx(n);
        if (W4 == 2 && (FULL || n + 8 <= p.N) && (p.ldc & 7) == 0) {
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
          uint32_t h[4], l[4];
          split2(v[0][0], v[0][1], h[0], l[0]);
          split2(v[0][2], v[0][3], h[1], l[1]);
          split2(v[W4 - 1][0], v[W4 - 1][1], h[2], l[2]);
          split2(v[W4 - 1][2], v[W4 - 1][3], h[3], l[3]);
          const u32x4 hi4 = {h[0], h[1], h[2], h[3]};
          EPI_ST((u32x4*)dst, hi4);
#if GR_SP
          const u32x4 lo4 = {l[0], l[1], l[2], l[3]};
          EPI_ST((u32x4*)(dst + 32), lo4);
#endif
        } else {
#pragma unroll
          for (int w = 0; w < W4; ++w)
            if (n + 4 * w < p.N) st4f((bf16_t*)p.C + orow * p.ldc, n + 4 * w, v[w]);
        }
      }
    }
  }
}
// SwiGLU over interleaved (gate, up) columns: 16 fused columns -> 8 bf16 outputs per thread-row; no loads
template <int COLS, int NIT, bool DEQ, bool FULL, class SR, class MR>
__device__ __forceinline__ void epi_rows_swiglu(const GemmArgs& p, const char* base, int c4, int n0, const EpiCols<4>& ec, SR rm, MR mm) {
  const int n = n0 + c4 * 4;
  float as[DEQ ? NIT : 1];
  if (DEQ) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = mm(it);
      as[DEQ ? it : 0] = (p.a_scale && (FULL || m < p.M)) ? p.a_scale[m] : 1.f;
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int m = mm(it);
    if (!FULL && m >= p.M) continue;
    long orow = m;
    if (!FULL && p.c_group > 0) orow = (long)(m / p.c_group) * p.c_group_stride + p.c_row_off + (m % p.c_group);
    uint32_t o[4], ol[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f32x4 x = stage_read4<COLS>(base, rm(it), c4 + w);
      if (DEQ) x = x * ec.wsc[w] * as[DEQ ? it : 0];
      x += ec.bias[w];
      split2(silu_f(x[0]) * x[1], silu_f(x[2]) * x[3], o[w], ol[w]);
    }
    bf16_t* dst = (bf16_t*)p.C + orow * p.ldc + sp_idx(n >> 1);
    if ((FULL || n + 16 <= p.N) && (p.ldc & 7) == 0) {
      *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);
#if GR_SP
      *(uint4*)(dst + 32) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
#endif
    } else {
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (n + 4 * w < p.N) {
          *(uint32_t*)(dst + 2 * w) = o[w];
#if GR_SP
          *(uint32_t*)(dst + 32 + 2 * w) = ol[w];
#endif
        }
    }
  }
}
// split-K partial tile -> workspace [z, M, N] f32 (raw accumulators; the reduce kernel applies the epilogue)
template <int COLS, int NIT, bool FULL, class SR, class MR>
__device__ __forceinline__ void epi_rows_splitk(const GemmArgs& p, const char* base, int c4, int n0, int z, SR rm, MR mm) {
  const int n = n0 + c4 * 4;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int m = mm(it);
    if (FULL || (m < p.M && n < p.N)) *(f32x4*)(p.ws + ((long)z * p.M + m) * p.N + n) = stage_read4<COLS>(base, rm(it), c4);
  }
}
// mode dispatch shared by both tile kernels: TPR1/2/4 = threads per staged row for 4/8/16 columns per thread
template <int COLS, int NTH, int ROWS, bool DEQ, bool FULL, class SR, class MR>
__device__ __forceinline__ void epi_dispatch_t(const GemmArgs& p, const char* base, int tid, int n0, int z, const EpiCols<4>& ec4,
                                               const EpiCols<2>& ec2, const EpiCols<1>& ec1, SR rm, MR mm) {
  constexpr int T4 = COLS / 16, T2 = COLS / 8, T1 = COLS / 4;          // threads per row
  constexpr int N4 = ROWS * T4 / NTH, N2 = ROWS * T2 / NTH, N1 = ROWS * T1 / NTH;  // rows per thread
  if (p.splits > 1) {
    epi_rows_splitk<COLS, N1, FULL>(p, base, tid % T1, n0, z, [&](int it) { return rm(it * (NTH / T1) + tid / T1); },
                                    [&](int it) { return mm(it * (NTH / T1) + tid / T1); });
  } else if (p.act == 3) {
    epi_rows_swiglu<COLS, N4, DEQ, FULL>(p, base, (tid % T4) * 4, n0, ec4, [&](int it) { return rm(it * (NTH / T4) + tid / T4); },
                                         [&](int it) { return mm(it * (NTH / T4) + tid / T4); });
  } else if (!p.out_f32) {
    auto r = [&](int it) { return rm(it * (NTH / T2) + tid / T2); };
    auto m = [&](int it) { return mm(it * (NTH / T2) + tid / T2); };
    if (p.resid) epi_rows<COLS, 2, N2, false, true, DEQ, FULL>(p, base, (tid % T2) * 2, n0, ec2, r, m);
    else epi_rows<COLS, 2, N2, false, false, DEQ, FULL>(p, base, (tid % T2) * 2, n0, ec2, r, m);
  } else {
    auto r = [&](int it) { return rm(it * (NTH / T1) + tid / T1); };
    auto m = [&](int it) { return mm(it * (NTH / T1) + tid / T1); };
    if (p.resid) epi_rows<COLS, 1, N1, true, true, DEQ, FULL>(p, base, tid % T1, n0, ec1, r, m);
    else epi_rows<COLS, 1, N1, true, false, DEQ, FULL>(p, base, tid % T1, n0, ec1, r, m);
  }
}
// `full` must be tile-uniform: the tile's rows [m0, m0 + tile) and columns [n0, n0 + tile) all lie inside M x N
template <int COLS, int NTH, int ROWS, bool DEQ, class SR, class MR>
__device__ __forceinline__ void epi_dispatch(const GemmArgs& p, const char* base, int tid, int n0, int z, const EpiCols<4>& ec4,
                                             const EpiCols<2>& ec2, const EpiCols<1>& ec1, SR rm, MR mm, bool full) {
  if (full) epi_dispatch_t<COLS, NTH, ROWS, DEQ, true>(p, base, tid, n0, z, ec4, ec2, ec1, rm, mm);
  else epi_dispatch_t<COLS, NTH, ROWS, DEQ, false>(p, base, tid, n0, z, ec4, ec2, ec1, rm, mm);
}

// skinny M<=8 streaming kernel (gemv_bf16.hip): requires p.splits == ceil(K/512) and p.ws
int gr_launch_gemv(const GemmArgs& p, hipStream_t stream);
int gr_launch_gemm_skinny(const GemmArgs& p, hipStream_t stream);      // gemm_skinny.hip: decode steps of 9..64 rows (tile == 3)
int gr_launch_gemm_skinny_fp8(const GemmArgs& p, hipStream_t stream);  // ... with e4m3 operands (gemm_skinny_fp8.hip)
// (256 | 192) x 256 x 64 ping-pong kernel (gemm_bf16_256.hip); p.tiles_m counts row tiles of height tile_rows
int gr_launch_gemm256(const GemmArgs& p, hipStream_t stream, int tile_rows);
// OCP-fp8 build of the same kernel (gemm_fp8_256.hip); A/W are e4m3 bytes, K % 128 == 0
int gr_launch_gemm256_fp8(const GemmArgs& p, hipStream_t stream);
