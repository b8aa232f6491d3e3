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
#include "gemm_common.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256
// time of one 256^2 K-step on a CU relative to one 128^2 K-step of two co-resident blocks (4x the MACs of one
// block = 2x the work per CU-interval, executed ~1.45x faster per flop)
#define G256_COST 1.41
// the 192-row form of the ping-pong kernel (round 4): per-K-step cost and per-tile overhead of a round relative to the 256-row
// form's 1.41 / 13, and the margin by which it must win before it is chosen.  Measured (profiles/r04_gemm_tile_rows.txt): the
// K loop of a 192-row tile is only 3.5 % shorter than that of a 256-row tile (0.99 vs 1.03 us per K-tile at M = 2328: the loop
// is bound by its load side -- B staging, DMA issue -- which does not shrink with the rows), the epilogue a third shorter.
#define PP192_C 1.36
#define PP192_O 9.0
#define PP_MARGIN 0.97
// one round of <= 256 128^2-tiles runs ONE workgroup per CU instead of two co-resident ones: 0.70-0.72 of the per-tile time
#define E128_SOLO 0.72

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
    a_src[i] = p.A + a_row_base(p, m) + kc * 8;
    int n = n0 + row;
    if (n > p.N - 1) n = p.N - 1;
    w_src[i] = p.W + (long)n * p.ldw + kc * 8;
  }
  const int lds_wave_off = wave * 1024;  // this wave's 64 lanes * 16 B within each 4 KB instruction group

  auto stage = [&](int ks, int buf) {
    const long k0 = (long)ks * BK;
    const long aoff = a_k_off(p, ks);
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
#if GR_SP
    // split operands: k-step 0 of the K-tile = hi halves of 32 logical k-values, k-step 1 = their lo halves (gr_common.h);
    // hi.hi + hi(W).lo(A) + lo(W).hi(A) into the same accumulators
    bf16x8 af[2][4], wf[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        af[kk][i] = *(const bf16x8*)(abuf + row * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        wf[kk][j] = *(const bf16x8*)(wbuf + row * 128 + ((c ^ (row & 7)) << 4));
      }
    }
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[j][i] = GR_MFMA_16x16x32(wf[pp == 1][j], af[pp == 2][i], acc[j][i]);
#else
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
          acc[j][i] = GR_MFMA_16x16x32(wf[j], af[i], acc[j][i]);
    }
#endif
  }

  // ---- epilogue through LDS (coalesced; see gemm_common.h) ----
  __syncthreads();  // every wave is done reading the last K-tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) stage_write4<BN>(smem, wm * 64 + i * 16 + fr, wn * 16 + j * 4 + fg, acc[j][i]);
  __syncthreads();
  EpiCols<4> ec4;
  EpiCols<2> ec2;
  EpiCols<1> ec1;
  if (p.act == 3) ec4.load(p, n0 + (tid & 7) * 16);
  else if (!p.out_f32 && p.splits == 1) ec2.load(p, n0 + (tid & 15) * 8);
  else ec1.load(p, n0 + (tid & 31) * 4);
  epi_dispatch<BN, NTHREADS, BM, false>(p, smem, tid, n0, z, ec4, ec2, ec1, [](int sr) { return sr; }, [&](int sr) { return m0 + sr; },
                                        m0 + BM <= p.M && n0 + BN <= p.N && p.c_group == 0 && p.resid_mod == 0);
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
    bf16_t* dst = (bf16_t*)p.C + orow * p.ldc + sp_idx(n >> 1);
    uint32_t hi, lo;
    split2(silu_f(v[0]) * v[1], silu_f(v[2]) * v[3], hi, lo);
    *(uint32_t*)dst = hi;
#if GR_SP
    *(uint32_t*)(dst + 32) = lo;
#endif
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
    st4f((bf16_t*)p.C + orow * p.ldc, n, v);
  }
}

// ---- timing hook (bench.py roofline leg): HIP events on the launch stream around every GEMM launch ----
#include <vector>
static bool g_prof_on = false;
struct ProfRec { hipEvent_t a, b; double flops; int M, N, K, tag; };
static std::vector<ProfRec> g_prof;

// the same hook for launches that do not go through gr_gemm_bf16 (gemv_fused.hip): returns a record index or -1 when the hook is off
int gr_prof_begin(hipStream_t stream, int M, int N, int K, int tag) {
  if (!g_prof_on) return -1;
  ProfRec rec;
  (void)hipEventCreate(&rec.a);
  (void)hipEventCreate(&rec.b);
  rec.flops = 2.0 * M * (double)N * K;
  rec.M = M; rec.N = N; rec.K = K; rec.tag = tag;
  (void)hipEventRecord(rec.a, stream);
  g_prof.push_back(rec);
  return (int)g_prof.size() - 1;
}
void gr_prof_end(hipStream_t stream, int idx) {
  if (idx >= 0 && idx < (int)g_prof.size()) (void)hipEventRecord(g_prof[idx].b, stream);
}

extern "C" int gr_abi_version(void) { return GROMA_HIP_ABI_VERSION; }
#if GR_SP
extern "C" int gr_operand_type(void) { return GR_OPERAND_SPLIT; }
#elif defined(GR_F16)
extern "C" int gr_operand_type(void) { return GR_OPERAND_F16; }
#else
extern "C" int gr_operand_type(void) { return GR_OPERAND_BF16; }
#endif
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

// per-launch variant of gr_prof_read: fills mnk[4*i..] = (M, N, K, tag) and ms[i]; drains the records
extern "C" int gr_prof_read_launches(long cap, int* mnk, float* ms, long* n_out) {
  long n = 0;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return GR_EINVAL;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return GR_EINVAL;
    if (n < cap && mnk && ms) {
      mnk[4 * n] = r.M; mnk[4 * n + 1] = r.N; mnk[4 * n + 2] = r.K; mnk[4 * n + 3] = r.tag;
      ms[n] = t;
    }
    ++n;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  if (n_out) *n_out = n;
  return GR_OK;
}

static thread_local int g_gemm_yield = 0;
extern "C" int gr_gemm_yield(int on) {
  g_gemm_yield = on != 0;
  return GR_OK;
}

extern "C" int gr_gemm_bf16(const gr_gemm_desc* d, hipStream_t stream) {
  if (!d || (!d->A && !d->a_parts) || !d->W || (!d->C && d->tile != 2)) return GR_EINVAL;
  if (d->a_parts && ((d->tile != 1 && d->tile != 2) || d->a_nsplit < 1 || d->a_hd < 8 || d->a_hd % 8 != 0 || d->K % d->a_hd != 0))
    return GR_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return GR_EINVAL;
  if (d->K % BK != 0 || d->N % 4 != 0) return GR_EINVAL;
  if (d->fp8 && (d->K % 128 != 0 || d->conv_C % 128 != 0 || !d->w_scale || d->tile == 1 || d->tile == 2)) return GR_EINVAL;  // (tile 3: gemm_skinny_fp8.hip)
  if (d->conv_C > 0 && (d->conv_C % BK != 0 || d->K % (9 * d->conv_C) != 0)) return GR_EINVAL;
  const int splits = d->splits > 1 ? d->splits : 1;
  if (splits > 1 && !d->ws) return GR_EINVAL;
  if (d->act == 3 && (d->resid || d->scale || d->out_f32)) return GR_EINVAL;
  if (d->act < 0 || d->act > 3) return GR_EINVAL;
  GemmArgs p;
  p.A = (const bf16_t*)d->A;
  p.W = (const bf16_t*)d->W;
  p.C = d->C;
  p.bias = d->bias;
  p.scale = d->scale;
  p.resid = d->resid;
  p.a_scale = d->fp8 ? d->a_scale : nullptr;
  p.w_scale = d->fp8 ? d->w_scale : nullptr;
  p.ws = d->ws;
  p.a_parts = d->a_parts; p.a_nsplit = d->a_nsplit; p.a_hd = d->a_hd;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
  p.act = d->act; p.out_f32 = d->out_f32; p.splits = splits;
  p.conv_H = d->conv_H; p.conv_W = d->conv_W; p.conv_C = d->conv_C;
  p.conv_seg_stride = d->conv_seg_stride;
#if GR_SP
  // Reference-precision build: every 16-bit operand is a (hi, lo) pair in 32-element blocks (gr_common.h), so a physical row holds
  // 2K elements and one 64-wide physical K-tile = 32 logical k-values.  The descriptor stays LOGICAL; the kernels see physical
  // extents.  The streaming decode kernels and the e4m3 path do not exist in this build.
  if (d->fp8 || d->tile == 1 || d->tile == 2 || d->a_parts) return GR_EINVAL;
  if (d->lda % 32 != 0 || d->ldw % 32 != 0 || (d->conv_C > 0 && d->conv_seg_stride % 32 != 0)) return GR_EINVAL;
  if (!d->out_f32 && (d->ldc % 32 != 0 || (d->act == 3 ? d->N / 2 : d->N) % 32 != 0)) return GR_EINVAL;  // rows = whole hi / lo block pairs
  p.K *= 2; p.lda *= 2; p.ldw *= 2; p.conv_C *= 2; p.conv_seg_stride *= 2;
  if (!d->out_f32) p.ldc *= 2;
#endif
  p.resid_mod = d->resid_mod;
  p.c_group = d->c_group; p.c_group_stride = d->c_group_stride; p.c_row_off = d->c_row_off;
  // ---- kernel choice: estimated time = rounds x per-slot tile time (slots: 512 for 128^2 at 2 blocks/CU, 256 for
  // 256^2 at 1 block/CU; per-CU throughput ratio measured on MI355X) ----
  bool use256 = false;  // the ping-pong kernel (gemm_bf16_256.hip), with row tiles of pp_rows
  int pp_rows = 256;
  const bool gemv = d->tile == 1 || d->tile == 2;  // decode-step shape: weights streamed once, no MFMA (gemv_bf16.hip)
  const bool partials_only = d->tile == 2;         // the caller's next kernel consumes ws[splits, M, N] (decode.hip)
  if (gemv && (p.M > 8 || p.conv_C > 0 || splits != gr_cdiv(p.K, 512) || !p.ws)) return GR_EINVAL;
  const bool skinny = d->tile == 3;                // decode steps of 9..64 rows: the weight stream on the matrix unit (gemm_skinny.hip)
  if (skinny && splits > 1) return GR_EINVAL;
  if (d->tile != 0 && d->tile != 1 && d->tile != 2 && d->tile != 3 && d->tile != 128 && d->tile != 256 && d->tile != GR_TILE_PP192) return GR_EINVAL;
  if (d->tile == 256 || (d->fp8 && !skinny)) use256 = true;  // fp8 exists for the 256x256 kernel (and the skinny stream) only
  else if (d->tile == GR_TILE_PP192) { use256 = true; pp_rows = 192; }
  else if (d->tile == 0) {
    const long t128 = (long)gr_cdiv(p.M, 128) * gr_cdiv(p.N, 128);
    const long tn256 = gr_cdiv(p.N, 256);
    const double ksteps = (double)(p.K / 64) / splits;
    // Estimated time = rounds x per-round cost, in units of one K-step of a round of 512 128^2-tiles.  Fitted on MI355X
    // (tests/gemm_microbench2.py at M = 14350 / 8148; tests/diag/gemm_tile_rows.py at M = 2328 / 582 -> profiles/r04_gemm_tile_rows.txt):
    // a round of 256 ping-pong tiles costs (ksteps x c + o) with c / o = 1.41 / 13 at 256 rows, PP192_C / PP192_O at 192 rows.  The
    // three kernels produce the same bits, so choosing by M is free of side effects.  (A 128-row x 256-column form of the ping-pong
    // kernel was built and measured too: never the best -- the 128x128 kernel wins at small M, 192 / 256 rows above -- and removed.)
    double e128 = (double)((t128 * splits + 511) / 512) * (ksteps * 1.00 + 5.0);         // 2 tiles of 128^2 per CU
    if (t128 * splits <= 256) e128 *= E128_SOLO;
    auto e_pp = [&](int rows, double c, double o) { return (double)(((long)gr_cdiv(p.M, rows) * tn256 * splits + 255) / 256) * (ksteps * c + o); };
    const double e256 = e_pp(256, G256_COST, 13.0), e192 = e_pp(192, PP192_C, PP192_O);
    double best = e128;
    if (e256 < best) { best = e256; use256 = true; pp_rows = 256; }
    if (e192 < best * PP_MARGIN) { best = e192; use256 = true; pp_rows = 192; }
  }
  p.yield = g_gemm_yield;
  p.tiles_m = gr_cdiv(p.M, use256 ? pp_rows : BM);
  p.tiles_n = gr_cdiv(p.N, use256 ? 256 : BN);
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
    rec.flops = 2.0 * p.M * (double)p.N * d->K;  // algorithmic (logical K; the split build issues 3x this in MFMA work); fp8 launches are tagged 16
    rec.M = p.M; rec.N = p.N; rec.K = d->K; rec.tag = (p.conv_C > 0 ? 1 : 0) | (splits > 1 ? 2 : 0) | (use256 ? 4 : 0) | (gemv || skinny ? 8 : 0) | (d->fp8 ? 16 : 0) | (skinny ? 64 : 0);
    (void)hipEventRecord(rec.a, stream);
  }
  if (skinny) {
    const int rc = d->fp8 ? gr_launch_gemm_skinny_fp8(p, stream) : gr_launch_gemm_skinny(p, stream);
    if (rc != GR_OK) return rc;
  } else if (gemv) {
    const int rc = gr_launch_gemv(p, stream);
    if (rc != GR_OK) return rc;
  } else if (use256) {
    const int rc = d->fp8 ? gr_launch_gemm256_fp8(p, stream) : gr_launch_gemm256(p, stream, pp_rows);
    if (rc != GR_OK) return rc;
  } else {
    hipLaunchKernelGGL(gemm_bf16_kernel, grid, dim3(NTHREADS), 65536, stream, p);
  }
  if (g_prof_on) {
    (void)hipEventRecord(rec.b, stream);
    g_prof.push_back(rec);
  }
  GR_CHECK_LAUNCH();
  if ((splits > 1 || gemv) && !partials_only) {
    const long tot = (long)p.M * (p.N >> 2);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(gr_cdiv(tot, 256)), dim3(256), 0, stream, p);
    GR_CHECK_LAUNCH();
  }
  return GR_OK;
}
