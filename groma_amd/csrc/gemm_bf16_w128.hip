// 256x256 bf16 MFMA GEMM, ONE wave per SIMD: 4 waves (2 x 2), each a 128x128 output block held in 256 accumulator
// registers (AGPRs), K advanced in steps of 32 through a 4-deep LDS ring.
//
// Why a third kernel: the ping-pong kernel (gemm_bf16_256.hip) time-shares every SIMD between two waves that may not
// issue MFMAs at the same time; each hand-off costs a block-wide barrier round trip (~90 clk) on the matrix pipe's
// critical path, and its load segments are as long as its MFMA segments (tests/diag/gemm_clk.py), so it sits at
// ~2400 clk per 64-deep K-tile against the 2048-clk MFMA floor.  Here a SIMD belongs to one wave for the whole tile:
//   * the wave issues its 64 MFMAs of a K-step back to back and slips the NEXT step's 16 ds_read_b128 and its share
//     (8 x 1 KB) of the LDS-DMA into the issue slots between them -- barriers no longer gate the matrix pipe, they only
//     publish landed data (one per K-step, usually already satisfied);
//   * 128x128 per wave halves the LDS bytes read per MFMA (256 B instead of 384 B);
//   * 4 stages x (A 256x32 + B 256x32) bf16 = 128 KB.  Step s computes from fragment registers that were read during
//     step s-1; during step s the wave reads the fragments of step s+1 (stage (s+1)&3, published by the barrier that
//     ended step s-1) and refills stage s&3 -- whose last reads were drained before that same barrier -- with step
//     s+4.  A DMA piece therefore has two full steps (>= 2048 clk) of flight before `s_waitcnt vmcnt(16)` needs it.
//   * LDS image: 64-B rows (4 chunks of 16 B).  Chunk position = k-chunk ^ ((-(row >> 2)) & 3): conflict-free for the
//     16-lane groups ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table); applied on the DMA's per-lane source
//     address (the LDS side of global_load_lds is lane-linear), mirrored on the fragment reads.
// Epilogue: the shared LDS-staged batched epilogue (gemm_common.h), four 64-row passes.
// Plain GEMM only (no implicit-conv gather, no fp8): gr_gemm_bf16 routes those to the ping-pong kernel.
//
// MEASURED (MI355X, tests/diag/w128_bench.py): bit-identical results, but 1.07-1.12 PF against the ping-pong kernel's
// 1.24-1.43 PF on the LLaMA shapes.  The ISA is exactly the intended interleave (2 ds_read + 1 LDS-DMA + 8 MFMA per
// chunk, accumulators in AGPRs, no scratch) -- what it shows is that with ONE wave on a SIMD every non-MFMA issue slot
// (an LDS-DMA issue costs ~60 clk, MI355X_MICROARCH.md) comes straight out of the matrix pipe's time, whereas the
// ping-pong partner hides exactly those.  Kept as a selectable variant (gr_gemm_desc.tile = 257 / GROMA_W128=1), not
// used by default.
#include "gemm_common.h"

#define WT 256
#define WNT 256          // threads
#define WKS 32           // K per step
#define WSTAGE 32768     // bytes per stage: A 256 x 64 B, then B 256 x 64 B
#define WB_OFF 16384

__global__ __launch_bounds__(WNT, 1) void gemm_bf16_w128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * WT, n0 = tn * WT;

  const int steps_total = p.K / WKS;
  const int z = blockIdx.y;
  const int per = (steps_total + p.splits - 1) / p.splits;
  const int s_begin = z * per;
  const int ns = min(steps_total, s_begin + per) - s_begin;

  // ---- DMA geometry: piece c (0..7) of a step = 16 rows x 64 B = 1 KB per wave; rows (c&3)*64 + wave*16 + (lane>>2)
  // of A (c < 4) or B (c >= 4); lane's chunk position lane&3 holds global k-chunk pos ^ swz(row)
  const int drow = lane >> 2, dpos = lane & 3;
  const char* src[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int r = (c & 3) * 64 + wave * 16 + drow;
    const int kc = dpos ^ ((-(r >> 2)) & 3);
    if (c < 4) {
      int m = m0 + r;
      if (m > p.M - 1) m = p.M - 1;
      src[c] = (const char*)p.A + ((long)m * p.lda + (long)s_begin * WKS) * 2 + kc * 16;
    } else {
      int n = n0 + r;
      if (n > p.N - 1) n = p.N - 1;
      src[c] = (const char*)p.W + ((long)n * p.ldw + (long)s_begin * WKS) * 2 + kc * 16;
    }
  }
  auto piece = [&](int c, int step) {  // wave-uniform destination, lane-linear 1 KB
    char* dst = smem + (step & 3) * WSTAGE + (c < 4 ? 0 : WB_OFF) + ((c & 3) * 4 + wave) * 1024;
    glds16(src[c] + (long)step * (WKS * 2), dst);
  };

  // ---- fragment geometry (swapped operands: W fragment first, so a lane owns 4 consecutive output columns)
  const int fr = lane & 15, fg = lane >> 4;
  const int coff = (fg ^ ((-(fr >> 2)) & 3)) << 4;
  const int a_lane = (wm * 128 + fr) * 64 + coff;
  const int b_lane = WB_OFF + (wn * 128 + fr) * 64 + coff;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 af[2][8], bf[2][8];
  auto read_frag = [&](int buf, int step, int which, int idx) {  // which 0: B fragment idx, 1: A fragment idx
    const char* st = smem + (step & 3) * WSTAGE;
    if (which == 0) bf[buf][idx] = *(const bf16x8*)(st + b_lane + idx * 1024);
    else af[buf][idx] = *(const bf16x8*)(st + a_lane + idx * 1024);
  };

  // ---- prologue: steps 0..3 in flight; steps 0 and 1 landed and published; fragments of step 0 in registers
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < ns)
#pragma unroll
      for (int c = 0; c < 8; ++c) piece(c, s);
  if (ns >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int idx = 0; idx < 8; ++idx) { read_frag(0, 0, 0, idx); read_frag(0, 0, 1, idx); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // one K-step; CUR = fragment buffer of this step (compile-time so the register arrays stay in registers)
#define W128_STEP(CUR, S)                                                                                    \
  {                                                                                                         \
    const int s = (S);                                                                                      \
    const bool more_frag = s + 1 < ns, more_dma = s + 4 < ns;                                               \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                         \
      if (more_frag) {                                                                                      \
        read_frag(1 - (CUR), s + 1, c >> 2, (c & 3) * 2);     /* chunks 0-3: the 8 B fragments, 4-7: the 8 A */ \
        read_frag(1 - (CUR), s + 1, c >> 2, (c & 3) * 2 + 1);                                               \
      }                                                                                                     \
      if (more_dma) piece(c, s + 4);                                                                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                         \
        acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[CUR][j], af[CUR][c], acc[c][j], 0, 0, 0);    \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    if (s + 4 < ns) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  /* step s+2 landed (s+3, s+4 may fly) */ \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          /* this step's fragment reads are drained */ \
    __builtin_amdgcn_s_barrier();                                                                           \
  }

  for (int s2 = 0; s2 < ns; s2 += 2) {
    W128_STEP(0, s2)
    if (s2 + 1 < ns) W128_STEP(1, s2 + 1)
  }

  // ---- epilogue through LDS in 4 passes of 64 rows: pass q stages m-tiles 2q, 2q+1 of every wave
  EpiCols<4> ec4;
  EpiCols<2> ec2;
  EpiCols<1> ec1;
  if (p.act == 3) ec4.load(p, n0 + (tid & 15) * 16);
  else if (!p.out_f32 && p.splits == 1) ec2.load(p, n0 + (tid & 31) * 8);
  else ec1.load(p, n0 + (tid & 63) * 4);
  // The pass loop stays rolled (one copy of the epilogue code); only the 16 accumulator -> LDS writes are written out
  // per pass, so acc[] is never indexed dynamically (that would push all 256 accumulators through scratch).
#define W128_STAGE(Q)                                                                                  \
  _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                         \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                       \
      stage_write4<WT>(buf, wm * 32 + e * 16 + fr, wn * 32 + j * 4 + fg, acc[2 * (Q) + e][j]);
#pragma nounroll
  for (int q = 0; q < 4; ++q) {
    char* buf = smem + (q & 1) * 65536;
    if (q == 0) { W128_STAGE(0) }
    else if (q == 1) { W128_STAGE(1) }
    else if (q == 2) { W128_STAGE(2) }
    else { W128_STAGE(3) }
    __syncthreads();
    epi_dispatch<WT, WNT, 64, false>(p, buf, tid, n0, z, ec4, ec2, ec1, [](int sr) { return sr; },
                                     [&](int sr) { return m0 + (sr >> 5) * 128 + q * 32 + (sr & 31); });
  }
}

int gr_launch_gemm_w128(const GemmArgs& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_w128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WSTAGE);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, p.splits);
  hipLaunchKernelGGL(gemm_bf16_w128_kernel, grid, dim3(WNT), 4 * WSTAGE, stream, p);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
