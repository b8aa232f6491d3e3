// 256x256 bf16 MFMA GEMM, ONE wave per SIMD: 4 waves (2 x 2), each a 128x128 output block held in 256 accumulator
// registers (AGPRs) as 4 x 4 tiles of v_mfma_f32_32x32x16_bf16; operands staged global -> VGPR -> LDS.
//
// Why: per flop a 128x128 wave block reads a third fewer LDS bytes than the ping-pong kernel's 128x64 blocks
// (gemm_bf16_256.hip) and no barrier ever gates the matrix pipe.  What the two earlier attempts at this shape taught:
//   * round 1 fed the LDS with LDS-DMA: 1.07-1.12 PF -- an LDS-DMA issue costs ~60 clk of the only wave's issue time;
//   * round 2, first version (git 5e7914d): plain `global_load_dwordx4` into spare VGPRs + `ds_write_b128`, 16x16x32
//     MFMAs, 32-deep steps: bit-identical to the ping-pong kernel, 1.15-1.23 PF.  Ablation builds: without the loop's
//     fragment reads 1.40 PF, without its global loads 1.51 PF -- a 16-clk MFMA gap hides ~one other instruction, and a
//     32-deep step fetches HALF of every 128-B line per request.
// Hence here: the 32x32x16 MFMA (32-clk gaps, half as many issues for the same flops) and 64-deep K-tiles fetched as
// whole 128-B rows.  (The vendor library's kernel for these shapes has the same outline: 256 threads, 256x256x64
// macro-tile, 512 registers, ~130 KB LDS -- profiles/r02_blas_yardstick.txt.)
//
// Pipeline.  K-tile S (64 k-values) lives in LDS stage S&1 (A 256 x 128 B, then W 256 x 128 B = 64 KB) and is computed
// in two sub-steps t = 2S, 2S+1 of 32 MFMAs each.  In sub-step t a wave
//   * computes from the fragment buffer t&1 and reads the fragments of sub-step t+1 into the other one;
//   * writes 8 of its 16 staging registers to LDS and refills each straight away:  odd t -> pieces 0-7 of K-tile
//     (t+3)/2, refilled with the same pieces of the tile after it;  even t -> pieces 8-15 of K-tile t/2+1, likewise.
//     A register therefore has 16 issue slots = two sub-steps (>= 2 x 1024 clk) of flight; `s_waitcnt vmcnt(15)`
//     before each write is exact (the loads are inline asm, so the counts are ours, not the compiler's).
//   * one block barrier per K-tile, after the even sub-step: it publishes tile S+1 (written during t = 2S-1 and 2S)
//     for the fragment reads of t = 2S+1, and it separates the last reads of a stage (during the even sub-step two
//     tiles earlier) from the first write that reuses it.
//   * the loop body is branch-free: past the end, loads re-fetch the last K-tile and writes land in a dead stage.
// LDS image: 128-B rows, 16-B chunk position = k-chunk ^ ((row >> 1) & 7): conflict-free for the 16-lane groups
// ds_read_b128 is serviced in when a fragment spans 32 consecutive rows; the global side applies the XOR to the per-lane
// source chunk so the ds_write_b128 is lane-linear (8 rows x 128 B per wave-instruction).
// Epilogue: the shared LDS-staged batched epilogue (gemm_common.h), four 64-row passes.
// Plain GEMM only (no implicit-conv gather, no fp8); operands must be addressable with 32-bit byte offsets.
#include "gemm_common.h"
#include <cstdlib>

#define WT 256
#define WNT 256        // threads
#define WKT 64         // K per tile (one 128-B LDS row)
#define WSTAGE 65536   // bytes per stage
#define WB_OFF 32768

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// timing ablations for tests/diag (results are wrong with any of them): -DW128_NO_R / _NO_W / _NO_G drop the loop's
// fragment reads / stage writes / global loads; _NO_WAIT the counted vmcnt waits; _NO_BAR the per-K-tile barrier
#ifdef W128_NO_R
#define W128_R(X)
#else
#define W128_R(X) X
#endif
#ifdef W128_NO_W
#define W128_W(X)
#else
#define W128_W(X) X
#endif
#ifdef W128_NO_WAIT
#define W128_WAIT(X)
#else
#define W128_WAIT(X) X
#endif
#ifdef W128_NO_BAR
#define W128_BAR(X)
#else
#define W128_BAR(X) X
#endif
#ifdef W128_NO_G
#define W128_G(X)
#else
#define W128_G(X) X
#endif

__global__ __launch_bounds__(WNT, 1) void gemm_bf16_w128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total_tiles = p.tiles_m * p.tiles_n;
  for (int vb = blockIdx.x; vb < total_tiles; vb += gridDim.x) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if (vb != (int)blockIdx.x) __syncthreads();  // the previous tile's epilogue has finished reading the stage buffers

  int tm, tn;
  tile_of_block(vb, total_tiles, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * WT, n0 = tn * WT;

  const int kt_total = p.K / WKT;
  const int z = blockIdx.y;
  const int per = (kt_total + p.splits - 1) / p.splits;
  const int kt_begin = z * per;
  const int nt = min(kt_total, kt_begin + per) - kt_begin;

  // ---- staging geometry: piece c (0..15) of a K-tile = 8 rows x 128 B = 1 KB per wave; rows (c&7)*32 + wave*8 + (lane>>3)
  // of A (c < 8) or W (c >= 8); the lane's 16-B slot lane&7 of its row holds global k-chunk slot ^ swz(row)
  const int drow = lane >> 3, dpos = lane & 7;
  unsigned goff[16];  // per-lane byte offset from the (uniform) operand base of the K-tile
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int r = (c & 7) * 32 + wave * 8 + drow;
    const int kc = dpos ^ ((r >> 1) & 7);
    if (c < 8) {
      int m = m0 + r;
      if (m > p.M - 1) m = p.M - 1;
      goff[c] = (unsigned)((long)m * p.lda * 2 + kc * 16);
    } else {
      int n = n0 + r;
      if (n > p.N - 1) n = p.N - 1;
      goff[c] = (unsigned)((long)n * p.ldw * 2 + kc * 16);
    }
  }
  const char* gA = (const char*)p.A + (long)kt_begin * (WKT * 2);
  const char* gW = (const char*)p.W + (long)kt_begin * (WKT * 2);
  // Inline asm so that the vmcnt waits are OURS.  Rules that follow: a staging register is only read behind an explicit
  // `s_waitcnt vmcnt(N)`, and the queue is drained (vmcnt(0)) before the registers are dead.
  auto gload = [&](int c, int tile) -> u32x4 {
    const int tc = tile < nt ? tile : nt - 1;  // branch-free tail: re-fetch the last K-tile
    const char* ptr = (c < 8 ? gA : gW) + (long)tc * (WKT * 2) + goff[c];
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
  };
  const int w_lane = wave * 1024 + lane * 16;
  auto lwrite = [&](int c, int tile, u32x4 v) {
    *(u32x4*)(smem + (tile & 1) * WSTAGE + w_lane + (c < 8 ? 0 : WB_OFF) + (c & 7) * 4096) = v;
  };

  // ---- fragment geometry (swapped operands: the W fragment is the MFMA's first operand, so a lane owns groups of 4
  // consecutive output columns of one row).  32x32x16: lane -> row lane&31, k-half lane>>5 of the 16-k slice.
  const int fr = lane & 31, hi = lane >> 5;
  const int coff = (hi ^ ((fr >> 1) & 7)) << 4;  // chunk (slice*2 + hi) ^ swz(row): the slice enters as XOR (slice << 5)
  const int a_lane = (wm * 128 + fr) * 128 + coff;
  const int b_lane = WB_OFF + (wn * 128 + fr) * 128 + coff;

  f32x16 acc[4][4];  // [mi][nj]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bf16x8 af[2][4][2], bf[2][4][2];  // [buffer][tile index][k-slice of the sub-step]
  u32x4 stg[16];
  // fragment `idx` (32 rows) of operand `which` (0: W, 1: A), k-slice kk of sub-step t, into buffer buf
  auto read_frag = [&](int buf, int t, int which, int idx, int kk) {
    const char* st = smem + ((t >> 1) & 1) * WSTAGE;
    const int x = (((t & 1) * 2 + kk) << 5);
    if (which == 0) bf[buf][idx][kk] = *(const bf16x8*)(st + (b_lane ^ x) + idx * 4096);
    else af[buf][idx][kk] = *(const bf16x8*)(st + (a_lane ^ x) + idx * 4096);
  };

  // ---- prologue: K-tile 0 and pieces 0-7 of K-tile 1 written and published; pieces 8-15 of tile 1 and 0-7 of tile 2
  // in flight (issued in the steady-state order); fragments of sub-step 0 in registers
  {
    u32x4 tmp[8];
#pragma unroll
    for (int c = 0; c < 16; ++c) stg[c] = gload(c, 0);
#pragma unroll
    for (int c = 0; c < 8; ++c) tmp[c] = gload(c, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 16; ++c) lwrite(c, 0, stg[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c) lwrite(c, 1, tmp[c]);
  }
#pragma unroll
  for (int c = 8; c < 16; ++c) stg[c] = gload(c, 1);
#pragma unroll
  for (int c = 0; c < 8; ++c) stg[c] = gload(c, 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int idx = 0; idx < 4; ++idx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { read_frag(0, 0, 0, idx, kk); read_frag(0, 0, 1, idx, kk); }

  // hipcc's waitcnt pass understands this builtin (not an asm string): with the prologue's reads retired here, the wait it
  // puts at the loop top is the one the back edge needs, not lgkmcnt(0)
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)

  // One sub-step = 32 k-values = 32 MFMAs in 8 chunks of 4; chunk c computes acc[c&3][0..3] from k-slice c>>2.  Every
  // non-MFMA instruction sits directly behind an MFMA issue (pinned with sched_barriers).  RD(r), r = 0..15: the
  // fragment reads of the next sub-step in the order it needs them (per slice: W 0-3, then A 0-3).
  // EVEN sub-step (ends with the block barrier): all its LDS traffic sits in chunks 0-5, so the lgkmcnt(0) before the
  // barrier finds nothing in flight -- with reads issued up to the last chunk that drain cost more than anything else
  // in the loop (ablation: 1.25 PF with the reads, 1.58 PF without).  Pieces 8-15: written in chunks 0-1, refilled in 2-3.
  // ODD sub-step: piece c written and refilled in chunk c.  Either way a piece has >= 14 chunks of flight and exactly 15
  // younger loads behind it when an odd sub-step writes it, 15-k when the even one writes its k-th.
#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(MI, NJ, KK) acc[MI][NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[CUR_][NJ][KK], af[CUR_][MI][KK], acc[MI][NJ], 0, 0, 0);
#define RD(R) W128_R(read_frag(1 - CUR_, t + 1, ((R) >> 2) & 1, (((R) >> 1) & 1) * 2 + ((R) & 1), (R) >> 3);)
#define VMWAIT(N) W128_G(W128_WAIT(asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");))
#define WR(PC) W128_W(lwrite(PC, wt, stg[PC]);)
#define GL(PC) W128_G(stg[PC] = gload(PC, wt + 1);)
#define CHUNK(C, F0, F1, F2, F3)                                                                             \
  MF((C) & 3, 0, (C) >> 2) SB F0 SB MF((C) & 3, 1, (C) >> 2) SB F1 SB MF((C) & 3, 2, (C) >> 2) SB F2 SB      \
  MF((C) & 3, 3, (C) >> 2) SB F3 SB
#define W128_EVEN(T)                                                                                         \
  {                                                                                                         \
    constexpr int CUR_ = 0;                                                                                 \
    const int t = (T);                                                                                      \
    const int wt = t / 2 + 1;                               /* K-tile whose pieces 8-15 are written here */   \
    CHUNK(0, VMWAIT(15) WR(8), VMWAIT(14) WR(9), VMWAIT(13) WR(10), VMWAIT(12) WR(11))                      \
    CHUNK(1, VMWAIT(11) WR(12), VMWAIT(10) WR(13), VMWAIT(9) WR(14), VMWAIT(8) WR(15))                      \
    CHUNK(2, RD(0) GL(8), RD(1) GL(9), RD(2) GL(10), RD(3) GL(11))                                          \
    CHUNK(3, RD(4) GL(12), RD(5) GL(13), RD(6) GL(14), RD(7) GL(15))                                        \
    CHUNK(4, RD(8), RD(9), RD(10), RD(11))                                                                  \
    CHUNK(5, RD(12), RD(13), RD(14), RD(15))                                                                \
    CHUNK(6, , , , )                                                                                        \
    CHUNK(7, , , , )                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of stage S and writes of S+1 are done */ \
    W128_BAR(__builtin_amdgcn_s_barrier();)                                                                 \
  }
#define W128_ODD(T)                                                                                          \
  {                                                                                                         \
    constexpr int CUR_ = 1;                                                                                 \
    const int t = (T);                                                                                      \
    const int wt = (t + 3) / 2;                             /* K-tile whose pieces 0-7 are written here */    \
    CHUNK(0, RD(0), RD(1), VMWAIT(15) WR(0), GL(0))                                                         \
    CHUNK(1, RD(2), RD(3), VMWAIT(15) WR(1), GL(1))                                                         \
    CHUNK(2, RD(4), RD(5), VMWAIT(15) WR(2), GL(2))                                                         \
    CHUNK(3, RD(6), RD(7), VMWAIT(15) WR(3), GL(3))                                                         \
    CHUNK(4, RD(8), RD(9), VMWAIT(15) WR(4), GL(4))                                                         \
    CHUNK(5, RD(10), RD(11), VMWAIT(15) WR(5), GL(5))                                                       \
    CHUNK(6, RD(12), RD(13), VMWAIT(15) WR(6), GL(6))                                                       \
    CHUNK(7, RD(14), RD(15), VMWAIT(15) WR(7), GL(7))                                                       \
  }

  for (int S = 0; S < nt; ++S) {
    W128_EVEN(2 * S)
    W128_ODD(2 * S + 1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's re-fetches still target the staging registers

  // ---- epilogue through LDS in 4 passes of 64 rows: pass q stages the 32-row tile mi = q of every wave
  __syncthreads();
  EpiCols<4> ec4;
  EpiCols<2> ec2;
  EpiCols<1> ec1;
  if (p.act == 3) ec4.load(p, n0 + (tid & 15) * 16);
  else if (!p.out_f32 && p.splits == 1) ec2.load(p, n0 + (tid & 31) * 8);
  else ec1.load(p, n0 + (tid & 63) * 4);
  // The pass loop stays rolled (one copy of the epilogue code); only the 16 accumulator -> LDS writes are written out
  // per pass, so acc[] is never indexed dynamically (that would push all 256 accumulators through scratch).
  // 32x32 accumulator layout: register 4g+e of lane (fr, hi) = row fr, column g*8 + hi*4 + e of the tile.
#define W128_STAGE(Q)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
      f32x4 v = {acc[Q][j][4 * g], acc[Q][j][4 * g + 1], acc[Q][j][4 * g + 2], acc[Q][j][4 * g + 3]};      \
      stage_write4<WT>(buf, wm * 32 + fr, wn * 32 + j * 8 + g * 2 + hi, v);                                 \
    }
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
  }  // persistent tile loop
}

bool gr_w128_eligible(const GemmArgs& p) {
  return p.conv_C == 0 && (long)p.M * p.lda * 2 < (1L << 32) && (long)p.N * p.ldw * 2 < (1L << 32) && p.K % WKT == 0 &&
         (p.K / WKT) % p.splits == 0 && p.K / WKT / p.splits >= 3;
}

int gr_launch_gemm_w128(const GemmArgs& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_w128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WSTAGE);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GR_EINVAL;
    n_cu = prop.multiProcessorCount > 8 ? (prop.multiProcessorCount & ~7) : 8;
  }
  static const bool persist_off = getenv("GROMA_G256_NO_PERSIST") != nullptr;
  const int tiles = p.tiles_m * p.tiles_n;
  dim3 grid(persist_off ? tiles : (tiles < n_cu ? tiles : n_cu), p.splits);
  hipLaunchKernelGGL(gemm_bf16_w128_kernel, grid, dim3(WNT), 2 * WSTAGE, stream, p);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
