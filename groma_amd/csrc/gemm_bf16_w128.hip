// 256x256 bf16 MFMA GEMM, ONE wave per SIMD: 4 waves (2 x 2), each a 128x128 output block held in 256 accumulator
// registers (AGPRs), K advanced in steps of 32 through an LDS ring, operands staged global -> VGPR -> LDS.
//
// Why: per flop a 128x128 wave block reads a third fewer LDS bytes than the ping-pong kernel's 128x64 blocks
// (gemm_bf16_256.hip), no barrier ever gates the matrix pipe, and -- the lesson of round 1's first attempt at this
// shape, which fed the ring with LDS-DMA and lost (1.07-1.12 PF: an LDS-DMA issue costs ~60 clk of the only wave's
// issue time, MI355X_MICROARCH.md) -- the refill uses plain `global_load_dwordx4` into spare VGPRs (a 512-register wave
// has 256 architectural VGPRs beside its 256 accumulators) and `ds_write_b128` two steps later: both issue in a few
// clocks in the shadow of the MFMAs.  (The vendor library's kernel for these shapes has the same outline: 256 threads,
// 256x256x64 macro-tile, 512 registers, ~130 KB LDS -- profiles/r02_blas_yardstick.txt.)
//
// Pipeline (step = 32 k-values; G = global loads into a register set, W = ds_write of that set, R = fragment reads,
// C = the 64 MFMAs):   G(k) in step k-5,  W(k) in step k-3,  R(k) in step k-1,  C(k) in step k.
//   * a load has two whole steps (>= 2 x 1088 clk) of flight before its ds_write needs it; two register sets
//     (2 x 8 x 16 B per lane) alternate by step parity, W(k+3)[c] followed by G(k+5)[c] into the same
//     registers;
//   * one block barrier per TWO steps (see the step macro) over a ring of four 32 KB stages;
//   * the loop body is branch-free: past the end, loads re-fetch the last step and writes land in a stage nobody reads.
//   * LDS image: 64-B rows (4 chunks of 16 B), chunk position = k-chunk ^ ((-(row >> 2)) & 3): conflict-free for the
//     16-lane groups ds_read_b128 is serviced in; the global side applies the XOR to the per-lane source chunk so the
//     ds_write_b128 is lane-linear (1 KB per wave-instruction, conflict-free).
// Epilogue: the shared LDS-staged batched epilogue (gemm_common.h), four 64-row passes.
// Plain GEMM only (no implicit-conv gather, no fp8); operands must be addressable with 32-bit byte offsets.
#include "gemm_common.h"
#include <cstdlib>

#define WT 256
#define WNT 256       // threads
#define WKS 32        // K per step
#define WSTAGE 32768  // bytes per stage: A 256 x 64 B, then B 256 x 64 B
#define WB_OFF 16384
#define WRING 4

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

  const int steps_total = p.K / WKS;
  const int z = blockIdx.y;
  const int per = (steps_total + p.splits - 1) / p.splits;
  const int s_begin = z * per;
  const int ns = min(steps_total, s_begin + per) - s_begin;

  // ---- staging geometry: piece c (0..7) of a step = 16 rows x 64 B = 1 KB per wave; rows (c&3)*64 + wave*16 + (lane>>2)
  // of A (c < 4) or W (c >= 4); the lane's 16-B slot lane&3 of its row holds global k-chunk slot ^ swz(row)
  const int drow = lane >> 2, dpos = lane & 3;
  unsigned goff[8];  // per-lane byte offset from the (uniform) operand base of the step
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int r = (c & 3) * 64 + wave * 16 + drow;
    const int kc = dpos ^ ((-(r >> 2)) & 3);
    if (c < 4) {
      int m = m0 + r;
      if (m > p.M - 1) m = p.M - 1;
      goff[c] = (unsigned)((long)m * p.lda * 2 + kc * 16);
    } else {
      int n = n0 + r;
      if (n > p.N - 1) n = p.N - 1;
      goff[c] = (unsigned)((long)n * p.ldw * 2 + kc * 16);
    }
  }
  const char* gA = (const char*)p.A + (long)s_begin * (WKS * 2);
  const char* gW = (const char*)p.W + (long)s_begin * (WKS * 2);
  // The loads are inline asm so that the vmcnt waits are OURS: hipcc's own counted waits for plain loads in this loop
  // came out as vmcnt(7) (half the intended flight).  Rules that follow from it: a register set is only read behind an
  // explicit `s_waitcnt vmcnt(N)`, and the queue is drained (vmcnt(0)) before the registers are dead.
  auto gload = [&](int c, int step) -> u32x4 {
    const int sc = step < ns ? step : ns - 1;  // branch-free tail: re-fetch the last step
    const char* ptr = (c < 4 ? gA : gW) + (long)sc * (WKS * 2) + goff[c];
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
  };
  const int w_lane = wave * 1024 + lane * 16;
  auto lwrite = [&](int c, int step, u32x4 v) {
    *(u32x4*)(smem + (step & (WRING - 1)) * WSTAGE + w_lane + (c < 4 ? 0 : WB_OFF) + (c & 3) * 4096) = v;
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
  u32x4 stg[2][8];
  auto read_frag = [&](int buf, int step, int which, int idx) {  // which 0: W fragment idx, 1: A fragment idx
    const char* st = smem + (step & (WRING - 1)) * WSTAGE;
    if (which == 0) bf[buf][idx] = *(const bf16x8*)(st + b_lane + idx * 1024);
    else af[buf][idx] = *(const bf16x8*)(st + a_lane + idx * 1024);
  };

  // ---- prologue: steps 0..2 written and published, steps 3 and 4 in flight, fragments of step 0 in registers
  {
    u32x4 tmp[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { stg[0][c] = gload(c, 0); stg[1][c] = gload(c, 1); tmp[c] = gload(c, 2); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 8; ++c) { lwrite(c, 0, stg[0][c]); lwrite(c, 1, stg[1][c]); lwrite(c, 2, tmp[c]); }
  }
  // (same issue order as the steady state -- all of one step, then all of the next -- so vmcnt(15) means the same thing)
#pragma unroll
  for (int c = 0; c < 8; ++c) stg[0][c] = gload(c, 3);
#pragma unroll
  for (int c = 0; c < 8; ++c) stg[1][c] = gload(c, 4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int idx = 0; idx < 8; ++idx) { read_frag(0, 0, 0, idx); read_frag(0, 0, 1, idx); }

  // One K-step (32 k-values); CUR = parity of the step = fragment buffer it computes from = register set it writes / refills.
  // Every non-MFMA instruction sits directly behind an MFMA issue (one per 16-clk gap, pinned with sched_barriers): bunched
  // in front of a chunk's MFMAs they leave the matrix pipe idle for as long as they take to issue (first version: 1.18 PF).
  // The block barrier comes after odd steps only: W(k) is issued in step k-3 and R(k) in step k-1, so one of the two step
  // boundaries in between is always a barrier, and so is one of the (at least two) between the last read of a ring slot
  // and its next write.
#define SB __builtin_amdgcn_sched_barrier(0);
// timing ablations for tests/diag (results are wrong with any of them): -DW128_NO_R / _NO_W / _NO_G drop the loop's
// fragment reads / stage writes / global loads
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
#ifdef W128_NO_G
#define W128_G(X)
#else
#define W128_G(X) X
#endif
#define MF(C, J) acc[C][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[CUR_][J], af[CUR_][C], acc[C][J], 0, 0, 0);
#define W128_STEP(CUR, S)                                                                                    \
  {                                                                                                         \
    constexpr int CUR_ = (CUR);                                                                             \
    const int s = (S);                                                                                      \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                         \
      MF(c, 0) SB                                                                                           \
      W128_R(read_frag(1 - CUR_, s + 1, c >> 2, (c & 3) * 2);)  /* chunks 0-3: the 8 W fragments, 4-7: the 8 A */ \
      SB MF(c, 1) SB                                                                                        \
      W128_R(read_frag(1 - CUR_, s + 1, c >> 2, (c & 3) * 2 + 1);)                                          \
      SB MF(c, 2) SB                                                                                        \
      W128_G(asm volatile("s_waitcnt vmcnt(15)" ::: "memory");) /* G(s+3)[c] landed: 15 younger loads may fly */ \
      W128_W(lwrite(c, s + 3, stg[CUR_][c]);)                                                               \
      SB MF(c, 3) SB                                                                                        \
      W128_G(stg[CUR_][c] = gload(c, s + 5);)                                                               \
      SB MF(c, 4) MF(c, 5) MF(c, 6) MF(c, 7) SB                                                             \
    }                                                                                                       \
    if (CUR_ == 1) {                                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* fragment reads and stage writes of both steps done */ \
      __builtin_amdgcn_s_barrier();                                                                         \
    }                                                                                                       \
  }

  // ns is even (gr_w128_eligible), so the second step's condition is always true -- it is kept because hipcc 7.2's
  // register allocator keeps the 256 accumulators in place across the back edge with this loop form and inserts
  // ~100 v_accvgpr_mov/read/write per iteration with the unconditional / do-while forms (checked in the ISA)
  for (int s2 = 0; s2 < ns; s2 += 2) {
    W128_STEP(0, s2)
    if (s2 + 1 < ns) W128_STEP(1, s2 + 1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's re-fetches still target the staging registers

  // ---- epilogue through LDS in 4 passes of 64 rows: pass q stages m-tiles 2q, 2q+1 of every wave
  __syncthreads();
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
  }  // persistent tile loop
}

bool gr_w128_eligible(const GemmArgs& p) {
  return p.conv_C == 0 && (long)p.M * p.lda * 2 < (1L << 32) && (long)p.N * p.ldw * 2 < (1L << 32) && p.K % (2 * WKS) == 0 &&
         (p.K / (2 * WKS)) % p.splits == 0 && p.K / WKS / p.splits >= 6;  // an even number of steps per split
}

int gr_launch_gemm_w128(const GemmArgs& p, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_w128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WRING * WSTAGE);
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
  hipLaunchKernelGGL(gemm_bf16_w128_kernel, grid, dim3(WNT), WRING * WSTAGE, stream, p);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
