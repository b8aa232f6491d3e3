// 256x256x64 bf16 MFMA GEMM for gfx950 -- "ping-pong" schedule.
//
// Why a second kernel: at 128x128 a CU at full MFMA rate pulls 64 B/clk of operands through L2 (2/128 bytes per
// flop), more than the ~34 TB/s the eight L2s deliver, so that structure tops out near 0.9 PF
// (cdna_hip_programming.md §5 ladder).  A 256x256 tile halves the operand traffic per flop.
//
// Structure (8 waves = 2(M) x 4(N), each wave a 128x64 output block = 8x4 tiles of v_mfma_f32_16x16x32_bf16):
//  * the two wave groups (wm = 0 / 1) that share each SIMD run one barrier interval apart: while one group issues
//    the 32 MFMAs of a phase, the other does its LDS fragment reads and global->LDS DMA issue, then they swap.  The
//    partner's MFMAs hide every non-MFMA issue slot (an LDS-DMA issue alone costs ~60 clk) -- the one-wave-per-SIMD
//    (a one-wave-per-SIMD variant measured what those cost when nothing hides them: DESIGN.md 3a);
//  * two phases per 64-deep K-tile:  X: A0 x (B0,B1) -> acc[0..3][*]   Y: A1 x (B0,B1) -> acc[4..7][*].  The B
//    fragments stay in registers across both; A0's registers are reused for A1.  (Round 1 started with four 16-MFMA
//    quadrant phases; each phase boundary costs a barrier round trip on the matrix pipe, halving them: +3.3 % e2e.)
//  * LDS = 2 stages x (A 256x64 + B 256x64) bf16 = 128 KB.  Each stage is four 16 KB "half-tiles"
//    {A0, B0, B1, A1} (the rows every wave needs for that fragment), refilled as P = {A0,B0,B1} -- issued in Y(t) for
//    tile t+2, after X(t)'s reads were drained before the barrier that ended X's load segment -- and Q = {A1}, issued in
//    X(t) for tile t+1.  Every piece has one full K-tile of flight.  Counted waits: vmcnt(6) in X (Q(t) landed, P(t+1)
//    may fly), vmcnt(2) in Y (P(t+1) landed, Q(t+1) may fly); the barrier that ends the segment publishes them.  The
//    DMA queue is never drained in steady state.  ds_reads are issued FIRST in a segment so they complete in the shadow
//    of the DMA wait/issue.
//  * LDS image: 128-B rows, 16-B chunk position = k-chunk ^ (row & 7): written lane-linearly by the DMA with the
//    XOR applied to the per-lane SOURCE address, mirrored on the ds_read_b128 side (conflict-free).
#include <type_traits>

#include "gemm_common.h"

#define T256 256
#define NT 512
#define STAGE_BYTES 65536
#define B_OFF 32768

enum { HT_A0 = 0, HT_B0 = 1, HT_B1 = 2, HT_A1 = 3 };

// The same source builds the OCP-fp8 (e4m3) variant (gemm_fp8_256.hip defines G256_FP8=1): operands are 1-byte
// elements, a 128-B LDS row then holds 128 k-values, and every 16-B fragment chunk feeds TWO
// v_mfma_f32_16x16x32_fp8_fp8 (low / high 8 bytes; both operands use the same k-order so the product is unchanged):
// twice the MFMA work per byte staged.  Per-row / per-column dequantisation scales are applied in the epilogue.
#ifndef G256_FP8
#define G256_FP8 0
#endif
#ifndef G256_FP8_PERSIST
#define G256_FP8_PERSIST 0
#endif
#if G256_FP8
#define G256_KERNEL gemm_fp8_256_kernel
#define G256_LAUNCH gr_launch_gemm256_fp8
#define ESZ 1
// One v_mfma_scale_f32_16x16x128_f8f6f4 per (m-tile, n-tile) and K-tile: the lane's two 16-B chunks of the 128-B row (chunk fg
// and chunk fg + 4) are the 32 k-values the instruction wants from it, A and B alike, so the fragments are used as they are
// read.  Both formats e4m3 (cbsz = blgp = 0), both block scales the e8m0 code 127 = 2^0: a plain e4m3 x e4m3 -> f32 product
// at the MX rate (8 passes for 128 k-values).  The unscaled v_mfma_f32_16x16x32_fp8_fp8 this replaces issues at the bf16
// rate -- 4 of them, 64 clk, for the same 128 k-values (MI355X_MICROARCH.md: "non-scaled fp8 = bf16 rate; MX-scaled K = 128 is
// the only path to the low-precision peak").
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef union { struct { bf16x8 c0, c1; } h; i32x8 v; } frag_mx;
__device__ __forceinline__ f32x4 mfma_mx_e4m3(bf16x8 a0, bf16x8 a1, bf16x8 b0, bf16x8 b1, f32x4 c) {
  frag_mx ua, ub;
  ua.h.c0 = a0; ua.h.c1 = a1;
  ub.h.c0 = b0; ub.h.c1 = b1;
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ua.v, ub.v, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
#else
#define G256_KERNEL gemm_bf16_256_kernel
#define G256_LAUNCH gr_launch_gemm256
#define ESZ 2
#define MFMA16(a, b, c) GR_MFMA_16x16x32(a, b, c)
#endif
#ifdef G256_ABL_NOMFMA  // timing ablation: the load segments alone (one MFMA per 16 kept so the fragments stay live)
__device__ __forceinline__ f32x4 abl_keep(bf16x8 b, bf16x8 a, f32x4 c) {
  asm volatile("" ::"v"(a), "v"(b));  // the fragments stay live (their LDS reads are not dead code)
  return c;
}
#define ABL_MFMA(b, a, c) ((i == 0 && j == 0) ? MFMA16(b, a, c) : abl_keep(b, a, c))
#else
#define ABL_MFMA(b, a, c) MFMA16(b, a, c)
#endif
#define KT (128 / ESZ)  // K elements per K-tile (one 128-B LDS row)
// (v_mfma_f32_32x32x16_bf16 was tried in place of 16x16x32 -- 1.22 vs 1.42 PF at 8192^3 -- and removed.)
#define LDS_SWZ(row) ((row) & 7)

#if defined(G256_CLK) && !G256_FP8  // diagnostic build only (tests/diag/build_clk.py): shader/wall clocks of block 0 at start, loop end, exit
__device__ unsigned long long g256_clk[40];
extern "C" int gr_diag_clk(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g256_clk), sizeof(g256_clk));
}
#define CLK_MARK(i) if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { g256_clk[i] = clock64(); g256_clk[12 + i] = wall_clock64(); }
// per-phase marks of K-tiles 8 and 9 (wave 0 of block 0): before / after the 16-MFMA segment of each phase
#define PH_DECL unsigned long long phm[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(i) if ((t == 8 || t == 9) && blockIdx.x == 0) phm[(t - 8) * 8 + (i)] = __builtin_readcyclecounter();
#define PH_FLUSH if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { for (int i = 0; i < 16; ++i) g256_clk[24 + i] = phm[i]; }
#else
#define CLK_MARK(i)
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH
#endif

// One tile's epilogue for ONE output mode (chosen once per tile, outside the passes): only that mode's per-thread column
// constants are live while the accumulators still occupy half the register file.  With the mode dispatch INSIDE each pass all
// three EpiCols sets stayed live across the passes and the allocator spilled address registers; every reload is a
// scratch_load, i.e. a VMEM op the compiler must wait for with `s_waitcnt vmcnt(0)` -- which, vmcnt being one in-order
// counter, also waited for the previous row's global store to COMPLETE (~850 clk each): the stores of a pass were serialised
// (measured 3000-4800 clk per 64-row pass for 4 store instructions per wave).  q and FULL are compile-time, so accumulator
// indices are constants and an interior tile's pass is a guard-free straight-line block.
enum { EM_SPLITK = 0, EM_SWIGLU = 1, EM_BF16 = 2, EM_BF16_RESID = 3, EM_F32 = 4, EM_F32_RESID = 5 };
template <int MODE>
struct EpiModeCols {
  static constexpr int W4 = MODE == EM_SWIGLU ? 4 : (MODE == EM_BF16 || MODE == EM_BF16_RESID) ? 2 : 1;
};
template <int MODE, bool FULL, int MT>
__device__ __forceinline__ void epi_tile256(const GemmArgs& p, char* smem, f32x4 (&acc)[MT][4], int tid, int wm, int wn, int fr,
                                            int fg, int m0, int n0, int z) {
  constexpr int W4 = EpiModeCols<MODE>::W4;
  constexpr bool DEQ = G256_FP8 != 0;
  constexpr int TPR = T256 / (4 * W4);        // threads per staged row
  constexpr int NR = 64 * TPR / NT;           // staged rows per thread and pass
  EpiCols<W4> ec;
  if constexpr (MODE != EM_SPLITK) ec.load(p, n0 + (tid % TPR) * 4 * W4);
  auto pass = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    char* buf = smem + (q & 1) * STAGE_BYTES;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) stage_write4<T256>(buf, wm * 32 + e * 16 + fr, wn * 16 + j * 4 + fg, acc[2 * q + e][j]);
    __syncthreads();
    CLK_MARK(3 + 2 * q)
    // staged row sr of this pass -> tile row (sr>>5)*H + q*32 + (sr&31), H = 16*MT rows per wave-row half of the tile
    auto rm = [&](int it) { return it * (NT / TPR) + tid / TPR; };
    auto mm = [&](int it) { const int sr = it * (NT / TPR) + tid / TPR; return m0 + (sr >> 5) * (16 * MT) + q * 32 + (sr & 31); };
    const int c4 = (tid % TPR) * W4;
    if constexpr (MODE == EM_SPLITK) epi_rows_splitk<T256, NR, FULL>(p, buf, c4, n0, z, rm, mm);
    else if constexpr (MODE == EM_SWIGLU) epi_rows_swiglu<T256, NR, DEQ, FULL>(p, buf, c4, n0, ec, rm, mm);
    else epi_rows<T256, W4, NR, (MODE == EM_F32 || MODE == EM_F32_RESID), (MODE == EM_BF16_RESID || MODE == EM_F32_RESID), DEQ, FULL>(
        p, buf, c4, n0, ec, rm, mm);
    CLK_MARK(4 + 2 * q)
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  if constexpr (MT >= 6) pass(std::integral_constant<int, 2>{});  // (MT = 6: three passes)
  if constexpr (MT >= 8) pass(std::integral_constant<int, 3>{});
}

// MT = 16-row m-tiles per wave: the tile is (32*MT) x 256 -- 256 rows (MT 8: the kernel of rounds 1-3) or 192 (MT 6).
// Round 4: at 4 images per call (M = 2328) the N = 4096 GEMMs are 10 x 16 = 160 tiles of 256 rows on 256 CUs; 13 x 16 = 208 tiles
// of 192 rows do 0.75 of the work each in the same single round (measured: o-proj 89.6 -> 78.6 us, down-proj 196.4 -> 181.7:
// the K loop itself is load-bound and barely shortens, the fp32-residual epilogue does; a 128-row form, MT 4, was measured too
// and lost to the 128x128 kernel at every shape it was meant for).  Everything scales with MT through the wave-row height H = 16*MT: each wave
// still owns MT x 4 MFMA tiles in two phases of MT/2 m-tiles, each A half-tile is still TWO LDS-DMA instructions per wave -- with
// 8*MT of the 64 lanes active, MT rows of 8 chunks -- so the counted vmcnt waits are unchanged; the epilogue runs MT/2 passes.
// An output element's K order is identical in all three (and in the 128x128 kernel): results are bitwise equal whatever tile
// the launcher picks, so the choice may depend on M without touching batch independence.
template <int MT>
__global__ __launch_bounds__(NT) void G256_KERNEL(GemmArgs p) {
  constexpr int H = 16 * MT;    // rows of the tile owned by one wave row (wm)
  constexpr int BM = 2 * H;     // tile rows
  constexpr int MTX = MT / 2;   // m-tiles per phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CLK_MARK(0)

  // Persistent tile loop: the grid is min(tiles, CUs) blocks and block b runs virtual blocks b, b + grid, b + 2*grid ...
  // (grid is a multiple of 8, so a virtual block keeps its XCD and the XCD-aware tile order is unchanged).  A CU goes from
  // one tile's last store to the next tile's first DMA without a workgroup retire / dispatch in between: measured
  // 1093 -> 1079 us on the LLaMA gate-up shape, 610 -> 604 us on QKV (tests/diag/persist_ab.py), neutral on the ViT shapes.
  const int total_tiles = p.tiles_m * p.tiles_n;
#if G256_FP8 && !G256_FP8_PERSIST  // e4m3 build: one tile per block (the tile loop measured neutral in round 3; -DG256_FP8_PERSIST=1 re-measures it)
  {
  const int vb = blockIdx.x;
#else
  for (int vb = blockIdx.x; vb < total_tiles; vb += gridDim.x) {
#endif
  // everything derived from the thread id is RE-derived per tile (opaque copy): hoisted out of the tile loop those values
  // would have to survive the register-hungry epilogue, and the kernel already sits at 246 of 256 VGPRs
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  if (vb != (int)blockIdx.x) __syncthreads();  // the previous tile's epilogue has finished reading the stage buffers
  int tm, tn;
  tile_of_block(vb, total_tiles, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * T256;

  const int ksteps_total = p.K / KT;
  const int z = blockIdx.y;
  const int ks_per = (ksteps_total + p.splits - 1) / p.splits;
  const int ks_begin = z * ks_per;
  const int nt = min(ksteps_total, ks_begin + ks_per) - ks_begin;

  // ---- staging geometry.  A half-tile = 2 x (H/2) rows x 8 chunks, B half-tile = 128 rows x 8 chunks; instruction i = 0, 1 of a
  // half-tile covers, per wave, MT rows of A (lanes with lrow < MT: all 64 at MT = 8) / 8 rows of B.
  // physical row inside the tile:  A0: i*H + wave*MT + lrow          A1: + H/2
  //                                B0: (hrow>>5)*64 + (hrow&31)       B1: +32     (hrow = i*64 + wave*8 + lrow)
  const int lrow = lane >> 3, lpos = lane & 7;
  const bool a_lane_on = lrow < MT;  // this lane takes part in the A instructions (rows beyond the half-tile belong to nobody)
  const char* src[4][2];  // [half-tile][i] : per-lane global source (row base + swizzled 16-B chunk), k-offset added later
  int ldsoff[4][2];         // wave-uniform LDS byte offset inside a stage
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = i * H + wave * MT;                             // A0 row of lane 0
    const int rb = (i * 2 + (wave >> 2)) * 64 + (wave & 3) * 8;   // B0 row of lane 0
    const int rows[4] = {ra, rb, rb + 32, ra + H / 2};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = rows[h] + lrow;
      const int kc = lpos ^ LDS_SWZ(r);
      const bool isA = (h == HT_A0 || h == HT_A1);
      if (isA) {
        int m = m0 + r;
        if (m > p.M - 1) m = p.M - 1;
        src[h][i] = (const char*)p.A + a_row_base(p, m) * ESZ + kc * 16;
      } else {
        int n = n0 + r;
        if (n > p.N - 1) n = p.N - 1;
        src[h][i] = (const char*)p.W + (long)n * p.ldw * ESZ + kc * 16;
      }
      ldsoff[h][i] = (isA ? 0 : B_OFF) + rows[h] * 128;
    }
  }

  // A-operand k offsets of K-tiles t+1 and t+2 are carried incrementally (the conv gather's (segment, tap, channel)
  // decomposition needs integer divisions otherwise -- too slow for the 16-MFMA shadow of a phase)
  long aoff1 = a_k_off(p, ks_begin + 1, KT), aoff2 = a_k_off(p, ks_begin + 2, KT);
  int cv_c = 0, cv_kx = 0, cv_ky = 0;
  long cv_base = 0;  // state of tile t+2
  if (p.conv_C > 0) {
    const long k0 = (long)(ks_begin + 2) * KT;
    const int tapc = (int)(k0 / p.conv_C);
    cv_c = (int)(k0 - (long)tapc * p.conv_C);
    const int seg = tapc / 9, tap = tapc - seg * 9;
    cv_ky = tap / 3;
    cv_kx = tap - cv_ky * 3;
    cv_base = (long)seg * p.conv_seg_stride;
  }
  auto advance = [&](auto mode_c) {  // (aoff1, aoff2) <- (aoff2, offset of the following K-tile); mode 0: test conv_C, 1: plain rows, 2: conv gather
    constexpr int AMODE = decltype(mode_c)::value;
    aoff1 = aoff2;
    if (AMODE == 2 || (AMODE == 0 && p.conv_C > 0)) {
      cv_c += KT;
      if (cv_c == p.conv_C) {
        cv_c = 0;
        if (++cv_kx == 3) {
          cv_kx = 0;
          if (++cv_ky == 3) { cv_ky = 0; cv_base += p.conv_seg_stride; }
        }
      }
      aoff2 = cv_base + (long)(cv_ky * (p.conv_W + 2) + cv_kx) * p.conv_C + cv_c;
    } else {
      aoff2 += KT;
    }
  };

  auto issue_at = [&](int h, int tile, long koff, bool in_range = false) {  // one half-tile of K-tile `tile` into stage tile&1
    if (!in_range && tile >= nt) return;  // (in_range: the caller knows tile < nt -- the steady-state K loop, no branch)
#ifdef G256_ABL_NODMA  // timing ablation (tests/diag): only the prologue's two K-tiles are ever staged; results are garbage
    if (tile >= 2) return;
#endif
    char* st = smem + (tile & 1) * STAGE_BYTES;
    if (MT == 8 || (h != HT_A0 && h != HT_A1) || a_lane_on) {  // (A instructions of the 192- / 128-row tiles: 8*MT lanes active)
      glds16(src[h][0] + koff * ESZ, st + ldsoff[h][0]);
      glds16(src[h][1] + koff * ESZ, st + ldsoff[h][1]);
    }
  };
  auto issue = [&](int h, int tile) {  // prologue form (offset from the tile index)
    const bool isA = (h == HT_A0 || h == HT_A1);
    issue_at(h, tile, isA ? a_k_off(p, ks_begin + tile, KT) : (long)(ks_begin + tile) * KT);
  };

  f32x4 acc[MT][4];  // [mi][ni]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int off0 = (fg ^ (fr & 7)) << 4;  // chunk fg of a row with (row&7) == (fr&7); the kk=1 chunk is off0 ^ 64
  const int a_lane = (wm * H + fr) * 128;
  const int b_lane = B_OFF + (wn * 64 + fr) * 128;

  // ---- prologue: P(0) = {A0,B0,B1}(0), Q(0) = {A1}(0), P(1) ; P(0) must have landed before the first reads
  issue(HT_A0, 0); issue(HT_B0, 0); issue(HT_B1, 0); issue(HT_A1, 0);
  issue(HT_A0, 1); issue(HT_B0, 1); issue(HT_B1, 1);
  if (nt >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave group by one interval

#define WAIT_VM(N)                                                      \
  do {                                                                  \
    if (steady) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               \
  } while (0)
  bf16x8 af[MTX][2];

  PH_DECL
  // Two 32-MFMA phases per K-tile instead of four 16-MFMA ones: every phase boundary costs ~80 clk of barrier round
  // trip on top of the MFMA segment (tests/diag/gemm_clk.py), so halving the boundaries is worth ~11 % of the loop.
  //   X(t): reads P(t) = A0,B0,B1 -> af, bf ; issues Q(t+1) = A1(t+1) ; 32 MFMA  acc[0..3][*] += A0 x B
  //   Y(t): reads Q(t) = A1       -> af     ; issues P(t+2)            ; 32 MFMA  acc[4..7][*] += A1 x B
  // A region is refilled one phase after BOTH wave groups finished reading it: the reads are drained (lgkmcnt(0))
  // before the barrier that ends their load segment.  vmcnt: at X only P(t+1) (6 pieces) may stay in flight, at Y
  // only Q(t+1) (2 pieces).
  bf16x8 bfr[4][2];
#define DRAIN_READS asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
// (Variants A/B-measured on one box and removed -- DESIGN.md 3a: handing the pipe over 6 MFMAs early, not draining the
// reads, splitting the LDS-DMA issue 4 + 4 over the two segments, the original four 16-MFMA phases per K-tile.)
// The MFMAs of one phase.  Unsplit operands: both 32-deep k-steps of the K-tile, 32 MFMAs.  Split operands (GR_SPLIT, the
// reference-precision build): the K-tile's first k-step holds the hi halves of 32 logical k-values and the second their lo
// halves (gr_common.h), and the phase issues hi.hi, hi(A).lo(W) and lo(A).hi(W): 48 MFMAs on the same fragments.
#if G256_FP8
#define PHASE_MFMAS(I0)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < MTX; ++i)                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
      acc[I0 + i][j] = mfma_mx_e4m3(bfr[j][0], bfr[j][1], af[i][0], af[i][1], acc[I0 + i][j]);
#elif GR_SP
#define PHASE_MFMAS(I0)                                                                                   \
  _Pragma("unroll") for (int pp = 0; pp < 3; ++pp)                                                        \
    _Pragma("unroll") for (int i = 0; i < MTX; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        acc[I0 + i][j] = ABL_MFMA(bfr[j][pp == 1], af[i][pp == 2], acc[I0 + i][j]);
#else
#define PHASE_MFMAS(I0)                                                                                   \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                        \
    _Pragma("unroll") for (int i = 0; i < MTX; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        acc[I0 + i][j] = ABL_MFMA(bfr[j][kk], af[i][kk], acc[I0 + i][j]);
#endif
#define PHASE32(P, I0)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                      \
  __builtin_amdgcn_s_barrier();                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                      \
  PH_MARK(2 * (P))                                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                          \
  PHASE_MFMAS(I0)                                                                                         \
  __builtin_amdgcn_s_setprio(0);                                                                          \
  PH_MARK(2 * (P) + 1)                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                      \
  __builtin_amdgcn_s_barrier();                                                                           \
  __builtin_amdgcn_sched_barrier(0);
  // Round 5: the K loop is PEELED into a steady-state loop (t + 2 < nt: every wait is the counted one, every staged tile exists)
  // and a two-iteration tail, so the steady state carries no scalar branch but its back-edge.  As one loop the `steady` /
  // `tile < nt` tests compiled to ~8 s_cbranch per K-tile inside the LOAD segments (seen in the ISA: wait-count variants and the
  // guarded LDS-DMA issues as separate basic blocks) -- taken branches on the segment that has to stay shorter than the partner's
  // 512-clk MFMA segment.  Same instructions in the same order per accumulator: bitwise identical.  -DG256_PEEL=0 = the old form.
  // Measured on one box, builds alternating in one process (tests/diag/gemm_variants.py, profiles/r05_gemm_peel_ab.txt): bf16
  // build -1.8 % over the step's GEMM launches (gate-up 1021 -> 1007 us, down 532 -> 523, lm_head 1563 -> 1531, 3x3 conv at 128^2
  // 2956 -> 2803, QKV unchanged); the operand-pair build, whose MFMA segments are 48 long and already cover its load segments,
  // is 1.9 % SLOWER peeled (fc1 317 -> 325 us, fc2 280 -> 288) and keeps the single loop.
#ifndef G256_PEEL
#define G256_PEEL (GR_SP ? 0 : 1)
#endif
  auto ktile = [&](int t, auto steady_c, auto amode_c) {
    constexpr bool STEADY_C = decltype(steady_c)::value;
    const char* st = smem + (t & 1) * STAGE_BYTES;
    const bool steady = G256_PEEL ? STEADY_C : (t + 2 < nt);
    const bool rng = G256_PEEL ? STEADY_C : false;
    // ===== X
#ifdef G256_ABL_NOREAD  // timing ablation: fragments are read for K-tile 0 only
    if (t == 0)
#endif
    {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bfr[j][0] = *(const bf16x8*)(st + b_lane + j * 2048 + off0);
      bfr[j][1] = *(const bf16x8*)(st + b_lane + j * 2048 + (off0 ^ 64));
    }
#pragma unroll
    for (int i = 0; i < MTX; ++i) {
      af[i][0] = *(const bf16x8*)(st + a_lane + i * 2048 + off0);
      af[i][1] = *(const bf16x8*)(st + a_lane + i * 2048 + (off0 ^ 64));
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    WAIT_VM(6);
    issue_at(HT_A1, t + 1, aoff1, rng);
    DRAIN_READS
    PHASE32(0, 0)
    // ===== Y
#ifdef G256_ABL_NOREAD
    if (t == 0)
#endif
    {
#pragma unroll
    for (int i = 0; i < MTX; ++i) {
      af[i][0] = *(const bf16x8*)(st + a_lane + (MTX + i) * 2048 + off0);
      af[i][1] = *(const bf16x8*)(st + a_lane + (MTX + i) * 2048 + (off0 ^ 64));
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    WAIT_VM(2);
    issue_at(HT_A0, t + 2, aoff2, rng);
    issue_at(HT_B0, t + 2, (long)(ks_begin + t + 2) * KT, rng);
    issue_at(HT_B1, t + 2, (long)(ks_begin + t + 2) * KT, rng);
    advance(amode_c);
    DRAIN_READS
    PHASE32(1, MTX)
  };
  {
    int t = 0;
#if G256_PEEL
    if (p.conv_C > 0) for (; t + 2 < nt; ++t) ktile(t, std::true_type{}, std::integral_constant<int, 2>{});
    else for (; t + 2 < nt; ++t) ktile(t, std::true_type{}, std::integral_constant<int, 1>{});
#endif
    for (; t < nt; ++t) ktile(t, std::false_type{}, std::integral_constant<int, 0>{});
  }
  PH_FLUSH
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the groups (same barrier count for every wave)

  // ---- epilogue through LDS in 4 passes of 64 rows (2 m-tiles per wave), double-buffered over the two stages ----
  __syncthreads();
  CLK_MARK(1)
  {
    // tile-uniform: interior tile of a plain row-major output -> guard-free, division-free straight-line passes
    const bool full = m0 + BM <= p.M && n0 + T256 <= p.N && p.c_group == 0 && p.resid_mod == 0;
#define EPI_GO(MODE)                                                                                          \
  do {                                                                                                        \
    if (full) epi_tile256<MODE, true, MT>(p, smem, acc, tid, wm, wn, fr, fg, m0, n0, z);                      \
    else {                                                                                                    \
      epi_tile256<MODE, false, MT>(p, smem, acc, tid, wm, wn, fr, fg, m0, n0, z);                             \
      __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): see below */                                           \
    }                                                                                                         \
  } while (0)
    if (p.splits > 1) EPI_GO(EM_SPLITK);
    else if (p.act == 3) EPI_GO(EM_SWIGLU);
    else if (!p.out_f32) { if (p.resid) EPI_GO(EM_BF16_RESID); else EPI_GO(EM_BF16); }
    else { if (p.resid) EPI_GO(EM_F32_RESID); else EPI_GO(EM_F32); }
#undef EPI_GO
    // The vmcnt(0) after an EDGE tile's guarded epilogue is for the compiler's wait-count pass: its guarded loads (residual
    // rows under `m < M`) leave "maybe pending" VGPR writes on some paths, the pass carries that state around the
    // persistent-loop back-edge into the K loop's header and puts an `s_waitcnt vmcnt(0)` THERE -- inside the K loop, in
    // front of the fragment reads, draining the LDS-DMA queue on every K-tile of every tile (seen in the ISA: +11 % K-tile
    // time).  An interior tile's straight-line epilogue consumes every load it issues, so it needs no such wait and its
    // stores keep draining under the next tile's first DMAs.
  }
  }  // persistent tile loop
  CLK_MARK(2)
}

#if G256_FP8
int G256_LAUNCH(const GemmArgs& p, hipStream_t stream) {
  const int tile_rows = 256;
#else
int G256_LAUNCH(const GemmArgs& p, hipStream_t stream, int tile_rows) {  // tile_rows: 256 | 192 (p.tiles_m counts tiles of that height)
#endif
  if (tile_rows != 256 && tile_rows != 192) return GR_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)G256_KERNEL<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
#if !G256_FP8
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)G256_KERNEL<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
#endif
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // persistent launch: one block per CU (the 128 KB of LDS allow one block per CU anyway), fewer when there are fewer tiles
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GR_EINVAL;
    n_cu = prop.multiProcessorCount > 8 ? (prop.multiProcessorCount & ~7) : 8;
  }
#ifdef G256_NO_PERSIST  // diagnostic build (tests/diag/build_variant.py name -DG256_NO_PERSIST): one tile per workgroup
  const bool persist_off = true;
#else
  const bool persist_off = (G256_FP8 != 0 && !G256_FP8_PERSIST) || p.yield != 0;
#endif
  const int tiles = p.tiles_m * p.tiles_n;
  dim3 grid(persist_off ? tiles : (tiles < n_cu ? tiles : n_cu), p.splits);
#if !G256_FP8
  if (tile_rows == 192) hipLaunchKernelGGL(G256_KERNEL<6>, grid, dim3(NT), 2 * STAGE_BYTES, stream, p);
  else
#endif
    hipLaunchKernelGGL(G256_KERNEL<8>, grid, dim3(NT), 2 * STAGE_BYTES, stream, p);
  GR_CHECK_LAUNCH();
  return GR_OK;
}
