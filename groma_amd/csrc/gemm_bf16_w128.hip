// 256x256 bf16 MFMA GEMM, ONE wave per SIMD: 4 waves (2 x 2), each a 128x128 output block held in 256 accumulator
// registers (AGPRs) as 8 x 8 tiles of v_mfma_f32_16x16x32_bf16; the LDS is refilled by LDS-DMA.
//
// Why: per flop a 128x128 wave block reads a third fewer LDS bytes than the ping-pong kernel's 128x64 blocks
// (gemm_bf16_256.hip) and no barrier gates the matrix pipe.  (The vendor library's kernel for these shapes has this
// outline: 256 threads, 256x256x64 macro-tile, 512 registers, direct-to-LDS loads -- profiles/r02_blas_yardstick.txt.)
// History of the shape in this repo, all bit-identical to the ping-pong kernel:
//   * round 1: LDS-DMA, "2 ds_read + 1 DMA + 8 MFMA" per chunk: 1.07-1.12 PF;
//   * round 2 (git 5e7914d): global_load -> VGPR -> ds_write_b128, 16x16x32, 32-deep steps: 1.15-1.23 PF;
//   * round 2 (git bbb44ae): the same with 32x32x16 MFMAs and 64-deep K-tiles: 1.19-1.26 PF.
// What the price list (tests/diag/filler_price.py: one instruction between two MFMAs of a wave that owns its SIMD, all
// four SIMDs running the same stream) says about them: a ds_read_b128 per 16-clk gap is free (+0.5 clk) but two in one
// gap cost +15; a ds_write_b128 costs ~15 clk wherever it sits (13 clk of store path per instruction, CU-wide); an LDS-DMA
// costs ~8 clk when there is one in every fourth gap -- and four times that when it shares its neighbourhood with LDS
// reads and writes.  Hence this version (v4): no ds_write in the loop at all, and the DMA issues spread out.
//   * v5 (tried after this one, not kept): the same with a ring of four 32-deep slots (64-B LDS rows) to give a DMA piece
//     two to three sub-steps of flight instead of one to two: 1.17 PF against v4's 1.36 PF.  A DMA instruction that covers
//     16 rows x 64 B touches sixteen 128-B lines instead of eight, and the texture-address path prices a wave-instruction
//     by the LINES it touches (~2 clk each): the refill of one K-tile then keeps it busy for as long as the MFMAs take.
//     That is also what capped round 1's kernel and git 5e7914d.  Fetch whole lines.
//   * v4b (not kept): landing wait split in two (W pieces first, `vmcnt(8)` + barrier at the end of the even sub-step;
//     A pieces behind a second `vmcnt(8)` + barrier in the middle of the odd one) so that every piece has >= 1.5 sub-steps
//     of flight: K = 11008 0.91 -> 0.93x, the ViT shapes 0.97 -> 1.00x, the big K = 4096 shapes 0.97 -> 0.95x.
// Where v4 stands: 0.97-0.98x of the ping-pong kernel on the K <= 4096 shapes, 0.91x at K = 11008.  With two 64 KB stages
// a piece cannot have more than one to two sub-steps (1100-2200 clk) of flight -- the stage is free for two sub-steps and
// issuing its 16 pieces takes one -- and operands that come from HBM need more; the ping-pong kernel frees its stage in
// four parts and gives every piece a whole K-tile.  Ablations of v4: without the DMA 1.60 PF, without the reads 1.51 PF.
// NOT the default: gr_gemm_bf16 uses it for tile = 257 or GROMA_W128=1 only.
//
// Pipeline.  K-tile S (64 k-values) lives in LDS stage S&1 (A 256 x 128 B, then W 256 x 128 B = 64 KB) and is computed in
// two sub-steps of 64 MFMAs (k-halves h = 0, 1), each from one of two fragment register sets:
//   even sub-step (S,0): reads the fragments of (S,1); then `vmcnt(0)` (tile S+1, issued a whole sub-step ago, has
//        landed), `lgkmcnt(0)`, block barrier -- the ONLY one per K-tile: it publishes tile S+1 and tells every wave that
//        stage S&1 has been read for the last time;
//   odd sub-step (S,1):  reads the fragments of (S+1,0) and issues this wave's 16 DMA pieces of tile S+2 into stage S&1.
// The loop body is branch-free: past the end the DMA re-fetches the last K-tile into a stage nobody reads.
// LDS image (as in the ping-pong kernel): 128-B rows, 16-B chunk position = k-chunk ^ (row & 7); the DMA writes
// lane-linearly (8 rows x 128 B per wave-instruction) with the XOR applied to the per-lane SOURCE chunk.
// Epilogue: the shared LDS-staged batched epilogue (gemm_common.h), four 64-row passes.
// Plain GEMM only (no implicit-conv gather, no fp8); operands must be addressable with 32-bit byte offsets.
#include "gemm_common.h"
#include <cstdlib>

#define WT 256
#define WNT 256        // threads
#define WKT 64         // K per tile (one 128-B LDS row)
#define WSTAGE 65536   // bytes per stage
#define WB_OFF 32768

// timing ablations for tests/diag (results are wrong with any of them)
#ifdef W128_NO_R
#define W128_R(X)
#else
#define W128_R(X) X
#endif
#ifdef W128_NO_G
#define W128_G(X)
#else
#define W128_G(X) X
#endif
#ifdef W128_NO_BAR
#define W128_BAR(X)
#else
#define W128_BAR(X) X
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

  // ---- DMA geometry: piece c (0..15) of a K-tile = 8 rows x 128 B = 1 KB per wave; rows (c&7)*32 + wave*8 + (lane>>3) of
  // A (c < 8) or W (c >= 8).  The row's low 3 bits are lane>>3, so the lane's LDS slot lane&7 holds k-chunk (lane&7)^(lane>>3).
  const int drow = lane >> 3, kc = (lane & 7) ^ (lane >> 3);
  unsigned goff[16];  // per-lane byte offset from the (uniform) operand base of the K-tile
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int r = (c & 7) * 32 + wave * 8 + drow;
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
  auto dma_src = [&](int c, int tile) -> const char* {
    const int tc = tile < nt ? tile : nt - 1;  // branch-free tail: re-fetch the last K-tile
    return (c < 8 ? gA : gW) + (long)tc * (WKT * 2) + goff[c];
  };
  auto dma_issue = [&](const char* src, int c, int tile) {  // piece c of K-tile `tile` into stage tile&1
    glds16(src, smem + (tile & 1) * WSTAGE + (c < 8 ? 0 : WB_OFF) + (c & 7) * 4096 + wave * 1024);
  };
  auto dma = [&](int c, int tile) { dma_issue(dma_src(c, tile), c, tile); };

  // ---- fragment geometry (swapped operands: the W fragment is the MFMA's first operand, so a lane owns 4 consecutive
  // output columns of one row)
  const int fr = lane & 15, fg = lane >> 4;
  const int off0 = (fg ^ (fr & 7)) << 4;  // k-half h enters as XOR (h << 6)
  const int a_lane = (wm * 128 + fr) * 128 + off0;
  const int b_lane = WB_OFF + (wn * 128 + fr) * 128 + off0;

  f32x4 acc[8][8];  // [mi][nj]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 af[2][8], bf[2][8];  // [fragment set][16-row tile]
  // fragment r (0-7: W tile r, 8-15: A tile r-8) of sub-step t into set `buf`
  auto read_frag = [&](int buf, int t, int r) {
    const char* st = smem + ((t >> 1) & 1) * WSTAGE;
    const int x = (t & 1) << 6;
    if (r < 8) bf[buf][r] = *(const bf16x8*)(st + (b_lane ^ x) + r * 2048);
    else af[buf][r - 8] = *(const bf16x8*)(st + (a_lane ^ x) + (r - 8) * 2048);
  };

  // ---- prologue: K-tiles 0 and 1 in their stages; fragments of sub-step 0 in registers
#pragma unroll
  for (int c = 0; c < 16; ++c) dma(c, 0);
#pragma unroll
  for (int c = 0; c < 16; ++c) dma(c, 1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // tile 0 landed (tile 1 may fly: waited for at the end of sub-step 0)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) read_frag(0, 0, r);
  // hipcc's waitcnt pass understands this builtin (not an asm string): with the prologue's reads retired here, the wait it
  // puts at the loop top is the one the back edge needs
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)

  // One sub-step = 64 MFMAs; MFMA g (0..63) computes acc[g>>3][g&7].  Every other instruction sits directly behind an MFMA
  // issue, never two in one gap (pinned with sched_barriers).
#define SB __builtin_amdgcn_sched_barrier(0);
#define MF(G) acc[(G) >> 3][(G) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[CUR_][(G) & 7], af[CUR_][(G) >> 3], acc[(G) >> 3][(G) & 7], 0, 0, 0);
  // EVEN: the 16 fragment reads of the odd sub-step in every 3rd gap (all issued by gap 45, so the drain before the
  // barrier finds them done)
#define W128_EVEN(T)                                                                                         \
  {                                                                                                         \
    constexpr int CUR_ = 0;                                                                                 \
    const int t = (T);                                                                                      \
    _Pragma("unroll") for (int g = 0; g < 64; ++g) {                                                        \
      MF(g) SB                                                                                              \
      if (g % 3 == 0 && g / 3 < 16) { W128_R(read_frag(1, t + 1, g / 3);) }                                 \
      SB                                                                                                    \
    }                                                                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); /* next K-tile landed; this wave's reads of the stage done */ \
    W128_BAR(__builtin_amdgcn_s_barrier();)                                                                 \
  }
  // ODD: fragment read r in gap 4r, DMA piece d in gap 4d+2 with its 64-bit source address computed in gap 4d+1 (+1 % over
  // having both in one gap)
#define W128_DMA_AT(G, TILE)                                                                                 \
  if (((G) & 3) == 1) { dsrc = dma_src((G) >> 2, TILE); asm volatile("" : "+v"(dsrc)); }                    \
  if (((G) & 3) == 2) dma_issue(dsrc, (G) >> 2, TILE);
#define W128_ODD(T)                                                                                          \
  {                                                                                                         \
    constexpr int CUR_ = 1;                                                                                 \
    const int t = (T);                                                                                      \
    const char* dsrc = nullptr;                                                                             \
    (void)dsrc;                                                                                             \
    _Pragma("unroll") for (int g = 0; g < 64; ++g) {                                                        \
      MF(g) SB                                                                                              \
      if ((g & 3) == 0) { W128_R(read_frag(0, t + 1, g >> 2);) }                                            \
      W128_G(W128_DMA_AT(g, (t + 3) / 2))                                                                   \
      SB                                                                                                    \
    }                                                                                                       \
  }

  for (int S = 0; S < nt; ++S) {
    W128_EVEN(2 * S)
    W128_ODD(2 * S + 1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's re-fetches still target the stage buffers

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
  // (per-lane byte offsets are 32-bit: the last row's last chunk must still fit)
  return p.conv_C == 0 && (long)p.M * p.lda * 2 <= (1L << 32) - 256 && (long)p.N * p.ldw * 2 <= (1L << 32) - 256 && p.K % WKT == 0 &&
         (p.K / WKT) % p.splits == 0 && p.K / WKT / p.splits >= 2;
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
