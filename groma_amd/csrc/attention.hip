// Fused softmax attention on MFMA for gfx950 (flash-style, online softmax).
//
// Serves DINOv2 self-attention (non-causal, hd 64, T=1025) and LLaMA
// prefill / decode attention (causal + right-padding, hd 128, KV cache; a20/a22;
// mask semantics of HF LlamaModel as called from groma/model/groma.py:389-397:
// key j visible to query i  <=>  j <= pos(i)  and  j < kv_len[b]).
//
// Layouts (bf16): Q [B,H,Lq,hd], K [B,H,kv_stride,hd], Vt [B,H,hd,kv_stride] (V is
// kept TRANSPOSED so both MFMA operands of P.V read 16-B contiguous LDS chunks),
// Out [B*Lq, H*hd] token-major.
//
// Block = 4 waves, 128 query rows (2 tiles of 16 per wave, sharing every K / Vt fragment read), KV tile 64,
// K/Vt tiles double-buffered in LDS (DMA of tile t+1 overlaps the MFMAs of tile t).  Scores are computed
// transposed, S^T = K.Q^T (v_mfma_f32_16x16x32_bf16, K rows as the A operand) so that
// a lane owns ONE query column: the row max/sum are in-lane + 2 shuffles, and P^T feeds
// the second MFMA (O^T = Vt.P^T) straight from registers -- no LDS round trip for P.
// K / Vt tiles are staged with global_load_lds into an XOR-swizzled, conflict-free image.
#include <type_traits>

#include "gr_common.h"
#include "../../include/groma_hip.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnArgs {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* out;
  const int* kv_len;  // [B] or null
  const int* pos_dev;  // or null: q_pos0 of row b = pos_dev[b * pos_stride], Skv = q_pos0 + Lq (device-resident step)
  int pos_stride;
  // q_ld > 0: q is read straight out of the fused projection buffer [B*Lq, q_ld] (row b*Lq+i, columns h*hd..) instead of
  // a packed [B,H,Lq,hd] copy, and, when cos/sin are given, HF rotate_half RoPE at position q_pos0 + i is applied while
  // the fragments are loaded (the partner d +- hd/2 of a lane's 8 values sits in the same lane, fragment kk +- hd/64)
  long q_ld;
  const float* rope_cos;
  const float* rope_sin;
  int B, H, Lq, Skv, kv_stride;
  int causal, q_pos0;
  float scale_log2;  // softmax scale * log2(e)
};

#ifdef G256_CLK  // diagnostic build (tests/diag/build_clk.py): per-iteration segment clocks of block 0 / wave 0
__device__ unsigned long long att_clk[64];
extern "C" int gr_diag_att_clk(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(att_clk), sizeof(att_clk)); }
#define ATT_MARK(i) if (blockIdx.y == 0 && blockIdx.x == 0 && t >= 2 && t < 6) am[(t - 2) * 5 + (i)] = __builtin_readcyclecounter();
#else
#define ATT_MARK(i)
#endif

// ATT_PIPE = 1: the pipelined main loop (see the loop itself); ATT_PIPE_PRE: soft-max steps issued ahead of its first MFMA
#ifndef ATT_PIPE
#define ATT_PIPE 0
#endif
#ifndef ATT_PIPE_PRE
#define ATT_PIPE_PRE 5
#endif
// (soft-max row statistics are reduced across the four 16-lane rows with rows_max / rows_sum, gr_common.h)
#ifndef ATT_G64
#define ATT_G64 2   // K / V^T fragments per prefetch group at head dim 64
#endif
#ifndef ATT_G128
#define ATT_G128 ((GR_SP || ATT_PIPE) ? 4 : 8)  // ... at head dim 128 (A/B: -5 % with 8; each fragment is a hi / lo pair in the split build)
#endif
#define ATT_G (HD == 128 ? ATT_G128 : ATT_G64)
// Split-operand build (gr_common.h): q, K, V^T and the context are (hi, lo) pairs in 32-element blocks, so a K row is 2*hd
// physical elements whose 16-B chunk 8*kk + g holds the hi halves of d = 32*kk + 8*g.. and chunk 8*kk + 4 + g their lo halves, a
// 64-key V^T tile row is 128 physical elements (chunk 8*tt + g = hi of keys 32*tt + 8*g.., + 4 = lo), and both contractions
// issue hi.hi + hi.lo + lo.hi.  P is split in registers.  Tiles are twice as large: one workgroup per CU.

// De-phasing the two workgroups that share a CU (round 5).  A wave-iteration is ~1.0 k clk of MFMA followed by ~1.2 k clk of
// soft-max VALU, and the two waves a SIMD holds (one from each resident workgroup) start together and then STAY in phase: while
// both are in their MFMA segment they share the matrix pipe, while both are in the soft-max they share the VALU, so the pair
// costs the SUM of the two segments (tests/diag/attn_clk.py) although the pipes are independent.  Two compile-time knobs break
// the symmetry by the wave's hardware slot on its SIMD (HW_ID[3:0]; the two co-resident waves differ in it):
//   ATT_PRIO_SLOT = p : odd slots run at s_setprio p -- the favoured wave gets through a contended segment first and the pair
//                       settles half an iteration apart (MFMA of one under the soft-max of the other);
//   ATT_SKEW = n      : the workgroup whose wave 0 sits in an odd slot sleeps n x 1024 clk before its first tile.
#ifndef ATT_PRIO_SLOT
#define ATT_PRIO_SLOT 0
#endif
#ifndef ATT_SKEW
#define ATT_SKEW 0
#endif
// ATT_PRIO_PHASE = 1: s_setprio 1 around the two MFMA loops of an iteration (S^T = K.Q^T and O^T += V^T.P^T), 0 in the soft-max.
// VALU issue between the waves of a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md "Two waves per SIMD", item 2): a
// wave in its MFMA segment that loses issue slots to a co-resident wave's soft-max VALU also loses matrix-pipe time, while a
// soft-max that yields to MFMA issues still gets the 3 of 4 issue slots an MFMA stream leaves free.
#ifndef ATT_PRIO_PHASE
#define ATT_PRIO_PHASE 0
#endif
#if ATT_PRIO_PHASE
#define ATT_MFMA_BEGIN __builtin_amdgcn_s_setprio(1);
#define ATT_MFMA_END __builtin_amdgcn_s_setprio(0);
#else
#define ATT_MFMA_BEGIN
#define ATT_MFMA_END
#endif

// (score sub-tile j, k-step kk) of fragment e of prefetch group g.  Round 5: inside a group the k-step is the SLOW index when the
// group spans several sub-tiles (hd 128: 8 fragments = 2 sub-tiles x 4 k-steps), so that consecutive MFMAs go to different
// accumulators (s[u][j]: 2 q-tiles x 2 sub-tiles = 4 independent chains) instead of four dependent MFMAs per accumulator in a row.
// Measured flat (profiles/r05_attn_variants.txt: 91.3 vs 91.6 us at 14 x 32 heads, 113.7 vs 112.2 us on the ViT shape, pair build 291.4
// vs 291.6): the dependent-MFMA spacing is not what the kernel waits for.  OFF by default (= round 4's accumulation order, bitwise).
#ifndef ATT_S_ORDER
#define ATT_S_ORDER 0
#endif
template <int GK, int NKK>
__device__ __forceinline__ constexpr int att_step_j(int g, int e) {
  constexpr int NJ = GK / NKK;  // whole sub-tiles per group (0: a group is part of one sub-tile)
  return (ATT_S_ORDER && NJ > 1) ? g * NJ + e % NJ : (g * GK + e) / NKK;
}
template <int GK, int NKK>
__device__ __forceinline__ constexpr int att_step_kk(int g, int e) {
  constexpr int NJ = GK / NKK;
  return (ATT_S_ORDER && NJ > 1) ? e / NJ : (g * GK + e) % NKK;
}
// ATT_SUM4: the soft-max row sum / row max of a lane's 16 scores as 4 independent partial chains (one per sub-tile) combined at the
// end, instead of one serial chain of 16 dependent adds / 8 dependent max3 (dependent VALU issues at ~0.6 of the independent rate)
// Measured neutral-to-slower (30.2 vs 29.0 us at 4 images, 115.0 vs 112.2 us on the ViT shape): OFF.
#ifndef ATT_SUM4
#define ATT_SUM4 0
#endif

template <int I, int N, class F>
__device__ __forceinline__ void att_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    att_static_for<I + 1, N>(f);
  }
}

template <int HD>
__global__ __launch_bounds__(256, GR_SP ? 1 : 2) void attention_kernel(AttnArgs p) {
  constexpr int KV = 64;
  constexpr int QT = 2;               // 16-row query tiles per wave -> 32 query rows / wave, 128 / block
  constexpr int SPW = GR_SPW;         // physical elements per logical element (2 in the split build)
  constexpr int KROW = HD * 2 * SPW;  // bytes per K row
  constexpr int KCH = KROW / 16;      // 16-B chunks per K row (8 or 16; split build 16 or 32)
  constexpr int KTILE = KV * KROW;    // bytes
  constexpr int VROW = KV * 2 * SPW;  // bytes per V^T tile row (128; split build 256)
  constexpr int VCH = VROW / 16;
  constexpr int VTILE = HD * VROW;    // Vt tile [HD][64] bf16
#ifdef ATT_R04_PAIR  // A/B only (tests/diag): round 4's pair-build soft-max split and V^T swizzle
  constexpr int VSWZ = 7;
#else
  constexpr int VSWZ = VCH - 1;       // V^T chunk swizzle mask: all chunk positions of a row (8, or 16 in the pair build)
#endif
  constexpr int STAGE = KTILE + VTILE;
  // Key order inside a 64-key tile: MFMA row i of score sub-tile j holds key (j>>1)*32 + (i>>2)*8 + (j&1)*4 + (i&3),
  // so that after S^T = K.Q^T a lane (k-group fg) owns keys 32*tt + fg*8 + {0..7} of query fr: exactly the 8
  // CONTIGUOUS keys of the standard MFMA k-slot, and the Vt fragment is ONE 16-B LDS read (ds_read_b128).
  // K-tile chunk swizzle (conflict-free for the 16 rows {0-3, 8-11, 16-19, 24-27}(+4) one ds_read_b128 touches):
  auto kswz = [](int row) { return KCH >= 16 ? ((row & 3) | (((row >> 3) & 3) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 3) << 1)); };
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages (double buffer)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
#if ATT_PRIO_SLOT || ATT_SKEW
  const unsigned hw_slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);  // HW_REG_HW_ID bits [3:0]: wave slot on this SIMD
#endif
#if ATT_PRIO_SLOT
  if (hw_slot & 1) __builtin_amdgcn_s_setprio(ATT_PRIO_SLOT);
#endif
  // Block order: (batch, head) fastest, query block slowest and -- when causal -- the block with the most visible keys
  // first, so the long blocks start in the first wave of workgroups and the short ones fill the tail of the launch
  // (A/B on one box, tests/diag/attn_bench.py: L = 582 causal 14 x 32 heads 97.5 -> 96 us, 4 x 32 heads 34.1 -> 26.5 us;
  // longest-first INSIDE a (batch, head) with the query block fastest was neutral-to-worse)
  const int bh = blockIdx.x;
  const int qb = p.causal ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * (64 * QT);

  const long q_base = p.q_ld > 0 ? (long)b * p.Lq * p.q_ld + h * HD : (long)bh * p.Lq * HD;  // logical element index
  const long q_rs = p.q_ld > 0 ? p.q_ld : HD;  // row stride
  const bf16_t* Kp = p.k + (long)bh * p.kv_stride * HD * SPW;
  const bf16_t* Vp = p.vt + (long)bh * HD * p.kv_stride * SPW;

  const int q_pos0 = p.pos_dev ? p.pos_dev[b * p.pos_stride] : p.q_pos0;
  const int Skv = p.pos_dev ? q_pos0 + p.Lq : p.Skv;
  int kvmax = Skv;
  if (p.kv_len) kvmax = min(kvmax, p.kv_len[b]);

  // loop bounds: block-level (staging + barriers) and wave-level (compute)
  const int blk_last = min(q0 + 64 * QT - 1, p.Lq - 1);
  const int wav_last = min(q0 + wave * (16 * QT) + 16 * QT - 1, p.Lq - 1);
  const int blk_limit = p.causal ? min(kvmax, q_pos0 + blk_last + 1) : kvmax;
  const int wav_limit = p.causal ? min(kvmax, q_pos0 + wav_last + 1) : kvmax;
  const int ntiles = (blk_limit + KV - 1) / KV;
  const bool wave_has_rows = q0 + wave * (16 * QT) < p.Lq;
  const int wav_first = min(q0 + wave * (16 * QT), p.Lq - 1);
  const int wav_min_limit = p.causal ? min(kvmax, q_pos0 + wav_first + 1) : kvmax;  // min over the wave's rows

  // stage K tile (lds chunk position pos holds logical chunk pos ^ (row&7)) and Vt tile of kv tile t into buffer t&1
  auto stage_k = [&](int t) {
    const int kv0 = t * KV;
    char* ksm = smem + (t & 1) * STAGE;
    constexpr int NCH = KV * KCH;
#pragma unroll
    for (int i = 0; i < NCH / 256; ++i) {
      const int q = i * 256 + tid;
      const int row = q / KCH, pos = q % KCH;
      const int c = pos ^ kswz(row);
      int kr = kv0 + row;
      if (kr > Skv - 1) kr = Skv - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(Kp + (long)kr * (HD * SPW) + c * 8), (lptr_t)(ksm + i * 4096 + wave * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](int t) {
    const int kv0 = t * KV;
    char* vsm = smem + (t & 1) * STAGE + KTILE;
    constexpr int NVC = HD * VCH;
#pragma unroll
    for (int i = 0; i < NVC / 256; ++i) {
      const int q = i * 256 + tid;
      const int row = q / VCH, pos = q % VCH;
      const int c = pos ^ (row & VSWZ);
      __builtin_amdgcn_global_load_lds((gptr_t)(Vp + ((long)row * p.kv_stride + kv0) * SPW + c * 8),
                                       (lptr_t)(vsm + i * 4096 + wave * 1024), 16, 0, 0);
    }
  };
  auto stage = [&](int t) {
    stage_k(t);
    stage_v(t);
  };

  // ATT_STAGE_FIRST: tile 0's LDS-DMA is requested BEFORE the query fragments (and their RoPE tables) are fetched, so the two
  // latencies of a workgroup's prologue overlap instead of adding up (a causal block at L = 582 runs only 2-10 tiles).
#ifndef ATT_STAGE_FIRST
#define ATT_STAGE_FIRST 0
#endif
#if ATT_STAGE_FIRST
  if (ntiles > 0) stage(0);
#endif
  // this lane's query rows (B-operand columns), one per q-tile -- clamp the tail
  int qi[QT], limit[QT];
  bool q_valid[QT];
  bf16x8 qf[QT][HD / 32];
#if GR_SP
  bf16x8 qfl[QT][HD / 32];  // lo halves
#endif
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    qi[u] = q0 + wave * (16 * QT) + u * 16 + fr;
    q_valid[u] = qi[u] < p.Lq;
    if (!q_valid[u]) qi[u] = p.Lq - 1;
#pragma unroll
    for (int kk = 0; kk < HD / 32; ++kk) {
      const bf16_t* qsrc = p.q + sp_idx(q_base + (long)qi[u] * q_rs + kk * 32 + fg * 8);
      qf[u][kk] = *(const bf16x8*)qsrc;
#if GR_SP
      qfl[u][kk] = *(const bf16x8*)(qsrc + 32);
#endif
    }
    if (p.rope_cos) {  // rotate in registers: out[d] = x[d]*cos - x[d+half]*sin (d < half), x[d]*cos + x[d-half]*sin (d >= half)
      constexpr int HALF = HD / 2, KH = HD / 64;  // fragments per half
      const float* cp = p.rope_cos + (long)(q_pos0 + qi[u]) * HALF + fg * 8;
      const float* sp = p.rope_sin + (long)(q_pos0 + qi[u]) * HALF + fg * 8;
      bf16x8 rot[HD / 32];
#if GR_SP
      bf16x8 rotl[HD / 32];
#endif
#pragma unroll
      for (int kk = 0; kk < HD / 32; ++kk) {
        const int kp = kk < KH ? kk + KH : kk - KH;
        const float sgn = kk < KH ? -1.f : 1.f;
        const int dc = (kk % KH) * 32;
        const f32x4 c0 = *(const f32x4*)(cp + dc), c1 = *(const f32x4*)(cp + dc + 4);
        const f32x4 s0 = *(const f32x4*)(sp + dc), s1 = *(const f32x4*)(sp + dc + 4);
        union { bf16x8 v; uint32_t w[4]; } o, ol;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const float ca = e < 4 ? c0[e] : c1[e - 4], cb = e < 4 ? c0[e + 1] : c1[e - 3];
          const float sa = e < 4 ? s0[e] : s1[e - 4], sb = e < 4 ? s0[e + 1] : s1[e - 3];
#if GR_SP
          const float x0 = bf2f((bf16_t)qf[u][kk][e]) + bf2f((bf16_t)qfl[u][kk][e]), x1 = bf2f((bf16_t)qf[u][kk][e + 1]) + bf2f((bf16_t)qfl[u][kk][e + 1]);
          const float y0 = bf2f((bf16_t)qf[u][kp][e]) + bf2f((bf16_t)qfl[u][kp][e]), y1 = bf2f((bf16_t)qf[u][kp][e + 1]) + bf2f((bf16_t)qfl[u][kp][e + 1]);
#else
          const float x0 = bf2f((bf16_t)qf[u][kk][e]), x1 = bf2f((bf16_t)qf[u][kk][e + 1]);
          const float y0 = bf2f((bf16_t)qf[u][kp][e]), y1 = bf2f((bf16_t)qf[u][kp][e + 1]);
#endif
          split2(x0 * ca + sgn * y0 * sa, x1 * cb + sgn * y1 * sb, o.w[e >> 1], ol.w[e >> 1]);
        }
        rot[kk] = o.v;
#if GR_SP
        rotl[kk] = ol.v;
#endif
      }
#pragma unroll
      for (int kk = 0; kk < HD / 32; ++kk) {
        qf[u][kk] = rot[kk];
#if GR_SP
        qfl[u][kk] = rotl[kk];
#endif
      }
    }
    limit[u] = p.causal ? min(kvmax, q_pos0 + qi[u] + 1) : kvmax;  // keys [0, limit) visible
  }
  f32x4 o[QT][HD / 16];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = -1e30f;
    l_run[u] = 0.f;
#pragma unroll
    for (int n = 0; n < HD / 16; ++n) o[u][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

#ifdef G256_CLK
  unsigned long long am[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if !ATT_STAGE_FIRST
  if (ntiles > 0) stage(0);
#endif
#if ATT_SKEW
  {
    __shared__ int skew;
    if (tid == 0) skew = (int)(hw_slot & 1);
    __syncthreads();
    if (skew)
      for (int i = 0; i < ATT_SKEW; ++i) __builtin_amdgcn_s_sleep(16);
  }
#endif
  // ---- the three pieces of an iteration (shared by the plain loop and the pipelined one, ATT_PIPE)
  // S^T tiles: rows = keys (4 sub-tiles of 16), col = this lane's query (per q-tile); K fragments shared.
  // K fragments are read one group AHEAD of the MFMAs that use them (double buffer, GK fragments per group): the
  // LDS latency of the next group hides behind this group's MFMAs instead of stalling every MFMA pair
  constexpr int GK = ATT_G;                   // fragments per group
  constexpr int NKK = HD / 32;                // k-steps per score sub-tile
  constexpr int NKG = 4 * NKK / GK;           // groups over (j, kk)
  auto load_k = [&](const char* ksm, int g, bf16x8* dst) {
#pragma unroll
    for (int e = 0; e < GK; ++e) {
      const int j = att_step_j<GK, NKK>(g, e), kk = att_step_kk<GK, NKK>(g, e);
      const int row = (j >> 1) * 32 + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3);
      dst[e] = *(const bf16x8*)(ksm + row * KROW + (((kk * 4 * SPW + fg) ^ kswz(row)) << 4));
#if GR_SP
      dst[GK + e] = *(const bf16x8*)(ksm + row * KROW + (((kk * 8 + 4 + fg) ^ kswz(row)) << 4));
#endif
    }
  };
  auto compute_S = [&](const char* ksm, f32x4 (&s)[QT][4]) {
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 kfr[2][GK * SPW];  // split build: [.., GK + e] = lo halves
    load_k(ksm, 0, kfr[0]);
    ATT_MFMA_BEGIN
#pragma unroll
    for (int g = 0; g < NKG; ++g) {
      if (g + 1 < NKG) load_k(ksm, g + 1, kfr[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#if GR_SP && ATT_S_ORDER
      // pair build: the three passes are the SLOW index, so an accumulator is revisited every GK * QT MFMAs, not every QT
#pragma unroll
      for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int e = 0; e < GK; ++e) {
          const int j = att_step_j<GK, NKK>(g, e), kk = att_step_kk<GK, NKK>(g, e);
#pragma unroll
          for (int u = 0; u < QT; ++u)
            s[u][j] = GR_MFMA_16x16x32(kfr[g & 1][pp == 2 ? GK + e : e], pp == 1 ? qfl[u][kk] : qf[u][kk], s[u][j]);
        }
#else
#pragma unroll
      for (int e = 0; e < GK; ++e) {
        const int j = att_step_j<GK, NKK>(g, e), kk = att_step_kk<GK, NKK>(g, e);
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][j] = GR_MFMA_16x16x32(kfr[g & 1][e], qf[u][kk], s[u][j]);
#if GR_SP
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][j] = GR_MFMA_16x16x32(kfr[g & 1][e], qfl[u][kk], s[u][j]);
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][j] = GR_MFMA_16x16x32(kfr[g & 1][GK + e], qf[u][kk], s[u][j]);
#endif
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    ATT_MFMA_END
  };
  // online softmax; lane holds keys kv0 + (j>>1)*32 + fg*8 + (j&1)*4 + r of query fr (per q-tile).
  // The running max is kept in RAW score units and the softmax scale is folded into one fma per element
  // (e = exp2(s*c - m*c)); masking (2 cmp + 2 select per element) is compiled only into boundary tiles.
  union PB { bf16x8 v; uint32_t w[4]; };
  PB pb[QT][2];
#if GR_SP
  PB pbl[QT][2];  // lo halves of P
#endif
  const float cs = p.scale_log2;
#if GR_SP
#ifdef ATT_R04_PAIR
#define ATT_SPLIT_P split2
#else
#define ATT_SPLIT_P split2_unit   // (P in [0, 1]: no saturation)
#endif
#endif
  // P^T of q-tile u, key half tt, as the B operand; k-slot (fg,e) <-> key 32*tt + fg*8 + e (standard contiguous slot)
  auto pack_p = [&](f32x4 (&s)[QT][4], int u, int tt) {
#if GR_SP
    ATT_SPLIT_P(s[u][2 * tt][0], s[u][2 * tt][1], pb[u][tt].w[0], pbl[u][tt].w[0]);
    ATT_SPLIT_P(s[u][2 * tt][2], s[u][2 * tt][3], pb[u][tt].w[1], pbl[u][tt].w[1]);
    ATT_SPLIT_P(s[u][2 * tt + 1][0], s[u][2 * tt + 1][1], pb[u][tt].w[2], pbl[u][tt].w[2]);
    ATT_SPLIT_P(s[u][2 * tt + 1][2], s[u][2 * tt + 1][3], pb[u][tt].w[3], pbl[u][tt].w[3]);
#else
    pb[u][tt].w[0] = pack2bf_unit(s[u][2 * tt][0], s[u][2 * tt][1]);
    pb[u][tt].w[1] = pack2bf_unit(s[u][2 * tt][2], s[u][2 * tt][3]);
    pb[u][tt].w[2] = pack2bf_unit(s[u][2 * tt + 1][0], s[u][2 * tt + 1][1]);
    pb[u][tt].w[3] = pack2bf_unit(s[u][2 * tt + 1][2], s[u][2 * tt + 1][3]);
#endif
  };
  auto softmax_tile = [&](auto masked_tag, f32x4 (&s)[QT][4], int kv0) {
    constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float mxp[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (MASKED) {
            const int key = kv0 + (j >> 1) * 32 + fg * 8 + (j & 1) * 4 + r;
            if (key >= limit[u]) s[u][j][r] = -1e30f;
          }
          mxp[ATT_SUM4 ? j : 0] = fmaxf(mxp[ATT_SUM4 ? j : 0], s[u][j][r]);
        }
      float mx = fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3]));
      mx = rows_max(mx);
      const float m_new = fmaxf(m_run[u], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * cs);
      const float mc = -m_new * cs;
      float rsp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // masked entries: s = -1e30 -> exp2(-huge) = 0 exactly, unless the whole row is masked so far (m_new = -1e30,
          // argument 0): select 0 there
          float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][j][r], cs, mc));
          if (MASKED) e = s[u][j][r] <= -1e30f ? 0.f : e;
          s[u][j][r] = e;
          rsp[ATT_SUM4 ? j : 0] += e;
        }
      float rs = (rsp[0] + rsp[1]) + (rsp[2] + rsp[3]);
      rs = rows_sum(rs);
      l_run[u] = l_run[u] * alpha + rs;
      m_run[u] = m_new;
#pragma unroll
      for (int n = 0; n < HD / 16; ++n) o[u][n] *= alpha;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) pack_p(s, u, tt);
    }
  };
  // O^T += Vt . P^T ; Vt fragments shared by the q-tiles; the next group's reads are issued before this group's MFMAs
  auto compute_PV = [&](const char* vsm) {
    constexpr int GV = ATT_G;
    constexpr int NG = 2 * (HD / 16) / GV;  // groups of GV (tt, n) steps
    bf16x8 vfr[2][GV * SPW];
    auto load_v = [&](int g, bf16x8* dst) {
#pragma unroll
      for (int e = 0; e < GV; ++e) {
        const int step = g * GV + e, tt = step / (HD / 16), n = step % (HD / 16);
        const int row = n * 16 + fr;      // d index
        const int c = tt * 4 * SPW + fg;  // keys 32*tt + fg*8 .. +7 = one 16-B chunk of the Vt row
        // chunk swizzle over ALL chunk positions of a V^T row: 8 in the 16-bit builds, 16 in the pair build (256-B rows start on
        // bank 0 there, and `row & 7` left the 16 rows a ds_read_b128 lane group touches 2-way conflicted -- round 5)
        dst[e] = *(const bf16x8*)(vsm + row * VROW + ((c ^ (row & VSWZ)) << 4));
#if GR_SP
        dst[GV + e] = *(const bf16x8*)(vsm + row * VROW + (((c + 4) ^ (row & VSWZ)) << 4));
#endif
      }
    };
    load_v(0, vfr[0]);
    ATT_MFMA_BEGIN
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) load_v(g + 1, vfr[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#if GR_SP && ATT_S_ORDER
#pragma unroll
      for (int pp = 0; pp < 3; ++pp)   // (passes slow, fragments fast: an accumulator is revisited every GV * QT MFMAs)
#pragma unroll
        for (int e = 0; e < GV; ++e) {
          const int step = g * GV + e, tt = step / (HD / 16), n = step % (HD / 16);
#pragma unroll
          for (int u = 0; u < QT; ++u)
            o[u][n] = GR_MFMA_16x16x32(vfr[g & 1][pp == 2 ? GV + e : e], pp == 1 ? pbl[u][tt].v : pb[u][tt].v, o[u][n]);
        }
#else
#pragma unroll
      for (int e = 0; e < GV; ++e) {
        const int step = g * GV + e, tt = step / (HD / 16), n = step % (HD / 16);
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][n] = GR_MFMA_16x16x32(vfr[g & 1][e], pb[u][tt].v, o[u][n]);
#if GR_SP
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][n] = GR_MFMA_16x16x32(vfr[g & 1][e], pbl[u][tt].v, o[u][n]);
#pragma unroll
        for (int u = 0; u < QT; ++u) o[u][n] = GR_MFMA_16x16x32(vfr[g & 1][GV + e], pb[u][tt].v, o[u][n]);
#endif
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    ATT_MFMA_END
  };

#if ATT_PIPE
  // ---- pipelined loop (round 6): iteration t issues the score MFMAs of tile t + 1 BETWEEN the soft-max instructions of tile t, one
  // MFMA per 16-clk matrix-pipe slot with the VALU work in its shadow (an in-order wave only overlaps the two pipes if its own
  // instruction stream alternates them; `sched_barrier` fences pin the order -- the compiler would cluster the MFMAs otherwise),
  // then P.V of tile t.  K is staged one tile ahead of V^T in the same two LDS stages.  Same operations in the same order per
  // accumulator as the plain loop: bitwise-identical results (tests/diag/attn_variants.py).  Boundary tiles (masked soft-max) and
  // the last tile run the plain pieces.
  static_assert(!ATT_S_ORDER && !ATT_SUM4, "the pipelined loop implements the default accumulation order");
  const int nt_w = wave_has_rows ? min(ntiles, (wav_limit + KV - 1) / KV) : 0;  // this wave computes tiles [0, nt_w)
  auto fused = [&](const char* ksm, f32x4 (&cur)[QT][4], f32x4 (&nxt)[QT][4]) {
    constexpr int PASSES = GR_SP ? 3 : 1;
    constexpr int NM = 4 * NKK * PASSES * QT;  // score MFMAs of a tile
    constexpr int NO = HD / 16;
    // soft-max steps per q-tile: 4 max, 1 row max, 18 of the exp phase (software-pipelined over the 16 scores: the fma of score k, the
    // exp of score k - 1 and the row-sum add of score k - 2 share a step, so a step holds no dependent pair), 1 row sum, NO rescales, 2 packs
    constexpr int NVU = 26 + NO;
    constexpr int NV = QT * NVU;
    constexpr int PRE = ATT_PIPE_PRE;          // soft-max steps ahead of the first MFMA (in the shadow of the first K fragment reads)
    bf16x8 kfr[2][GK * SPW];
    float mx[QT], m_new[QT], alpha[QT], mc[QT], rs[QT];
    auto valu = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int u = k / NVU, q = k % NVU;
      if constexpr (q < 4) {
        const float m4 = fmaxf(fmaxf(cur[u][q][0], cur[u][q][1]), fmaxf(cur[u][q][2], cur[u][q][3]));
        mx[u] = q == 0 ? fmaxf(-1e30f, m4) : fmaxf(mx[u], m4);
      } else if constexpr (q == 4) {
        mx[u] = rows_max(mx[u]);
        m_new[u] = fmaxf(m_run[u], mx[u]);
        alpha[u] = __builtin_amdgcn_exp2f((m_run[u] - m_new[u]) * cs);
        mc[u] = -m_new[u] * cs;
      } else if constexpr (q < 23) {
        constexpr int x = q - 5;  // 0..17
        if constexpr (x >= 2) {   // add of score x - 2 (in score order: the plain loop's summation order)
          constexpr int j = (x - 2) / 4, r = (x - 2) % 4;
          rs[u] = x == 2 ? 0.f + cur[u][j][r] : rs[u] + cur[u][j][r];
        }
        if constexpr (x >= 1 && x <= 16) {
          constexpr int j = (x - 1) / 4, r = (x - 1) % 4;
          cur[u][j][r] = __builtin_amdgcn_exp2f(cur[u][j][r]);
        }
        if constexpr (x <= 15) {
          constexpr int j = x / 4, r = x % 4;
          cur[u][j][r] = __builtin_fmaf(cur[u][j][r], cs, mc[u]);
        }
      } else if constexpr (q == 23) {
        rs[u] = rows_sum(rs[u]);
        l_run[u] = l_run[u] * alpha[u] + rs[u];
        m_run[u] = m_new[u];
      } else if constexpr (q < 24 + NO) {
        o[u][q - 24] *= alpha[u];
      } else {
        pack_p(cur, u, q - 24 - NO);
      }
    };
    auto mfma = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int u = i % QT, pass = (i / QT) % PASSES, ge = i / (QT * PASSES);
      constexpr int g = ge / GK, e = ge % GK, j = ge / NKK, kk = ge % NKK;
      if constexpr (e == 0 && pass == 0 && u == 0 && g + 1 < NKG) load_k(ksm, g + 1, kfr[(g + 1) & 1]);
      const f32x4 c = (kk == 0 && pass == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : nxt[u][j];
#if GR_SP
      nxt[u][j] = GR_MFMA_16x16x32(kfr[g & 1][pass == 2 ? GK + e : e], pass == 1 ? qfl[u][kk] : qf[u][kk], c);
#else
      nxt[u][j] = GR_MFMA_16x16x32(kfr[g & 1][e], qf[u][kk], c);
#endif
    };
    load_k(ksm, 0, kfr[0]);
    __builtin_amdgcn_sched_barrier(0);
    att_static_for<0, PRE>(valu);
    __builtin_amdgcn_sched_barrier(0);
    att_static_for<0, NM>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      mfma(ic);
      __builtin_amdgcn_sched_barrier(0);
      att_static_for<PRE + i * (NV - PRE) / NM, PRE + (i + 1) * (NV - PRE) / NM>(valu);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  f32x4 sA[QT][4], sB[QT][4];
  if (ntiles > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile 0 landed for every wave
    if (ntiles > 1) stage_k(1);
    if (nt_w > 0) compute_S(smem, sA);
  }
  auto iter = [&](int t, f32x4 (&cur)[QT][4], f32x4 (&nxt)[QT][4]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K(t+1) and V^T(t) landed for every wave; everyone finished iteration t-1 (read K(t), V^T(t-1))
    if (t + 2 < ntiles) stage_k(t + 2);
    if (t + 1 < ntiles) stage_v(t + 1);
    if (t >= nt_w) return;  // wave-uniform: every key of this tile is masked for all of this wave's rows, or it has no query row
    const int kv0 = t * KV;
    const char* ksm_n = smem + ((t + 1) & 1) * STAGE;
    const char* vsm = smem + (t & 1) * STAGE + KTILE;
    const bool unmasked = kv0 + KV <= wav_min_limit;  // every key of the tile visible to every row
    if (ATT_PIPE == 1 && t + 1 < nt_w && unmasked) {
      fused(ksm_n, cur, nxt);
    } else {
      if (t + 1 < nt_w) compute_S(ksm_n, nxt);
      if (unmasked) softmax_tile(std::false_type{}, cur, kv0);
      else softmax_tile(std::true_type{}, cur, kv0);
    }
    compute_PV(vsm);
  };
  for (int t = 0; t < ntiles; t += 2) {
    iter(t, sA, sB);
    if (t + 1 < ntiles) iter(t + 1, sB, sA);
  }
#else
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * KV;
    ATT_MARK(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile t landed for every wave; everyone finished reading the other buffer
    ATT_MARK(1)
    if (t + 1 < ntiles) stage(t + 1);
    // wave-uniform skips: every key of this tile is masked for all of this wave's rows, or the wave has no query row at all
    // (the last query block of T = 1025 holds one row: three of its four waves only help staging the tiles)
    if (kv0 >= wav_limit || !wave_has_rows) continue;
    const char* ksm = smem + (t & 1) * STAGE;
    const char* vsm = ksm + KTILE;
    f32x4 s[QT][4];
    compute_S(ksm, s);
    ATT_MARK(2)
    if (kv0 + KV <= wav_min_limit) softmax_tile(std::false_type{}, s, kv0);  // every key of the tile visible to every row
    else softmax_tile(std::true_type{}, s, kv0);
    ATT_MARK(3)
    compute_PV(vsm);
    ATT_MARK(4)
  }
#endif
#ifdef G256_CLK
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == 0 && HD == 128)
    for (int i = 0; i < 20; ++i) att_clk[i] = am[i];
#endif

  // ---- epilogue: lane holds d = n*16 + fg*4 + r for query fr
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (!q_valid[u]) continue;
    const float inv = l_run[u] > 0.f ? 1.0f / l_run[u] : 0.f;
    const long orow = ((long)b * p.Lq + qi[u]) * (p.H * HD) + h * HD;  // logical element index
#pragma unroll
    for (int n = 0; n < HD / 16; ++n) st4f(p.out, orow + n * 16 + fg * 4, o[u][n] * inv);
  }
}

extern "C" int gr_attention_bf16(const void* q, const void* k, const void* vt, void* out, const int* kv_len, int B, int H,
                                 int Lq, int Skv, int kv_stride, int head_dim, int causal, int q_pos0, float scale,
                                 const int* pos_dev, int pos_stride, long q_ld, const float* rope_cos,
                                 const float* rope_sin, hipStream_t stream) {
  if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Lq <= 0 || Skv <= 0) return GR_EINVAL;
  if (kv_stride % 64 != 0 || kv_stride < Skv) return GR_EINVAL;  // Vt tile reads run to the next multiple of 64
  if (GR_SP && q_ld % 32 != 0) return GR_EINVAL;
  AttnArgs p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.out = (bf16_t*)out;
  p.kv_len = kv_len;
  p.pos_dev = pos_dev; p.pos_stride = pos_stride;
  if (q_ld < 0 || (q_ld > 0 && (q_ld % 8 != 0 || q_ld < (long)H * head_dim)) || (rope_cos == nullptr) != (rope_sin == nullptr))
    return GR_EINVAL;
  p.q_ld = q_ld; p.rope_cos = rope_cos; p.rope_sin = rope_sin;
  p.B = B; p.H = H; p.Lq = Lq; p.Skv = Skv; p.kv_stride = kv_stride;
  p.causal = causal; p.q_pos0 = q_pos0;
  p.scale_log2 = scale * 1.44269504088896340736f;
  dim3 grid(B * H, gr_cdiv(Lq, 128));
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attention_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * GR_SPW) != hipSuccess ||
        hipFuncSetAttribute((const void*)attention_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * GR_SPW) != hipSuccess)
      return GR_EINVAL;
    attr_set = true;
  }
  if (head_dim == 128) hipLaunchKernelGGL(attention_kernel<128>, grid, dim3(256), 65536 * GR_SPW, stream, p);
  else if (head_dim == 64) hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 32768 * GR_SPW, stream, p);
  else return GR_EINVAL;
  GR_CHECK_LAUNCH();
  return GR_OK;
}
