// Shared device helpers for the groma_amd HIP kernels (gfx950 / CDNA4 only).
// No torch types anywhere in csrc/: raw pointers + sizes + hipStream_t.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GR_OK 0
#define GR_EINVAL 22

// The 16-bit operand type of the library is a BUILD choice: libgroma_hip.so carries bfloat16 (the reference's training /
// benchmark dtype), libgroma_hip_f16.so -- the same sources compiled with -DGR_F16, the same C ABI -- carries IEEE half, the
// dtype the reference's own inference entry points autocast to (R: groma/eval/run_groma.py:82, serve/model_worker.py:256).
// MFMA runs both at the same rate; half has 3 more mantissa bits.  Every kernel converts through the three helpers below and
// multiplies through GR_MFMA_16x16x32, so `bf16_t` reads "the library's 16-bit storage type" everywhere else.
#ifdef GR_F16
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2_hw;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8_hw;
// f32 -> f16 round-to-nearest-even (v_cvt_f16_f32), saturating at +-65504 instead of producing inf (one v_med3_f32)
__device__ __forceinline__ float sat_h16(float f) { return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)sat_h16(f)); }
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const h16x2_hw v = {(_Float16)sat_h16(a), (_Float16)sat_h16(b)};
  return __builtin_bit_cast(uint32_t, v);
}
// for values known to lie in [0, 1] (soft-max probabilities): no saturation needed
__device__ __forceinline__ uint32_t pack2bf_unit(float a, float b) {
  const h16x2_hw v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, v);
}
#define GR_MFMA_16x16x32(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8_hw, a), __builtin_bit_cast(h16x8_hw, b), c, 0, 0, 0)
// c + a.lo * b.lo + a.hi * b.hi on two packed 16-bit pairs (v_dot2c_f32_f16): the VALU dot product of the streaming GEMV
#define GR_DOT2(a_u32, b_u32, c) __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_hw, a_u32), __builtin_bit_cast(h16x2_hw, b_u32), c, false)
#else
// f32 -> bf16, round-to-nearest-even (same rule as torch .to(bfloat16)): the __bf16 casts lower to ONE
// v_cvt_pk_bf16_f32 per pair on gfx950 (the bit-twiddling form costs ~7 VALU ops per element).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const bf16x2_hw v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack2bf_unit(float a, float b) { return pack2bf(a, b); }
#define GR_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define GR_DOT2(a_u32, b_u32, c) __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, a_u32), __builtin_bit_cast(bf16x2_hw, b_u32), c, false)
#endif

// ---- split-operand storage: the reference-precision build (libgroma_hip_ref.so, -DGR_F16 -DGR_SPLIT) ------------------------
// A third build of the same sources whose "16-bit operand" is a PAIR of halves: x ~= hi + lo with hi = f16(x), lo = f16(x - hi),
// 22 mantissa bits instead of 11 (bf16: 8).  Every contraction multiplies hi.hi + hi.lo + lo.hi into the same fp32 MFMA
// accumulators (the dropped lo.lo term is 2^-22 relative), so a GEMM costs 3x the MFMA work and 2x the operand bytes and is
// within ~1e-6 of an fp32 GEMM -- what north_star's "logits within 1e-3 of reference" needs at 32 layers of depth, where one
// 16-bit rounding per operand already costs 3e-3 (fp16) / 2.6e-2 (bf16): DESIGN.md 4.
// Layout: a logical tensor of n 16-bit elements is stored as 2n, interleaved in blocks of 32 -- logical element i sits at
// physical element (i / 32) * 64 + i % 32 (hi) and 32 further (lo).  One 64-element physical block is exactly one 128-B K-tile
// row of the GEMM kernels: its first MFMA k-step (32 deep) reads hi, the second lo, so the three products are three MFMAs on
// fragments the unsplit kernel already holds.  Every innermost extent on the path is a multiple of 32 (K % 64 == 0 is an ABI
// requirement, head dims are 64 / 128, KV capacities multiples of 64), so the map is a function of the FLAT logical index.
// All dimension / stride arguments of the C ABI stay LOGICAL in this build; buffers are twice as large.
#ifdef GR_SPLIT
#define GR_SP 1
#else
#define GR_SP 0
#endif
#define GR_SPW (1 + GR_SP)  // physical 16-bit elements per logical element
#if GR_SP
// The MFMA kernels of this build do 3 passes per contraction on twice the operand bytes: they carry their own names, so that a
// trace of a precision="hybrid" model (the ViT from this library, everything else from the bf16 one, in ONE process) keeps the
// two apart -- rocprofv3 --stats aggregates by kernel name, and bench.py's roofline block is per kernel name.
#define gemm_bf16_256_kernel gemm_pair_256_kernel
#define gemm_bf16_kernel gemm_pair_kernel
#define attention_kernel attention_pair_kernel
#define norm_rows_kernel norm_rows_pair_kernel
#define qkv_split_kernel qkv_split_pair_kernel
#endif
__device__ __host__ __forceinline__ long sp_idx(long i) { return GR_SP ? (((i >> 5) << 6) + (i & 31)) : i; }
// (a, b) -> packed hi pair (+ packed lo pair in the split build)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2bf(a, b);
#if GR_SP
  lo = pack2bf(a - bf2f((bf16_t)(hi & 0xffffu)), b - bf2f((bf16_t)(hi >> 16)));
#else
  lo = 0;
#endif
}
// the same for values known to lie in [0, 1] (soft-max probabilities): neither half can leave the half range, so no saturation --
// the general form costs 4 v_med3 + 4 scalar conversions + a pack per pair, and P is split for EVERY score of the pair attention
// (round 5: that split was 2/3 of the kernel's VALU work, which is what bounds it)
__device__ __forceinline__ void split2_unit(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2bf_unit(a, b);
#if GR_SP
  const h16x2_hw h = __builtin_bit_cast(h16x2_hw, hi);
  lo = pack2bf_unit(a - (float)h[0], b - (float)h[1]);
#else
  lo = 0;
#endif
}
__device__ __forceinline__ float ld1f(const bf16_t* base, long i) {
  const long p = sp_idx(i);
  float v = bf2f(base[p]);
#if GR_SP
  v += bf2f(base[p + 32]);
#endif
  return v;
}
__device__ __forceinline__ void st1f(bf16_t* base, long i, float v) {
  const long p = sp_idx(i);
  const bf16_t h = f2bf(v);
  base[p] = h;
#if GR_SP
  base[p + 32] = f2bf(v - bf2f(h));
#endif
}
// 4 consecutive logical elements (i % 4 == 0): one 8-B store (two in the split build)
__device__ __forceinline__ void st4f(bf16_t* base, long i, f32x4 v) {
  const long p = sp_idx(i);
  uint2 h, l;
  split2(v[0], v[1], h.x, l.x);
  split2(v[2], v[3], h.y, l.y);
  *(uint2*)(base + p) = h;
#if GR_SP
  *(uint2*)(base + p + 32) = l;
#endif
}
// 8 consecutive logical elements (i % 8 == 0): one 16-B access (two in the split build)
__device__ __forceinline__ void st8f(bf16_t* base, long i, const float* v) {
  const long p = sp_idx(i);
  union { bf16x8 x; uint32_t u[4]; } h, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) split2(v[2 * k], v[2 * k + 1], h.u[k], l.u[k]);
  *(bf16x8*)(base + p) = h.x;
#if GR_SP
  *(bf16x8*)(base + p + 32) = l.x;
#endif
}
__device__ __forceinline__ void ld8f(const bf16_t* base, long i, float* o) {
  const long p = sp_idx(i);
  const bf16x8 h = *(const bf16x8*)(base + p);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = bf2f((bf16_t)h[k]);
#if GR_SP
  const bf16x8 l = *(const bf16x8*)(base + p + 32);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] += bf2f((bf16_t)l[k]);
#endif
}

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-roundoff class): ~14 VALU ops instead of the
// ~60 of ocml erff -- the exact-erf GELU (HF "gelu") epilogue of the ViT fc1 / bridge GEMMs is VALU-visible otherwise.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);  // v_rcp_f32: 1 ulp, plenty under the 1.5e-7 bound
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return x < 0.f ? -r : r;
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float silu_f(float x) {  // x * sigmoid(x); v_exp + v_rcp (~2 ulp), bf16 output downstream
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Reductions across the four 16-lane rows of a wave (lanes that differ in bits 4 and 5) inside the VALU: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange register halves, where __shfl_xor(x, 16 | 32) compiles to ds_bpermute_b32 (an
// LDS-crossbar round trip).  With both operands = x the swap leaves (x[lane], x[lane ^ 16|32]) in the two results, in either
// order, so a commutative op gives the same bits as the shuffle form.
// Inline asm on purpose: with hipcc (ROCm 7.2) __builtin_amdgcn_permlane{16,32}_swap returns its FIRST result for both elements
// of the result pair (the ISA read v_add v, v0, v0 / dropped the max altogether), which silently breaks every reduction.
__device__ __forceinline__ void xswap16(float x, float& a, float& b) {
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // rows 1,3 of a <-> rows 0,2 of b
}
__device__ __forceinline__ void xswap32(float x, float& a, float& b) {
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));  // lanes 32..63 of a <-> lanes 0..31 of b
}
__device__ __forceinline__ float rows_max(float x) {
  float a, b;
  xswap16(x, a, b);
  x = fmaxf(a, b);
  xswap32(x, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float rows_sum(float x) {
  float a, b;
  xswap16(x, a, b);
  x = a + b;
  xswap32(x, a, b);
  return a + b;
}

// Wave-wide all-reduce: inside a 16-lane row by DPP rotations (row_ror 8, 4, 2, 1: VALU data path), across the four rows by the
// permlane swaps above.  (__shfl_xor compiles to ds_bpermute_b32 for every distance on this compiler -- six dependent LDS-crossbar
// round trips per reduction, which is most of the latency of the small decode / DDETR kernels.)
// PRECONDITION of wave_sum / wave_max / rows_sum / rows_max: ALL 64 lanes of the wave are active at the call (no divergent
// early exit, no partial tail wave).  bound_ctrl makes an inactive source lane read as 0 -- not the neutral element of max --
// and the permlane swaps do not exchange with inactive lanes.  Every caller in csrc/ satisfies it by construction: one wave
// per row with a wave-uniform `if (row >= rows) return;` (norm.hip, fp8.hip), full 1024-thread blocks (decode.hip), full
// waves per weight-row group (gemv_bf16.hip), whole-wave query tiles (attention.hip, ddetr.hip).  Lanes without data must
// contribute the neutral element themselves (0 for sums, -inf / -1e30 for maxima) instead of skipping the call.
template <int N>
__device__ __forceinline__ float dpp_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_ror<8>(v);
  v += dpp_ror<4>(v);
  v += dpp_ror<2>(v);
  v += dpp_ror<1>(v);
  return rows_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_ror<8>(v));
  v = fmaxf(v, dpp_ror<4>(v));
  v = fmaxf(v, dpp_ror<2>(v));
  v = fmaxf(v, dpp_ror<1>(v));
  return rows_max(v);
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold 16 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

static inline int gr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define GR_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

// ---- OCP e4m3 packing (v_cvt_pk_fp8_f32: round-to-nearest-even).  The conversion does not saturate, so a value quantised with a
// STATIC scale (the conv activations of the e4m3 path) is clamped to the largest finite e4m3 magnitude first; per-row dynamic
// scales (fp8.hip) map the row maximum to 448 and need no clamp.
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}
__device__ __forceinline__ float sat448(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
// 8 fp32 values * inv (clamped) -> 8 e4m3 bytes at q[idx .. idx + 7]
__device__ __forceinline__ void st8q(uint8_t* q, long idx, const float* o, float inv) {
  uint2 w;
  w.x = pack4_fp8(sat448(o[0] * inv), sat448(o[1] * inv), sat448(o[2] * inv), sat448(o[3] * inv));
  w.y = pack4_fp8(sat448(o[4] * inv), sat448(o[5] * inv), sat448(o[6] * inv), sat448(o[7] * inv));
  *(uint2*)(q + idx) = w;
}
