// Diagnostic (not part of the product library): what does ONE other instruction cost when it sits between two MFMAs of a
// wave that has its SIMD to itself (the 512-register one-wave-per-SIMD GEMM outline)?  For each filler kind: clocks per
// MFMA of a long stream "MFMA, filler, MFMA, filler, ..." (different accumulators), 256 blocks x 256 threads, one block
// per CU (128 KB of LDS), against the bare stream.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC filler_price.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SB __builtin_amdgcn_sched_barrier(0);

// filler kinds
enum { F_NONE, F_DSREAD, F_DSWRITE, F_GLOAD, F_DMA, F_VALU, F_SNOP, F_DSREAD_VALU, F_GLOAD_ADDR, F_WAIT_DSWRITE, F_MIX_RRWG,
       F_SALU, F_DSREAD2, F_GLOAD_SADDR, F_DMA_SALU_ALT, F_GLOAD_Q, F_DSWRITE_Q, F_DMA_Q, F_MIX_RRWD, F_MIX_RRW_, F_COUNT };

template <int KIND>
__device__ __forceinline__ void filler(int slot, char* lds, const char* g, unsigned& vaddr, u32x4 (&r)[4], int& sacc,
                                       unsigned lane_off) {
  const int q = slot & 3;
  if (KIND == F_DSREAD || KIND == F_DSREAD_VALU || KIND == F_DSREAD2) {
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r[q]) : "v"(vaddr) : "memory");
    if (KIND == F_DSREAD2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(r[(q + 1) & 3]) : "v"(vaddr) : "memory");
    if (KIND == F_DSREAD_VALU) asm volatile("v_add_u32 %0, %0, 0" : "+v"(vaddr));
  } else if (KIND == F_DSWRITE) {
    asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[q]) : "memory");
  } else if (KIND == F_WAIT_DSWRITE) {
    asm volatile("s_waitcnt vmcnt(15)\n\tds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[q]) : "memory");
  } else if (KIND == F_GLOAD) {
    const char* p = g + lane_off;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(p) : "memory");
  } else if (KIND == F_GLOAD_ADDR) {
    const char* p = g + lane_off + (long)(slot & 7) * 4096;
    asm volatile("" : "+v"(p));  // keep the 64-bit address computation in this gap
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(p) : "memory");
  } else if (KIND == F_GLOAD_SADDR) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[q]) : "v"(lane_off), "s"(g) : "memory");
  } else if (KIND == F_DMA) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + lane_off),
                                     (void __attribute__((address_space(3)))*)(lds + 16384 + (slot & 7) * 1024), 16, 0, 0);
  } else if (KIND == F_DMA_SALU_ALT) {
    if (slot & 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
    else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + lane_off),
                                          (void __attribute__((address_space(3)))*)(lds + 16384), 16, 0, 0);
  } else if (KIND == F_VALU) {
    asm volatile("v_add_u32 %0, %0, 0" : "+v"(vaddr));
  } else if (KIND == F_SALU) {
    asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
  } else if (KIND == F_SNOP) {
    asm volatile("s_nop 0");
  } else if (KIND == F_GLOAD_Q) {
    if (q == 3) { const char* p = g + lane_off; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[3]) : "v"(p) : "memory"); }
  } else if (KIND == F_DSWRITE_Q) {
    if (q == 3) asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[q]) : "memory");
  } else if (KIND == F_DMA_Q) {
    if (q == 3) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + lane_off),
                                                 (void __attribute__((address_space(3)))*)(lds + 16384), 16, 0, 0);
  } else if (KIND == F_MIX_RRWD) {
    if (q == 0 || q == 1) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r[q]) : "v"(vaddr) : "memory");
    else if (q == 2) asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[2]) : "memory");
    else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + lane_off),
                                          (void __attribute__((address_space(3)))*)(lds + 16384), 16, 0, 0);
  } else if (KIND == F_MIX_RRW_) {
    if (q == 0 || q == 1) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r[q]) : "v"(vaddr) : "memory");
    else if (q == 2) asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[2]) : "memory");
  } else if (KIND == F_MIX_RRWG) {
    if (q == 0 || q == 1) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(r[q]) : "v"(vaddr) : "memory");
    else if (q == 2) asm volatile("s_waitcnt vmcnt(15)\n\tds_write_b128 %0, %1 offset:8192" ::"v"(vaddr), "v"(r[2]) : "memory");
    else {
      const char* p = g + lane_off;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[3]) : "v"(p) : "memory");
    }
  }
}

template <int SHAPE, int KIND>
__global__ __launch_bounds__(256, 1) void price_loop(int iters, const char* g, float* sink, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (threadIdx.x * 8 + i) * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    a[i] = (short)((h & 0x807f) | 0x3e00);
    h *= 0x9e3779b1u; h ^= h >> 16;
    b[i] = (short)((h & 0x807f) | 0x3e00);
  }
  unsigned vaddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;  // lane-linear, conflict-free
  const unsigned lane_off = threadIdx.x * 16 + (blockIdx.x & 63) * 4096;    // L2-resident 256 KB window
  u32x4 r[4];
  for (int i = 0; i < 4; ++i) r[i] = (u32x4){1u, 2u, 3u, 4u};
  int sacc = 0;
  unsigned long long c0 = 0, w0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
  float res = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        SB filler<KIND>(i, lds, g, vaddr, r, sacc, lane_off); SB
      }
    }
    for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
        SB filler<KIND>(u, lds, g, vaddr, r, sacc, lane_off); SB
      }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) res += acc[i][e];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  res += (float)(r[0][0] + r[1][1] + r[2][2] + r[3][3] + vaddr + sacc);
  if (res == 123.456f) sink[0] = res;
}

template <int SHAPE>
static void launch(int kind, int iters, const char* g, float* sink, unsigned long long* clk, hipStream_t s) {
#define CASE(K)                                                                                              \
  case K: {                                                                                                  \
    (void)hipFuncSetAttribute((const void*)price_loop<SHAPE, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    hipLaunchKernelGGL((price_loop<SHAPE, K>), dim3(256), dim3(256), 131072, s, iters, g, sink, clk);         \
  } break;
  switch (kind) {
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19)
    default: break;
  }
}

// MFMAs per wave per launch
extern "C" long diag_filler_price(int shape, int kind, int iters, const char* g, float* sink, unsigned long long* clk, hipStream_t s) {
  if (shape == 16) launch<16>(kind, iters, g, sink, clk, s);
  else launch<32>(kind, iters, g, sink, clk, s);
  return (long)iters * (shape == 16 ? 16 : 8);
}
