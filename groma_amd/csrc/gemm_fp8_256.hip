// OCP fp8 (e4m3) build of the 256x256 ping-pong GEMM (BASELINE.json configs[4]): same schedule, 1-byte operands,
// two v_mfma_f32_16x16x32_fp8_fp8 per 16-B fragment chunk, dequantisation scales in the epilogue.
#define G256_FP8 1
#include "gemm_bf16_256.hip"
