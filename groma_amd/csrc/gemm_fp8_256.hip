// OCP fp8 (e4m3) build of the 256x256 ping-pong GEMM (BASELINE.json configs[4]): same schedule, 1-byte operands, 128-deep
// K-tiles, one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per MFMA tile and K-tile, dequantisation scales in the
// epilogue.  (Rounds 1-3 used v_mfma_f32_16x16x32_fp8_fp8, which issues at the bf16 rate: 1.6 PF in the model against 2.2-2.3 PF
// with the K = 128 instruction, profiles/r04_gemm_shapes_fp8_b14.txt.)
#define G256_FP8 1
#include "gemm_bf16_256.hip"
