// The 256x256 ping-pong bf16 GEMM with the fused-QKV epilogue (gr_gemm_desc.act == 4): same main loop, its own kernel.
#define G256_QKV 1
#include "gemm_bf16_256.hip"
