// OCP e4m3 build of the matrix-unit weight stream for decode steps of 9..64 rows (gemm_skinny.hip: same loads, same x image, four
// v_mfma_scale_f32_16x16x128_f8f6f4 per slice instead of eight 16x16x32, dequantisation scales in the epilogue): an fp8 = True model's
// decode steps past 8 rows read 6.6 GB of weight bytes per tick instead of falling back to the prefill kernels.
#define SK_FP8 1
#include "gemm_skinny.hip"
