/*
 * groma_hip.h -- C ABI of libgroma_hip.so, the MI355X (gfx950) operator library behind
 * groma_amd.GromaModel (the drop-in for reference groma/model/groma.py:86-427).
 *
 * Boundary rules (inner boundary = the reference's mmcv `_ext` op plugin,
 * mmcv/ops/csrc/pytorch/pybind.cpp:175,191,596,611):
 *   - extern "C", plain device pointers + sizes + hipStream_t; no torch / ATen types;
 *   - the caller owns every buffer; ops never allocate, never synchronise, never throw;
 *   - work is enqueued on the given stream (the reference ops use the current CUDA stream:
 *     mmcv/ops/csrc/pytorch/cuda/roi_align_cuda.cu:15-16);
 *   - return 0 on success, GR_EINVAL (22) for a rejected argument set, or the hipError_t of a failed
 *     launch (the reference raises RuntimeError via TORCH_CHECK; the Python host maps non-zero to
 *     RuntimeError the same way).
 * bf16 tensors are raw uint16 bit patterns; "f32" = IEEE binary32.  All row-major.
 *
 * Two builds of this ABI exist, from the same sources: libgroma_hip.so, whose 16-bit operand type is bfloat16, and
 * libgroma_hip_f16.so (-DGR_F16), whose 16-bit operand type is IEEE binary16 -- the dtype the reference's own inference entry
 * points autocast to (groma/eval/run_groma.py:82, groma/serve/model_worker.py:256).  In the second build every parameter this
 * header calls "bf16" holds half bit patterns (conversions saturate at +-65504); names and signatures are identical, and
 * gr_operand_type() tells a host which build it has loaded.
 *
 * A third build, libgroma_hip_ref.so (-DGR_F16 -DGR_SPLIT), is the reference-precision path: its "16-bit operand" is a PAIR of
 * halves, x ~= hi + lo with hi = f16(x), lo = f16(x - hi) (22 mantissa bits), and every contraction issues hi.hi + hi.lo + lo.hi
 * into the same fp32 MFMA accumulators -- 3x the MFMA work, within ~1e-6 of an fp32 GEMM, which is what a 32-layer-deep chain
 * needs to stay inside north_star's 1e-3 of the reference's fp32 path (groma/eval/eval_rec.py:69 loads fp32 weights).
 * Storage: a logical 16-bit tensor of n elements occupies 2n; logical element i (flat) sits at physical element
 * (i / 32) * 64 + i % 32 (hi) and 32 elements further (lo).  Every size / stride / index argument of this header stays
 * LOGICAL; only the buffers double.  Innermost extents and row strides of 16-bit tensors must be multiples of 32 there
 * (GR_EINVAL otherwise).  Entry points without a split form (the weight-streaming decode kernels, tile 1 / 2 of gr_gemm_bf16,
 * the e4m3 path) return GR_EINVAL in that build; a decode step runs through the general kernels.
 */
#ifndef GROMA_HIP_H
#define GROMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define GROMA_HIP_ABI_VERSION 9
int gr_abi_version(void);
#define GR_OPERAND_BF16 0
#define GR_OPERAND_F16 1
#define GR_OPERAND_SPLIT 2 /* libgroma_hip_ref.so: (hi, lo) pairs of halves, see below */
/* the 16-bit operand type this build of the library was compiled for */
int gr_operand_type(void);
/* kernel timing hook used by bench.py: when enabled, every gr_gemm_bf16 launch is bracketed by HIP
 * events on its own stream; gr_prof_read drains them (sync) and returns total ms + launch count. */
int gr_prof_enable(int on);
int gr_prof_read(double* total_ms, long* launches, double* flops);
int gr_prof_read_launches(long cap, int* mnk, float* ms, long* n_out);
/* Launch geometry of the ping-pong GEMM for the CALLING THREAD's subsequent gr_gemm_bf16 launches (round 6).  The kernel normally runs
 * as a persistent grid (one workgroup per CU walking the tile list), which holds every CU until the whole GEMM is done; on != 0 switches
 * to one workgroup per tile, so that workgroups of kernels queued on OTHER streams are dispatched whenever a tile retires.  The path
 * sets it around the region pyramid's convolutions (side stream), whose multi-millisecond launches otherwise stall the latency-bound
 * proposer chain on the main stream (groma/model/groma.py:240-280 runs beside roi_align.py:180-193), and the serving loop around
 * an admission prefill that shares the GPU with the decode ticks of live rows (groma_amd/serving.py).  Same tiles, same bits. */
int gr_gemm_yield(int on);

/* ------------------------------------------------------------------ dense contractions (MFMA) -- */
/* C[M,N] = epilogue(A[M,K] . W[N,K]^T).  Replaces every nn.Linear / nn.Conv2d on the path:
 * HF Dinov2 / LLaMA projections, img_txt_bridge (groma/model/groma.py:112-116), lm_head (+) extra_lm_head
 * (groma.py:399-402), MLVLFuseModule / MlvlRoIExtractor convs (groma/model/roi_align.py:128-143,251-264). */
typedef struct gr_gemm_desc {
  const void* A;      /* bf16 [M,K] (lda), or zero-bordered NHWC maps when conv_C > 0              */
  const void* W;      /* bf16 [N,K] (ldw) -- nn.Linear.weight layout                               */
  void* C;            /* bf16 or f32 [M,N] (ldc)                                                   */
  const float* bias;  /* [N] or NULL                                                               */
  const float* scale; /* [N] or NULL (DINOv2 LayerScale), applied after bias/activation            */
  const float* resid; /* f32 [*,N] (ldr) or NULL, added last                                       */
  float* ws;          /* split-K workspace f32 [splits,M,N] (required when splits > 1)             */
  int M, N, K;        /* K % 64 == 0, N % 4 == 0                                                   */
  long lda, ldw, ldc, ldr;
  int act;            /* 0 none, 1 GELU(erf), 2 ReLU, 3 SwiGLU over interleaved (gate,up) rows of W:
                         C is [M, N/2] bf16 = silu(gate)*up */
  int out_f32;        /* C element type                                                            */
  int splits;         /* split-K factor >= 1                                                       */
  /* implicit 3x3 / pad 1 convolution gather for A (conv_C > 0): row m = output pixel (img,y,x) of
   * [imgs,conv_H,conv_W]; A = [imgs,conv_H+2,conv_W+2,conv_C] bf16 with a zero border; K = S*9*conv_C with
   * k = ((s*9 + ky*3+kx)*conv_C + c); segment s (summed source maps) is conv_seg_stride elements apart. */
  int conv_H, conv_W, conv_C;
  long conv_seg_stride;
  int resid_mod;      /* > 0: residual row = m % resid_mod (position-embedding broadcast)          */
  /* output row remap: row(m) = (m / c_group)*c_group_stride + c_row_off + m % c_group (c_group > 0) */
  int c_group, c_group_stride, c_row_off;
#define GR_TILE_PP192 192 /* ping-pong kernel with 192-row tiles */
  int tile;           /* 0 = choose per shape (all MFMA kernels produce the same bits); 128 / 256 force the 128x128 / 256x256
                         kernel, GR_TILE_PP192 the 192-row form of the latter; 1 = skinny decode
                         kernel (M <= 8; requires splits == ceil(K/512) and ws); 2 = the same kernel but
                         the split-K partials are LEFT in ws [splits, M, N] f32 for the caller
                         (round 1-3's decode step; the step now runs on gr_gemv_fused): C and the epilogue fields are unused;
                         3 = the weight stream on the matrix unit for decode steps of 9..64 rows (round 6, csrc/gemm_skinny.hip:
                         M <= 64, K % 32 == 0, plain row-major A, no split / conv / row remap / scale; epilogues bias, act 0-3,
                         f32 or 16-bit out, fp32 residual; a row's result is independent of M; GR_EINVAL in the operand-pair build);
                         any other value: GR_EINVAL */
  /* OCP fp8 (e4m3) operands (BASELINE configs[4]): A, W are 1-byte elements, K % 128 == 0 (conv gather: conv_C % 128 == 0);
   * the result is dequantised as acc * a_scale[m] * w_scale[n] before the rest of the epilogue */
  int fp8;
  const float* a_scale; /* [M] per-row scale of A, or NULL (= 1)                                     */
  const float* w_scale; /* [N] per-output-channel scale of W (required when fp8)                     */
  /* tile 1/2 only: A is the not-yet-merged output of gr_decode_attention(nsplit > 1): f32
   * [M * K/a_hd heads][a_nsplit][a_hd + 2] = (un-normalised o, running max, running sum) per key slice; the kernel
   * merges the slices (in slice order) while loading its x operand and rounds to bf16 exactly as the nsplit = 1 path */
  const float* a_parts;
  int a_nsplit, a_hd;
} gr_gemm_desc;
int gr_gemm_bf16(const gr_gemm_desc* d, hipStream_t stream);

/* Decode-step weight streaming with fused prologue / epilogue (gemv_fused.hip): y[M <= 8, N] = x[M,K] . W[N,K]^T where a
 * workgroup owns whole rows of W (no split-K partials leave the kernel) -- 5 launches per LLaMA layer at L = 1
 * (R: groma/model/groma.py:376-379 -> HF LlamaDecoderLayer).  K % 64 == 0.
 *   x_mode 0: A bf16 [M,K] (lda).
 *   x_mode 1: x = bf16(gamma * h * rsqrt(mean(h^2) + eps)) from the f32 residual rows h [M,K] (ldh): HF LlamaRMSNorm fused.
 *   x_mode 2: x = the merge of gr_decode_attention's key slices a_parts (see gr_gemm_desc.a_parts), rounded to bf16.
 *   epi 0: C f32 [M,N] (ldc) = y.                 epi 1: resid f32 [M,N] (ldr) += y (in place residual update).
 *   epi 2: C bf16 [M, N/2] (ldc) = silu(y[2j]) * y[2j+1] (SwiGLU over interleaved (gate, up) rows of W).
 *   epi 3: W = fused QKV [3*H*HD, K]: y rounded to bf16, HF rotate_half RoPE at position pos (cosT / sinT [pos, HD/2], or
 *          NULL), then q -> q [M,H,HD], k -> kc [M,H,kv_stride,HD] row pos, v -> vt [M,H,HD,kv_stride] column pos;
 *          pos = pos_dev ? pos_dev[m * pos_stride] : pos0 (device-resident for hipGraph replay / ragged rows). */
typedef struct gr_gemv_desc {
  const void* W; long ldw; int M, N, K;
  int x_mode; const void* A; long lda; const float* h; long ldh; const float* gamma; float eps;
  const float* a_parts; int a_nsplit, a_hd;
  int epi; void* C; long ldc; float* resid; long ldr;
  void* q; void* kc; void* vt; const float* cosT; const float* sinT; int H, HD, pos0, kv_stride; const int* pos_dev; int pos_stride;
  /* ABI 9 -- OCP e4m3 weights (BASELINE configs[4] at L = 1): w8 != 0: W is [N, K] e4m3 BYTES (ldw in elements) with the per-row
   * scale w_scale [N]; the operand is quantised by the prologue exactly as the e4m3 prefill path forms it (per row, s = max|x| / 448,
   * q = e4m3_rne(x * (1 / s)): x_mode 1 from the fp32 normalisation output, x_mode 0 / 2 from the 16-bit-rounded activation) and
   * y = acc * w_scale[n] * s[m] before the epilogue.  Limits (GR_EINVAL beyond; engine.LlamaEngine.forward `fits8` checks them and
   * falls back to the general kernels): the staged operand MB * (K + 16) bytes (MB = M rounded up to 4 / 8; + MB * K * 2 bytes for
   * x_mode 2, whose merged context is held 16-bit before it is quantised) <= 128 KB; K % 128 == 0; K <= 12288 for x_mode 0 / 2 (the
   * register-resident row quantiser) and K <= 4096 for x_mode 1; N % 16 == 0; HD % 32 == 0 for epi 3. */
  int w8; const float* w_scale;
} gr_gemv_desc;
int gr_gemv_fused(const gr_gemv_desc* d, hipStream_t stream);

/* exact fp32 GEMM (f32-input MFMA): C = act(A.W^T + bias) (+ resid); K % 16 == 0.  DDETR linears
 * (HF 4.32 DeformableDetr* layers used from groma/model/ddetr_transformer.py:299-359). act: 0 | 2 (ReLU) */
int gr_gemm_f32(const float* A, const float* W, float* C, const float* bias, const float* resid, int M, int N, int K,
                long lda, long ldw, long ldc, int act, hipStream_t stream);

/* ------------------------------------------------------------------------------ normalisation -- */
/* out = LN(x (+ add)) * gamma + beta over C (C % 4 == 0; register-resident kernel for C = 256 * 2^k <= 4096, a streaming
 * kernel for every other width); out bf16 or f32. */
int gr_layernorm(const float* x, const float* add, const float* gamma, const float* beta, void* out, int rows, int C,
                 long ldx, long ldo, float eps, int out_bf16, int relu_in, hipStream_t stream);
/* fp8 row quantisation: q = e4m3(x / s), s[m] = max|x[m,:]| / 448 ; and the fused norm -> fp8 variant */
int gr_quant_rows_fp8(const void* x, int x_is_f32, void* q, float* scale, int rows, int K, long ldx, hipStream_t stream);
int gr_norm_fp8(const float* x, const float* gamma, const float* beta, void* q, float* scale, int rows, int C, float eps,
                int rms, hipStream_t stream);
/* HF LlamaRMSNorm */
int gr_rmsnorm(const float* x, const float* gamma, void* out, int rows, int C, long ldx, long ldo, float eps,
               int out_bf16, hipStream_t stream);

/* ---------------------------------------------------------------------------------- attention -- */
/* q [B,H,Lq,hd], k [B,H,kv_stride,hd], vt [B,H,hd,kv_stride] bf16 -> out [B*Lq, H*hd] bf16.
 * key j visible to query i <=> j < Skv, j < kv_len[b] (if given), and (causal) j <= q_pos0 + i.  hd in {64,128}.
 * pos_dev (optional, device): q_pos0 of batch row b = pos_dev[b * pos_stride] and Skv = q_pos0 + Lq -- the step position
 * lives in device memory so a decode step can be captured once in a hipGraph and replayed (pos_stride 0 = one shared
 * counter, 1 = ragged per-row positions); Skv is then only the capacity bound that is validated.
 * q_ld > 0: q points at the fused projection buffer [B*Lq, q_ld] (row b*Lq+i, columns h*hd + d) and is read in place;
 * with rope_cos/rope_sin [pos, hd/2] HF rotate_half RoPE at position q_pos0 + i is applied while loading (the q copy
 * and its round trip that gr_qkv_split would make disappear: pass q = NULL there). */
int gr_attention_bf16(const void* q, const void* k, const void* vt, void* out, const int* kv_len, int B, int H, int Lq,
                      int Skv, int kv_stride, int head_dim, int causal, int q_pos0, float scale, const int* pos_dev,
                      int pos_stride, long q_ld, const float* rope_cos, const float* rope_sin, hipStream_t stream);
/* fused-QKV split (+ HF rotate_half RoPE when cos/sin given) into the layouts above / the KV cache */
int gr_qkv_split(const void* qkv, void* q, void* k, void* vt, const float* cosT, const float* sinT, int B, int H, int L,
                 int head_dim, int pos0, int kv_stride, const int* pos_dev, int pos_stride, hipStream_t stream);

/* ------------------------------------------------------------------------------ decode step -- */
/* One new token per row (groma/model/groma.py:376-379 + HF LlamaDecoderLayer at L = 1): gr_gemv_fused (above) streams each
 * weight matrix with its producer and consumer fused in; between the o-proj and the QKV stream sits single-query attention:
 *   gr_decode_attention  : out[b, h*hd + d] = softmax(q.K^T * scale)[0 .. pos] . V   (Smax <= 8192 = LDS score buffer)
 *                          nsplit > 1 cuts every row's keys into nsplit slices (one block each, so B*H*nsplit blocks
 *                          fill the chip at small B) and writes parts f32 [B*H][nsplit][hd+2] instead of out; the
 *                          o-proj stream merges them while building its operand (gr_gemv_desc.a_parts, x_mode 2) -- no
 *                          cross-block hand-off inside the launch, hence no device-scope fence (an L2 write-back on a
 *                          multi-XCD part)
 * Position of row b = pos_dev ? pos_dev[b * pos_stride] : q_pos0 (see gr_attention_bf16). */
int gr_decode_attention(const void* q, const void* k, const void* vt, void* out, const int* kv_len, int B, int H, int Smax,
                        int kv_stride, int head_dim, int q_pos0, float scale, const int* pos_dev, int pos_stride,
                        int nsplit, float* parts, hipStream_t stream);

/* ---------------------------------------------------------------------- image preprocessing -- */
/* PIL Image.resize (BICUBIC, 8-bit fixed point, two passes; groma/eval/run_groma.py:79) on interleaved uint8 RGB, with
 * Pillow's host-computed window table bounds [out,2] = (first index, count) and 22-bit fixed-point coef [out,ksize];
 * the vertical pass optionally emits the HF processor's rescale+normalise through lut f32 [3,256] as f32 [3,Hout,W]. */
int gr_resize_h_u8(const void* in, void* out, const int* bounds, const int* coef, int H, int Win, int Wout, int ksize,
                   hipStream_t stream);
int gr_resize_v_norm(const void* in, void* out_u8, float* out_f32, const int* bounds, const int* coef, const float* lut,
                     int Hin, int Hout, int W, int ksize, hipStream_t stream);

/* the eval datasets' mmdet pipeline (groma/data/datasets/refcoco_rec.py:38-65): cv2.resize INTER_LINEAR of the uint8 HWC (BGR)
 * image (mmcv/image/geometric.py:51-101) fused with mmcv.imnormalize(mean, std, to_rgb) (mmcv/image/photometric.py:9-45).
 * xofs [Wout] / xalpha int16 [Wout,2] and yofs [Hout] / ybeta int16 [Hout,2]: OpenCV's per-axis tap index and 11-bit fixed-point
 * coefficients, computed on the host as OpenCV does (groma_amd/preprocess.py); an exact 2x down-scale takes the INTER_AREA fast
 * path and needs no tables.  out_u8 [Hout,Wout,3] (the resized image, optional) and/or out_f32 [3,Hout,Wout] (normalised). */
int gr_cv2_resize_norm(const void* in, int Hin, int Win, const int* xofs, const short* xalpha, const int* yofs,
                       const short* ybeta, void* out_u8, float* out_f32, const double* mean, const double* stdinv, int to_rgb,
                       int Hout, int Wout, hipStream_t stream);

/* ------------------------------------------------------------------------- packing / movement -- */
int gr_patchify(const float* images, void* out, int B, int S, int P, int Kpad, hipStream_t stream);
int gr_fill_rows_f32(const float* src, float* dst, int rows, int C, long ld_dst, hipStream_t stream);
int gr_mean4_tokens(const float* h0, const float* h1, const float* h2, const float* h3, float* out, int B, int T, int C,
                    hipStream_t stream);
int gr_s2d_pack(const float* h, void* out, int B, int G, int C, hipStream_t stream);
int gr_upsample_coord_pack(const float* h, void* out, int B, int G, int Ho, int C, int Cpad, hipStream_t stream);
/* sums: f32 [imgs, gr_gn_stats_blocks(HW), C, 2] partials (no atomics: bit-reproducible) */
int gr_gn_stats_blocks(int HW);
int gr_gn_stats(const void* x, float* sums, int imgs, int HW, int C, hipStream_t stream);
int gr_gn_finalize(const float* sums, const float* gamma, const float* beta, float* coef, int imgs, int HW, int C,
                   int groups, float eps, hipStream_t stream);
int gr_fuse_shuffle(const void* tar, const float* tar_coef, int tarS, const void* top, const float* top_coef, int topS,
                    const void* down, const float* down_coef, int downS, void* out, int imgs, int C, int shuffle, int pad,
                    hipStream_t stream);
/* the same map as OCP e4m3 bytes of value * inv_scale (clamped to +-448): the A operand of an e4m3 implicit-GEMM 3x3 conv
 * (gr_gemm_desc.fp8 with conv_C > 0) whose activation scale 1 / inv_scale is a constant folded into that GEMM's w_scale
 * (BASELINE configs[4] extended to the region encoder's convs; groma/model/roi_align.py:150-193) */
int gr_fuse_shuffle_fp8(const void* tar, const float* tar_coef, int tarS, const void* top, const float* top_coef, int topS,
                        const void* down, const float* down_coef, int downS, void* out, int imgs, int C, int shuffle, int pad,
                        float inv_scale, hipStream_t stream);
int gr_cast_f32_bf16(const float* a, const float* b, void* out, long n, hipStream_t stream);
int gr_add_rows_f32(const float* a, const float* b, float* out, long rows, int C, int b_mod, hipStream_t stream);
int gr_embed_gather(const long* ids, const void* table0, const void* table1, float* out, long n, int C, int V0, int V1,
                    hipStream_t stream);
int gr_scatter_rows_f32(const float* src, const int* row_idx, float* dst, long n, int C, hipStream_t stream);
int gr_argmax_rows(const float* x, long* out, int rows, int V, long ld, hipStream_t stream);
/* the serving loop's sampler (groma/serve/model_worker.py:307-311): inv_temp[row] == 0 (or inv_temp NULL) -> arg-max, else
 * token ~ softmax(logits * inv_temp[row]) by inverse CDF with u = splitmix64(seed[row], pos[row*pos_stride] + pos_off) in
 * [0,1) -- a counter-based draw keyed on the request's seed and the new token's absolute position (independent of batching,
 * replayable in a hipGraph, recomputable on the host). */
int gr_sample_rows(const float* x, long* out, int rows, int V, long ld, const float* inv_temp, const long* seed,
                   const int* pos, int pos_stride, int pos_off, hipStream_t stream);
/* one step of HF 4.32 GenerationMixin.greedy_search bookkeeping on the device (reference: the loop HF runs around
 * groma/model/groma.py:176-200): n = unfinished ? nxt : pad; seq[:, *step] = n; tok = n; unfinished &= n != eos;
 * ++*step; pos[0..pos_rows) += inc_pos; *n_unfinished = sum(unfinished).  eos < 0 = no stopping token.
 * inc_pos == 2 (pos_rows == rows) is the ragged (continuous-batching) form: only rows with unfinished != 0 advance
 * their own position and emit a token; idle rows emit pad and stay put. */
int gr_greedy_advance(const long* nxt, long* tok, long* unfinished, long* seq, int* pos, int* step, int* n_unfinished,
                      int rows, long eos, long pad, int seq_ld, int pos_rows, int inc_pos, hipStream_t stream);

/* --------------------------------------------------------------- region proposer (fp32, DDETR) -- */
int gr_msda_f32(const float* value, const float* offw, const float* ref, float* out, int B, int Q, int heads,
                int n_points, int Hs, int Ws, int ld, int rdim, int ref_batched, hipStream_t stream);
int gr_mha32_f32(const float* qk, const float* v, float* out, int B, int Q, int heads, int ldqk, float scale,
                 hipStream_t stream);
int gr_ddetr_topk_gather(const int* idx, const float* delta, const float* prop, float* ref, float* pos, int B, int S,
                         int Kq, int npf, hipStream_t stream);
int gr_box_refine(const float* tmp, const float* ref, float* out, long n, hipStream_t stream);
int gr_score_fuse(const float* coco, const float* sa1b, float* out, long n, long ld, hipStream_t stream);

/* ------------------------------------------------------------------- selection (index-exact) -- */
/* replaces torch.topk at groma/model/ddetr_transformer.py:556; order (value desc, index asc); S <= 4096
 * (one workgroup per image: LDS bitonic sort, 1024- or 4096-entry capacity) */
int gr_topk_desc(const float* x, int* out_idx, int B, int S, int K, long ldx, hipStream_t stream);
/* replaces mmcv `_ext.nms` + NMSop glue (mmcv/ops/nms.py:14-33, csrc/pytorch/cpu/nms.cpp:5-54) for the BATCHED call at
 * groma/model/groma.py:266-272: boxes are (cx,cy,w,h) (converted with HF center_to_corners_format inside), score
 * threshold applied only when > 0 (nms.py:21), first max_num kept; keep is int64 [B,max_num], -1 padded.
 * n <= 512: one workgroup per image, no workspace (ws may be NULL).  512 < n <= 4096: three launches over a caller-owned
 * workspace of gr_nms_workspace_bytes(B, n) bytes. */
long gr_nms_workspace_bytes(int B, int n);
int gr_nms_f32(const float* boxes_cxcywh, const float* scores, int B, int n, float iou_thr, float score_thr, int max_num,
               const int* n_valid, long* keep, int* n_keep, void* ws, hipStream_t stream);
/* The reference op itself, argument for argument: `Tensor nms(Tensor boxes, Tensor scores, float iou_threshold,
 * int offset)` (mmcv/ops/csrc/pytorch/pybind.cpp:175; CPU semantics csrc/pytorch/cpu/nms.cpp:5-54; called from
 * NMSop.forward, mmcv/ops/nms.py:26-27).  boxes f32 [n,4] = (x1,y1,x2,y2), scores f32 [n], offset in {0,1}.
 * The returned Tensor becomes two caller-owned outputs: keep int64 [n] receives the kept indices (into the input, by
 * descending score, ties by ascending index; entries past *n_keep are -1) and n_keep (device int32) their count.
 * n == 0 is legal (count 0).  ws as for gr_nms_f32 with B = 1. */
int gr_nms(const float* boxes_xyxy, const float* scores, int n, float iou_threshold, int offset, long* keep, int* n_keep,
           void* ws, hipStream_t stream);

/* ---------------------------------------------------------------------- fused RoIAlign + pack -- */
/* replaces mmcv `_ext.roi_align_forward` (mmcv/ops/roi_align.py:93-104; kernel
 * csrc/common/cuda/roi_align_cuda_kernel.cuh:17-108) for the calls at groma/model/roi_align.py:299-305.
 * feat bf16 NHWC [imgs,H,W,C]; rois f32 [R,5] = (img, x1,y1,x2,y2) exactly as the reference passes them;
 * out [R, PH+2*pad, PW+2*pad, C] bf16 (or f32), interior written, border left untouched (zero it once). */
int gr_roi_align_pack(const void* feat_nhwc, const float* rois, void* out, int R, int C, int H, int W, int pooled_h,
                      int pooled_w, float spatial_scale, int sampling_ratio, int aligned, int pad, int out_f32,
                      hipStream_t stream);
/* the same tiles as OCP e4m3 bytes of value * inv_scale (clamped to +-448): the A operand of the e4m3 per-ROI conv
 * (groma/model/roi_align.py:312-315) */
int gr_roi_align_pack_fp8(const void* feat_nhwc, const float* rois, void* out, int R, int C, int H, int W, int pooled_h,
                          int pooled_w, float spatial_scale, int sampling_ratio, int aligned, int pad, float inv_scale,
                          hipStream_t stream);

/* The reference op itself, argument for argument: `void roi_align_forward(Tensor input, Tensor rois, Tensor output,
 * Tensor argmax_y, Tensor argmax_x, int aligned_height, int aligned_width, float spatial_scale, int sampling_ratio,
 * int pool_mode, bool aligned)` (mmcv/ops/csrc/pytorch/pybind.cpp:596; caller mmcv/ops/roi_align.py:93-104; arithmetic
 * of csrc/common/cuda/roi_align_cuda_kernel.cuh:17-108).  Tensors become pointers + the sizes ATen would carry:
 * input f32 NCHW [N,C,H,W], rois f32 [K,5] = (batch, x1,y1,x2,y2), output f32 [K,C,aligned_height,aligned_width]
 * (caller-allocated; every element is written), argmax_y/argmax_x f32 like output -- written when pool_mode == 0 (max),
 * ignored (may be NULL) when pool_mode == 1 (avg).  Same numbers as gr_roi_align_pack; this entry keeps the reference's
 * NCHW fp32 layout for an mmcv-side binding, the packed NHWC bf16 entry above is the one the hot path uses. */
int gr_roi_align_forward(const float* input, const float* rois, float* output, float* argmax_y, float* argmax_x, int K,
                         int C, int H, int W, int aligned_height, int aligned_width, float spatial_scale,
                         int sampling_ratio, int pool_mode, int aligned, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GROMA_HIP_H */
