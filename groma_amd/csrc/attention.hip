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
// Block = 4 waves, 64 query rows (16 per wave), KV tile 64.  Scores are computed
// transposed, S^T = K.Q^T (v_mfma_f32_16x16x32_bf16, K rows as the A operand) so that
// a lane owns ONE query column: the row max/sum are in-lane + 2 shuffles, and P^T feeds
// the second MFMA (O^T = Vt.P^T) straight from registers -- no LDS round trip for P.
// K / Vt tiles are staged with global_load_lds into an XOR-swizzled, conflict-free image.
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
  int B, H, Lq, Skv, kv_stride;
  int causal, q_pos0;
  float scale_log2;  // softmax scale * log2(e)
};

template <int HD>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
  constexpr int KV = 64;
  constexpr int KROW = HD * 2;        // bytes per K row
  constexpr int KCH = KROW / 16;      // 16-B chunks per K row (8 or 16)
  constexpr int KTILE = KV * KROW;    // bytes
  constexpr int VTILE = HD * KV * 2;  // Vt tile [HD][64] bf16, 128-B rows
  __shared__ __attribute__((aligned(16))) char smem[KTILE + VTILE];
  char* ksm = smem;
  char* vsm = smem + KTILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 64;

  const bf16_t* Qp = p.q + (long)bh * p.Lq * HD;
  const bf16_t* Kp = p.k + (long)bh * p.kv_stride * HD;
  const bf16_t* Vp = p.vt + (long)bh * HD * p.kv_stride;

  // this lane's query row (B operand column) -- clamp the tail
  int qi = q0 + wave * 16 + fr;
  const bool q_valid = qi < p.Lq;
  if (!q_valid) qi = p.Lq - 1;
  bf16x8 qf[HD / 32];
#pragma unroll
  for (int kk = 0; kk < HD / 32; ++kk) qf[kk] = *(const bf16x8*)(Qp + (long)qi * HD + kk * 32 + fg * 8);

  int limit = p.Skv;  // keys [0, limit) visible to this lane's query
  if (p.kv_len) limit = min(limit, p.kv_len[b]);
  if (p.causal) limit = min(limit, p.q_pos0 + qi + 1);
  // block-level loop bound: max over the block's queries
  int blk_limit = p.Skv;
  if (p.kv_len) blk_limit = min(blk_limit, p.kv_len[b]);
  if (p.causal) blk_limit = min(blk_limit, p.q_pos0 + min(q0 + 63, p.Lq - 1) + 1);
  const int ntiles = (blk_limit + KV - 1) / KV;

  f32x4 o[HD / 16];
#pragma unroll
  for (int n = 0; n < HD / 16; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * KV;
    __syncthreads();  // previous tile fully consumed
    // ---- stage K tile: KV rows x KCH chunks; lds chunk position p holds logical chunk p ^ (row&7)
    {
      constexpr int NCH = KV * KCH;  // 512 or 1024 chunks
#pragma unroll
      for (int i = 0; i < NCH / 256; ++i) {
        const int q = i * 256 + tid;
        const int row = q / KCH, pos = q % KCH;
        const int c = pos ^ (row & 7);
        int kr = kv0 + row;
        if (kr > p.Skv - 1) kr = p.Skv - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(Kp + (long)kr * HD + c * 8), (lptr_t)(ksm + i * 4096 + wave * 1024), 16,
                                         0, 0);
      }
      // ---- stage Vt tile: HD rows x 8 chunks (64 keys)
      constexpr int NVC = HD * 8;
#pragma unroll
      for (int i = 0; i < NVC / 256; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 3, pos = q & 7;
        const int c = pos ^ (row & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(Vp + (long)row * p.kv_stride + kv0 + c * 8),
                                         (lptr_t)(vsm + i * 4096 + wave * 1024), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- S^T tile: rows = keys (4 tiles of 16), col = this lane's query
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int row = j * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < HD / 32; ++kk) {
        const int c = kk * 4 + fg;
        const bf16x8 kf = *(const bf16x8*)(ksm + row * KROW + ((c ^ (row & 7)) << 4));
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[j], 0, 0, 0);
      }
    }
    // lane holds keys kv0 + j*16 + fg*4 + r for query fr
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + j * 16 + fg * 4 + r;
        float v = s[j][r] * p.scale_log2;
        v = key < limit ? v : -1e30f;
        s[j][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kv0 + j * 16 + fg * 4 + r;
        const float e = key < limit ? exp2f(s[j][r] - m_new) : 0.f;
        s[j][r] = e;
        rs += e;
      }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int n = 0; n < HD / 16; ++n) o[n] *= alpha;

    // ---- O^T += Vt . P^T ; k-slot (fg,e): e<4 -> key 32*tt + fg*4 + e ; e>=4 -> key 32*tt + 16 + fg*4 + (e-4)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      union {
        bf16x8 v;
        uint32_t u[4];
      } pb;
      pb.u[0] = pack2bf(s[2 * tt][0], s[2 * tt][1]);
      pb.u[1] = pack2bf(s[2 * tt][2], s[2 * tt][3]);
      pb.u[2] = pack2bf(s[2 * tt + 1][0], s[2 * tt + 1][1]);
      pb.u[3] = pack2bf(s[2 * tt + 1][2], s[2 * tt + 1][3]);
#pragma unroll
      for (int n = 0; n < HD / 16; ++n) {
        const int row = n * 16 + fr;  // d index
        // keys 32*tt + fg*4 .. +3  -> byte offset within the 128-B row
        const int off0 = (tt * 32 + fg * 4) * 2;
        const int off1 = (tt * 32 + 16 + fg * 4) * 2;
        const int sw = (row & 7) << 4;
        const char* base = vsm + row * 128;
        union {
          bf16x8 v;
          uint2 h[2];
        } vf;
        vf.h[0] = *(const uint2*)(base + (((off0 & ~15) ^ sw) | (off0 & 15)));
        vf.h[1] = *(const uint2*)(base + (((off1 & ~15) ^ sw) | (off1 & 15)));
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pb.v, o[n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds d = n*16 + fg*4 + r for query fr
  if (q_valid) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = p.out + ((long)b * p.Lq + qi) * (p.H * HD) + h * HD;
#pragma unroll
    for (int n = 0; n < HD / 16; ++n) {
      uint2 pk;
      pk.x = pack2bf(o[n][0] * inv, o[n][1] * inv);
      pk.y = pack2bf(o[n][2] * inv, o[n][3] * inv);
      *(uint2*)(orow + n * 16 + fg * 4) = pk;
    }
  }
}

extern "C" int gr_attention_bf16(const void* q, const void* k, const void* vt, void* out, const int* kv_len, int B, int H,
                                 int Lq, int Skv, int kv_stride, int head_dim, int causal, int q_pos0, float scale,
                                 hipStream_t stream) {
  if (!q || !k || !vt || !out || B <= 0 || H <= 0 || Lq <= 0 || Skv <= 0) return GR_EINVAL;
  if (kv_stride % 64 != 0 || kv_stride < Skv) return GR_EINVAL;  // Vt tile reads run to the next multiple of 64
  AttnArgs p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.out = (bf16_t*)out;
  p.kv_len = kv_len;
  p.B = B; p.H = H; p.Lq = Lq; p.Skv = Skv; p.kv_stride = kv_stride;
  p.causal = causal; p.q_pos0 = q_pos0;
  p.scale_log2 = scale * 1.44269504088896340736f;
  dim3 grid(gr_cdiv(Lq, 64), B * H);
  if (head_dim == 128) hipLaunchKernelGGL(attention_kernel<128>, grid, dim3(256), 0, stream, p);
  else if (head_dim == 64) hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, stream, p);
  else return GR_EINVAL;
  GR_CHECK_LAUNCH();
  return GR_OK;
}
