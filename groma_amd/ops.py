"""Tensor-level wrappers over the C ABI (include/groma_hip.h).

torch is used for device memory and the current stream only; every computation below is a HIP kernel
in groma_amd/csrc.  All wrappers raise (TypeError / RuntimeError) on bad inputs -- there is no eager path.
"""
import ctypes

import torch

from . import _lib
from ._lib import GemmDesc

F32, I32, I64 = torch.float32, torch.int32, torch.int64
FP8 = torch.float8_e4m3fn  # OCP e4m3 (gfx950 native)


def H16():
    """torch dtype of the active 16-bit operand type: bfloat16 (libgroma_hip.so) or float16 (libgroma_hip_f16.so, and the
    halves of libgroma_hip_ref.so's operand pairs)"""
    return torch.bfloat16 if _lib.PRECISION[0] == "bf16" else torch.float16


def SP():
    """physical 16-bit elements per logical operand element: 2 under precision "ref" (libgroma_hip_ref.so: every operand is a
    (hi, lo) pair of halves, 32-element blocks interleaved along the innermost axis -- csrc/gr_common.h), else 1.  16-bit
    buffers are allocated SP() times as wide; every size handed to the C ABI stays logical."""
    return 2 if _lib.PRECISION[0] == "ref" else 1


# Largest magnitude the half-based operand storages hold: IEEE half 65504; a (hi, lo) pair hi = 65504 saturated + lo up to 65504.
H16_MAX = 65504.0
PAIR_MAX = 2 * 65504.0


class OperandOverflow(ValueError):
    """a tensor packed for a half-based build (precision "fp16" / "ref" / the pair ViT of "hybrid") holds values the storage cannot
    represent: raised at LOAD time for weights (weights.bf), so a checkpoint with out-of-range parameters is reported instead of
    silently saturated.  Activations are converted on the device, where the conversions saturate at +-65504 (gr_common.h sat_h16) --
    where the reference's own fp16 autocast (R: groma/eval/run_groma.py:82) would produce inf."""


def _check_range(t, limit, what, on_overflow):
    if on_overflow == "saturate" or t.numel() == 0:
        return
    m = float(t.abs().max())   # (load-time only: one host sync per packed tensor)
    if m > limit or m != m:
        msg = f"{what or 'tensor'}: max |x| = {m:.6g} exceeds the {limit:.6g} the half-based operand storage holds"
        if on_overflow == "raise":
            raise OperandOverflow(msg)
        import warnings
        warnings.warn(msg + " -- saturated")


def split_pack(t, on_overflow="saturate", what=None):
    """f32 [..., K] (K % 32 == 0) -> the split-operand storage of libgroma_hip_ref.so: half [..., 2K] with
    hi = f16(x), lo = f16(x - hi) interleaved in blocks of 32 (load-time / test plumbing; the kernels write this layout
    themselves on the forward path).
    Range and precision of a pair (tests/test_operand_range.py): |x - (hi + lo)| <= max(2^-22 |x|, 2^-25) for |x| <= 65504 -- 22
    mantissa bits while lo is a normal half (|x| >= 2^-3), an ABSOLUTE 2^-25 below that (lo is then a subnormal half: quantum
    2^-24), so an operand of magnitude 0.02 keeps ~19 bits and anything below 3e-8 is flushed; 65504 < |x| <= 131008 is still
    exact to 2^-11 relative (hi saturated, lo takes the rest), beyond that the pair saturates.
    on_overflow: "saturate" (what the device-side conversions do) | "warn" | "raise" (OperandOverflow) for |x| > 131008."""
    t = t.float()
    _check_range(t, PAIR_MAX, what, on_overflow)
    t = t.clamp(-PAIR_MAX, PAIR_MAX)
    K = t.shape[-1]
    if K % 32:
        raise ValueError(f"split operand rows must be multiples of 32 elements, got {K}")
    hi = t.clamp(-H16_MAX, H16_MAX).to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    return torch.stack((hi.reshape(*t.shape[:-1], K // 32, 32), lo.reshape(*t.shape[:-1], K // 32, 32)), dim=-2) \
        .reshape(*t.shape[:-1], 2 * K).contiguous()


def unsplit(t):
    """inverse view of split_pack: half [..., 2K] -> f32 [..., K] = hi + lo (exact in fp32)"""
    K2 = t.shape[-1]
    v = t.reshape(*t.shape[:-1], K2 // 64, 2, 32).float()
    return (v[..., 0, :] + v[..., 1, :]).reshape(*t.shape[:-1], K2 // 2)


def to_h16(t, on_overflow="saturate", what=None):
    """f32 values -> the active operand storage (bf16 / fp16 cast, or the split pair layout).
    on_overflow ("saturate" | "warn" | "raise"): what to do with values beyond the storage's range (weights.bf passes "raise")"""
    if SP() == 2:
        return split_pack(t, on_overflow, what)
    if H16() == torch.float16:
        t = t.float()
        _check_range(t, H16_MAX, what, on_overflow)
        t = t.clamp(-H16_MAX, H16_MAX)  # the device-side conversions saturate too (gr_common.h sat_h16): never inf
    return t.to(H16()).contiguous()


def from_h16(t):
    """operand storage -> f32 values (tests / the legacy KV view)"""
    return unsplit(t) if SP() == 2 else t.float()


class precision:
    """with ops.precision("fp16"): ...  -- the kernels launched inside come from the library built for that 16-bit operand type
    and the wrappers allocate / expect the matching torch dtype (re-entrant, restores on exit).  A GromaModel is built for one
    precision and wraps its entry points in this; "bf16" is the process default."""

    def __init__(self, p):
        if p not in ("bf16", "fp16", "ref"):
            raise ValueError(f"unknown precision {p!r}")
        self.p = p

    def __enter__(self):
        self.prev, _lib.PRECISION[0] = _lib.PRECISION[0], self.p

    def __exit__(self, *a):
        _lib.PRECISION[0] = self.prev


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/HIP tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise TypeError(f"{name}: must be contiguous")
    return t


_GEMV_WS = {}


def _gemv_ws(splits, M, N, device):
    key = (splits, M, N, str(device))
    t = _GEMV_WS.get(key)
    if t is None:
        t = torch.empty((splits, M, N), dtype=F32, device=device)
        _GEMV_WS[key] = t
    return t


# ---- GEMM plan: which launches are split along K ----------------------------------------------------------------------------
# Split-K changes the fp32 summation order, so the plan must not depend on anything but the GEMM's own (N, K): an image's
# logits, its selected regions and a served request's tokens are then bitwise independent of who shares the batch -- under
# EITHER plan (tests/test_fullsize_properties_gpu.py, tests/test_serving_gpu.py at Groma-7B width).
#   "throughput" (default): never split.  The benchmark's 14 images per GPU fill the chip with whole tiles.
#   "latency": splits = f(N, K) -- the factor that fills the 512 tile slots of the 128x128 kernel for ONE request-sized row
#              block (640 rows: a 582-token prompt, a 1025-token ViT image rounds to the same factor), applied at every M.
#              One image per call: 39 -> 47 img/s (o-proj / down-proj / ViT fc2 / bridge leave most CUs idle otherwise; in the operand-pair
#              build K counts PHYSICAL k-values, so the hybrid ViT's K = 1024 GEMMs split in two: 43.4 -> 44.3 img/s, round 6); at
#              large M it costs the partial-sum traffic, which is why it is a plan the CALLER picks (GromaModel.gemm_plan),
#              not something inferred from the batch.
_PLAN = _lib.ThreadSlot("throughput")  # per thread, like the operand type (groma_amd/_lib.py)
_PLAN_REF_ROWS = 640
PAIR_PLAN_PHYSICAL_K = True


def plan_splits(N, K, plan=None):
    """split-K factor of a plain bf16 GEMM with this (N, K) under `plan` (default: the active plan); a function of (N, K) only"""
    plan = plan or _PLAN[0]
    if PAIR_PLAN_PHYSICAL_K:
        K = K * SP()   # operand-pair build: a logical k-value is two physical ones, and the K-steps the rule counts are physical (round 6)
    if plan == "throughput" or K < 2048:
        return 1
    t128 = -(-_PLAN_REF_ROWS // 128) * -(-N // 128)
    if 2 * t128 > 512:
        return 1
    return max(1, min(512 // t128, K // 1024, 8))  # at least 16 K-steps of 64 per split


class gemm_plan:
    """with ops.gemm_plan("latency"): ...  -- select the plan for the GEMMs launched inside (re-entrant, restores on exit)"""

    def __init__(self, plan):
        if plan not in ("throughput", "latency"):
            raise ValueError(f"unknown GEMM plan {plan!r}")
        self.plan = plan

    def __enter__(self):
        self.prev, _PLAN[0] = _PLAN[0], self.plan

    def __exit__(self, *a):
        _PLAN[0] = self.prev


class gemm_yield:
    """with ops.gemm_yield(): ...  -- the ping-pong GEMMs launched inside (by this thread) run one workgroup per tile instead of a
    persistent grid, so kernels queued on other streams get CUs whenever a tile retires (include/groma_hip.h gr_gemm_yield); re-entrant"""
    _depth = _lib.ThreadSlot(0)

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            gemm_yield._depth[0] += 1
            if gemm_yield._depth[0] == 1:
                _lib.check(_lib.load().gr_gemm_yield(1), "gr_gemm_yield")

    def __exit__(self, *a):
        if self.on:
            gemm_yield._depth[0] -= 1
            if gemm_yield._depth[0] == 0:
                _lib.check(_lib.load().gr_gemm_yield(0), "gr_gemm_yield")


_SPLIT_WS = {}


def _split_ws(splits, M, N, device):
    """fp32 partial-sum workspace [splits, M, N] for EAGER launches, one growing arena per (device, stream): concurrent launches
    on the two streams of a forward never share it, launches on one stream are ordered.  Never (re)allocated inside a hipGraph
    capture: an arena born in a graph's private pool would be baked into every later graph by address and freed with its
    owner -- launch sequences that are captured own their workspace (`split_ws=`: engine.Workspace, part of the graph key)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    n = splits * M * N
    t = _SPLIT_WS.get(key)
    if t is None or t.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K workspace requested inside a hipGraph capture: the captured sequence must pass its own "
                               "`split_ws` (sized with ops.plan_ws_elems)")
        if t is not None:
            torch.cuda.synchronize(device)
        t = _SPLIT_WS[key] = torch.empty((max(n, 0 if t is None else t.numel() * 3 // 2),), dtype=F32, device=device)
    return t[:n].view(splits, M, N)


def plan_ws_elems(M, shapes, plan=None):
    """f32 elements of split-K workspace the plain GEMMs `shapes` = [(N, K), ...] need at M rows under `plan` (0: none split)"""
    need = 0
    for N, K in shapes:
        sp = plan_splits(N, K, plan)
        if sp > 1:
            need = max(need, sp * M * N)
    return need


def gemm(a, w, *, out=None, bias=None, scale=None, resid=None, act=0, out_f32=False, splits=1, ws=None,
         M=None, lda=None, conv=None, resid_mod=0, row_map=None, ldc=None, ldr=None, tile=0,
         a_scale=None, w_scale=None, split_ws=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T).  `conv=(imgs,H,W,C,seg_stride)` switches A to the implicit
    3x3 gather over zero-bordered NHWC maps (then M = imgs*H*W, K = w.shape[1]).
    split_ws: caller-owned flat f32 buffer (>= ops.plan_ws_elems) used when the active GEMM plan splits this launch along K --
    required for launch sequences captured in a hipGraph (the shared arena of eager launches is never touched in a capture)."""
    lib = _lib.load()
    fp8 = w.dtype == FP8
    if fp8:
        _chk(a, FP8, "a"); _chk(w, FP8, "w")
        if w_scale is None:
            raise ValueError("fp8 gemm needs w_scale")
    else:
        _chk(a, H16(), "a"); _chk(w, H16(), "w")
    sp = 1 if fp8 else SP()
    N, K = w.shape[0], w.shape[1] // sp  # logical K
    d = GemmDesc()
    d.fp8 = int(fp8)
    d.a_scale = _chk(a_scale, F32, "a_scale").data_ptr() if a_scale is not None else None
    d.w_scale = _chk(w_scale, F32, "w_scale").data_ptr() if w_scale is not None else None
    if conv is not None:
        imgs, H, W_, C, seg = conv
        M = imgs * H * W_
        d.conv_H, d.conv_W, d.conv_C, d.conv_seg_stride = H, W_, C, seg
        d.lda = 0
    else:
        if M is None:
            M = a.numel() // a.shape[-1]
        d.lda = lda if lda is not None else a.shape[-1] // sp
        if a.shape[-1] != K * sp and lda is None:
            raise ValueError(f"gemm: K mismatch {a.shape[-1] // sp} vs {K}")
    if tile == 0 and conv is None and not fp8 and M <= 8 and splits == 1 and N * K >= (1 << 20) and sp == 1:
        tile, splits = 1, (K + 511) // 512  # decode step: weight-streaming kernel + deterministic split-K reduce
        ws = _gemv_ws(splits, M, N, a.device)
    # (the implicit-conv GEMMs are never split by the plan: measured gain at one image per GPU was within box-to-box spread)
    elif tile == 0 and conv is None and not fp8 and splits == 1 and ws is None and M > 8:
        splits = plan_splits(N, K)
    n_out = N // 2 if act == 3 else N
    osp = 1 if out_f32 else sp  # 16-bit outputs are operand pairs too
    if out is None:
        out = torch.empty((M, n_out * osp), dtype=F32 if out_f32 else H16(), device=a.device)
    _chk(out, F32 if out_f32 else H16(), "out")
    if splits > 1 and ws is None:
        if split_ws is not None:
            if split_ws.numel() < splits * M * N:
                raise ValueError("split_ws is smaller than ops.plan_ws_elems asks for")
            ws = _chk(split_ws, F32, "split_ws").view(-1)[: splits * M * N].view(splits, M, N)
        else:
            ws = _split_ws(splits, M, N, a.device)
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = _chk(bias, F32, "bias").data_ptr() if bias is not None else None
    d.scale = _chk(scale, F32, "scale").data_ptr() if scale is not None else None
    d.resid = _chk(resid, F32, "resid").data_ptr() if resid is not None else None
    d.ws = ws.data_ptr() if ws is not None else None
    d.M, d.N, d.K = M, N, K
    d.ldw = w.stride(0) // sp
    d.ldc = ldc if ldc is not None else out.shape[-1] // osp
    d.ldr = ldr if ldr is not None else (resid.shape[-1] if resid is not None else 0)
    d.act, d.out_f32, d.splits = act, int(out_f32), splits
    d.resid_mod = resid_mod
    d.tile = tile
    if row_map is not None:
        d.c_group, d.c_group_stride, d.c_row_off = row_map
    _lib.check(lib.gr_gemm_bf16(ctypes.byref(d), _stream()), "gr_gemm_bf16")
    return out


def gemv_partials(a, w, M=None, a_parts=None):
    """Decode-step weight streaming: returns (part f32 [splits, M, N], splits) with a @ w^T = sum_z part[z]; the
    reduction is left to the caller (tests merge decode_attention's key slices through it; the decode step itself runs on gemv_fused).
    a_parts = (parts, nsplit, hd, M): the operand is the un-merged output of decode_attention(nsplit > 1)."""
    lib = _lib.load()
    _chk(w, H16(), "w")
    N, K = w.shape
    if a_parts is not None:
        parts, nsplit, hd, M = a_parts
        _chk(parts, F32, "a_parts")
    else:
        _chk(a, H16(), "a")
    if M is None:
        M = a.numel() // a.shape[-1]
    splits = (K + 511) // 512
    ws = _gemv_ws(splits, M, N, w.device)
    d = GemmDesc()
    d.A, d.W, d.C, d.ws = (a.data_ptr() if a is not None else None), w.data_ptr(), None, ws.data_ptr()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, K, w.stride(0), N
    if a_parts is not None:
        d.a_parts, d.a_nsplit, d.a_hd = parts.data_ptr(), nsplit, hd
    d.splits, d.tile = splits, 2
    _lib.check(lib.gr_gemm_bf16(ctypes.byref(d), _stream()), "gr_gemm_bf16")
    return ws, splits


def gemv_fused(w, *, M, x=None, norm=None, a_parts=None, out=None, resid=None, swiglu_out=None, qkv=None, w_scale=None):
    """One decode-step weight stream y[M <= 8, N] = x . w[N,K]^T with its producer and consumer fused (csrc/gemv_fused.hip; with
    e4m3 weights + w_scale csrc/gemv_fp8.hip: the operand is quantised per row in the prologue, the products run on the matrix unit).
    Operand (exactly one): x = 16-bit [M,K]; norm = (h f32 [M,K], gamma, eps) -> RMSNorm in the prologue; a_parts = the
    un-merged output of decode_attention(nsplit > 1).
    Result (exactly one): out f32 [M,N]; resid f32 [M,N] (+= y in place); swiglu_out 16-bit [M, N/2] (interleaved gate / up rows);
    qkv = dict(q, k, vt, cos, sin, H, hd, pos0, pos_dev, pos_stride): RoPE + q / K-cache row / V^T-cache column."""
    lib = _lib.load()
    if w.dtype == FP8:   # e4m3 weight stream (w_scale = the per-row scale): half the bytes per token, operand quantised in the prologue
        if w_scale is None:
            raise ValueError("an e4m3 weight stream needs w_scale")
        _chk(w, FP8, "w")
    else:
        _chk(w, H16(), "w")
    N, K = w.shape
    d = _lib.GemvDesc()
    d.W, d.ldw, d.M, d.N, d.K = w.data_ptr(), w.stride(0), M, N, K
    if w.dtype == FP8:
        d.w8, d.w_scale = 1, _chk(w_scale, F32, "w_scale").data_ptr()
    if norm is not None:
        h, gamma, eps = norm
        _chk(h, F32, "h"); _chk(gamma, F32, "gamma")
        d.x_mode, d.h, d.ldh, d.gamma, d.eps = 1, h.data_ptr(), h.shape[-1], gamma.data_ptr(), eps
    elif a_parts is not None:
        parts, nsplit, hd, Mp = a_parts
        _chk(parts, F32, "a_parts")
        d.x_mode, d.a_parts, d.a_nsplit, d.a_hd = 2, parts.data_ptr(), nsplit, hd
    else:
        _chk(x, H16(), "x")
        d.x_mode, d.A, d.lda = 0, x.data_ptr(), x.shape[-1]
    if out is not None:
        _chk(out, F32, "out")
        d.epi, d.C, d.ldc = 0, out.data_ptr(), out.shape[-1]
    elif resid is not None:
        _chk(resid, F32, "resid")
        d.epi, d.resid, d.ldr = 1, resid.data_ptr(), resid.shape[-1]
    elif swiglu_out is not None:
        _chk(swiglu_out, H16(), "swiglu_out")
        d.epi, d.C, d.ldc = 2, swiglu_out.data_ptr(), swiglu_out.shape[-1]
    else:
        q, k, vt = qkv["q"], qkv["k"], qkv["vt"]
        _chk(q, H16(), "q"); _chk(k, H16(), "k"); _chk(vt, H16(), "vt")
        d.epi, d.q, d.kc, d.vt = 3, q.data_ptr(), k.data_ptr(), vt.data_ptr()
        cos, sin = qkv.get("cos"), qkv.get("sin")
        d.cosT, d.sinT = (cos.data_ptr() if cos is not None else None), (sin.data_ptr() if sin is not None else None)
        d.H, d.HD, d.pos0, d.kv_stride = qkv["H"], qkv["hd"], qkv.get("pos0", 0), k.shape[2]
        pd = qkv.get("pos_dev")
        if pd is not None:
            _chk(pd, I32, "pos_dev")
            d.pos_dev, d.pos_stride = pd.data_ptr(), qkv.get("pos_stride", 0)
    _lib.check(lib.gr_gemv_fused(ctypes.byref(d), _stream()), "gr_gemv_fused")


_DEC_ATT_WS = {}


def decode_attention(q, k, vt, out, *, Smax, q_pos0=0, kv_len=None, scale=None, pos_dev=None, pos_stride=0, nsplit=None):
    """q [B,H,1,hd] against the cache k [B,H,kv_stride,hd] / vt [B,H,hd,kv_stride] -> out bf16 [B, H*hd].
    nsplit None = enough key slices per (row, head) to put a block on every CU.  With nsplit > 1 `out` is NOT written:
    the return value is (parts, nsplit, hd, B) for gemv_partials(..., a_parts=...) (the o-proj merges the slices)."""
    lib = _lib.load()
    _chk(q, H16(), "q"); _chk(k, H16(), "k"); _chk(vt, H16(), "vt"); _chk(out, H16(), "out")
    B, H, _, hd = q.shape
    if scale is None:
        scale = hd ** -0.5
    if pos_dev is not None:
        _chk(pos_dev, I32, "pos_dev")
    if kv_len is not None:
        _chk(kv_len, I32, "kv_len")
    if nsplit is None:
        # measured on MI355X (tests/diag/dec_attn_bench.py): one block per (row, head) streams the cache at ~4.2 TB/s
        # marginal once B*H >= 128; slicing only pays when fewer blocks than that exist (the consumer-side merge costs
        # ~5 us in the o-proj GEMV).  Independent of S, so eager and graph-replayed steps slice identically.
        nsplit = 1 if B * H >= 128 else max(1, min(4, 128 // (B * H)))
    parts = None
    if nsplit > 1:
        key = (B * H, nsplit, hd, str(q.device))
        if key not in _DEC_ATT_WS:
            _DEC_ATT_WS[key] = torch.empty((B * H * nsplit * (hd + 2),), dtype=F32, device=q.device)
        parts = _DEC_ATT_WS[key]
    _lib.check(lib.gr_decode_attention(_p(q), _p(k), _p(vt), _p(out), _p(kv_len), B, H, Smax, k.shape[2], hd, q_pos0,
                                       scale, _p(pos_dev), pos_stride, nsplit, _p(parts), _stream()),
               "gr_decode_attention")
    if nsplit > 1:
        return parts, nsplit, hd, B
    return out


def _h16_shape(shape):
    return tuple(shape[:-1]) + (shape[-1] * SP(),)


def gemm_f32(a, w, *, bias=None, resid=None, act=0, out=None, M=None, lda=None, ldc=None):
    lib = _lib.load()
    _chk(a, F32, "a"); _chk(w, F32, "w")
    N, K = w.shape
    if M is None:
        M = a.numel() // a.shape[-1]
    lda = lda if lda is not None else a.shape[-1]
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=a.device)
    ldc = ldc if ldc is not None else out.shape[-1]
    _lib.check(lib.gr_gemm_f32(_p(a), _p(w), _p(out), _p(bias), _p(resid), M, N, K, lda, w.stride(0), ldc, act,
                               _stream()), "gr_gemm_f32")
    return out


def layernorm(x, gamma, beta, eps, *, add=None, out_bf16=False, relu_in=False, out=None):
    lib = _lib.load()
    _chk(x, F32, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty(_h16_shape(x.shape) if out_bf16 else x.shape, dtype=H16() if out_bf16 else F32, device=x.device)
    _lib.check(lib.gr_layernorm(_p(x), _p(add), _p(gamma), _p(beta), _p(out), rows, C, C, C, eps, int(out_bf16),
                                int(relu_in), _stream()), "gr_layernorm")
    return out


def rmsnorm(x, gamma, eps, *, out_bf16=True, out=None):
    lib = _lib.load()
    _chk(x, F32, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty(_h16_shape(x.shape) if out_bf16 else x.shape, dtype=H16() if out_bf16 else F32, device=x.device)
    _lib.check(lib.gr_rmsnorm(_p(x), _p(gamma), _p(out), rows, C, C, C, eps, int(out_bf16), _stream()), "gr_rmsnorm")
    return out


def attention(q, k, vt, *, Skv, causal, q_pos0=0, kv_len=None, scale=None, out=None, pos_dev=None, pos_stride=0,
              fused=None):
    """q [B,H,Lq,hd]; k [B,H,kv_stride,hd]; vt [B,H,hd,kv_stride] -> [B*Lq, H*hd].
    pos_dev (i32 device tensor): row b attends keys [0, pos_dev[b*pos_stride] + Lq) -- device-resident decode position."""
    lib = _lib.load()
    _chk(q, H16(), "q"); _chk(k, H16(), "k"); _chk(vt, H16(), "vt")
    q_ld, cos, sin = 0, None, None
    if fused is not None:  # q = the fused projection buffer [B*Lq, ld]; fused = dict(B, H, Lq, hd, cos=None, sin=None)
        B, H, Lq, hd = fused["B"], fused["H"], fused["Lq"], fused["hd"]
        q_ld, cos, sin = q.shape[-1] // SP(), fused.get("cos"), fused.get("sin")
    else:
        B, H, Lq, hd = q.shape
        hd //= SP()
    kv_stride = k.shape[2]
    if scale is None:
        scale = hd ** -0.5
    if out is None:
        out = torch.empty((B * Lq, H * hd * SP()), dtype=H16(), device=q.device)
    if kv_len is not None:
        _chk(kv_len, I32, "kv_len")
    if pos_dev is not None:
        _chk(pos_dev, I32, "pos_dev")
    _lib.check(lib.gr_attention_bf16(_p(q), _p(k), _p(vt), _p(out), _p(kv_len), B, H, Lq, Skv, kv_stride, hd,
                                     int(causal), q_pos0, scale, _p(pos_dev), pos_stride, q_ld, _p(cos), _p(sin), _stream()),
               "gr_attention_bf16")
    return out


def qkv_split(qkv, q, k, vt, *, B, H, L, hd, pos0=0, cos=None, sin=None, pos_dev=None, pos_stride=0):
    """q=None: only k / v^T are written (attention(fused=...) reads q from qkv)"""
    lib = _lib.load()
    _chk(qkv, H16(), "qkv")
    if pos_dev is not None:
        _chk(pos_dev, I32, "pos_dev")
    _lib.check(lib.gr_qkv_split(_p(qkv), _p(q), _p(k), _p(vt), _p(cos), _p(sin), B, H, L, hd, pos0, k.shape[2],
                                _p(pos_dev), pos_stride, _stream()), "gr_qkv_split")


def patchify(images, P, Kpad, out=None):
    lib = _lib.load()
    _chk(images, F32, "images")
    B, _, S, _ = images.shape
    G = S // P
    if out is None:
        out = torch.empty((B * G * G, Kpad * SP()), dtype=H16(), device=images.device)
    _chk(out, H16(), "out")
    _lib.check(lib.gr_patchify(_p(images), _p(out), B, S, P, Kpad, _stream()), "gr_patchify")
    return out


def fill_rows(src, dst, rows, ld_dst):
    lib = _lib.load()
    _lib.check(lib.gr_fill_rows_f32(_p(src), _p(dst), rows, src.numel(), ld_dst, _stream()), "gr_fill_rows_f32")


def mean4_tokens(h0, h1, h2, h3):
    lib = _lib.load()
    B, T, C = h0.shape
    out = torch.empty((B * (T - 1), C), dtype=F32, device=h0.device)
    _lib.check(lib.gr_mean4_tokens(_p(h0), _p(h1), _p(h2), _p(h3), _p(out), B, T, C, _stream()), "gr_mean4_tokens")
    return out


def s2d_pack(h, G):
    lib = _lib.load()
    B, T, C = h.shape
    out = torch.empty((B * (G // 2) ** 2, 4 * C * SP()), dtype=H16(), device=h.device)
    _lib.check(lib.gr_s2d_pack(_p(h), _p(out), B, G, C, _stream()), "gr_s2d_pack")
    return out


def upsample_coord_pack(h, G, Ho, Cpad, out=None):
    lib = _lib.load()
    B, T, C = h.shape
    if out is None:
        out = torch.empty((B * Ho * Ho, Cpad * SP()), dtype=H16(), device=h.device)
    _chk(out, H16(), "out")
    _lib.check(lib.gr_upsample_coord_pack(_p(h), _p(out), B, G, Ho, C, Cpad, _stream()), "gr_upsample_coord_pack")
    return out


def gn_stats_blocks(HW):
    return int(_lib.load().gr_gn_stats_blocks(HW))


def gn_coef(x, imgs, HW, C, groups, gamma, beta, eps, sums=None, coef=None):
    """GroupNorm statistics of a conv output x bf16 [imgs*HW, C] -> per-(image, channel) affine y = x*a + b,
    f32 [imgs, 2, C] (consumed by fuse_shuffle).  sums / coef: caller-owned buffers ([imgs, gn_stats_blocks(HW), C, 2] / [imgs, 2, C])"""
    lib = _lib.load()
    if sums is None:
        sums = torch.empty((imgs, lib.gr_gn_stats_blocks(HW), C, 2), dtype=F32, device=x.device)
    _chk(sums, F32, "sums")
    _lib.check(lib.gr_gn_stats(_p(x), _p(sums), imgs, HW, C, _stream()), "gr_gn_stats")
    if coef is None:
        coef = torch.empty((imgs, 2, C), dtype=F32, device=x.device)
    _chk(coef, F32, "coef")
    _lib.check(lib.gr_gn_finalize(_p(sums), _p(gamma), _p(beta), _p(coef), imgs, HW, C, groups, eps, _stream()),
               "gr_gn_finalize")
    return coef


def fuse_shuffle(tar, top, down, out, *, imgs, C, shuffle, pad, q_inv=None):
    """each of tar/top/down = (map bf16 [imgs*S*S, C], coef or None, S).  q_inv: write the map as e4m3 bytes of value * q_inv
    (out is a float8_e4m3fn buffer; the conv that consumes it carries 1 / q_inv in its w_scale)"""
    lib = _lib.load()
    t, tp, dn = tar, top or (None, None, 0), down or (None, None, 0)
    if q_inv is not None:
        _chk(out, FP8, "out")
        _lib.check(lib.gr_fuse_shuffle_fp8(_p(t[0]), _p(t[1]), t[2], _p(tp[0]), _p(tp[1]), tp[2], _p(dn[0]), _p(dn[1]), dn[2],
                                           _p(out), imgs, C, int(shuffle), pad, float(q_inv), _stream()), "gr_fuse_shuffle_fp8")
        return out
    _lib.check(lib.gr_fuse_shuffle(_p(t[0]), _p(t[1]), t[2], _p(tp[0]), _p(tp[1]), tp[2], _p(dn[0]), _p(dn[1]), dn[2],
                                   _p(out), imgs, C, int(shuffle), pad, _stream()), "gr_fuse_shuffle")
    return out


def cast_bf16(a, b=None):
    lib = _lib.load()
    _chk(a, F32, "a")
    out = torch.empty(_h16_shape(a.shape), dtype=H16(), device=a.device)
    _lib.check(lib.gr_cast_f32_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "gr_cast_f32_bf16")
    return out


def add_rows(a, b, b_mod=0, out=None):
    lib = _lib.load()
    _chk(a, F32, "a"); _chk(b, F32, "b")
    C = a.shape[-1]
    if out is None:
        out = torch.empty_like(a)
    _lib.check(lib.gr_add_rows_f32(_p(a), _p(b), _p(out), a.numel() // C, C, b_mod, _stream()), "gr_add_rows_f32")
    return out


def embed_gather(ids, table0, table1, out=None):
    lib = _lib.load()
    _chk(ids, I64, "ids")
    C = table0.shape[1] // SP()
    if out is None:
        out = torch.empty((ids.numel(), C), dtype=F32, device=ids.device)
    _lib.check(lib.gr_embed_gather(_p(ids), _p(table0), _p(table1), _p(out), ids.numel(), C, table0.shape[0],
                                   table1.shape[0], _stream()), "gr_embed_gather")
    return out


def scatter_rows(src, row_idx, dst):
    lib = _lib.load()
    _chk(src, F32, "src"); _chk(row_idx, I32, "row_idx"); _chk(dst, F32, "dst")
    _lib.check(lib.gr_scatter_rows_f32(_p(src), _p(row_idx), _p(dst), row_idx.numel(), src.shape[-1], _stream()),
               "gr_scatter_rows_f32")


def argmax_rows(x, V, out=None):
    lib = _lib.load()
    _chk(x, F32, "x")
    rows = x.numel() // x.shape[-1]
    if out is None:
        out = torch.empty((rows,), dtype=I64, device=x.device)
    _lib.check(lib.gr_argmax_rows(_p(x), _p(out), rows, V, x.shape[-1], _stream()), "gr_argmax_rows")
    return out


def sample_uniform(seed, counter):
    """host twin of the device draw (splitmix64 of the request seed and the token position) -> float in [0, 1)"""
    M = (1 << 64) - 1
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(counter) + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return (z >> 40) / 16777216.0


def sample_rows(x, V, inv_temp, seed, pos=None, pos_stride=0, pos_off=0, out=None):
    """x f32 [rows, ld] -> token ids int64 [rows]: arg-max where inv_temp[row] == 0, else a temperature sample
    (model_worker.py:307-311) drawn with sample_uniform(seed[row], pos[row*pos_stride] + pos_off)"""
    lib = _lib.load()
    _chk(x, F32, "x"); _chk(inv_temp, F32, "inv_temp"); _chk(seed, I64, "seed")
    if pos is not None:
        _chk(pos, I32, "pos")
    rows = x.numel() // x.shape[-1]
    if out is None:
        out = torch.empty((rows,), dtype=I64, device=x.device)
    _lib.check(lib.gr_sample_rows(_p(x), _p(out), rows, V, x.shape[-1], _p(inv_temp), _p(seed), _p(pos), pos_stride, pos_off,
                                  _stream()), "gr_sample_rows")
    return out


def greedy_advance(nxt, tok, unfinished, seq, pos, step, n_unfinished, *, eos, pad, inc_pos):
    """HF greedy_search bookkeeping of one step on the device (see include/groma_hip.h)"""
    lib = _lib.load()
    _chk(nxt, I64, "nxt"); _chk(tok, I64, "tok"); _chk(unfinished, I64, "unfinished"); _chk(seq, I64, "seq")
    _chk(pos, I32, "pos"); _chk(step, I32, "step"); _chk(n_unfinished, I32, "n_unfinished")
    _lib.check(lib.gr_greedy_advance(_p(nxt), _p(tok), _p(unfinished), _p(seq), _p(pos), _p(step), _p(n_unfinished),
                                     tok.numel(), -1 if eos is None else int(eos), 0 if pad is None else int(pad),
                                     seq.shape[-1], pos.numel(), int(inc_pos), _stream()), "gr_greedy_advance")


def msda(value, offw, ref, *, B, Q, heads, n_points, Hs, Ws, rdim, ref_batched):
    lib = _lib.load()
    out = torch.empty((B * Q, heads * 32), dtype=F32, device=value.device)
    _lib.check(lib.gr_msda_f32(_p(value), _p(offw), _p(ref), _p(out), B, Q, heads, n_points, Hs, Ws, offw.shape[-1],
                               rdim, int(ref_batched), _stream()), "gr_msda_f32")
    return out


def mha32(qk, v, *, B, Q, heads, scale):
    lib = _lib.load()
    out = torch.empty((B * Q, heads * 32), dtype=F32, device=qk.device)
    _lib.check(lib.gr_mha32_f32(_p(qk), _p(v), _p(out), B, Q, heads, qk.shape[-1], scale, _stream()), "gr_mha32_f32")
    return out


def ddetr_topk_gather(idx, delta, prop, *, B, S, Kq, npf):
    lib = _lib.load()
    ref = torch.empty((B * Kq, 4), dtype=F32, device=delta.device)
    pos = torch.empty((B * Kq, 4 * npf), dtype=F32, device=delta.device)
    _lib.check(lib.gr_ddetr_topk_gather(_p(idx), _p(delta), _p(prop), _p(ref), _p(pos), B, S, Kq, npf, _stream()),
               "gr_ddetr_topk_gather")
    return ref, pos


def box_refine(tmp, ref):
    lib = _lib.load()
    out = torch.empty_like(ref)
    _lib.check(lib.gr_box_refine(_p(tmp), _p(ref), _p(out), ref.numel(), _stream()), "gr_box_refine")
    return out


def score_fuse(coco, sa1b, n, ld=1):
    lib = _lib.load()
    out = torch.empty((n,), dtype=F32, device=coco.device)
    _lib.check(lib.gr_score_fuse(_p(coco), _p(sa1b), _p(out), n, ld, _stream()), "gr_score_fuse")
    return out


def topk_desc(x, K):
    """x f32 [B,S] -> int32 [B,K], order (value desc, index asc)"""
    lib = _lib.load()
    _chk(x, F32, "x")
    B, S = x.shape
    out = torch.empty((B, K), dtype=I32, device=x.device)
    _lib.check(lib.gr_topk_desc(_p(x), _p(out), B, S, K, S, _stream()), "gr_topk_desc")
    return out


_NMS_WS = {}


def _nms_ws(B, n, device):
    """caller-owned workspace of the general NMS path (n > 512); None on the one-workgroup fast path"""
    nbytes = _lib.load().gr_nms_workspace_bytes(B, n)
    if nbytes == 0:
        return None
    key = (str(device), nbytes)
    if key not in _NMS_WS:
        _NMS_WS.clear()
        _NMS_WS[key] = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    return _NMS_WS[key]


def nms_workspace(B, n, device):
    """a workspace of the general NMS path (n > 512) that the CALLER owns (None when the one-workgroup path needs none): for
    launches captured in a hipGraph, whose pointer arguments must outlive the shared cache above"""
    nbytes = _lib.load().gr_nms_workspace_bytes(B, n)
    return torch.empty((nbytes,), dtype=torch.uint8, device=device) if nbytes else None


def nms(boxes_cxcywh, scores, iou_thr, score_thr, max_num, n_valid=None, workspace=None):
    """boxes [B,n,4] (cx,cy,w,h), scores [B,n] -> keep int64 [B,max_num] (-1 padded), n_keep int32 [B]"""
    lib = _lib.load()
    _chk(boxes_cxcywh, F32, "boxes"); _chk(scores, F32, "scores")
    B, n = scores.shape
    keep = torch.empty((B, max_num), dtype=I64, device=scores.device)
    n_keep = torch.empty((B,), dtype=I32, device=scores.device)
    ws = workspace if workspace is not None else _nms_ws(B, n, scores.device)
    _lib.check(lib.gr_nms_f32(_p(boxes_cxcywh), _p(scores), B, n, iou_thr, score_thr, max_num, _p(n_valid), _p(keep),
                              _p(n_keep), _p(ws), _stream()), "gr_nms_f32")
    return keep, n_keep


def nms_xyxy(boxes_xyxy, scores, iou_threshold, offset):
    """mmcv `_ext.nms` contract (pybind.cpp:175): boxes f32 [n,4] corners, scores [n] -> (keep int64 [n] with -1 tail,
    n_keep int32 [1]) on the device; no host sync here (groma_amd.mmcv_ext.nms slices the result like the reference)."""
    lib = _lib.load()
    _chk(boxes_xyxy, F32, "boxes"); _chk(scores, F32, "scores")
    n = scores.shape[0]
    if boxes_xyxy.shape != (n, 4):
        raise ValueError(f"boxes must be [{n},4], got {tuple(boxes_xyxy.shape)}")
    keep = torch.empty((max(n, 1),), dtype=I64, device=scores.device)
    n_keep = torch.empty((1,), dtype=I32, device=scores.device)
    _lib.check(lib.gr_nms(_p(boxes_xyxy), _p(scores), n, float(iou_threshold), int(offset), _p(keep), _p(n_keep),
                          _p(_nms_ws(1, n, scores.device)), _stream()), "gr_nms")
    return keep, n_keep


def roi_align_forward(input, rois, output, argmax_y, argmax_x, aligned_height, aligned_width, spatial_scale,
                      sampling_ratio, pool_mode, aligned):
    """mmcv `_ext.roi_align_forward` contract (pybind.cpp:596): NCHW f32 input, [K,5] rois, writes `output`
    (and argmax_y / argmax_x for max pooling) in place on the current stream."""
    lib = _lib.load()
    _chk(input, F32, "input"); _chk(rois, F32, "rois"); _chk(output, F32, "output")
    N, C, H, W = input.shape
    K = rois.shape[0]
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise ValueError("rois must be [K,5]")
    if tuple(output.shape) != (K, C, aligned_height, aligned_width):
        raise ValueError("output must be [K,C,aligned_height,aligned_width]")
    if pool_mode == 0:
        _chk(argmax_y, F32, "argmax_y"); _chk(argmax_x, F32, "argmax_x")
    _lib.check(lib.gr_roi_align_forward(_p(input), _p(rois), _p(output), _p(argmax_y) if pool_mode == 0 else None,
                                        _p(argmax_x) if pool_mode == 0 else None, K, C, H, W, aligned_height,
                                        aligned_width, float(spatial_scale), int(sampling_ratio), int(pool_mode),
                                        int(bool(aligned)), _stream()), "gr_roi_align_forward")
    return output


def roi_align_pack(feat_nhwc, rois, out, *, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned=True, pad=1,
                   out_f32=False, q_inv=None):
    lib = _lib.load()
    _chk(feat_nhwc, H16(), "feat"); _chk(rois, F32, "rois")
    if q_inv is not None:  # e4m3 tiles of value * q_inv (the per-ROI conv's A operand in fp8 mode)
        _chk(out, FP8, "out")
        _lib.check(lib.gr_roi_align_pack_fp8(_p(feat_nhwc), _p(rois), _p(out), rois.shape[0], C, H, W, ph, pw, spatial_scale,
                                             sampling_ratio, int(aligned), pad, float(q_inv), _stream()), "gr_roi_align_pack_fp8")
        return out
    _lib.check(lib.gr_roi_align_pack(_p(feat_nhwc), _p(rois), _p(out), rois.shape[0], C, H, W, ph, pw, spatial_scale,
                                     sampling_ratio, int(aligned), pad, int(out_f32), _stream()), "gr_roi_align_pack")
    return out


def prof_enable(on):
    _lib.check(_lib.load().gr_prof_enable(int(on)), "gr_prof_enable")


def prof_read():
    ms, n, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
    _lib.check(_lib.load().gr_prof_read(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "gr_prof_read")
    return ms.value, n.value, fl.value


def prof_read_launches(cap=65536):
    """-> list of (M, N, K, tag, ms) per gr_gemm_bf16 launch since prof_enable(True) (drains the records)"""
    import numpy as np
    mnk = np.zeros((cap, 4), dtype=np.int32)
    ms = np.zeros((cap,), dtype=np.float32)
    n = ctypes.c_long()
    _lib.check(_lib.load().gr_prof_read_launches(cap, mnk.ctypes.data, ms.ctypes.data, ctypes.byref(n)),
               "gr_prof_read_launches")
    k = min(n.value, cap)
    return [(int(mnk[i, 0]), int(mnk[i, 1]), int(mnk[i, 2]), int(mnk[i, 3]), float(ms[i])) for i in range(k)]


def quant_rows_fp8(x, out=None):
    """x bf16|f32 [rows, K] -> (q e4m3 [rows, K], scale f32 [rows]) with x ~= q * scale[:, None].
    out = (q, s): caller-owned buffers (launch sequences replayed from a hipGraph bake their addresses)"""
    lib = _lib.load()
    K = x.shape[-1]
    rows = x.numel() // K
    if out is not None:
        q, s = _chk(out[0], FP8, "q"), _chk(out[1], F32, "s")
        if q.numel() != rows * K or s.numel() != rows:
            raise ValueError("quant_rows_fp8: out buffers do not match the input")
    else:
        q = torch.empty((rows, K), dtype=FP8, device=x.device)
        s = torch.empty((rows,), dtype=F32, device=x.device)
    _lib.check(lib.gr_quant_rows_fp8(_p(x), int(x.dtype == F32), _p(q), _p(s), rows, K, K, _stream()), "gr_quant_rows_fp8")
    return q, s


def norm_fp8(x, gamma, beta, eps, rms, out=None):
    """RMSNorm / LayerNorm of f32 rows, emitted as e4m3 + per-row scale; out = (q, s): caller-owned buffers"""
    lib = _lib.load()
    _chk(x, F32, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is not None:
        q, s = _chk(out[0], FP8, "q"), _chk(out[1], F32, "s")
        if q.numel() != rows * C or s.numel() != rows:
            raise ValueError("norm_fp8: out buffers do not match the input")
    else:
        q = torch.empty((rows, C), dtype=FP8, device=x.device)
        s = torch.empty((rows,), dtype=F32, device=x.device)
    _lib.check(lib.gr_norm_fp8(_p(x), _p(gamma), _p(beta), _p(q), _p(s), rows, C, eps, int(rms), _stream()), "gr_norm_fp8")
    return q, s
