"""Device-side stages of the Groma forward path, as sequences of HIP kernels (groma_amd/ops.py -> C ABI).

Stage            reference                                                    SURVEY §8a rows
VitEngine        HF Dinov2Model, groma/model/groma.py:222                     a1
ProposerEngine   groma/model/ddetr.py + ddetr_transformer.py, groma.py:240-249 a4-a10
RegionEngine     groma/model/roi_align.py:97-327                               a14-a17
LlamaEngine      HF LlamaModel + heads, groma.py:389-402                       a19-a22
Residual streams are fp32 in HBM; GEMM operands bf16 (fp32 accumulate); the proposer is fp32 end to end.
"""
import collections
import math

import torch

from . import ops

F32, I32, I64 = torch.float32, torch.int32, torch.int64
H16 = ops.H16  # dtype of the active 16-bit operand type (bf16 / fp16: ops.precision)


Q_IN_PLACE = True  # prefill attention reads q (and applies RoPE) straight from the fused QKV projection
FUSED_DECODE = True  # a decode step of <= 8 rows runs on the fused weight streams (False: the general kernels -- tests compare the two)
WIDE_DECODE = True   # a decode step of 9..64 rows runs on the matrix-unit weight stream (csrc/gemm_skinny.hip); False: the general kernels
YIELD_PYRAMID = True  # the region pyramid's convolutions (side stream) run one workgroup per tile, so the proposer chain on the main stream is not
                      # stalled behind multi-millisecond persistent launches (ops.gemm_yield); False: persistent grids (tests/diag A-B)


def _ru(x, m):
    return (x + m - 1) // m * m


def h16(*shape):
    """physical shape of a 16-bit buffer with this LOGICAL shape under the active operand type: the innermost extent doubles
    when the operands are (hi, lo) pairs (precision "ref": ops.SP() == 2, csrc/gr_common.h); kernels take logical sizes"""
    return tuple(shape[:-1]) + (shape[-1] * ops.SP(),)


def normal_mode(fn):
    """Run an entry point of the boundary with autograd off and torch.inference_mode DISABLED, whatever the caller set.
    The reference's callers wrap generate() in `with torch.inference_mode():` (groma/eval/eval_rec.py:92, run_groma.py:82,
    serve/model_worker.py:256).  This path keeps persistent device state across calls -- workspace arenas, KV / decode arenas,
    the device-resident loop counters of the captured decode graph -- and a tensor allocated under inference mode could never
    be updated in place by a later call made outside it.  Inside, every tensor is an ordinary no-grad tensor; results handed
    back are ordinary tensors too (usable inside or outside the caller's inference-mode block)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        with torch.inference_mode(False), torch.no_grad():
            return fn(*a, **kw)
    return wrapped


def model_entry(precision_of):
    """normal_mode + the model's 16-bit operand type active (ops.precision) for the duration of the call.
    precision_of(self, *args, **kwargs) -> "bf16" | "fp16" | "ref"."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(self, *a, **kw):
            with torch.inference_mode(False), torch.no_grad(), ops.precision(precision_of(self, *a, **kw)):
                return fn(self, *a, **kw)
        return wrapped
    return deco


def inplace_copy(dst, src):
    """dst.copy_(src) for a tensor the CALLER owns (the reference mutates its input_ids, groma.py:295,307): an inference
    tensor may only be written under inference mode"""
    with torch.inference_mode(dst.is_inference()):
        dst.copy_(src)


# Parity instrumentation: set engine.TRACE = {} and the next forward stores a copy of every intermediate that crosses a
# kernel boundary in the FIRST layer of each stage (tests/test_fullwidth_parity_gpu.py feeds each kernel's actual input to
# the oracle's version of that one operation).  None (the default) = no copies, no overhead.
TRACE = None


def _trace(name, t):
    if TRACE is not None:
        TRACE[name] = t.detach().clone()
    return t


def _trace_q8(tag, x8, sx):
    """e4m3 operand of a GEMM: the quantised rows and their scales (tests/test_fp8_width_gpu.py feeds exactly these to the oracle)"""
    if TRACE is not None and tag:
        TRACE[tag + ".q8"], TRACE[tag + ".s8"] = x8.detach().clone(), sx.detach().clone()


def _zeros(shape, dtype, device):
    """torch.zeros, with e4m3 buffers made as bytes (0x00 is e4m3 zero; fill kernels for the 8-bit float types are not assumed)"""
    if dtype == ops.FP8:
        return torch.zeros(shape, dtype=torch.uint8, device=device).view(ops.FP8)
    return torch.zeros(shape, dtype=dtype, device=device)


class Workspace:
    """Scratch arenas, one per buffer NAME, grown geometrically and handed out as a view of the first prod(shape)
    elements -- so ragged request shapes (L = P + 256 + 2*N_regions changes with every image, R = sum N_i) do not
    each pin their own permanent set of buffers (HBM would grow without bound in an eval / serving loop).

    zero=True: the consumer relies on untouched elements being zero (the 1-pixel borders of the 3x3-conv inputs, the
    64-key padding of the ViT K / V^T tiles).  A fresh arena is zero-filled; when the requested shape changes the view is
    cleared again (one memset per shape change, none in steady state).
    exact=True: a dedicated tensor per (name, shape) that is never moved -- for buffers whose address is baked into a
    captured hipGraph (the decode step)."""

    def __init__(self, device):
        self.device, self._arena, self._zshape, self._exact = device, {}, {}, {}

    def get(self, name, shape, dtype, zero=False, exact=False):
        shape = tuple(int(x) for x in shape)
        if exact:
            key = (name, shape, dtype)
            t = self._exact.get(key)
            if t is None:
                t = self._exact[key] = _zeros(shape, dtype, self.device)
            return t
        n = math.prod(shape)
        key = (name, dtype)
        buf = self._arena.get(key)
        if buf is None or buf.numel() < n:
            if buf is not None:
                torch.cuda.synchronize(self.device)  # the old arena may still be read on the side stream
            cap = n if buf is None else max(n, buf.numel() * 3 // 2)
            buf = self._arena[key] = _zeros((cap,), dtype, self.device)
            self._zshape[key] = shape
        view = buf[:n].view(shape)
        if zero and self._zshape.get(key) != shape:
            (view.view(torch.uint8) if dtype == ops.FP8 else view).zero_()
            self._zshape[key] = shape
        elif not zero:
            self._zshape[key] = None
        return view

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in list(self._arena.values()) + list(self._exact.values()))


class GraphPool:
    """Captured hipGraphs of the shape-static launch sequences of a forward (the 24 ViT layers, the 32 LLaMA prefill layers).

    A launch sequence is a pure function of what is baked into its kernel arguments: buffer addresses and scalars.  The
    caller folds ALL of them into `key` (every scratch / cache / input address, the shape, the GEMM plan), so a graph is
    only ever replayed onto exactly the memory the eager launches would have used -- an arena that was regrown, or a new KV
    cache, simply produces a new key.  A key is captured the third time it is seen (one-off shapes of an eval loop stay
    eager; a capture costs about one eager pass) and the pool keeps the `cap` most recently used graphs; an evicted key
    starts counting again, so a loop cycling through more shapes than `cap` captures at most once per 3 * cap calls.

    Why: the kernels are identical either way, but eagerly launched kernels show a box-dependent 0-7 us dispatch gap between
    consecutive kernels (profiles/r03_timeline_b14.txt: 436 gaps = 3 ms of a 145 ms step on some boxes, none on others);
    nodes of a replayed graph start back to back on every box, and ~700 host launches per step disappear."""

    enabled = True

    CAPTURE_AT = 3
    _first_sight = ops._lib.ThreadSlot(False)   # per thread: capture a key the FIRST time it is seen (GraphPool.first_sight())

    @classmethod
    def first_sight(cls):
        """with GraphPool.first_sight(): ...  -- every pool captures a new key on its first sighting inside the block instead of
        its third.  For warm-up passes whose shapes are known to recur (serving.ContinuousBatcher.warm_admission): the capture
        passes then happen before live traffic, not inside the first three live requests of a shape."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, cls._first_sight[0] = cls._first_sight[0], True
            try:
                yield
            finally:
                cls._first_sight[0] = prev
        return ctx()

    _eager = ops._lib.ThreadSlot(False)          # per thread: launch eagerly, leave every pool alone (GraphPool.eager())

    @classmethod
    def eager(cls):
        """with GraphPool.eager(): ...  -- the calling thread's launch sequences run eagerly and are neither counted nor captured
        (the settling pass of a warm-up: arenas grow, kernels see their first launch)"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, cls._eager[0] = cls._eager[0], True
            try:
                yield
            finally:
                cls._eager[0] = prev
        return ctx()

    def drop(self, pred):
        """forget the graphs (and sighting counts) whose key satisfies pred(key) -- keys that bake in memory which no longer exists"""
        for d in (self._graphs, self._seen):
            for key in [k for k in d if pred(k)]:
                d.pop(key, None)
                self._bytes.pop(key, None)

    def __init__(self, cap=16, byte_budget=2 << 30):
        # cap: graphs kept (LRU); byte_budget: device memory the graphs' private pools may pin in total (what a capture reserved is
        # measured around it) -- the least recently used graphs go first when either bound is exceeded
        self.cap, self.byte_budget = cap, byte_budget
        self._graphs, self._seen, self._bytes = collections.OrderedDict(), collections.OrderedDict(), {}
        self.replays = self.captures = 0

    def run(self, key, launch):
        if not GraphPool.enabled or GraphPool._eager[0] or TRACE is not None or torch.cuda.is_current_stream_capturing():
            launch()
            return
        g = self._graphs.get(key)
        if g is not None:
            self._graphs.move_to_end(key)
            g.replay()
            self.replays += 1
            return
        n = self._seen.pop(key, 0) + 1
        if n < GraphPool.CAPTURE_AT and not GraphPool._first_sight[0]:
            self._seen[key] = n
            if len(self._seen) > 256:
                self._seen.popitem(last=False)
            launch()
            return
        self._capture(key, launch).replay()

    def warm(self, key, launch):
        """capture `key` now (admission-time warm-up of a serving loop: keeps the capture pass out of the first live requests)"""
        if GraphPool.enabled and TRACE is None and key not in self._graphs and not torch.cuda.is_current_stream_capturing():
            self._capture(key, launch)

    def _capture(self, key, launch):
        """Capture on a stream of our own in thread-local error mode.  Only the CALLER's stream is waited for -- not the device:
        torch.cuda.graph() would synchronise the whole device, collect garbage and empty the allocator cache, stalling every
        other stream (another model's forward, a serving loop's decode ticks) for a capture that concerns none of them."""
        cur = torch.cuda.current_stream()
        cur.synchronize()
        dev = cur.device
        before = torch.cuda.memory_reserved(dev)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        err = None
        with torch.cuda.stream(side):
            g.capture_begin(capture_error_mode="thread_local")
            try:
                launch()
            except BaseException as e:   # a launch that raises (GR_EINVAL, a workspace requested inside the capture) invalidates the capture
                err = e
            try:
                g.capture_end()          # always end it: a dangling capture would poison every later launch on this stream
            except Exception:
                if err is None:
                    raise
        cur.wait_stream(side)            # (the streams are re-joined whatever happened)
        if err is not None:
            self._seen.pop(key, None)    # nothing is kept; the key starts counting again
            raise err                    # the ORIGINAL error, not the secondary one from ending a broken capture
        self._graphs[key] = g
        self._bytes[key] = max(0, torch.cuda.memory_reserved(dev) - before)
        self.captures += 1
        while len(self._graphs) > 1 and (len(self._graphs) > self.cap or sum(self._bytes.values()) > self.byte_budget):
            old, _ = self._graphs.popitem(last=False)
            self._bytes.pop(old, None)
        return g

    def clear(self):
        self._graphs.clear(), self._seen.clear(), self._bytes.clear()


# ------------------------------------------------------------------------------------------------ DINOv2
class VitEngine:
    def __init__(self, w, cfg, ws):
        self.w, self.ws = w, ws
        vc = cfg.perceiver_cfg.vis_encoder_cfg
        self.D, self.H, self.P, self.eps = vc.hidden_size, vc.num_attention_heads, vc.patch_size, vc.layer_norm_eps
        self.hd = self.D // self.H
        self.G = w["grid"]
        self.T = 1 + self.G * self.G
        self.keep = 4  # hidden states the path consumes: -1..-4 (groma.py:224,240,312)
        self.graphs = GraphPool()

    def forward(self, images):
        """images f32 [bs,3,S,S] -> list of the last `keep` hidden states, each f32 [bs,T,D] (pre final-LN, T7).
        bf16 models: the launch sequence is replayed from a captured hipGraph once a batch shape repeats (GraphPool)."""
        w, ws = self.w, self.ws
        bs = images.shape[0]
        D, T, G, H, hd = self.D, self.T, self.G, self.H, self.hd
        M = bs * T
        nl = len(w["layers"])
        fp8 = w["fp8"]
        Tp = _ru(T, 64)
        I = w["layers"][0]["w1"][0].shape[0]
        # every buffer the launches address, taken up front: the zero-padding bookkeeping of k / vt runs on every call
        # (replayed or not) and the addresses key the graph
        ops._chk(images, F32, "images")
        graph = TRACE is None and GraphPool.enabled
        if graph:  # the caller's tensor moves from call to call: stage it
            img = ws.get("vit_img", images.shape, F32)
            img.copy_(images)
        else:
            img = images
        a = ws.get("vit_patch", h16(bs * G * G, w["Kpad"]), H16())
        kept = [ws.get(f"vit_h{i}", (bs, T, D), F32) for i in range(self.keep)]
        scratch = [ws.get(f"vit_s{i}", (bs, T, D), F32) for i in range(2)]
        mid = ws.get("vit_mid", (bs, T, D), F32)
        q = None if Q_IN_PLACE else ws.get("vit_q", h16(bs, H, T, hd), H16())
        k = ws.get("vit_k", h16(bs, H, Tp, hd), H16(), zero=True)
        vt = ws.get("vit_vt", h16(bs, H, hd, Tp), H16(), zero=True)
        x_b = ws.get("vit_x", h16(M, D), H16())
        qkv_b = ws.get("vit_qkv", h16(M, 3 * D), H16())
        ctx_b = ws.get("vit_ctx", h16(M, D), H16())
        y_b = ws.get("vit_y", h16(M, I), H16())
        # split-K partial sums of the plan-split GEMMs (latency plan): an arena of our own whose address keys the graph, never
        # the shared eager arena of ops (which must not be born or regrown inside a capture)
        n_sk = ops.plan_ws_elems(M, [(3 * D, D), (D, D), (I, D), (D, I)])
        skw = ws.get("vit_splitk", (n_sk,), F32) if n_sk else None
        # e4m3 operands: the quantised rows and their scales live in arenas too (round 5: the e4m3 launch sequence is captured like
        # the 16-bit one; rounds 3-4 allocated them per launch and ran eagerly)
        q8 = {K_: (ws.get(f"vit_q8_{K_}", (M, K_), ops.FP8), ws.get(f"vit_s8_{K_}", (M,), F32)) for K_ in ((D, I) if fp8 else ())}

        def out_buf(layer_out_index):  # hidden_states index (0 = embeddings ... nl = last layer)
            j = layer_out_index - (nl + 1 - self.keep)
            return kept[j] if j >= 0 else scratch[layer_out_index & 1]

        def lin(x_f32, ln_g, ln_b, wt, tag=None, **kw):
            """LayerNorm -> GEMM (bf16 operands, or e4m3 operands with dynamic per-row activation scales)"""
            if fp8:
                x8, sx = ops.norm_fp8(x_f32, ln_g, ln_b, self.eps, False, out=q8[x_f32.shape[-1]])
                _trace_q8(tag, x8, sx)
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], **kw)
            x = ops.layernorm(x_f32, ln_g, ln_b, self.eps, out_bf16=True, out=x_b)
            if tag:
                _trace(tag, x)
            return ops.gemm(x, wt[0], split_ws=skw, **kw)

        def lin_bf16(x_bf16, wt, tag=None, **kw):
            if fp8:
                x8, sx = ops.quant_rows_fp8(x_bf16, out=q8[x_bf16.shape[-1]])
                _trace_q8(tag, x8, sx)
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], **kw)
            return ops.gemm(x_bf16, wt[0], split_ws=skw, **kw)

        def launch():
            ops.patchify(img, self.P, w["Kpad"], out=a)
            h = out_buf(0)
            ops.fill_rows(w["cls_pos0"], h, bs, T * D)
            ops.gemm(a, w["patch_w"], bias=w["patch_b"], resid=w["pos_patch"], resid_mod=G * G, out=h, out_f32=True,
                     row_map=(G * G, T, 1))
            for i, L in enumerate(w["layers"]):
                t0 = TRACE is not None and i == 0
                if t0:
                    _trace("vit0.h_in", h)
                qkv = lin(h, L["ln1_g"], L["ln1_b"], L["wqkv"], tag="vit0.ln1" if t0 else None, bias=L["bqkv"], out=qkv_b)
                ops.qkv_split(qkv, q, k, vt, B=bs, H=H, L=T, hd=hd)
                if Q_IN_PLACE:  # attention reads q straight from the fused projection
                    ctx = ops.attention(qkv, k, vt, Skv=T, causal=False, out=ctx_b, fused=dict(B=bs, H=H, Lq=T, hd=hd))
                else:
                    ctx = ops.attention(q, k, vt, Skv=T, causal=False, out=ctx_b)
                lin_bf16(ctx, L["wo"], tag="vit0.ctx" if t0 else None, bias=L["bo"], scale=L["ls1"], resid=h, out=mid,
                         out_f32=True)
                y = lin(mid, L["ln2_g"], L["ln2_b"], L["w1"], tag="vit0.ln2" if t0 else None, bias=L["b1"], act=1, out=y_b)
                hn = out_buf(i + 1)
                lin_bf16(y, L["w2"], tag="vit0.fc1" if t0 else None, bias=L["b2"], scale=L["ls2"], resid=mid, out=hn,
                         out_f32=True)
                if t0:
                    _trace("vit0.qkv", qkv), _trace("vit0.ctx", ctx), _trace("vit0.mid", mid), _trace("vit0.fc1", y)
                    _trace("vit0.out", hn)
                h = hn

        if graph:
            bufs = [img, a, mid, k, vt, x_b, qkv_b, ctx_b, y_b] + kept + scratch + ([] if q is None else [q]) + ([] if skw is None else [skw]) \
                + [t for pair in q8.values() for t in pair]
            self.graphs.run(("vit", bs, ops._PLAN[0], fp8) + tuple(t.data_ptr() for t in bufs), launch)
        else:
            launch()
        first = nl + 1 - self.keep
        return [out_buf(j) for j in range(max(first, 0), nl + 1)]


# ------------------------------------------------------------------------------------------------ DDETR proposer
class ProposerEngine:
    def __init__(self, w, cfg, ws):
        self.w, self.ws = w, ws
        dc = cfg.perceiver_cfg.ddetr_cfg
        self.dc = dc
        self.d, self.g = w["d"], w["g"]
        self.heads_e, self.heads_d = dc.encoder_attention_heads, dc.decoder_attention_heads
        if self.d // self.heads_e != 32 or self.d // self.heads_d != 32:
            raise NotImplementedError("MSDA / decoder attention kernels are built for head_dim 32")

    @staticmethod
    def _mlp3(x, layers):
        x = ops.gemm_f32(x, layers[0][0], bias=layers[0][1], act=2)
        x = ops.gemm_f32(x, layers[1][0], bias=layers[1][1], act=2)
        return ops.gemm_f32(x, layers[2][0], bias=layers[2][1])

    def forward(self, hidden4, debug=None):
        """hidden4: the last 4 ViT hidden states (f32 [bs,T,D]).  Returns pred_boxes f32 [bs,Q,4] (cxcywh),
        fused scores f32 [bs,Q], topk_idx int32 [bs,Q]."""
        w, dc = self.w, self.dc
        bs, T, D = hidden4[0].shape
        g, d = self.g, self.d
        HW = g * g
        x_in = ops.mean4_tokens(*hidden4)
        src = ops.gemm_f32(x_in, w["proj_w"], bias=w["proj_b"])
        x = ops.layernorm(src, w["proj_ln"][0], w["proj_ln"][1], 1e-6)
        src = x
        for L in w["enc"]:
            a = L["att"]
            qin = ops.add_rows(x, w["pos"], b_mod=HW)
            offw = ops.gemm_f32(qin, a["offw_w"], bias=a["offw_b"])
            val = ops.gemm_f32(x, a["v_w"], bias=a["v_b"])
            samp = ops.msda(val, offw, w["enc_ref"], B=bs, Q=HW, heads=self.heads_e, n_points=dc.encoder_n_points, Hs=g,
                            Ws=g, rdim=2, ref_batched=False)
            y = ops.gemm_f32(samp, a["o_w"], bias=a["o_b"])
            x = ops.layernorm(x, L["ln1"][0], L["ln1"][1], 1e-5, add=y)
            y = ops.gemm_f32(ops.gemm_f32(x, L["fc1"][0], bias=L["fc1"][1], act=2), L["fc2"][0], bias=L["fc2"][1])
            x = ops.layernorm(x, L["ln2"][0], L["ln2"][1], 1e-5, add=y)
        memory = x
        oq = ops.gemm_f32(memory, w["enc_output"][0], bias=w["enc_output"][1])
        oq = ops.layernorm(oq, w["enc_output_norm"][0], w["enc_output_norm"][1], 1e-5)
        enc_class = ops.gemm_f32(oq, w["class_enc"][0], bias=w["class_enc"][1])  # [bs*HW, 1]
        delta = self._mlp3(oq, w["bbox_enc"])
        Q = dc.two_stage_num_proposals
        idx = ops.topk_desc(enc_class.view(bs, HW), Q)
        ref0, pose = ops.ddetr_topk_gather(idx, delta, w["proposals"], B=bs, S=HW, Kq=Q, npf=d // 2)
        pt = ops.gemm_f32(pose, w["pos_trans"][0], bias=w["pos_trans"][1])
        pt = ops.layernorm(pt, w["pos_trans_norm"][0], w["pos_trans_norm"][1], 1e-5)
        qpos = pt[:, :d].contiguous()
        hs = w["target"].unsqueeze(0).expand(bs, -1, -1).reshape(bs * Q, d).contiguous()
        n = len(w["dec"])
        ref_prev = ref0
        pred = None
        for i, L in enumerate(w["dec"]):
            qk = ops.gemm_f32(ops.add_rows(hs, qpos), L["qk_w"], bias=L["qk_b"])
            v = ops.gemm_f32(hs, L["v"][0], bias=L["v"][1])
            att = ops.mha32(qk, v, B=bs, Q=Q, heads=self.heads_d, scale=32 ** -0.5)
            y = ops.gemm_f32(att, L["o"][0], bias=L["o"][1])
            hs = ops.layernorm(hs, L["ln1"][0], L["ln1"][1], 1e-5, add=y)
            a = L["att"]
            offw = ops.gemm_f32(ops.add_rows(hs, qpos), a["offw_w"], bias=a["offw_b"])
            val = ops.gemm_f32(memory, a["v_w"], bias=a["v_b"])
            samp = ops.msda(val, offw, ref0, B=bs, Q=Q, heads=self.heads_d, n_points=dc.decoder_n_points, Hs=g, Ws=g,
                            rdim=4, ref_batched=True)
            y = ops.gemm_f32(samp, a["o_w"], bias=a["o_b"])
            hs = ops.layernorm(hs, L["ln2"][0], L["ln2"][1], 1e-5, add=y)
            y = ops.gemm_f32(ops.gemm_f32(hs, L["fc1"][0], bias=L["fc1"][1], act=2), L["fc2"][0], bias=L["fc2"][1])
            hs = ops.layernorm(hs, L["ln3"][0], L["ln3"][1], 1e-5, add=y)
            # box refinement; the reference never feeds refined points back (ddetr_transformer.py:163, T3), so only
            # the last two levels reach pred_boxes (ddetr_transformer.py:696-728)
            if i == n - 2:
                ref_prev = ops.box_refine(self._mlp3(hs, w["bbox_prev"]), ref0)
            if i == n - 1:
                pred = ops.box_refine(self._mlp3(hs, w["bbox_last"]), ref_prev)
        coco = ops.gemm_f32(hs, w["class_coco"][0], bias=w["class_coco"][1])
        sa1b = ops.gemm_f32(hs, w["class_sa1b"][0], bias=w["class_sa1b"][1])
        scores = ops.score_fuse(coco, sa1b, bs * Q)
        if debug is not None:
            debug.update(enc_class=enc_class.view(bs, HW), memory=memory.view(bs, HW, d), src=src.view(bs, HW, d),
                         ref0=ref0.view(bs, Q, 4), coco=coco.view(bs, Q), sa1b=sa1b.view(bs, Q), last_hidden=hs.view(bs, Q, d))
        return pred.view(bs, Q, 4), scores.view(bs, Q), idx


# ------------------------------------------------------------------------------------------------ region encoder
class RegionEngine:
    STRIDES = (14 / 8, 14 / 4, 14 / 2)  # groma/model/roi_align.py:204 (2x off the true strides: T2, reproduced)

    def __init__(self, w, cfg, ws):
        self.w, self.ws, self.rc = w, ws, cfg.region_cfg
        self.D = cfg.perceiver_cfg.vis_encoder_cfg.hidden_size
        self.G = cfg.image_size // cfg.perceiver_cfg.vis_encoder_cfg.patch_size
        self.img = cfg.image_size
        if self.rc.num_levels != 3:
            raise NotImplementedError("3 pyramid levels (reference: MLVLROIQueryModule(num_levels=3))")
        self.graphs = GraphPool(cap=4)

    def fuse(self, hidden3):
        """MLVLFuseModule: 3 ViT hidden states -> 3 NHWC bf16 maps [bs,S,S,D], S = 4G, 2G, G.
        The ~100 launches of the pyramid (shape-static, no host value in any argument) are replayed from a captured hipGraph once a
        batch shape repeats, like the ViT / LLaMA-prefill layers (GraphPool): the inputs are the ViT's arena views and the outputs
        the `reg_feat*` arenas, whose addresses key the graph; everything allocated in between belongs to the graph's pool."""
        ws, rc, D, G = self.ws, self.rc, self.D, self.G
        bs = hidden3[0].shape[0]
        S = [G * 4, G * 2, G]
        feats = [ws.get(f"reg_feat{l}", h16(bs, S[l], S[l], D), H16()) for l in range(3)]
        pads = [ws.get(f"reg_pad{l}", h16(bs, S[l] + 2, S[l] + 2, D), H16(), zero=True) for l in range(3)]
        if self.w["fp8"]:
            pads += [ws.get(f"reg_pad8{l}", (bs, S[l] + 2, S[l] + 2, D), ops.FP8, zero=True) for l in range(3)]
        bufs = list(hidden3) + feats + pads + [ws.get(f"reg_in{l}", h16(bs * S[l] * S[l], D), H16()) for l in range(3)] + \
            [ws.get(f"reg_conv{l}_{r}", h16(bs * S[l] * S[l], D), H16()) for l in range(3) for r in range(min(2, rc.num_fuse))]
        # every temporary of the launch sequence lives in an arena too (round 5): the packed inputs of the 1x1 convs
        # (0.65 GB at 14 images), the GroupNorm partial sums / coefficients and the split-K workspace of the plan-split GEMMs --
        # none of them is born inside a capture any more, so a captured batch shape pins nothing in a private graph pool
        bufs += self._fuse_temps(bs, S)
        def launch():
            with ops.gemm_yield(YIELD_PYRAMID):
                self._fuse_launch(hidden3)
        self.graphs.run(("fuse", bs, ops._PLAN[0], YIELD_PYRAMID) + tuple(t.data_ptr() for t in bufs), launch)
        return feats, S

    def _fuse_temps(self, bs, S):
        """the arena views _fuse_launch works in besides its maps (the same views on every call with the same shapes)"""
        ws, rc, D, Cpad = self.ws, self.rc, self.D, self.w["Cpad"]
        t = {}
        for l in range(3):
            HW = S[l] * S[l]
            t[f"up{l}"] = ws.get(f"reg_up{l}", h16(bs * HW, Cpad), H16())
            for r in range(min(2, rc.num_fuse)):   # the coefficients of round r are read while round r + 1 writes its own
                t[f"sums{l}_{r}"] = ws.get(f"reg_gns{l}_{r}", (bs, ops.gn_stats_blocks(HW), D, 2), F32)
                t[f"coef{l}_{r}"] = ws.get(f"reg_gnc{l}_{r}", (bs, 2, D), F32)
        n_sk = max(ops.plan_ws_elems(bs * S[l] * S[l], [(D, Cpad)]) for l in range(3))
        if n_sk:
            t["splitk"] = ws.get("reg_splitk", (n_sk,), F32)
        self._ft = t
        return list(t.values())

    def _fuse_launch(self, hidden3):
        w, ws, rc, D, G = self.w, self.ws, self.rc, self.D, self.G
        bs = hidden3[0].shape[0]
        S = [G * 4, G * 2, G]
        maps, sums = [], [None, None, None]
        ft = self._ft   # (set by fuse() for exactly these shapes)
        for l in range(3):
            a = _trace(f"reg.up{l}", ops.upsample_coord_pack(hidden3[l], G, S[l], w["Cpad"], out=ft[f"up{l}"]))
            maps.append(_trace(f"reg.in{l}", ops.gemm(a, w["in_w"][l], bias=w["in_b"][l], split_ws=ft.get("splitk"),
                                                      out=ws.get(f"reg_in{l}", h16(bs * S[l] * S[l], D), H16()))))
        for r in range(rc.num_fuse):
            new_maps, new_coef = [], []
            for l in range(3):
                top, dow = min(l + 1, 2), max(l - 1, 0)
                out = ws.get(f"reg_conv{l}_{r & 1}", h16(bs * S[l] * S[l], D), H16())
                srcs = ((maps[l], sums[l], S[l]), (maps[top], sums[top], S[top]), (maps[dow], sums[dow], S[dow]))
                fw = w["fuse"][r]
                if "w8" in fw:  # fp8 mode, round >= 1: the shuffled relu(GroupNorm(.)) map is written as e4m3 with the static scale
                    pad = ws.get(f"reg_pad8{l}", (bs, S[l] + 2, S[l] + 2, D), ops.FP8, zero=True)  # of weights.conv_act_scale
                    ops.fuse_shuffle(*srcs, pad, imgs=bs, C=D, shuffle=True, pad=1, q_inv=fw["q_inv"])
                    ops.gemm(pad, fw["w8"], conv=(bs, S[l], S[l], D, 0), out=out, w_scale=fw["ws8"])
                    if r == 1 and TRACE is not None:  # (tests/test_fp8_width_gpu.py: the quantiser and the e4m3 conv, teacher-forced)
                        _trace(f"reg8.map{l}", maps[l]), _trace(f"reg8.coef{l}", sums[l])
                        _trace(f"reg8.pad{l}", pad), _trace(f"reg8.conv{l}", out)
                else:
                    pad = ws.get(f"reg_pad{l}", h16(bs, S[l] + 2, S[l] + 2, D), H16(), zero=True)
                    ops.fuse_shuffle(*srcs, pad, imgs=bs, C=D, shuffle=True, pad=1)
                    ops.gemm(pad, fw["w"], conv=(bs, S[l], S[l], D, 0), out=out)
                if r == 0:
                    _trace(f"reg.pad{l}", pad), _trace(f"reg.conv{l}", out)
                new_maps.append(out)
                # GN of THIS round's conv (fuse_convs[r].gn), applied where the map is consumed next
                new_coef.append(ops.gn_coef(out, bs, S[l] * S[l], D, rc.gn_groups, w["fuse"][r]["g"], w["fuse"][r]["b"], 1e-5,
                                            sums=ft[f"sums{l}_{r & 1}"], coef=ft[f"coef{l}_{r & 1}"]))
            maps, sums = new_maps, new_coef
        feats = []
        for l in range(3):
            f = ws.get(f"reg_feat{l}", h16(bs, S[l], S[l], D), H16())
            ops.fuse_shuffle((maps[l], sums[l], S[l]), None, None, f, imgs=bs, C=D, shuffle=False, pad=0)
            if rc.num_fuse == 1:  # then maps[l] is the traced round-0 conv output: feat = ReLU(GN(conv))
                _trace(f"reg.feat{l}", f)
            if w["fp8"]:
                _trace(f"reg8.feat{l}", f)
            feats.append(f)

    def extract(self, feats, S, boxes, img_idx):
        """MlvlRoIExtractor: boxes f32 [R,4] normalised cxcywh (device), img_idx f32 [R] -> region tokens f32 [R, T]."""
        w, ws, rc, D = self.w, self.ws, self.rc, self.D
        R = boxes.shape[0]
        P = rc.roi_size
        rois = torch.cat([img_idx[:, None], boxes * float(self.img)], dim=1).contiguous()  # (idx, "x1,y1,x2,y2") -- T1
        fp8 = w["fp8"]
        tiles = ws.get("reg_tiles8", (3, R, P + 2, P + 2, D), ops.FP8, zero=True) if fp8 else \
            ws.get("reg_tiles", h16(3, R, P + 2, P + 2, D), H16(), zero=True)
        for l in range(3):
            ops.roi_align_pack(feats[l], rois, tiles[l], C=D, H=S[l], W=S[l], ph=P, pw=P, spatial_scale=1.0 / self.STRIDES[l],
                               sampling_ratio=2, aligned=True, pad=1, q_inv=w["pconv_q_inv"] if fp8 else None)
        pc_out = ws.get("reg_pc", h16(R * P * P, D), H16())
        conv = (R, P, P, D, R * (P + 2) * (P + 2) * D)
        if fp8:  # e4m3 tiles x e4m3 weights; the tiles' static scale rides in w_scale (weights.pack_region)
            pc = ops.gemm(tiles, w["pconv_w8"], bias=w["pconv_b"], act=2, conv=conv, out=pc_out, w_scale=w["pconv_ws8"])
        else:
            pc = ops.gemm(tiles, w["pconv_w"], bias=w["pconv_b"], act=2, conv=conv, out=pc_out)
        _trace("reg.rois", rois), _trace("reg.tiles", tiles), _trace("reg.pc", pc)
        # pos_embedd(rois) on the UNSCALED cxcywh boxes (roi_align.py:278)
        b16 = torch.zeros((R, 16), dtype=F32, device=boxes.device)
        b16[:, :4] = boxes
        pe = ops.gemm_f32(b16, w["pe0_w"], bias=w["pe0_b"], act=2)
        pe = ops.layernorm(pe, w["pe_ln1"][0], w["pe_ln1"][1], 1e-5)
        pe = ops.gemm_f32(pe, w["pe3_w"], bias=w["pe3_b"], act=2)
        pe = ops.layernorm(pe, w["pe_ln2"][0], w["pe_ln2"][1], 1e-5)
        K = P * P * D
        splits = max(1, min(32, K // 4096))
        fl = ops.gemm(pc.view(R, K * ops.SP()), w["flat_w"], bias=w["flat_b"], resid=pe, splits=splits)
        _trace("reg.pe", pe), _trace("reg.fl", fl)
        return _trace("reg.out", ops.gemm(fl, w["up_w"], bias=w["up_b"], out_f32=True))


# ------------------------------------------------------------------------------------------------ LLaMA
class KVCache:
    """Device KV cache: K [bs,H,Smax,hd], V transposed [bs,H,hd,Smax] per layer.  Indexing gives the reference's legacy
    tuple view: cache[l][0].shape == [bs, H, S, hd] (groma/model/groma.py:377-378, groma/serve/model_worker.py:298)."""

    def __init__(self, n_layers, bs, H, hd, smax, device):
        self.k = [torch.zeros(h16(bs, H, smax, hd), dtype=H16(), device=device) for _ in range(n_layers)]
        self.vt = [torch.zeros(h16(bs, H, hd, smax), dtype=H16(), device=device) for _ in range(n_layers)]
        self.seq_len, self.smax, self.bs = 0, smax, bs
        self.sp = ops.SP()  # 2: (hi, lo) operand pairs, innermost extents doubled (precision "ref")
        self._addr = None

    def __len__(self):
        return len(self.k)

    def __getitem__(self, l):
        S = self.seq_len
        if self.sp == 2:  # reference-precision cache: hand out the values the pairs stand for (f32)
            return (ops.unsplit(self.k[l])[:, :, :S], ops.unsplit(self.vt[l])[:, :, :, :S].transpose(2, 3))
        return (self.k[l][:, :, :S], self.vt[l][:, :, :, :S].transpose(2, 3))

    def __bool__(self):
        return True

    def grow(self, smax):
        if smax <= self.smax:
            return
        for l in range(len(self.k)):
            k = torch.zeros((self.bs,) + tuple(self.k[l].shape[1:2]) + (smax, self.k[l].shape[3]), dtype=self.k[l].dtype,
                            device=self.k[l].device)
            k[:, :, : self.smax] = self.k[l]
            # (capacities are multiples of 64, so the old columns are a whole-block prefix of the new row in either layout)
            v = torch.zeros(tuple(self.vt[l].shape[:3]) + (smax * self.sp,), dtype=self.vt[l].dtype, device=self.k[l].device)
            v[..., : self.smax * self.sp] = self.vt[l]
            self.k[l], self.vt[l] = k, v
        self.smax, self._addr = smax, None


def cache_addresses(cache):
    """every layer's K / V^T address (+ capacity) of a KVCache-like object (KVCache, serving's row view): part of the key of a
    captured prefill graph.  Memoised on the object; KVCache.grow() drops the memo."""
    a = getattr(cache, "_addr", None)
    if a is None:
        a = cache._addr = (cache.smax,) + tuple(t.data_ptr() for t in cache.k) + tuple(t.data_ptr() for t in cache.vt)
    return a


class LlamaEngine:
    def __init__(self, w, cfg, ws, prec=None):
        """prec: None, or the operand type of each of the stack's three stages {"attn", "mlp", "head"} (round 6,
        groma_amd.groma.parse_precision).  A stage's kernels run under ops.precision(its type) on weights packed in that storage;
        the stages exchange the fp32 residual stream only, so nothing is converted between them.  The KV cache belongs to "attn"."""
        self.w, self.ws, self.prec = w, ws, prec
        lc = cfg.llm_cfg
        self.T, self.H, self.eps, self.I = lc.hidden_size, lc.num_attention_heads, lc.rms_norm_eps, lc.intermediate_size
        self.hd = self.T // self.H
        self.V, self.Vpad = w["V"], w["Vpad"]
        self.V0 = lc.vocab_size
        self.graphs = GraphPool()

    def embed(self, ids, out=None):
        return ops.embed_gather(ids.reshape(-1).contiguous(), self.w["embed"], self.w["new_embed"], out=out)

    def _st(self, stage):
        """the operand type of `stage` active for the kernels launched inside (no-op for a single-type stack)"""
        import contextlib
        return ops.precision(self.prec[stage]) if self.prec is not None else contextlib.nullcontext()

    def new_cache(self, bs, smax, device):
        with self._st("attn"):   # K / V^T are the attention stage's operands
            return KVCache(len(self.w["layers"]), bs, self.H, self.hd, _ru(smax, 64), device)

    def forward(self, h, bs, L, cache, kv_len=None, all_logits=True, pos_dev=None, pos_stride=0, states=None):
        """h: f32 [bs*L, T] input embeddings (consumed as the residual stream, updated in place).
        states: a list that receives a copy of the residual stream in front of every layer (HF `all_hidden_states`, R:
        groma/model/groma.py:389-397 with output_hidden_states=True); the launches then run eagerly.
        Appends L positions to `cache`.  Returns logits f32 [bs, L or 1, V] (view into a Vpad-wide buffer).
        pos_dev (i32, device): the append position is read on the device (row b: pos_dev[b*pos_stride]) instead of
        cache.seq_len -- no host value enters the kernel arguments, so the step can be captured in a hipGraph; the
        caller then owns the position counter and must have sized the cache."""
        w, ws, T, H, hd = self.w, self.ws, self.T, self.H, self.hd
        M = bs * L
        dyn = pos_dev is not None
        past = 0 if dyn else cache.seq_len
        if not dyn and past + L > cache.smax:
            cache.grow(_ru(past + L + 64, 64))
        # (e4m3 weights, round 5: the streams read the e4m3 bytes on the matrix unit -- csrc/gemv_fp8.hip, half the HBM traffic per
        #  token; the quantised operand is staged as M x K bytes in LDS, plus the merged context as 16-bit values in the o-proj)
        fits8 = not w["fp8"] or (_ru(bs, 4) * (max(T, self.I) + 16) <= 128 * 1024 and _ru(bs, 4) * (3 * T + 16) <= 128 * 1024 and T <= 4096
                                 and max(T, self.I) <= 12288 and T % 128 == 0 and self.I % 128 == 0 and self.Vpad % 16 == 0
                                 and self.hd % 32 == 0)   # (the e4m3 QKV stream rotates heads in 32-dim halves)
        if FUSED_DECODE and L == 1 and M <= 8 and fits8 and T <= 8192 and cache.smax <= 8192 and ops.SP() == 1 and states is None \
                and self.prec is None:
            return self._decode_forward(h, bs, cache, kv_len, pos_dev, pos_stride, past)
        if WIDE_DECODE and L == 1 and 8 < M <= 64 and ops.SP() == 1 and self.prec is None and states is None \
                and T % 32 == 0 and self.I % 32 == 0 and cache.smax <= 8192 and (not w["fp8"] or (T % 128 == 0 and self.I % 128 == 0)):
            return self._decode_forward_wide(h, bs, cache, kv_len, pos_dev, pos_stride, past)
        # A decode step that does not fit the weight-streaming paths (more than 64 rows, pair operands, e4m3 beyond 8 rows, > 8192 keys)
        # runs the general kernels below.  Its scratch may be baked into GreedyDecoder's captured hipGraph, so -- like _decode_forward --
        # it uses dedicated never-moved tensors (a later, larger prefill regrows the shared arenas and would free memory the
        # graph still addresses) and leaves its logits in the same `dec_logits` buffer the sampler reads.
        dec = dyn or L == 1

        def buf(name, shape, dtype):
            return ws.get(name + "_dec", shape, dtype, exact=True) if dec else ws.get(name, shape, dtype)

        fp8 = w["fp8"]
        st = self._st
        with st("attn"):
            q = None if Q_IN_PLACE else buf("llm_q", h16(bs, H, L, hd), H16())
            x_b, qkv_b = buf("llm_x", h16(M, T), H16()), buf("llm_qkv", h16(M, 3 * T), H16())
            ctx_b = buf("llm_ctx", h16(M, T), H16())
        with st("mlp"):
            x2_b = buf("llm_x2", h16(M, T), H16()) if self.prec is not None else x_b
            y_b = buf("llm_y", h16(M, self.I), H16())
        with st("head"):
            xh_b = buf("llm_xh", h16(M, T), H16()) if self.prec is not None else x_b
        n_sk = ops.plan_ws_elems(M, [(3 * T, T), (T, T), (2 * self.I, T), (T, self.I)]) if M > 8 else 0
        skw = buf("llm_splitk", (n_sk,), F32) if n_sk else None  # caller-owned split-K workspace: see VitEngine.forward
        # a prefill whose shape and memory repeat is replayed from a captured hipGraph (GraphPool): everything the launches
        # bake in goes into the key; the ragged-row lengths are staged into a buffer of our own
        graph = not dec and TRACE is None and GraphPool.enabled and states is None
        q8 = {K_: (buf(f"llm_q8_{K_}", (M, K_), ops.FP8), buf(f"llm_s8_{K_}", (M,), F32)) for K_ in ((T, self.I) if fp8 else ())}
        if graph and kv_len is not None:
            kvl = ws.get("llm_kvlen", (bs,), I32)
            kvl.copy_(kv_len)
            kv_len = kvl

        def lin(x_f32, gain, wt, tag=None, xb=None, **kw):
            """RMSNorm -> GEMM (bf16 operands, or e4m3 operands with dynamic per-row activation scales)"""
            if fp8:
                x8, sx = ops.norm_fp8(x_f32, gain, None, self.eps, True, out=q8[x_f32.shape[-1]])
                _trace_q8(tag, x8, sx)
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], **kw)
            x = ops.rmsnorm(x_f32, gain, self.eps, out=x_b if xb is None else xb)
            if tag:
                _trace(tag, x)
            return ops.gemm(x, wt[0], split_ws=skw, **kw)

        def lin_bf16(x_bf16, wt, tag=None, **kw):
            if fp8:
                x8, sx = ops.quant_rows_fp8(x_bf16, out=q8[x_bf16.shape[-1]])
                _trace_q8(tag, x8, sx)
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], **kw)
            return ops.gemm(x_bf16, wt[0], split_ws=skw, **kw)

        def launch():
            for i, Lw in enumerate(w["layers"]):
                t0 = TRACE is not None and i == 0
                if t0:
                    _trace("llm0.h_in", h)
                if states is not None:
                    states.append(h.view(bs, L, T).clone())
                with st("attn"):
                    qkv = lin(h, Lw["n1"], Lw["wqkv"], tag="llm0.n1" if t0 else None, out=qkv_b)
                    ops.qkv_split(qkv, q, cache.k[i], cache.vt[i], B=bs, H=H, L=L, hd=hd, pos0=past,
                                  cos=w["cos"], sin=w["sin"], pos_dev=pos_dev, pos_stride=pos_stride)
                    att_kw = dict(Skv=cache.smax if dyn else past + L, causal=True, q_pos0=past, kv_len=kv_len,
                                  out=ctx_b, pos_dev=pos_dev, pos_stride=pos_stride)
                    if Q_IN_PLACE:  # q is read (and rotated) in place: no packed q copy, no round trip
                        ctx = ops.attention(qkv, cache.k[i], cache.vt[i],
                                            fused=dict(B=bs, H=H, Lq=L, hd=hd, cos=w["cos"], sin=w["sin"]), **att_kw)
                    else:
                        ctx = ops.attention(q, cache.k[i], cache.vt[i], **att_kw)
                    if t0:
                        _trace("llm0.qkv", qkv), _trace("llm0.ctx", ctx)
                    lin_bf16(ctx, Lw["wo"], tag="llm0.ctx" if t0 else None, resid=h, out=h, out_f32=True)
                if t0:
                    _trace("llm0.h_attn", h)
                with st("mlp"):
                    y = lin(h, Lw["n2"], Lw["wgu"], tag="llm0.n2" if t0 else None, xb=x2_b, act=3, out=y_b)
                    if t0:
                        _trace("llm0.act", y)
                    lin_bf16(y, Lw["wd"], tag="llm0.act" if t0 else None, resid=h, out=h, out_f32=True)
                if t0:
                    _trace("llm0.h_out", h)

        if graph:
            bufs = [h, x_b, x2_b, qkv_b, ctx_b, y_b] + ([] if q is None else [q]) + ([] if kv_len is None else [kv_len]) + ([] if skw is None else [skw]) \
                + [t for pair in q8.values() for t in pair]
            self.graphs.run(("llm", bs, L, past, kv_len is None, ops._PLAN[0], fp8, ops.SP()) + tuple(t.data_ptr() for t in bufs)
                            + cache_addresses(cache), launch)
        else:
            launch()
        if not dyn:
            cache.seq_len = past + L
        with st("head"):
            return self._head(h, bs, L, xh_b, all_logits)

    def _head(self, h, bs, L, x_b, all_logits):
        """a21: final RMSNorm -> lm_head (+) extra_lm_head (under the head stage's operand type)"""
        w, T, fp8 = self.w, self.T, self.w["fp8"]
        hn = ops.rmsnorm(h, w["norm"], self.eps, out=x_b)
        if len(w["layers"]) == 1:
            _trace("llm.final_norm", hn)
        if fp8 and "head8" in w:  # a21 in e4m3: the head reads the final norm quantised straight from fp32, like every other norm -> GEMM
            def head(rows_f32, **kw):
                x8, sx = ops.norm_fp8(rows_f32, w["norm"], None, self.eps, True)
                _trace_q8("llm.head_in", x8, sx)
                return ops.gemm(x8, w["head8"][0], a_scale=sx, w_scale=w["head8"][1], out_f32=True, **kw)
            if L == 1:
                logits = head(h, out=self.decode_logits(bs))
                return logits.view(bs, 1, self.Vpad)[:, :, : self.V], hn
            if not all_logits:
                logits = head(h.view(bs, L, T)[:, -1].contiguous())
                return logits.view(bs, 1, self.Vpad)[:, :, : self.V], hn.view(bs, L, -1)[:, -1].contiguous()
            logits = head(h)
            return logits.view(bs, L, self.Vpad)[:, :, : self.V], hn
        if L == 1:  # decode step: the sampler (GreedyDecoder._step, serving) reads this buffer
            logits = ops.gemm(hn, w["head"], out_f32=True, out=self.decode_logits(bs))
            return logits.view(bs, 1, self.Vpad)[:, :, : self.V], hn
        if not all_logits and L > 1:
            hn = hn.view(bs, L, -1)[:, -1].contiguous()   # (2T physical columns per row in the operand-pair build)
            logits = ops.gemm(hn, w["head"], out_f32=True)
            return logits.view(bs, 1, self.Vpad)[:, :, : self.V], hn
        # prefill logits are handed to the caller (GromaModel.forward returns them): a fresh tensor per call, never a
        # recycled arena view (the caching allocator re-serves the block once the caller drops the previous result)
        logits = ops.gemm(hn, w["head"], out_f32=True)
        return logits.view(bs, L, self.Vpad)[:, :, : self.V], hn

    def decode_logits(self, bs):
        """the padded f32 [bs, Vpad] buffer every single-position step (either path) leaves its logits in: a dedicated tensor,
        so a captured decode graph and its sampler keep addressing the same memory"""
        return self.ws.get("dec_logits", (bs, self.Vpad), F32, exact=True)

    def _decode_forward_wide(self, h, bs, cache, kv_len, pos_dev, pos_stride, past):
        """One new position per row for 9..64 rows (round 6: continuous batching past the 8-row streams).  Every weight matrix is
        still read ONCE per step -- csrc/gemm_skinny.hip puts the weights through the matrix unit against up to 64 batch rows -- so a
        step costs about what an 8-row step does and the tokens per second scale with the rows.  7 launches per layer: RMSNorm, QKV,
        RoPE + cache write, single-query attention, o-proj (+ residual), RMSNorm, gate/up (+ SwiGLU), down (+ residual); the norms are
        kernels of their own here (a 64-row operand no longer fits a workgroup's prologue).  Never-moved buffers: the step is captured."""
        w, ws, T, H, hd = self.w, self.ws, self.T, self.H, self.hd
        dyn = pos_dev is not None
        fp8 = w["fp8"]
        x = ws.get("decw_x", (bs, T), H16(), exact=True)
        qkv = ws.get("decw_qkv", (bs, 3 * T), H16(), exact=True)
        q = ws.get("decw_q", (bs, H, 1, hd), H16(), exact=True)
        ctx = ws.get("decw_ctx", (bs, T), H16(), exact=True)
        y = ws.get("decw_y", (bs, self.I), H16(), exact=True)
        # e4m3 models (csrc/gemm_skinny_fp8.hip): operands quantised per row exactly as the prefill forms them -- a normalisation output
        # straight from fp32 (gr_norm_fp8), a stored 16-bit activation through the row quantiser -- into never-moved buffers
        q8 = {K_: (ws.get(f"decw_q8_{K_}", (bs, K_), ops.FP8, exact=True), ws.get(f"decw_s8_{K_}", (bs,), F32, exact=True))
              for K_ in ((T, self.I) if fp8 else ())}

        def lin(h_f32, gain, wt, **kw):       # RMSNorm -> GEMM
            if fp8:
                x8, sx = ops.norm_fp8(h_f32, gain, None, self.eps, True, out=q8[T])
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], tile=3, **kw)
            return ops.gemm(ops.rmsnorm(h_f32, gain, self.eps, out=x), wt[0], tile=3, **kw)

        def lin16(a16, wt, **kw):             # stored 16-bit activation -> GEMM
            if fp8:
                x8, sx = ops.quant_rows_fp8(a16, out=q8[a16.shape[-1]])
                return ops.gemm(x8, wt[0], a_scale=sx, w_scale=wt[1], tile=3, **kw)
            return ops.gemm(a16, wt[0], tile=3, **kw)

        for i, Lw in enumerate(w["layers"]):
            lin(h, Lw["n1"], Lw["wqkv"], out=qkv)
            ops.qkv_split(qkv, q, cache.k[i], cache.vt[i], B=bs, H=H, L=1, hd=hd, pos0=past, cos=w["cos"], sin=w["sin"],
                          pos_dev=pos_dev, pos_stride=pos_stride)
            ops.decode_attention(q, cache.k[i], cache.vt[i], ctx, Smax=cache.smax if dyn else past + 1, q_pos0=past, kv_len=kv_len,
                                 pos_dev=pos_dev, pos_stride=pos_stride, nsplit=1)
            lin16(ctx, Lw["wo"], resid=h, out=h, out_f32=True)
            lin(h, Lw["n2"], Lw["wgu"], act=3, out=y)
            lin16(y, Lw["wd"], resid=h, out=h, out_f32=True)
        if not dyn:
            cache.seq_len = past + 1
        logits = self.decode_logits(bs)
        if fp8 and "head8" in w:
            lin(h, w["norm"], w["head8"], out_f32=True, out=logits)
            return logits.view(bs, 1, self.Vpad)[:, :, : self.V], None
        hn = ops.rmsnorm(h, w["norm"], self.eps, out=x)
        ops.gemm(hn, w["head"], out_f32=True, out=logits, tile=3)
        return logits.view(bs, 1, self.Vpad)[:, :, : self.V], hn

    def _decode_forward(self, h, bs, cache, kv_len, pos_dev, pos_stride, past):
        """One new position per row: 5 launches per layer.  Every weight matrix is ONE streaming kernel whose
        prologue builds its operand (RMSNorm of the fp32 residual rows / the merge of the attention's key slices) and whose
        epilogue is its consumer (RoPE + cache write, in-place residual update, SwiGLU): csrc/gemv_fused.hip."""
        w, ws, T, H, hd = self.w, self.ws, self.T, self.H, self.hd
        dyn = pos_dev is not None
        q = ws.get("dec_q", h16(bs, H, 1, hd), H16(), exact=True)
        ctx = ws.get("dec_ctx", h16(bs, T), H16(), exact=True)
        y = ws.get("dec_y", h16(bs, self.I), H16(), exact=True)
        ws8 = (lambda wt: dict(w_scale=wt[1])) if w["fp8"] else (lambda wt: {})   # e4m3 weights carry their per-row scale
        for i, Lw in enumerate(w["layers"]):
            t0 = TRACE is not None and i == 0
            if t0:
                _trace("dec0.h_in", h)
            ops.gemv_fused(Lw["wqkv"][0], M=bs, norm=(h, Lw["n1"], self.eps), **ws8(Lw["wqkv"]),
                           qkv=dict(q=q, k=cache.k[i], vt=cache.vt[i], cos=w["cos"], sin=w["sin"], H=H, hd=hd, pos0=past,
                                    pos_dev=pos_dev, pos_stride=pos_stride))
            att = ops.decode_attention(q, cache.k[i], cache.vt[i], ctx, Smax=cache.smax if dyn else past + 1, q_pos0=past,
                                       kv_len=kv_len, pos_dev=pos_dev, pos_stride=pos_stride)
            if t0:
                _trace("dec0.q", q)
                TRACE["dec0.nsplit"] = att[1] if isinstance(att, tuple) else 1
                if not isinstance(att, tuple):
                    _trace("dec0.ctx", ctx)
            if isinstance(att, tuple):  # key slices on separate blocks: the o-proj merges them while building its operand
                ops.gemv_fused(Lw["wo"][0], M=bs, a_parts=att, resid=h, **ws8(Lw["wo"]))
            else:
                ops.gemv_fused(Lw["wo"][0], M=bs, x=ctx, resid=h, **ws8(Lw["wo"]))
            if t0:
                _trace("dec0.h_attn", h)
            ops.gemv_fused(Lw["wgu"][0], M=bs, norm=(h, Lw["n2"], self.eps), swiglu_out=y, **ws8(Lw["wgu"]))
            if t0:
                _trace("dec0.act", y)
            ops.gemv_fused(Lw["wd"][0], M=bs, x=y, resid=h, **ws8(Lw["wd"]))
        if not dyn:
            cache.seq_len = past + 1
        if TRACE is not None and len(w["layers"]) == 1:
            _trace("dec.h_out", h)
        logits = self.decode_logits(bs)
        if w["fp8"] and "head8" in w:
            ops.gemv_fused(w["head8"][0], M=bs, norm=(h, w["norm"], self.eps), out=logits, w_scale=w["head8"][1])
        else:
            ops.gemv_fused(w["head"], M=bs, norm=(h, w["norm"], self.eps), out=logits)
        return logits.view(bs, 1, self.Vpad)[:, :, : self.V], None


class GreedyDecoder:
    """Decode arena for a fixed batch of rows: KV cache + device-resident loop state + ONE captured hipGraph of the
    per-token step (embed -> 32 layers -> lm_head -> arg-max -> HF greedy_search bookkeeping).

    Reference behaviour reproduced (HF 4.32 GenerationMixin.greedy_search around groma/model/groma.py:176-200,376-379):
    every row appends at the same position past+1 (all-ones mask over the padded prefix); finished rows
    emit `pad`; the loop ends when every row has produced `eos`.  Nothing position-dependent is a kernel argument: the
    kernels read the step position from `pos` (gr_qkv_split / gr_attention_bf16 pos_dev), so the graph is replayed
    unchanged for every token instead of ~330 launches per token going through the host."""

    def __init__(self, llm, bs, smax, max_new, eos, pad, device):
        self.llm, self.bs, self.eos, self.pad = llm, bs, eos, pad
        self.cache = llm.new_cache(bs, smax, device)
        I64, I32 = torch.int64, torch.int32
        self.tok = torch.zeros((bs,), dtype=I64, device=device)
        self.nxt = torch.zeros((bs,), dtype=I64, device=device)
        self.unfinished = torch.ones((bs,), dtype=I64, device=device)
        self.seq = torch.zeros((bs, max_new), dtype=I64, device=device)
        self.pos = torch.zeros((1,), dtype=I32, device=device)
        self.step = torch.zeros((1,), dtype=I32, device=device)
        self.n_unf = torch.zeros((1,), dtype=I32, device=device)
        self.h = torch.zeros((bs, llm.T), dtype=F32, device=device)
        # sampling state (groma/serve/model_worker.py:307-311): 1/temperature per row (0 = greedy arg-max) and the seed of the
        # counter-based draw; device tensors, so switching greedy <-> sampling never re-captures the graph
        self.inv_temp = torch.zeros((bs,), dtype=F32, device=device)
        self.seed = torch.zeros((bs,), dtype=I64, device=device)
        self.graph = None

    def set_sampling(self, temperature=0.0, seeds=None):
        self.inv_temp.fill_(0.0 if temperature is None or temperature < 1e-4 else 1.0 / float(temperature))
        if seeds is not None:
            self.seed.copy_(torch.as_tensor(seeds, dtype=I64).reshape(-1).to(self.seed.device))

    def _advance(self, inc_pos):
        ops.greedy_advance(self.nxt, self.tok, self.unfinished, self.seq, self.pos, self.step, self.n_unf,
                           eos=self.eos, pad=self.pad, inc_pos=inc_pos)

    def _step(self):
        llm = self.llm
        ops.embed_gather(self.tok, llm.w["embed"], llm.w["new_embed"], out=self.h)
        llm.forward(self.h, self.bs, 1, self.cache, pos_dev=self.pos, pos_stride=0)
        # the token sampled from the logits at position `pos` sits at pos + 1 (pos_off = 1)
        ops.sample_rows(llm.decode_logits(self.bs), llm.V, self.inv_temp, self.seed,
                        pos=self.pos, pos_stride=0, pos_off=1, out=self.nxt)
        self._advance(1)

    def capture(self):
        side = torch.cuda.Stream(device=self.tok.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up: workspace buffers, GEMV split-K scratch, function attributes
            for _ in range(3):
                self.pos.zero_()
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        self.graph = g

    def run(self, first_logits, L, max_new):
        """first_logits f32 [bs, V]: last-position logits of the prefill that filled cache[0:L).  Returns the number of
        tokens produced; they are in self.seq[:, :n]."""
        if max_new > self.seq.shape[1]:
            raise ValueError("max_new exceeds the arena")
        if self.graph is None:
            raise RuntimeError("GreedyDecoder.capture() must run before the prefill fills the arena")
        self.pos.fill_(L)
        self.step.zero_()
        self.unfinished.fill_(1)
        ops.sample_rows(first_logits.contiguous(), first_logits.shape[-1], self.inv_temp, self.seed, pos=self.pos,
                        pos_stride=0, pos_off=0, out=self.nxt)  # first new token: position L
        self._advance(0)
        n = 1
        while n < max_new:
            if self.eos is not None and int(self.n_unf.item()) == 0:
                break
            self.graph.replay()
            n += 1
        return n
