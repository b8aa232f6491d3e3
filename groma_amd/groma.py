"""GromaModel / GromaConfig -- the drop-in boundary (reference: groma/model/groma.py:31-431).

Same constructor-level names, `forward()` / `generate()` / `prepare_inputs_for_generation()` /
`init_special_token_id()` signatures, argument meaning, return packaging and error behaviour (asserts) as the
reference, so `groma/eval/*.py` and `groma/serve/model_worker.py` call it unchanged:

    model = GromaModel.from_pretrained(path).cuda(); model.init_special_token_id(tokenizer)
    out = model.generate(input_ids, images=image, use_cache=True, do_sample=False, max_new_tokens=3,
                         return_dict_in_generate=True, output_hidden_states=True,
                         generation_config=model.generation_config)
    out.sequences;  out.hidden_states[0][-1]['pred_boxes'][0]

Host code is Python; every tensor computation is a HIP kernel behind the C ABI (no eager fallback: a missing
libgroma_hip.so raises).  The host keeps only index bookkeeping the reference also does in Python
(placeholder splice, randperm, refer-box matching on a handful of boxes).
"""
import glob
import json
import os
from types import SimpleNamespace

import torch

from . import engine, ops, weights
from .config import GromaConfig
from .constants import DEFAULT_TOKENS, REGION_IDX_TOKENS, IGNORE_INDEX

F32, I32, I64 = torch.float32, torch.int32, torch.int64


class CausalLMOutputWithPast(dict):
    """Attribute + index access like transformers.modeling_outputs.CausalLMOutputWithPast."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.__dict__.update(kw)

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)


class GenerateOutput(SimpleNamespace):
    pass


def _c2c(b):  # HF center_to_corners_format
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def _box_iou(b1, b2):  # torchvision.ops.box_iou on a handful of host-side boxes
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


# precision="hybrid" / "hybrid-fp16": (operand type behind the ViT, operand type of the ViT)
HYBRID_MODES = {"hybrid": ("bf16", "ref"), "hybrid-fp16": ("fp16", "ref")}
# The operand type is a PER-STAGE choice (round 5: the ViT; round 6: every stage that exchanges fp32 tensors with its neighbours).
# A stage is a run of kernels whose inputs and outputs are fp32 (residual streams, ViT states, region / image tokens, logits), so
# nothing is converted where two stages of different operand types meet:
#   vit     DINOv2 (a1)                       region  pyramid + RoIAlign + per-ROI conv + flatten + updims (a14-a17)
#   bridge  img_txt_bridge (a3)               attn    RMSNorm -> QKV -> RoPE / KV cache -> attention -> o-proj (+ residual)
#   mlp     RMSNorm -> gate/up -> SwiGLU -> down (+ residual)          head    final RMSNorm -> lm_head (+) extra_lm_head
# precision = "<base>[+<stage>:<type>]..." with base one of bf16 | fp16 | ref | hybrid | hybrid-fp16 and type bf16 | fp16 | ref,
# e.g. "hybrid-fp16+attn:ref" (tests/diag/precision_ablation.py measures what each stage's operand rounding costs the logits).
STAGES = ("vit", "region", "bridge", "attn", "mlp", "head")
_TYPES = ("bf16", "fp16", "ref")


def parse_precision(spec, vit_precision=None):
    """-> (stage -> operand type, base type behind the ViT).  `spec`: a name as above, or a dict {stage: type, "base": type}."""
    if isinstance(spec, dict):
        base = spec.get("base", "bf16")
        table = {s: spec.get(s, base) for s in STAGES}
    else:
        name, *over = str(spec).split("+")
        base, vit = HYBRID_MODES.get(name, (name, name))
        table = {s: base for s in STAGES}
        table["vit"] = vit
        for o in over:
            stage, _, typ = o.partition(":")
            if stage not in STAGES:
                raise ValueError(f"precision {spec!r}: unknown stage {stage!r} (stages: {', '.join(STAGES)})")
            table[stage] = typ
    if vit_precision is not None:
        table["vit"] = vit_precision
    if base not in _TYPES or any(t not in _TYPES for t in table.values()):
        raise ValueError(f"precision must be 'bf16', 'fp16', 'ref', 'hybrid' or 'hybrid-fp16' (+ '<stage>:<type>' overrides), got {spec!r}")
    return table, base
SPECULATIVE_EXTRACT = True   # forward(): queue the region extraction before the NMS counts reach the host (False: tests / tests/diag A-B only)

def _img_of(counts):
    """[0] * counts[0] + [1] * counts[1] + ... as an int64 tensor.  Not torch.repeat_interleave: with a many-thread intra-op pool it
    opens a parallel region for a few hundred elements -- measured 5.2 ms against 23 us single-threaded (8 threads on 8 busy cores),
    inside the one window of a forward in which the GPU waits for the host"""
    return torch.cat([torch.full((int(c),), i, dtype=I64) for i, c in enumerate(counts)]) if len(counts) else torch.empty((0,), dtype=I64)


_entry = engine.model_entry(lambda self, *a, **kw: self.precision)  # every boundary entry point: see engine.normal_mode / ops.precision


class GromaModel:
    config_class = GromaConfig

    def __init__(self, config: GromaConfig, source=None, device="cuda", fp8=False, precision="bf16", vit_precision=None):
        self.config = config
        # 16-bit operand type of the GEMM / attention kernels and of every 16-bit buffer (KV cache, feature maps):
        # "bf16" (libgroma_hip.so: BASELINE's benchmark dtype) or "fp16" (libgroma_hip_f16.so: what the reference's inference
        # scripts autocast to, R: groma/eval/run_groma.py:82; same MFMA rate, 3 more mantissa bits).  Accumulation, residual
        # streams, norms and the proposer are fp32 either way.
        # "ref" (libgroma_hip_ref.so) is the reference-precision path: every operand is a (hi, lo) pair of halves (22 mantissa
        # bits) and every contraction -- GEMMs, implicit-GEMM convs, both attention products -- issues hi.hi + hi.lo + lo.hi
        # into the fp32 MFMA accumulators: 3x the MFMA work, within ~1e-6 of fp32 per contraction, which is what keeps the
        # 24 + 32-layer chain inside north_star's 1e-3 of the reference's fp32 forward (R: groma/eval/eval_rec.py:69 loads fp32
        # weights).  Same kernels, same launch sequence; 16-bit buffers are twice as wide.
        # "hybrid" (round 5) selects the operand type PER STAGE: the ViT -- the only 16-bit stage in front of the fp32 proposer, 724
        # of the 11 850 GFLOP of an image -- runs on operand pairs ("ref"), everything behind it (bridge, region encoder, LLaMA) on
        # bf16 ("hybrid") or fp16 ("hybrid-fp16") operands, or e4m3 with fp8=True.  The proposer then sees ViT states within
        # ~3e-6 of the reference's fp32 ones, so the INDEX-valued results of the path -- top-300 proposal ids, NMS keep ids, the
        # shuffled selection, the spliced token ids -- equal the reference's end to end, with no stage chaining
        # (R: groma/model/groma.py:222-280 in one fp32 pass; tests/test_e2e_unchained_gpu.py), at ~0.9x the bf16 throughput
        # instead of "ref"'s 0.4x.  The logits keep the 16-bit format's distance.
        self.stage_precision, precision = parse_precision(precision, vit_precision)
        vit_precision = self.stage_precision["vit"]
        behind = {s: t for s, t in self.stage_precision.items() if s != "vit"}
        if fp8 and any(t == "ref" for t in behind.values()):
            raise ValueError("fp8=True and operand pairs ('ref') behind the ViT are exclusive")
        if fp8 and len(set(behind.values())) > 1:
            raise ValueError("fp8=True takes one 16-bit type behind the ViT")
        # `precision`: the BASE operand type behind the ViT (the type the entry points switch to; what a stage without an override
        # runs on); `vit_precision`: the ViT's.  Equal except under "hybrid".  `stage_precision`: the whole table.
        self.precision, self.vit_precision = precision, vit_precision
        self.decode_graph = True  # generate(): replay one captured hipGraph per token (False = eager per-kernel launches)
        # output_hidden_states=True: False = hidden_states[0] is the 1-tuple (final normed state,) -- every reference caller only
        # reads hidden_states[-1]['pred_boxes'] and passes the flag for that (R: groma/eval/eval_rec.py:93-101), so the 33
        # per-layer copies (305 MB per image at Groma-7B) are not made by default; True = the reference's full tuple (embeddings,
        # the residual stream after layers 1..n-1, the final normed state), R: groma/model/groma.py:389-397,421-427
        self.all_hidden_states = False
        # proposer chain (input_proj -> DDETR encoder x6 -> two-stage top-300 -> decoder x6 -> heads -> score fusion -> NMS,
        # ~330 launches of fp32 kernels that are launch-latency-bound) captured once per batch size and replayed
        self.proposer_graph = True
        self.fp8 = bool(fp8)  # BASELINE configs[4]: OCP e4m3 operands for the DINOv2 / LLaMA GEMMs, lm_head and the region encoder's 3x3 convs
        # which GEMMs are split along K (ops.plan_splits): "throughput" = none (batched eval / the benchmark), "latency" = a
        # fixed per-(N, K) factor tuned for one request per call.  Either way a function of the layer shape only, so results
        # never depend on batch composition; the two plans differ from each other by fp32 summation order (bf16-noise level).
        self.gemm_plan = "throughput"
        self.device = torch.device(device)
        self.training = False
        self.pad_token_id = None
        self.img_token_id = None
        self.reg_token_id = None
        self.refer_box_token_id = None
        self.refer_feat_token_id = None
        self.ground_box_token_id = None
        self.box_idx_token_ids = None
        lc = config.llm_cfg
        self.generation_config = SimpleNamespace(pad_token_id=lc.pad_token_id, bos_token_id=lc.bos_token_id,
                                                 eos_token_id=lc.eos_token_id, do_sample=False, max_new_tokens=20)
        self._ws = None
        self._loaded = False
        if source is not None:
            self._load(source)

    # ------------------------------------------------------------------ construction / loading
    def _load(self, source):
        with ops.precision(self.precision):
            self._load_packed(source)

    def _load_packed(self, source):
        ops._lib.load()  # fail loudly if the HIP library is missing
        cfg = self.config
        self._ws = engine.Workspace(self.device)
        with ops.precision(self.vit_precision):   # ("hybrid": the ViT's weights are operand pairs, and never e4m3)
            self.vit = engine.VitEngine(weights.pack_vit(source, cfg, self.fp8 and self.vit_precision != "ref"), cfg, self._ws)
        sp = self.stage_precision
        self.proposer = engine.ProposerEngine(weights.pack_ddetr(source, cfg), cfg, self._ws)
        with ops.precision(sp["region"]):
            self.region = engine.RegionEngine(weights.pack_region(source, cfg, self.fp8), cfg, self._ws)
        with ops.precision(sp["bridge"]):
            self.bridge = weights.pack_bridge(source, cfg)
        # (one type for the whole LLaMA stack = the stage switches are no-ops and the streaming decode kernels apply)
        llm_prec = None if sp["attn"] == sp["mlp"] == sp["head"] == self.precision else {k: sp[k] for k in ("attn", "mlp", "head")}
        self.llm = engine.LlamaEngine(weights.pack_llm(source, cfg, self.fp8, prec=llm_prec), cfg, self._ws, prec=llm_prec)
        self._loaded = True

    @property
    def mode(self):
        """the name this model's per-stage operand types go by: "bf16" | "fp16" | "ref" | "hybrid" | "hybrid-fp16" (+ "+e4m3")"""
        name = next((k for k, v in HYBRID_MODES.items() if v == (self.precision, self.vit_precision)), None)
        if name is None:
            name = self.precision if self.precision == self.vit_precision else f"{self.precision}+vit:{self.vit_precision}"
        name += "".join(f"+{s}:{t}" for s, t in self.stage_precision.items() if s != "vit" and t != self.precision)
        return name + ("+e4m3" if self.fp8 else "")

    def _vit_forward(self, images):
        """a1 under the ViT's own operand type.  Its results are fp32 hidden states [bs, T, D] in either build (the residual
        stream), so nothing is converted at the hand-over: the proposer, the bridge and the region pyramid read them as they are."""
        with ops.precision(self.vit_precision):
            return self.vit.forward(images)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _scratch_cache(self, bs, L):
        c = getattr(self, "_kv_scratch", None)
        if c is None or c.bs != bs or c.smax < L:
            c = self.llm.new_cache(bs, L, self.device)
            self._kv_scratch = c
        c.seq_len = 0
        return c

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", fp8=False, precision="bf16"):
        return cls(config, weights.Source.from_state_dict(state_dict, torch.device(device)), device, fp8=fp8, precision=precision)

    @classmethod
    def from_synthetic(cls, config, seed=0, device="cuda", fp8=False, precision="bf16"):
        """Random-init weights of the configured architecture, generated on the device (benchmark path)."""
        return cls(config, weights.Source.synthetic(config, seed, torch.device(device)), device, fp8=fp8, precision=precision)

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, device="cuda", **kw):
        """Reads a reference checkpoint directory: config.json + *.safetensors / pytorch_model*.bin shards with the
        reference's parameter names (groma/eval/eval_rec.py:69).  Weights are repacked to 16-bit device layouts:
        torch_dtype=torch.float16 (what groma/eval/run_groma.py and the model worker pass) selects the fp16 operand build, an
        EXPLICIT torch_dtype=torch.float32 the reference-precision build ("ref": split operands, ~fp32 results at 3x the MFMA
        work), anything else (None / "auto" / bfloat16) bf16; `precision="bf16" | "fp16" | "ref" | "hybrid" | "hybrid-fp16"`
        overrides ("hybrid": the ViT on operand pairs so that box indices / token ids equal the fp32 reference's, the rest 16-bit).
        The selection is logged (logger "groma_amd"); the implicit float32 -> "ref" one also warns, because it costs 2.5x the
        bf16 step, twice the weight / KV bytes, and decodes through the general GEMM kernels."""
        if torch_dtype not in (None, "auto", torch.float32, torch.float16, torch.bfloat16):
            raise NotImplementedError(f"torch_dtype={torch_dtype}: the MI355X path computes with bf16 or fp16 operands (fp32 accumulate)")
        # torch_dtype states how the CALLER would have held the weights (fp32 in eval_rec.py:69, fp16 in run_groma.py): norms,
        # biases and the proposer stay fp32, the GEMM operands take the 16-bit type
        precision = kw.get("precision") or ("fp16" if torch_dtype == torch.float16 else "ref" if torch_dtype == torch.float32 else "bf16")
        import logging
        logging.getLogger("groma_amd").info("from_pretrained(%s): torch_dtype=%s -> precision=%r", path, torch_dtype, precision)
        if torch_dtype == torch.float32 and not kw.get("precision"):
            import warnings
            warnings.warn("GromaModel.from_pretrained(torch_dtype=torch.float32) selects precision='ref' (operand pairs, three MFMA passes: "
                          "~fp32 results, 2.5x the bf16 step time, 2x the weight and KV-cache bytes, no streaming decode kernels).  Pass "
                          "precision='hybrid' for reference-exact box indices / token ids at ~0.9x the bf16 speed, or precision='bf16'.",
                          stacklevel=2)
        for unsupported in ("load_in_8bit", "load_in_4bit", "quantization_config"):
            if kw.get(unsupported):
                raise NotImplementedError(f"{unsupported} is not supported by the MI355X path (bf16 / fp32 only)")
        config = GromaConfig.from_pretrained(path)
        sd = {}
        st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if st:
            from safetensors.torch import load_file
            for f in st:
                sd.update(load_file(f))
        else:
            for f in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
                sd.update(torch.load(f, map_location="cpu"))
        if not sd:
            raise FileNotFoundError(f"no weight shards under {path}")
        return cls.from_state_dict(config, sd, device, fp8=bool(kw.get("fp8", False)), precision=precision)

    def _same_device(self, device):
        if device is None:
            return
        d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if d.type != "cuda" or (d.index is not None and self.device.index is not None and d.index != self.device.index):
            raise NotImplementedError(f"weights were packed for {self.device}; load the model on the target device "
                                      f"(from_pretrained(..., device=...)) instead of moving it to {d}")

    def cuda(self, device=None):
        self._same_device(device)
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        """Device moves are rejected (the packed weights live where they were loaded); dtype requests are ignored: the
        path always computes with bf16 operands / fp32 accumulation and fp32 residual streams."""
        for x in list(a) + [k.get("device")]:
            if isinstance(x, (str, int, torch.device)):
                self._same_device(x)
        return self

    def init_special_token_id(self, tokenizer):  # groma/model/groma.py:136-144
        self.pad_token_id = tokenizer.pad_token_id
        self.img_token_id = tokenizer.convert_tokens_to_ids([DEFAULT_TOKENS['image']])[0]
        self.reg_token_id = tokenizer.convert_tokens_to_ids([DEFAULT_TOKENS['region']])[0]
        self.refer_box_token_id = tokenizer.convert_tokens_to_ids([DEFAULT_TOKENS['rbox']])[0]
        self.refer_feat_token_id = tokenizer.convert_tokens_to_ids([DEFAULT_TOKENS['rfeat']])[0]
        self.ground_box_token_id = tokenizer.convert_tokens_to_ids([DEFAULT_TOKENS['gbox']])[0]
        self.box_idx_token_ids = tokenizer.convert_tokens_to_ids(REGION_IDX_TOKENS)
        self.generation_config.pad_token_id = tokenizer.pad_token_id
        return

    @_entry
    def get_input_embeddings(self, input_ids):  # groma/model/groma.py:165-174
        bs, L = input_ids.shape
        return self.llm.embed(input_ids.to(self.device)).view(bs, L, -1)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):  # groma/model/groma.py:176-200
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({
            "past_key_values": past_key_values, "attention_mask": attention_mask, "use_cache": kwargs.get("use_cache"),
            "images": kwargs.get("images", None), "refer_boxes": kwargs.get("refer_boxes", None),
            "ground_boxes": kwargs.get("ground_boxes", None)})
        return model_inputs

    # ------------------------------------------------------------------ stages
    @_entry
    def perceive(self, images, refer_boxes=None, ground_boxes=None, debug=None):
        """Steps A-E (groma.py:218-280): ViT -> proposer -> NMS -> shuffle.  Returns (hidden4, selected_boxes list of
        device f32 [N_i,4], aux dict)."""
        images = images.to(device=self.device, dtype=F32).contiguous()
        hidden4 = self._vit_forward(images)
        selected, aux = self.propose(hidden4, refer_boxes, ground_boxes, debug)
        return hidden4, selected, aux

    @_entry
    def propose(self, hidden4, refer_boxes=None, ground_boxes=None, debug=None, seeds=None, _defer=False):
        """Steps C-E: DDETR proposer -> fused scores -> on-device NMS -> host randperm (one D2H of <1 KB/image).
        _defer=True (internal): returns (spec, finish) right after the launches -- see the comment at the copy below."""
        cfg = self.config
        bs = hidden4[0].shape[0]
        dev = self.device
        n_extra = [(refer_boxes[i].shape[0] if refer_boxes is not None else 0) +
                   (ground_boxes[i].shape[0] if ground_boxes is not None else 0) for i in range(bs)]
        if self.proposer_graph and debug is None and max(n_extra) == 0:
            # fixed shapes, no host value in any kernel argument: replay the captured chain
            pred_boxes, scores, topk_idx, keep, n_keep = self._propose_graph(hidden4)
            # the graph owns its output buffers and the next replay overwrites them: what leaves this call (aux, _last_aux,
            # the selected boxes) is the caller's own copy (KB-sized)
            pred_boxes, scores, topk_idx = pred_boxes.clone(), scores.clone(), topk_idx.clone()
            Q = pred_boxes.shape[1]
            boxes_all, scores_all = pred_boxes, scores
        else:
            pred_boxes, scores, topk_idx = self.proposer.forward(hidden4, debug=debug)
            Q = pred_boxes.shape[1]
            if refer_boxes is None:
                refer_boxes = [torch.empty((0, 4), device=dev) for _ in range(bs)]
            if ground_boxes is None:
                ground_boxes = [torch.empty((0, 4), device=dev) for _ in range(bs)]
            nmax = Q + max(n_extra)
            if max(n_extra) == 0:
                boxes_all, scores_all, n_valid = pred_boxes.contiguous(), scores.contiguous(), None
            else:
                boxes_all = torch.zeros((bs, nmax, 4), dtype=F32, device=dev)
                scores_all = torch.zeros((bs, nmax), dtype=F32, device=dev)
                boxes_all[:, :Q], scores_all[:, :Q] = pred_boxes, scores
                for i in range(bs):  # groma.py:259-264: refer boxes score 1.0, ground boxes 0.2
                    r, g = refer_boxes[i].to(dev, F32), ground_boxes[i].to(dev, F32)
                    boxes_all[i, Q:Q + len(r)], scores_all[i, Q:Q + len(r)] = r, 1.0
                    boxes_all[i, Q + len(r):Q + len(r) + len(g)] = g
                    scores_all[i, Q + len(r):Q + len(r) + len(g)] = 0.2
                n_valid = torch.tensor([Q + e for e in n_extra], dtype=I32, device=dev)
            keep, n_keep = ops.nms(boxes_all, scores_all, float(cfg.nms_thres), float(cfg.box_score_thres),
                                   int(cfg.max_region_num), n_valid=n_valid)
        nmax = boxes_all.shape[1]
        # Speculation (round 4): the host work between the NMS result and the first region-extraction launch is GPU-idle time
        # (profiles/r04_timeline_b14.txt: 1.3 ms).  The usual outcome is that every image keeps max_region_num boxes, and then the
        # CPU-RNG shuffles (T4) need no device value at all: draw them NOW, while the proposer is still running, upload them and
        # queue the two device gathers behind the NMS kernel; after the sync only the counts are checked.  If any image kept fewer
        # boxes the global RNG is put back exactly where it was and the ordinary path below runs -- same draws, same results.
        spec = self._speculate_shuffle(keep, boxes_all, bs, nmax, seeds) if max(n_extra) == 0 else None
        # ONE host round trip: kept indices + counts in one D2H copy (the reference syncs at nms / len / randperm too), the
        # CPU-RNG shuffles (T4), then ONE pinned H2D copy of the flat selection and ONE device gather -- the per-image
        # index_select / .to(device) sequence this replaces cost ~1.2 ms of idle GPU per forward in pageable synchronous copies
        kk = torch.cat([keep, n_keep.to(I64)[:, None]], dim=1)
        # _defer (round 5, forward() only): the copy goes over the copy stream into pinned memory and the caller gets `finish` back
        # BEFORE waiting for it -- it queues the speculative region extraction on the main stream first, so the GPU works through
        # the RoIAlign / per-ROI conv launches while the host waits for the counts and does its glue (the timeline showed ~1 ms of
        # idle GPU per step around this sync under the profiler, ~0.5 ms without)
        wait = self._to_host_async(kk, "kk") if (_defer and spec is not None) else None

        def finish():
            kk_h = wait().clone() if wait is not None else kk.cpu()   # (polling an event instead of a blocking copy measured no gain: profiles/r04_host_sync_ab.txt)
            return self._propose_finish(kk_h, spec, pred_boxes, scores, topk_idx, boxes_all, scores_all, nmax, Q, n_extra, seeds, bs)
        if _defer:
            return spec, finish
        return finish()

    def _propose_finish(self, kk_h, spec, pred_boxes, scores, topk_idx, boxes_all, scores_all, nmax, Q, n_extra, seeds, bs):
        """the host half of propose(): counts checked, CPU-RNG shuffles drawn (or the speculative ones accepted), selection gathered"""
        dev = self.device
        keep_h, n_keep_l = kk_h[:, :-1], kk_h[:, -1].tolist()
        if spec is not None:
            if all(int(nk) == spec["n"] for nk in n_keep_l):
                sel = keep_h.gather(1, spec["perms"])                      # [bs, n]: what keep_h[i].index_select(0, perm_i) gives
                sel_idx = list(sel.unbind(0))
                aux = dict(pred_boxes=pred_boxes, scores=scores, topk_idx=topk_idx, nms_keep=list(keep_h[:, : spec["n"]].unbind(0)),
                           sel_idx=sel_idx, boxes_cat=spec["boxes_cat"], img_idx=spec["img_idx"], spec_hit=True)
                return list(spec["boxes_cat"].split([spec["n"]] * bs)), aux
            if spec["rng"] is not None:
                torch.set_rng_state(spec["rng"])   # mis-speculated: nothing was consumed as far as the ordinary path can tell
        sel_idx = []
        if seeds is None and all(int(nk) > 0 for nk in n_keep_l):
            sel_idx = self._host_select(keep_h, n_keep_l, nmax)[0]   # (the same draws in the same order as the loop below)
        for i, nk in enumerate(n_keep_l if not sel_idx else ()):   # (a handful of host ops per image: this loop sits between the sync and the next launch)
            if nk > 0:  # groma.py:273-276 -- torch.randperm on the CPU global RNG (T4)
                if seeds is not None and seeds[i] is not None:
                    # serving: each request owns its shuffle seed.  A local generator yields exactly what the global RNG
                    # would after torch.manual_seed(seed), without reseeding the process-wide CPU / device generators
                    perm = torch.randperm(nk, generator=torch.Generator().manual_seed(int(seeds[i])))
                else:
                    perm = torch.randperm(nk)
                inds = keep_h[i].index_select(0, perm)   # perm < nk: only kept entries are read
            else:       # groma.py:277-279
                nv = Q + n_extra[i]
                inds = torch.max(scores_all[i, :nv], dim=0).indices.reshape(1).cpu()
            sel_idx.append(inds)
        n_sel = [int(x.numel()) for x in sel_idx]
        R = sum(n_sel)
        stage = self._pinned("sel", 2 * R)
        counts = torch.tensor(n_sel)
        img_of = _img_of(n_sel)
        torch.add(torch.cat(sel_idx), img_of, alpha=nmax, out=stage[:R])
        stage[R:2 * R] = img_of
        sel_dev = stage[:2 * R].to(dev, non_blocking=True)
        boxes_cat = boxes_all.reshape(bs * nmax, 4).index_select(0, sel_dev[:R])   # [R, 4], image-major
        selected = list(boxes_cat.split(n_sel))
        n_keep_h = n_keep_l
        aux = dict(pred_boxes=pred_boxes, scores=scores, topk_idx=topk_idx, nms_keep=[keep_h[i, :int(n_keep_h[i])] for i in range(bs)],
                   sel_idx=sel_idx, boxes_cat=boxes_cat, img_idx=sel_dev[R:].to(F32))
        return selected, aux

    def _speculate_shuffle(self, keep, boxes_all, bs, nmax, seeds):
        """see propose(): per-image torch.randperm(max_region_num) drawn before the NMS result is known, uploaded, and the
        selection gathered on the device; returns None when speculation is pointless (fewer candidates than max_region_num)"""
        n = int(self.config.max_region_num)
        if n <= 0 or n > nmax or keep.shape[1] != n:
            return None
        dev = self.device
        rng = torch.get_rng_state() if (seeds is None or any(s is None for s in seeds)) else None
        perms = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(int(seeds[i])))
                             if (seeds is not None and seeds[i] is not None) else torch.randperm(n) for i in range(bs)])   # image order (T4)
        R = bs * n
        stage = self._pinned("spec", 2 * R)
        img_of = torch.arange(bs)[:, None].expand(bs, n).reshape(-1)   # (not repeat_interleave: see _img_of)
        torch.add(perms.reshape(-1), img_of, alpha=n, out=stage[:R])      # position of the j-th selected entry inside keep [bs, n]
        stage[R:2 * R] = img_of
        sd = stage[:2 * R].to(dev, non_blocking=True)
        picked = keep.reshape(-1).index_select(0, sd[:R]).clamp_(min=0)   # (-1 padding if an image kept fewer: discarded then, never an OOB read)
        flat = picked + sd[R:] * nmax
        boxes_cat = boxes_all.reshape(bs * nmax, 4).index_select(0, flat)  # [R, 4], image-major
        return dict(n=n, perms=perms, rng=rng, boxes_cat=boxes_cat, img_idx=sd[R:].to(F32))

    def _pinned(self, name, n):
        """a reusable page-locked int64 staging buffer (H2D copies from it are asynchronous on the stream).  One buffer per
        use site: a site's previous copy has always completed by the time the site is reached again (every forward syncs on
        the NMS result in between)"""
        pool = self.__dict__.setdefault("_pin", {})
        t = pool.get(name)
        if t is None or t.numel() < n:
            t = pool[name] = torch.empty((max(n, 1024),), dtype=I64).pin_memory()
        return t

    def _ids_to_host_async(self, input_ids):
        """Start the D2H copy of the caller's device-resident prompt ids on a copy stream of our own, into pinned memory; returns a
        callable that waits for it and hands out the host tensor (None for ids that already live on the host).  The forward needs
        the ids on the host right after the NMS sync; fetched there, `input_ids.cpu()` was a second synchronous round trip inside
        the one window in which the GPU sits idle (profiles/r04_timeline_b14.txt: 0.6 ms between the two copies)."""
        if not input_ids.is_cuda or input_ids.dtype != I64:
            return None
        return self._to_host_async(input_ids, "ids_in")

    def _to_host_async(self, t, name):
        """int64 device tensor -> the pinned buffer `name`, over the copy stream; returns wait() -> the host view (valid until the
        next copy into the same buffer)"""
        cp = self.__dict__.get("_copy_stream")
        if cp is None:
            cp = self._copy_stream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream()
        host = self._pinned(name, t.numel())[: t.numel()].view(t.shape)  # (reused page-locked buffer)
        cp.wait_stream(cur)        # (whoever produced the values did so on the caller's stream)
        with torch.cuda.stream(cp):
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cp)
        t.record_stream(cp)

        def wait():
            ev.synchronize()
            return host
        return wait

    def _propose_graph(self, hidden4):
        """Replay (capturing on first use) the hipGraph of the proposer chain + NMS for these input buffers.  The graph
        reads the ViT states in place (workspace arenas: stable addresses) and owns its outputs; thresholds are baked in, so
        the key carries them.  At most 4 graphs are kept (batch sizes seen most recently)."""
        cfg = self.config
        key = (tuple(h.data_ptr() for h in hidden4), tuple(hidden4[0].shape), float(cfg.nms_thres), float(cfg.box_score_thres),
               int(cfg.max_region_num))
        pool = self.__dict__.setdefault("_pgraphs", {})
        ent = pool.get(key)
        if ent is None:
            # the general NMS path (> 512 candidates) takes a workspace pointer: the graph gets its own, kept alive in `ent`
            nms_ws = ops.nms_workspace(hidden4[0].shape[0], int(cfg.perceiver_cfg.ddetr_cfg.two_stage_num_proposals), self.device)

            def chain():
                pred, scores, idx = self.proposer.forward(hidden4)
                keep, n_keep = ops.nms(pred.contiguous(), scores.contiguous(), float(cfg.nms_thres), float(cfg.box_score_thres),
                                       int(cfg.max_region_num), workspace=nms_ws)
                return pred, scores, idx, keep, n_keep
            chain()  # warm-up: lazy per-kernel attributes, workspaces
            torch.cuda.synchronize(self.device)  # nothing else of this forward (side stream) may overlap the capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = chain()
            if len(pool) >= 4:
                pool.pop(next(iter(pool)))
            ent = pool[key] = (g, outs, nms_ws)
        ent[0].replay()
        return ent[1]

    def _splice(self, input_ids_h, n_img_tok, n_reg):
        """groma.py:317-357 on the host (index bookkeeping only)."""
        # (round 5: the placeholder runs are slices of two cached tensors and the positions come from plain-Python scans of the row --
        #  per image the reference's tensor-by-tensor form cost ~130 us of dispatch here, 1.8 ms per 14-image batch inside the window in
        #  which the GPU waits for the host; same ids, tests/test_host_logic.py::test_splice_matches_oracle...)
        ph = self.__dict__.get("_splice_ph")
        if ph is None or ph[0] != (n_img_tok, self.img_token_id, self.reg_token_id, tuple(self.box_idx_token_ids)):
            pairs = torch.tensor([[b, self.reg_token_id] for b in self.box_idx_token_ids], dtype=I64).reshape(-1)
            ph = self._splice_ph = ((n_img_tok, self.img_token_id, self.reg_token_id, tuple(self.box_idx_token_ids)),
                                    torch.full((n_img_tok,), self.img_token_id, dtype=I64), pairs)
        img_ph, pairs = ph[1], ph[2]
        rows = input_ids_h.tolist()
        new_ids = []
        for i, row in enumerate(rows):
            ids = input_ids_h[i]
            assert self.img_token_id in row and self.reg_token_id in row
            img_pos, reg_pos = row.index(self.img_token_id), row.index(self.reg_token_id)
            if row.count(self.img_token_id) != 1 or row.count(self.reg_token_id) != 1:
                # the reference evaluates `assert img_pos < reg_pos` on position TENSORS: more than one <image> / <region> fails there with
                raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous")   # (same class, same text)
            pad_pos = row.index(self.pad_token_id) if self.pad_token_id in row else len(row)
            assert img_pos < reg_pos
            if n_reg[i] > len(self.box_idx_token_ids):
                raise IndexError("more regions than <r_k> tokens")
            new_ids.append(torch.cat((ids[:img_pos], img_ph, ids[img_pos + 1: reg_pos], pairs[: 2 * n_reg[i]], ids[reg_pos + 1: pad_pos])))
        if len({int(x.numel()) for x in new_ids}) == 1:
            out = torch.stack(new_ids)
        else:
            out = torch.nn.utils.rnn.pad_sequence(new_ids, batch_first=True, padding_value=self.pad_token_id)
        return out, out.ne(self.pad_token_id)

    def _host_splice_plan(self, ids_h, n_img_tok, n_reg):
        """The host side of groma.py:317-369 for one batch: spliced ids + mask, and the rows of the flattened sequence that receive
        image / region features.  Pure host code (no device value enters or leaves): this and _host_select() are everything a
        forward does on the CPU between the NMS result and the LLaMA launches -- what `bench.py --dry-exchange --host-glue` times
        under N-way host contention."""
        new_ids_h, mask_h = self._splice(ids_h, n_img_tok, n_reg)
        flat = new_ids_h.reshape(-1)
        img_rows_h = (flat == self.img_token_id).nonzero(as_tuple=True)[0]
        reg_rows_h = (flat == self.reg_token_id).nonzero(as_tuple=True)[0]
        return new_ids_h, mask_h, flat, img_rows_h, reg_rows_h

    @staticmethod
    def _host_select(keep_h, n_keep_l, nmax):
        """The CPU-RNG shuffle of the kept proposals (groma.py:273-276, T4) and the flat gather list the device selection uses, as
        propose() builds them on a mis-speculated / ragged batch (the general path; host tensors in, host tensors out)"""
        sel_idx = [keep_h[i].index_select(0, torch.randperm(int(nk))) for i, nk in enumerate(n_keep_l)]
        img_of = _img_of([int(x.numel()) for x in sel_idx])
        return sel_idx, torch.cat(sel_idx) + img_of * nmax, img_of

    # ------------------------------------------------------------------ forward
    @_entry
    def forward(self, input_ids=None, inputs_embeds=None, labels=None, attention_mask=None, images=None,
                refer_boxes=None, ground_boxes=None, past_key_values=None, use_cache=False, output_attentions=False,
                output_hidden_states=False, return_dict=False, _last_logits_only=False, _reserve=0, _cache=None,
                _seeds=None):
        if not self._loaded:
            raise RuntimeError("GromaModel has no weights: use from_pretrained / from_state_dict / from_synthetic")
        if output_attentions:
            raise NotImplementedError("output_attentions is not available from the fused attention kernel")
        dev = self.device
        vis_outputs = None
        with ops.gemm_plan(self.gemm_plan):
            if past_key_values is None:
                images = images.to(device=dev, dtype=F32).contiguous()
                ids_early = self._ids_to_host_async(input_ids)  # the prompt ids are needed on the host after the NMS sync: fetch them now
                hidden4 = self._vit_forward(images)
                # Two HIP streams: the region-encoder pyramid (5 rounds of MFMA-bound 3x3 convs) and the bridge MLP only
                # need the ViT states, so they run beside the launch-latency-bound fp32 proposer + NMS + host sync.
                main = torch.cuda.current_stream()
                side = self._side_stream()
                side.wait_stream(main)
                sp = self.stage_precision
                with torch.cuda.stream(side):
                    with ops.precision(sp["region"]):
                        feats, S = self.region.fuse(hidden4[-3:])
                    last = hidden4[self.config.perceiver_cfg.vis_output_layer]
                    with ops.precision(sp["bridge"]):
                        s2d = engine._trace("bridge.s2d", ops.s2d_pack(last, self.vit.G))
                        mid = engine._trace("bridge.mid", ops.gemm(s2d, self.bridge["w0"], bias=self.bridge["b0"], act=1))
                        image_features = ops.gemm(mid, self.bridge["w2"], bias=self.bridge["b2"], out_f32=True)
                    image_features.record_stream(main)
                # region tokens (groma.py:312-315).  With CPU-RNG shuffles and every image expected to keep >= nmax boxes the
                # selection is already on the device before the NMS counts reach the host (propose(): _speculate_shuffle), so the
                # extraction is queued BEFORE the host waits for them and the GPU never drains; a mis-speculation (an image kept
                # fewer boxes) discards it and extracts again from the ordinary selection -- same results either way.
                spec, finish = self.propose(hidden4, refer_boxes, ground_boxes, seeds=_seeds, _defer=True)
                main.wait_stream(side)
                region_features = None
                if spec is not None and SPECULATIVE_EXTRACT and engine.TRACE is None:   # (a trace wants each tensor recorded once)
                    with ops.precision(sp["region"]):
                        region_features = self.region.extract(feats, S, spec["boxes_cat"], spec["img_idx"])
                selected_boxes, aux = finish()
                bs = len(selected_boxes)
                if region_features is None or not aux.get("spec_hit"):
                    # launched FIRST -- nothing below changes its inputs, and every host op placed between the NMS sync and this
                    # launch is GPU-idle time
                    with ops.precision(sp["region"]):
                        region_features = self.region.extract(feats, S, aux["boxes_cat"], aux["img_idx"])  # f32 [R, T]
                ids_h = ids_early() if ids_early is not None else input_ids.cpu()
                writeback = input_ids.is_cuda
                if not input_ids.is_cuda and input_ids.is_inference():  # a CPU tensor the caller made under inference mode:
                    ids_h, writeback = ids_h.clone(), True              # edit a copy, write it back under that mode
                # replace <refer_box>/<ground_box> placeholders by matched <r_k> ids (groma.py:283-309), in place
                refer_box_inds = []
                # (one vectorised membership test per placeholder kind instead of 2 x bs tensor `in` checks: this code sits between
                #  the NMS host sync and the first region-extraction launch, where the GPU has nothing queued)
                has_ref = (ids_h == self.refer_box_token_id).any(dim=1).tolist()
                has_gnd = (ids_h == self.ground_box_token_id).any(dim=1).tolist()
                need_boxes = any(has_ref) or any(has_gnd)
                sel_h = [b.cpu() for b in selected_boxes] if need_boxes else None
                box_ids = torch.tensor(self.box_idx_token_ids)
                for i in range(bs):
                    if has_ref[i]:
                        ious = _box_iou(_c2c(refer_boxes[i].cpu().float()), _c2c(sel_h[i]))
                        matched = torch.max(ious, dim=-1).indices
                        refer_box_inds.append(matched)
                        ids_h[i].masked_scatter_(ids_h[i] == self.refer_box_token_id, box_ids[matched])
                    else:
                        refer_box_inds.append([])
                    if has_gnd[i]:
                        ious = _box_iou(_c2c(ground_boxes[i].cpu().float()), _c2c(sel_h[i]))
                        matched = torch.max(ious, dim=-1).indices
                        mask = ids_h[i] == self.ground_box_token_id
                        ids_h[i].masked_scatter_(mask, box_ids[matched])
                        if labels is not None:
                            with torch.inference_mode(labels.is_inference()):
                                labels[i].masked_scatter_(mask.to(labels.device), box_ids[matched].to(labels.device))
                assert len(refer_box_inds) == bs
                if writeback and need_boxes:
                    engine.inplace_copy(input_ids, ids_h)  # the reference mutates the caller's input_ids (groma.py:295,307)
                n_reg = [b.shape[0] for b in selected_boxes]
                n_img_tok = (self.vit.G // 2) ** 2  # image tokens: groma.py:224-237, :361 (computed on the side stream)
                # splice placeholders, embed, inject (groma.py:317-369)
                new_ids_h, mask_h, flat, img_rows_h, reg_rows_h = self._host_splice_plan(ids_h, n_img_tok, n_reg)
                if labels is not None:
                    labels = self._splice_labels(ids_h, labels.cpu(), n_img_tok, n_reg)
                L = new_ids_h.shape[1]
                assert img_rows_h.numel() == image_features.shape[0] and reg_rows_h.numel() == region_features.shape[0]
                # spliced ids + the two scatter-row lists in ONE pinned, asynchronous H2D copy
                n0, n1, n2 = flat.numel(), img_rows_h.numel(), reg_rows_h.numel()
                stage = self._pinned("ids", n0 + n1 + n2)
                stage[:n0], stage[n0:n0 + n1], stage[n0 + n1:n0 + n1 + n2] = flat, img_rows_h, reg_rows_h
                ids_dev = stage[:n0 + n1 + n2].to(dev, non_blocking=True)
                new_ids = ids_dev[:n0].view(bs, L)
                # f32 [bs*L, T], in an arena of our own: the residual stream's address keys the captured prefill graph
                emb = self.llm.embed(new_ids, out=self._ws.get("llm_h", (bs * L, self.llm.T), F32))
                img_rows, reg_rows = ids_dev[n0:n0 + n1].to(I32), ids_dev[n0 + n1:].to(I32)
                ops.scatter_rows(image_features, img_rows, emb)
                ops.scatter_rows(region_features, reg_rows, emb)
                ref_rows = (flat == self.refer_feat_token_id).nonzero(as_tuple=True)[0]
                if ref_rows.numel() > 0:
                    offs = [0]
                    for n in n_reg:
                        offs.append(offs[-1] + n)
                    gather = torch.cat([torch.as_tensor(ind, dtype=I64) + offs[i] for i, ind in enumerate(refer_box_inds)
                                        if len(ind) > 0])
                    ops.scatter_rows(region_features.index_select(0, gather.to(dev)).contiguous(),
                                     ref_rows.to(I32).to(dev), emb)
                kv_len = mask_h.sum(-1).to(I32).to(dev) if not bool(mask_h.all()) else None
                if _cache is not None:  # generate(): prefill straight into the decode arena the captured graph reads
                    cache = _cache
                    cache.seq_len = 0
                    if cache.bs != bs or cache.smax < L + max(int(_reserve), 0):
                        raise RuntimeError("decode arena too small for this prompt")
                elif use_cache or _reserve:
                    cache = self.llm.new_cache(bs, L + max(int(_reserve), 0), dev)
                else:  # no cache requested: recycle one scratch KV buffer instead of zero-filling 2x32 tensors per call
                    cache = self._scratch_cache(bs, L)
                vis_outputs = {'pred_boxes': selected_boxes,
                               'image_features': image_features.view(bs, n_img_tok, -1),
                               'region_features': region_features}
                aux["lengths"] = mask_h.sum(-1).tolist()  # expanded length of every row (right padding excluded)
                aux["hidden4"] = hidden4  # the four ViT states the path consumed (arena views: valid until the next forward)
                aux["input_ids"] = new_ids_h
                if getattr(self, "capture_embeds", False):  # parity tests: the LLM stage's exact input (emb is consumed in place)
                    aux["inputs_embeds"] = emb.view(bs, L, -1).clone()
                self._last_aux = aux
            else:
                cache = past_key_values
                bs = cache.bs
                L = 1 if inputs_embeds is None else inputs_embeds.shape[1]
                if inputs_embeds is None:
                    emb = self.llm.embed(input_ids[:, -1:].to(dev))
                else:
                    emb = inputs_embeds.to(dev, F32).reshape(bs * L, -1).contiguous()
                kv_len = None  # the reference rebuilds an all-ones mask over past+1 (groma.py:376-379, T6): every cached key is visible
            states = [] if (output_hidden_states and self.all_hidden_states) else None
            logits, hn = self.llm.forward(emb, bs, L, cache, kv_len=kv_len, all_logits=not _last_logits_only, states=states)

        loss = None
        if labels is not None:  # groma.py:404-415 (training-side convenience; not on the inference hot path)
            lab = labels.to(dev)
            loss = torch.nn.functional.cross_entropy(logits[..., :-1, :].reshape(-1, self.config.vocab_size).float(),
                                                     lab[..., 1:].reshape(-1), ignore_index=IGNORE_INDEX)
        # Everything returned is the caller's: decode-step logits and the final hidden state live in recycled scratch
        # buffers internally, so the boundary hands out copies (KB-sized); `_last_logits_only` marks the internal
        # generate() loop, which consumes the views immediately.
        if past_key_values is not None and not _last_logits_only:
            logits = logits.clone()
        hidden_states = None
        if output_hidden_states:
            with ops.precision(self.stage_precision["head"]):   # (the final norm's output is the head stage's operand)
                if hn is None:  # decode step: the final norm lives in the head GEMV's prologue; `emb` is the residual stream, updated in place
                    hn = ops.rmsnorm(emb, self.llm.w["norm"], self.llm.eps)
                # (operand pairs: hand out the f32 values they stand for)
                hidden_states = tuple(states or ()) + ((ops.unsplit(hn) if ops.SP() == 2 else hn.clone()).view(bs, -1, self.llm.T),)
        if not use_cache and past_key_values is None:
            cache = None  # HF returns past_key_values=None without use_cache; the scratch KV buffer is recycled
        if not return_dict:
            output = (logits, cache)
            return (loss,) + output if loss is not None else output
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache,
                                      hidden_states=(hidden_states, vis_outputs), attentions=None)

    __call__ = forward

    def _splice_labels(self, ids_h, labels_h, n_img_tok, n_reg):
        new_labels = []
        for i in range(ids_h.shape[0]):
            ids = ids_h[i]
            img_pos = (ids == self.img_token_id).nonzero(as_tuple=True)[0]
            reg_pos = (ids == self.reg_token_id).nonzero(as_tuple=True)[0]
            pad_pos = (ids == self.pad_token_id).nonzero(as_tuple=True)[0]
            pad_pos = pad_pos[0] if len(pad_pos) > 0 else len(ids)
            new_labels.append(torch.cat((labels_h[i][:img_pos], torch.full((n_img_tok,), IGNORE_INDEX),
                                         labels_h[i][img_pos + 1: reg_pos], torch.full((n_reg[i] * 2,), IGNORE_INDEX),
                                         labels_h[i][reg_pos + 1: pad_pos])))
        return torch.nn.utils.rnn.pad_sequence(new_labels, batch_first=True, padding_value=IGNORE_INDEX)

    # ------------------------------------------------------------------ hipGraph decode loop
    def _decoder(self, bs, smax, max_new, eos, pad):
        key = (bs, smax, max_new, eos, pad, self.gemm_plan)  # the plan is baked into the captured launches
        pool = self.__dict__.setdefault("_decoders", {})
        dec = pool.get(key)
        if dec is None:
            if len(pool) >= 2:  # each arena owns a KV cache: keep the two most recent shapes
                pool.pop(next(iter(pool)))
            dec = engine.GreedyDecoder(self.llm, bs, smax, max_new, eos, pad, self.device)
            dec.capture()  # before any prefill lands in the arena: the warm-up steps scribble on KV rows 0..2
            pool[key] = dec
        return dec

    def _generate_graph(self, seqs, ids, images, refer_boxes, ground_boxes, max_new_tokens, eos, pad, return_dict,
                        output_hidden_states, temperature=0.0, draw_seeds=None):
        """generate() with the per-token step captured once in a hipGraph (engine.GreedyDecoder): the step position,
        the token fed back, the finished mask and the output ids all live on the device, so one replay = one token
        and the host only reads the unfinished-row count when an EOS id is configured."""
        bs, P = ids.shape
        pc = self.config.perceiver_cfg
        n_img_tok = (self.config.image_size // pc.vis_encoder_cfg.patch_size // 2) ** 2
        bound = P + n_img_tok + 2 * self.config.max_region_num + (sum(len(b) for b in refer_boxes) if refer_boxes else 0)
        smax = engine._ru(bound + max_new_tokens + 1, 256)
        dec = self._decoder(bs, smax, engine._ru(max_new_tokens, 64), eos, pad)
        first = self.forward(input_ids=ids, images=images, refer_boxes=refer_boxes, ground_boxes=ground_boxes,
                             use_cache=True, output_hidden_states=output_hidden_states, return_dict=True,
                             _last_logits_only=True, _reserve=max_new_tokens, _cache=dec.cache)
        L = dec.cache.seq_len
        dec.set_sampling(temperature, draw_seeds() if draw_seeds is not None else None)
        n = dec.run(first.logits[:, -1, :], L, max_new_tokens)
        dec.cache.seq_len = L + n - 1
        seqs = torch.cat([seqs, dec.seq[:, :n]], dim=-1)
        if not return_dict:
            return seqs
        hs = (first.hidden_states,) if output_hidden_states else None
        return GenerateOutput(sequences=seqs, hidden_states=hs, past_key_values=dec.cache)

    # ------------------------------------------------------------------ generate (HF 4.32 greedy_search semantics)
    @_entry
    def generate(self, *a, **kw):
        with ops.gemm_plan(self.gemm_plan):
            return self._generate(*a, **kw)

    def _generate(self, input_ids, images=None, refer_boxes=None, ground_boxes=None, use_cache=True, do_sample=False,
                  max_new_tokens=None, return_dict_in_generate=False, output_hidden_states=False, generation_config=None,
                  **kw):
        """Greedy decoding as HF GenerationMixin.greedy_search drives the reference model
        (groma/eval/eval_rec.py:93-104): the next token is the arg-max of the LAST position of the right-padded
        expanded sequence; finished rows emit pad; `sequences` = original prompt + new ids."""
        gc = generation_config if generation_config is not None else self.generation_config
        if do_sample is None:
            do_sample = bool(getattr(gc, "do_sample", False))
        if kw.get("num_beams", 1) != 1:
            raise NotImplementedError("beam search is not implemented")
        temperature = 0.0
        if do_sample:
            # the sampler the reference serves with (groma/serve/model_worker.py:307-311): softmax(logits / T) + multinomial;
            # HF's top-k / top-p warpers are not part of it
            if kw.get("top_k") not in (None, 0) or kw.get("top_p") not in (None, 1.0):
                raise NotImplementedError("top_k / top_p sampling is not implemented (temperature sampling only)")
            temperature = float(kw.get("temperature", getattr(gc, "temperature", 1.0) or 1.0))
        # The rows' counter-based samplers are seeded by ONE draw of the CPU global RNG taken AFTER the prefill (so the region
        # shuffle's torch.randperm, trap T4, sees the same RNG state as in a greedy call; torch.manual_seed reproduces a run)
        draw_seeds = (lambda: torch.randint(0, 2 ** 62, (input_ids.shape[0],), dtype=I64)) if do_sample else (lambda: None)
        if max_new_tokens is None:
            max_new_tokens = getattr(gc, "max_new_tokens", 20) or 20
        eos = getattr(gc, "eos_token_id", None)
        pad = getattr(gc, "pad_token_id", None)
        if eos is not None and pad is None:
            pad = eos
        dev = self.device
        seqs = input_ids.to(dev).clone()
        ids_for_model = input_ids.to(dev)
        if self.decode_graph and max_new_tokens > 1:
            return self._generate_graph(seqs, ids_for_model, images, refer_boxes, ground_boxes, max_new_tokens, eos, pad,
                                        return_dict_in_generate, output_hidden_states, temperature, draw_seeds)
        out = self.forward(input_ids=ids_for_model, images=images, refer_boxes=refer_boxes, ground_boxes=ground_boxes,
                           use_cache=True, output_hidden_states=output_hidden_states, return_dict=True,
                           _last_logits_only=True, _reserve=max_new_tokens)
        first = out
        cache = out.past_key_values
        logits = out.logits
        unfinished = torch.ones(seqs.shape[0], dtype=I64, device=dev)
        bs = seqs.shape[0]
        inv_t = torch.full((bs,), 0.0 if temperature < 1e-4 else 1.0 / temperature, dtype=F32, device=dev)
        seeds = draw_seeds()
        seed_t = (seeds if seeds is not None else torch.zeros((bs,), dtype=I64)).to(dev)
        pos_t = torch.zeros((1,), dtype=I32, device=dev)
        L0 = cache.seq_len  # expanded prompt length: the first new token sits at position L0
        for step in range(max_new_tokens):
            V = logits.shape[-1]
            last = logits[:, -1, :]
            pos_t.fill_(L0 + step)
            nxt = ops.sample_rows(last if last.is_contiguous() else last.contiguous(), V, inv_t, seed_t, pos=pos_t)
            if eos is not None:
                nxt = nxt * unfinished + pad * (1 - unfinished)
            seqs = torch.cat([seqs, nxt[:, None]], dim=-1)
            if eos is not None:
                unfinished = unfinished * (nxt != eos).long()
                if int(unfinished.max()) == 0:
                    break
            if step == max_new_tokens - 1:
                break
            o = self.forward(input_ids=nxt[:, None], past_key_values=cache, use_cache=True, return_dict=True)
            logits = o.logits
        if not return_dict_in_generate:
            return seqs
        hs = (first.hidden_states,) if output_hidden_states else None
        return GenerateOutput(sequences=seqs, hidden_states=hs, past_key_values=cache)
