"""Full-DEPTH Groma-7B (24 ViT layers, 6+6 DDETR, 5 fusion rounds, 32 LLaMA layers, 32 114-wide head) on the MI355X against the
CPU oracle, one image, with DISTINCT weights in every layer.

Every parameter is drawn on the GPU from its own seeded generator (groma_amd.weights.Source.synthetic -- what
GromaModel.from_synthetic packs) and handed to the oracle through a lazy state dict that re-draws the tensor and copies it to
the host when the oracle asks for it: the device model and the oracle see bit-identical fp32 parameters, and the 28 GB host
copy never exists (a layer's tensors are dropped as soon as the oracle has used them).

Three distances are reported for every stage (relative L2): device <-> fp32 oracle, device <-> bf16-rounded oracle, and
bf16-rounded oracle <-> fp32 oracle -- the last one is what the bf16 FORMAT costs any implementation at this depth, so the
first must not be materially worse than it.  The oracles consume the device's ViT states for the stages behind the ViT (stage
chaining, SURVEY 'Hard parts') and run their own 24-layer ViT for the ViT comparison; what survives WITHOUT chaining (the fp32
oracle's proposer fed its own ViT states) is reported beside it: fraction of the top-300 / NMS ids that agree, next to the
oracle's smallest adjacent logit gap and the device's logit error.  ~1.5 minutes, most of it the two oracle passes on the host
cores.

precision="ref" (operand pairs, csrc/gr_common.h) is compared UNCHAINED only: one fp32 oracle pass that runs its own ViT end to
end, which is the reference's computation (R: groma/model/groma.py:222-280,389-402 in one fp32 pass); there is no rounded
oracle for it -- pure fp32 is its oracle.
precision="hybrid" / "hybrid-fp16" (round 5: the ViT on operand pairs, everything behind it bf16 / fp16) is compared UNCHAINED too:
both oracle passes -- fp32, and operands rounded to the 16-bit type BEHIND the ViT -- consume the oracle's own fp32 ViT states,
nothing of the device's."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from groma_amd import config as gconfig, constants, synth, weights
from groma_amd.groma import GromaModel
from oracle import groma_oracle as O
from tests import util


class LazyDeviceStateDict:
    """sd[name] -> the fp32 tensor weights.Source.synthetic(cfg, seed, device) produces for `name`, copied to the host"""

    def __init__(self, cfg, seed, device):
        self._src = weights.Source.synthetic(cfg, seed, device)
        self._names = {n for n, _, _ in synth.param_spec(cfg)}
        self.bytes_served = 0

    def __contains__(self, k):
        return k in self._names

    def __getitem__(self, k):
        if k not in self._names:
            raise KeyError(k)
        t = self._src._get(k).float().cpu()
        self.bytes_served += t.numel() * 4
        return t

    def get(self, k, default=None):
        return self[k] if k in self._names else default


def run(seed=0, precision="bf16"):
    """-> dict of the measured numbers (also printed).  precision: the operand type of the device model AND of the rounded
    oracle it is compared with ("bf16" | "fp16"), or "ref": device vs the fp32 oracle, unchained"""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    full = gconfig.groma_7b(box_score_thres=0.0)
    dev = torch.device("cuda")
    tk = util.TokenIds()
    images, ids = synth.make_inputs(full, tk, 1, seed=1234)
    t = time.time()
    hybrid = precision.startswith("hybrid")
    base = {"hybrid": "bf16", "hybrid-fp16": "fp16"}.get(precision, precision)   # operand type behind the ViT
    model = GromaModel.from_synthetic(full, seed=seed, device=dev, precision=precision)
    model.init_special_token_id(constants.SyntheticTokenizer())
    sd = LazyDeviceStateDict(full, seed, dev)
    # spot check: the lazy dict serves exactly what the device model packed (distinct per layer)
    a, b = sd["llm.model.layers.0.mlp.down_proj.weight"], sd["llm.model.layers.31.mlp.down_proj.weight"]
    from groma_amd import ops
    with ops.precision(base):
        assert not torch.equal(a, b) and torch.equal(ops.to_h16(a), model.llm.w["layers"][0]["wd"][0].cpu())
        assert torch.equal(ops.to_h16(b), model.llm.w["layers"][31]["wd"][0].cpu())
    print(f"device model (distinct per-layer weights) packed in {time.time() - t:.1f} s")
    rel = util.relerr
    cd = full.to_dict()
    with torch.no_grad():
        torch.manual_seed(77)
        out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True, use_cache=True)
        aux = model._last_aux
        dev_h = [h.float().cpu() for h in aux["hidden4"]]
        dbg = {}
        model.proposer.forward(aux["hidden4"], debug=dbg)   # the device's own class logits (arena views of this forward)
        dev_cls = dbg["enc_class"].float().cpu()
        unchained = precision == "ref" or hybrid
        ref, ref_v = {}, {}
        for mode in ((None,) if precision == "ref" else (None, base)):
            t = time.time()
            with O.rounding(None if hybrid else mode):   # (hybrid: the ViT is NOT a 16-bit stage -- both passes take the fp32 ViT)
                own = ref_v[None] if (hybrid and None in ref_v) else O.vit_forward(sd, cd, images)
                ref_v[mode] = own if hybrid else own[-4:]
            with O.rounding(mode):
                torch.manual_seed(77)
                ref[mode] = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images,
                                            hidden_states=tuple(own) if unchained else tuple(dev_h))
            print(f"oracle ({'fp32' if mode is None else base + '-rounded'}{', UNCHAINED (its own ViT states)' if unchained else ''}): "
                  f"24-layer ViT + proposer + region encoder + 32-layer LLaMA in {time.time() - t:.1f} s")
        if hybrid:
            ref_v = {m: v[-4:] for m, v in ref_v.items()}
        # what the index-valued results do WITHOUT stage chaining: the fp32 oracle's proposer on the oracle's own ViT states
        torch.manual_seed(77)
        per = ref[None] if unchained else O.perceive(sd, cd, images, hidden_states=tuple(ref_v[None]))
    o_cls = per["det"]["enc_class"]
    srt = torch.sort(o_cls, dim=1, descending=True, stable=True)[0]
    Q = aux["topk_idx"].shape[1]
    d_ids, o_ids = aux["topk_idx"].cpu().long(), per["det"]["topk_idx"]
    un = dict(topk_pos_equal=(d_ids == o_ids).float().mean().item(),
              topk_set_overlap=len(set(d_ids[0].tolist()) & set(o_ids[0].tolist())) / Q,
              nms_equal=torch.equal(aux["nms_keep"][0], per["nms_inds"][0]),
              nms_set_overlap=len(set(aux["nms_keep"][0].tolist()) & set(per["nms_inds"][0].tolist())) / max(1, len(per["nms_inds"][0])),
              min_gap=(srt[:, :Q] - srt[:, 1:Q + 1]).min().item(), cls_err=(dev_cls - o_cls).abs().max().item())
    print(f"UNCHAINED (oracle runs its own fp32 ViT): top-300 ids equal at {un['topk_pos_equal']:.3f} of slots, set overlap {un['topk_set_overlap']:.3f}; "
          f"NMS ids equal: {un['nms_equal']} (set overlap {un['nms_set_overlap']:.3f}); oracle min adjacent gap of the top-301 logits "
          f"{un['min_gap']:.2e}, device class-logit max abs error {un['cls_err']:.2e}")
    r32, r16 = ref[None], ref[None if precision == "ref" else base]
    precision_name, precision = precision, base   # (the prints below name the operand type of the rounded oracle)
    eq = dict(topk_equal=torch.equal(aux["topk_idx"].cpu().long(), r32["det"]["topk_idx"]),
              nms_equal=torch.equal(aux["nms_keep"][0], r32["nms_inds"][0]),
              ids_equal=torch.equal(aux["input_ids"], r32["input_ids"]) and torch.equal(r16["input_ids"], r32["input_ids"]))
    print("top-300 ids equal:", eq["topk_equal"], "| NMS keep ids equal:", eq["nms_equal"], "| spliced ids equal:", eq["ids_equal"],
          "| L =", r32["input_ids"].shape[1])
    vis = out.hidden_states[1]

    def three(name, d, f):
        a, b, c = rel(d, f(r32)), rel(d, f(r16)), rel(f(r16), f(r32))
        if precision_name == "ref":
            print(f"{name:42s} device<->fp32 oracle (unchained) {a:.3e}")
        else:
            print(f"{name:42s} device<->fp32 {a:.3e} | device<->{precision}-rounded {b:.3e} | {precision}-rounded<->fp32 {c:.3e} | ratio {a / c:.2f}")
        return a, b, c
    res = {"unchained": un}
    res["vit"] = [(rel(a, b), rel(a, c), rel(c, b)) for a, b, c in zip(dev_h, ref_v[None], ref_v[None if unchained else precision])]
    for i, (a, b, c) in enumerate(res["vit"]):
        if unchained:
            print(f"{'ViT state ' + str(21 + i) + ' layers deep':42s} device<->fp32 oracle {a:.3e}")
        else:
            print(f"{'ViT state ' + str(21 + i) + ' layers deep':42s} device<->fp32 {a:.3e} | device<->{precision}-rounded {b:.3e} | {precision}-rounded<->fp32 {c:.3e} | ratio {a / c:.2f}")
    res["image_tokens"] = three("image tokens (s2d + bridge)", vis["image_features"], lambda r: r["image_features"])
    res["region_tokens"] = three("region tokens (5 fusion rounds + RoI)", vis["region_features"], lambda r: r["region_features"])
    res["k0"] = three("K cache layer 0", out.past_key_values[0][0], lambda r: r["past"][0][0])
    res["k31"] = three("K cache layer 31", out.past_key_values[31][0], lambda r: r["past"][31][0])
    lg_d = out.logits.float().cpu()
    res["logits"] = three("logits (32 layers deep, all 582 positions)", lg_d, lambda r: r["logits"])
    res["region_logits"] = three("last-position region logits <r0..r99>", lg_d[:, -1, 32014:32114], lambda r: r["logits"][:, -1, 32014:32114])
    lg_r = r32["logits"]
    err = (lg_d - lg_r).abs().max().item()
    top2 = lg_r.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * err
    agree = (lg_d.argmax(-1) == lg_r.argmax(-1))
    agree16 = (r16["logits"].argmax(-1) == lg_r.argmax(-1))
    print(f"logits max abs err {err:.3e} (max |logit| {lg_r.abs().max().item():.2f}); arg-max equal to the fp32 oracle's at {agree.float().mean().item():.3f} of "
          f"positions ({precision}-rounded oracle: {agree16.float().mean().item():.3f}), at {agree[clear].float().mean().item() if clear.any() else float('nan'):.3f} of the "
          f"{clear.float().mean().item():.3f} clear-margin positions")
    print(f"host bytes served by the lazy state dict: {sd.bytes_served / 1e9:.1f} GB")
    res.update(eq, argmax_agree=agree.float().mean().item(), argmax_agree_bf16_oracle=agree16.float().mean().item(),
               argmax_agree_clear=agree[clear].float().mean().item() if clear.any() else 1.0, L=r32["input_ids"].shape[1])
    return res


def run_fp8(seed=0, oracle=True):
    """BASELINE configs[4] at the real depth: the e4m3 model ("hybrid" + fp8=True: DINOv2 on operand pairs, the
    LLaMA linears, lm_head and the region encoder's 3x3 convs on OCP e4m3 with per-row / static scales) against
      (a) the bf16 device path of the same weights ("hybrid") -- configs[4]'s "logits within stated tol vs bf16",
      (b) the fp32 oracle, unchained,  (c) the e4m3-rounded oracle (oracle.groma_oracle.rounding("e4m3")) on the same fp32 ViT states,
    and (d) what the e4m3 FORMAT costs the oracle itself at this depth (e4m3-rounded oracle <-> fp32 oracle)."""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    full = gconfig.groma_7b(box_score_thres=0.0)
    dev = torch.device("cuda")
    tk = util.TokenIds()
    images, ids = synth.make_inputs(full, tk, 1, seed=1234)
    sd = LazyDeviceStateDict(full, seed, dev)
    cd = full.to_dict()
    rel = util.relerr
    outs = {}
    with torch.no_grad():
        for name, kw in (("bf16", dict(precision="hybrid")), ("e4m3", dict(precision="hybrid", fp8=True))):
            t = time.time()
            model = GromaModel.from_synthetic(full, seed=seed, device=dev, **kw)
            model.init_special_token_id(constants.SyntheticTokenizer())
            torch.manual_seed(77)
            o = model.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True)
            aux = model._last_aux
            outs[name] = dict(logits=o.logits.float().cpu(), ids=aux["input_ids"], topk=aux["topk_idx"].cpu().long(), nms=aux["nms_keep"][0],
                              k31=o.past_key_values[31][0].float().cpu(), region=o.hidden_states[1]["region_features"].float().cpu(), mode=model.mode)
            print(f"device {model.mode}: packed + forward in {time.time() - t:.1f} s")
            del model, o, aux
            torch.cuda.empty_cache()
        if not oracle:   # the test's form: the two device paths against each other (the oracle passes are profiles/r06_fulldepth_fp8.txt)
            d8, d16 = outs["e4m3"], outs["bf16"]
            res = dict(ids_equal=torch.equal(d8["ids"], d16["ids"]), topk_equal=torch.equal(d8["topk"], d16["topk"]), nms_equal=torch.equal(d8["nms"], d16["nms"]),
                       logits=rel(d8["logits"], d16["logits"]), region_logits=rel(d8["logits"][:, -1, 32014:32114], d16["logits"][:, -1, 32014:32114]),
                       k31=rel(d8["k31"], d16["k31"]), region_tokens=rel(d8["region"], d16["region"]),
                       argmax=(d8["logits"].argmax(-1) == d16["logits"].argmax(-1)).float().mean().item())
            print("e4m3 device <-> bf16 device at full depth:", res)
            return res
        t = time.time()
        own = O.vit_forward(sd, cd, images)
        ref = {}
        for mode in (None, "e4m3"):
            torch.manual_seed(77)
            with O.rounding(mode):
                ref[mode] = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(own))
            print(f"oracle ({'fp32' if mode is None else 'e4m3-rounded'}, its own fp32 ViT): {time.time() - t:.1f} s")
    d8, d16, o32, o8 = outs["e4m3"], outs["bf16"], ref[None], ref["e4m3"]
    eq = dict(topk_equal=torch.equal(d8["topk"], o32["det"]["topk_idx"]), nms_equal=torch.equal(d8["nms"], o32["nms_inds"][0]),
              ids_equal=torch.equal(d8["ids"], o32["input_ids"]) and torch.equal(d16["ids"], o32["input_ids"]))
    print("e4m3 model: top-300 ids equal:", eq["topk_equal"], "| NMS ids equal:", eq["nms_equal"], "| spliced ids equal:", eq["ids_equal"])
    res = dict(eq)
    for name, f_dev, f_or in (("region tokens", lambda d: d["region"], lambda r: r["region_features"]),
                              ("K cache layer 31", lambda d: d["k31"], lambda r: r["past"][31][0]),
                              ("logits (32 layers deep, all 582 positions)", lambda d: d["logits"], lambda r: r["logits"]),
                              ("last-position region logits <r0..r99>", lambda d: d["logits"][:, -1, 32014:32114], lambda r: r["logits"][:, -1, 32014:32114])):
        a = dict(vs_bf16_device=rel(f_dev(d8), f_dev(d16)), vs_fp32=rel(f_dev(d8), f_or(o32)), vs_e4m3_oracle=rel(f_dev(d8), f_or(o8)),
                 format=rel(f_or(o8), f_or(o32)), bf16_device_vs_fp32=rel(f_dev(d16), f_or(o32)))
        res[name] = a
        print(f"{name:44s} e4m3 device<->bf16 device {a['vs_bf16_device']:.3e} | <->fp32 oracle {a['vs_fp32']:.3e} | <->e4m3-rounded oracle {a['vs_e4m3_oracle']:.3e} | "
              f"e4m3-rounded oracle<->fp32 (the format) {a['format']:.3e} | bf16 device<->fp32 {a['bf16_device_vs_fp32']:.3e}")
    am = lambda a, b: (a.argmax(-1) == b.argmax(-1)).float().mean().item()
    res["argmax_vs_bf16_device"], res["argmax_vs_fp32"], res["argmax_oracle_e4m3_vs_fp32"] = am(d8["logits"], d16["logits"]), am(d8["logits"], o32["logits"]), am(o8["logits"], o32["logits"])
    print(f"arg-max of the e4m3 model equal to the bf16 device's at {res['argmax_vs_bf16_device']:.3f} of positions, to the fp32 oracle's at {res['argmax_vs_fp32']:.3f} "
          f"(e4m3-rounded oracle vs fp32: {res['argmax_oracle_e4m3_vs_fp32']:.3f})")
    return res


if __name__ == "__main__":
    a = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    run_fp8() if a == "fp8" else run(precision=a)
