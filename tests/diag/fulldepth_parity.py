"""Full-DEPTH Groma-7B (24 ViT layers, 6+6 DDETR, 5 fusion rounds, 32 LLaMA layers, 32 114-wide head) on the MI355X against the
CPU oracle, one image, with DISTINCT weights in every layer.

Every parameter is drawn on the GPU from its own seeded generator (groma_amd.weights.Source.synthetic -- what
GromaModel.from_synthetic packs) and handed to the oracle through a lazy state dict that re-draws the tensor and copies it to
the host when the oracle asks for it: the device model and the oracle see bit-identical fp32 parameters, and the 28 GB host
copy never exists (a layer's tensors are dropped as soon as the oracle has used them).

Three distances are reported for every stage (relative L2): device <-> fp32 oracle, device <-> bf16-rounded oracle, and
bf16-rounded oracle <-> fp32 oracle -- the last one is what the bf16 FORMAT costs any implementation at this depth, so the
first must not be materially worse than it.  The oracles consume the device's ViT states for the stages behind the ViT (stage
chaining, SURVEY 'Hard parts') and run their own 24-layer ViT for the ViT comparison.  ~1.5 minutes, most of it the two oracle
passes on the host cores."""
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
    """-> dict of the measured numbers (also printed).  precision: the 16-bit operand type of the device model AND of the
    rounded oracle it is compared with ("bf16" | "fp16")"""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    full = gconfig.groma_7b(box_score_thres=0.0)
    dev = torch.device("cuda")
    tk = util.TokenIds()
    images, ids = synth.make_inputs(full, tk, 1, seed=1234)
    t = time.time()
    model = GromaModel.from_synthetic(full, seed=seed, device=dev, precision=precision)
    model.init_special_token_id(constants.SyntheticTokenizer())
    sd = LazyDeviceStateDict(full, seed, dev)
    # spot check: the lazy dict serves exactly what the device model packed (distinct per layer)
    a, b = sd["llm.model.layers.0.mlp.down_proj.weight"], sd["llm.model.layers.31.mlp.down_proj.weight"]
    h16 = torch.float16 if precision == "fp16" else torch.bfloat16
    assert not torch.equal(a, b) and torch.equal(a.to(h16), model.llm.w["layers"][0]["wd"][0].cpu())
    assert torch.equal(b.to(h16), model.llm.w["layers"][31]["wd"][0].cpu())
    print(f"device model (distinct per-layer weights) packed in {time.time() - t:.1f} s")
    rel = util.relerr
    cd = full.to_dict()
    with torch.no_grad():
        torch.manual_seed(77)
        out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True, use_cache=True)
        aux = model._last_aux
        dev_h = [h.float().cpu() for h in aux["hidden4"]]
        ref, ref_v = {}, {}
        for mode in (None, precision):
            t = time.time()
            with O.rounding(mode):
                ref_v[mode] = O.vit_forward(sd, cd, images)[-4:]
                torch.manual_seed(77)
                ref[mode] = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(dev_h))
            print(f"oracle ({'fp32' if mode is None else precision + '-rounded'}): 24-layer ViT + proposer + region encoder + 32-layer LLaMA in {time.time() - t:.1f} s")
    r32, r16 = ref[None], ref[precision]
    eq = dict(topk_equal=torch.equal(aux["topk_idx"].cpu().long(), r32["det"]["topk_idx"]),
              nms_equal=torch.equal(aux["nms_keep"][0], r32["nms_inds"][0]),
              ids_equal=torch.equal(aux["input_ids"], r32["input_ids"]) and torch.equal(r16["input_ids"], r32["input_ids"]))
    print("top-300 ids equal:", eq["topk_equal"], "| NMS keep ids equal:", eq["nms_equal"], "| spliced ids equal:", eq["ids_equal"],
          "| L =", r32["input_ids"].shape[1])
    vis = out.hidden_states[1]

    def three(name, d, f):
        a, b, c = rel(d, f(r32)), rel(d, f(r16)), rel(f(r16), f(r32))
        print(f"{name:42s} device<->fp32 {a:.3e} | device<->{precision}-rounded {b:.3e} | {precision}-rounded<->fp32 {c:.3e} | ratio {a / c:.2f}")
        return a, b, c
    res = {}
    res["vit"] = [(rel(a, b), rel(a, c), rel(c, b)) for a, b, c in zip(dev_h, ref_v[None], ref_v[precision])]
    for i, (a, b, c) in enumerate(res["vit"]):
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


if __name__ == "__main__":
    run(precision=sys.argv[1] if len(sys.argv) > 1 else "bf16")
