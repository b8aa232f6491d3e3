"""Full-DEPTH Groma-7B (24 ViT layers, 6+6 DDETR, 5 fusion rounds, 32 LLaMA layers, 32 114-wide head) on the MI355X against the
fp32 CPU oracle, one image.  Per-layer weights of the three deep stacks are aliased to one materialised layer each (bench.py's
_AliasedLayers: same arithmetic per layer, 3 GB of host state instead of 30 GB); the oracle consumes the device's ViT states for
the stages behind the ViT (stage chaining) and runs its own 24-layer ViT for the ViT comparison.  ~1 minute of CPU work."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from groma_amd import config as gconfig, constants, synth
from groma_amd.groma import GromaModel
from oracle import groma_oracle as O
from tests import util



def run():
    """-> dict of the measured numbers (also printed)"""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    full = gconfig.groma_7b(box_score_thres=0.0)
    d = full.to_dict(); d.pop("vocab_size")
    d["perceiver_cfg"]["vis_encoder_cfg"]["num_hidden_layers"] = 1
    d["region_cfg"]["num_fuse"] = 1
    d["llm_cfg"]["num_hidden_layers"] = 1
    small = gconfig.GromaConfig(**d)
    sd = bench._AliasedLayers(synth.make_state_dict(small, 0))
    tk = util.TokenIds()
    images, ids = synth.make_inputs(full, tk, 1, seed=1234)
    t = time.time()
    model = GromaModel.from_state_dict(full, sd, "cuda")
    model.init_special_token_id(constants.SyntheticTokenizer())
    print(f"device model packed in {time.time() - t:.1f} s")
    rel = util.relerr
    with torch.no_grad():
        torch.manual_seed(77)
        out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True, use_cache=True)
        aux = model._last_aux
        dev_h = [h.float().cpu() for h in aux["hidden4"]]
        t = time.time()
        ref_v = O.vit_forward(sd, full.to_dict(), images)[-4:]
        print(f"oracle ViT (24 layers) {time.time() - t:.1f} s; device ViT states vs fp32 oracle:", [f"{rel(a, b):.2e}" for a, b in zip(dev_h, ref_v)])
        t = time.time()
        torch.manual_seed(77)
        ref = O.groma_forward(sd, full.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(dev_h))
        print(f"oracle proposer + region encoder + 32-layer LLaMA {time.time() - t:.1f} s")
    print("top-300 ids equal:", torch.equal(aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"]),
          "| NMS keep ids equal:", torch.equal(aux["nms_keep"][0], ref["nms_inds"][0]),
          "| spliced ids equal:", torch.equal(aux["input_ids"], ref["input_ids"]), "| L =", ref["input_ids"].shape[1])
    vis = out.hidden_states[1]
    print(f"image tokens {rel(vis['image_features'], ref['image_features']):.2e}  region tokens {rel(vis['region_features'], ref['region_features']):.2e}")
    lg_d, lg_r = out.logits.float().cpu(), ref["logits"]
    err = (lg_d - lg_r).abs().max().item()
    top2 = lg_r.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * err
    agree = (lg_d.argmax(-1) == lg_r.argmax(-1))
    print(f"logits (32 layers deep) rel-L2 {rel(lg_d, lg_r):.2e}, max abs err {err:.3e} (max |logit| {lg_r.abs().max().item():.2f}); "
          f"arg-max equal at {agree.float().mean().item():.3f} of positions, at {agree[clear].float().mean().item() if clear.any() else float('nan'):.3f} of the "
          f"{clear.float().mean().item():.3f} clear-margin positions")
    print(f"last-position region logits rel-L2 {rel(lg_d[:, -1, 32014:32114], lg_r[:, -1, 32014:32114]):.2e}")
    print(f"K cache layer 0 / 31 rel-L2 {rel(out.past_key_values[0][0], ref['past'][0][0]):.2e} / {rel(out.past_key_values[31][0], ref['past'][31][0]):.2e}")
    return dict(vit=[rel(a, b) for a, b in zip(dev_h, ref_v)], topk_equal=torch.equal(aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"]),
                nms_equal=torch.equal(aux["nms_keep"][0], ref["nms_inds"][0]), ids_equal=torch.equal(aux["input_ids"], ref["input_ids"]),
                image_tokens=rel(vis["image_features"], ref["image_features"]), region_tokens=rel(vis["region_features"], ref["region_features"]),
                logits=rel(lg_d, lg_r), argmax_agree=agree.float().mean().item(),
                argmax_agree_clear=agree[clear].float().mean().item() if clear.any() else 1.0, L=ref["input_ids"].shape[1])


if __name__ == "__main__":
    run()
