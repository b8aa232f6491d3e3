import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import groma_oracle as O
from tests import util
from groma_amd import synth, constants
from groma_amd.groma import GromaModel
cfg, sd, tk = util.tiny_setup(seed=0)
images, ids = synth.make_inputs(cfg, tk, bs=2, seed=1234)
for fp8 in (False, True):
    m = GromaModel.from_state_dict(cfg, sd, device="cuda", fp8=fp8)
    m.init_special_token_id(constants.SyntheticTokenizer())
    for rep in range(2):
        torch.manual_seed(77)
        out = m.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True)
        torch.cuda.synchronize()
        aux = m._last_aux
        dev_h = [m._ws.get(f"vit_h{i}", (2, m.vit.T, m.vit.D), torch.float32).cpu() for i in range(4)]
        det = O.ddetr_forward(sd, cfg.to_dict(), O.ddetr_inputs_from_hidden(tuple(dev_h)))
        sc = O.fuse_scores(det["logits_coco"], det["logits_sa1b"])
        print("fp8", fp8, "rep", rep, "scores relerr", util.relerr(aux["scores"], sc), "boxes relerr", util.relerr(aux["pred_boxes"], det["pred_boxes"]),
              "topk equal", torch.equal(aux["topk_idx"].cpu().long(), det["topk_idx"]))
        # recompute the proposer on the device from the SAME buffers after the forward finished
        h4 = [m._ws.get(f"vit_h{i}", (2, m.vit.T, m.vit.D), torch.float32) for i in range(4)]
        pb, s2, ti = m.proposer.forward(h4)
        print("   re-run proposer on final buffers: scores relerr vs oracle", util.relerr(s2, sc), "vs first run", util.relerr(s2, aux["scores"]))
        dbg = {}
        m.proposer.forward(h4, debug=dbg)
        print("   absmax of states", [float(h.abs().max()) for h in dev_h], "finite", all(bool(torch.isfinite(h).all()) for h in dev_h))
        for k in ("src", "memory", "enc_class"):
            print("   ", k, "relerr", util.relerr(dbg[k], det[k]), "absmax", float(det[k].abs().max()))
