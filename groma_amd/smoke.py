"""__graft_entry__.smoke(): one tiny invocation of the hot path on cuda:0, checked against the oracle."""
import torch


def run_smoke():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from groma_amd import config, constants, synth
    from groma_amd.groma import GromaModel
    from oracle import groma_oracle as O

    cfg = config.groma_tiny(box_score_thres=0.0)
    sd = synth.make_state_dict(cfg, 0)
    model = GromaModel.from_state_dict(cfg, sd, "cuda:0")
    tk = constants.SyntheticTokenizer()
    model.init_special_token_id(tk)
    images, ids = synth.make_inputs(cfg, model, bs=1, seed=1)
    torch.manual_seed(3)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True)
    torch.cuda.synchronize()
    hs = tuple(model._ws.get(f"vit_h{i}", (1, model.vit.T, model.vit.D), torch.float32).cpu() for i in range(4))
    tok = dict(pad_token_id=model.pad_token_id, img_token_id=model.img_token_id, reg_token_id=model.reg_token_id,
               refer_box_token_id=model.refer_box_token_id, refer_feat_token_id=model.refer_feat_token_id,
               ground_box_token_id=model.ground_box_token_id, box_idx_token_ids=model.box_idx_token_ids)
    torch.manual_seed(3)
    ref = O.groma_forward(sd, cfg.to_dict(), tok, ids.clone(), images, hidden_states=hs)
    assert torch.equal(model._last_aux["nms_keep"][0], ref["nms_inds"][0]), "NMS indices differ from the oracle"
    a, b = out.logits.float().cpu(), ref["logits"]
    err = ((a - b).norm() / b.norm()).item()
    assert err < 2e-2, f"logits relative error {err}"
    print(f"smoke ok: logits {tuple(a.shape)} rel-L2 vs oracle {err:.2e}, N={ref['pred_boxes'][0].shape[0]} regions")
