"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.

Groma-7B WIDTH at reduced depth (config.groma_7b_width: every GEMM / conv / attention shape of the benchmark -- ViT 1024 x 16
heads, 1024-channel pyramid, 27 648-deep per-ROI conv, LLaMA 4096 / 11 008 / 32 heads, 32 114-wide head, 6+6 DDETR -- with 3 ViT
layers, 1 fusion round, 1 LLaMA layer), one image: the fp32 CPU oracle finishes in seconds."""
import os

import torch


def run_smoke():
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    from groma_amd import config, constants, synth
    from groma_amd.groma import GromaModel
    from oracle import groma_oracle as O

    torch.set_num_threads(max(1, min(64, os.cpu_count() or 8)))
    cfg = config.groma_7b_width(box_score_thres=0.0)
    sd = synth.make_state_dict(cfg, 0)
    model = GromaModel.from_state_dict(cfg, sd, "cuda:0")
    tk = constants.SyntheticTokenizer()
    model.init_special_token_id(tk)
    # image seed 648: the first committed UNCHAINED fixture at this width (tests/golden/e2e_seeds.json: the oracle's smallest adjacent gap
    # among its top-301 class logits is 1.4e-4, ~10x the distance between two fp32 evaluations of the proposer), so the torch.equal
    # assertions on index-valued results below are decided by the implementation, never by a near-tie (what happens on unselected
    # seeds is measured: profiles/r06_index_survival.txt, and reported per run by bench.py's `parity` block)
    images, ids = synth.make_inputs(cfg, model, bs=1, seed=648)
    torch.manual_seed(3)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True)
    torch.cuda.synchronize()
    hs = tuple(h.float().cpu() for h in model._last_aux["hidden4"])
    tok = dict(pad_token_id=model.pad_token_id, img_token_id=model.img_token_id, reg_token_id=model.reg_token_id,
               refer_box_token_id=model.refer_box_token_id, refer_feat_token_id=model.refer_feat_token_id,
               ground_box_token_id=model.ground_box_token_id, box_idx_token_ids=model.box_idx_token_ids)
    torch.manual_seed(3)
    with torch.no_grad():
        ref = O.groma_forward(sd, cfg.to_dict(), tok, ids.clone(), images, hidden_states=hs)
    assert torch.equal(model._last_aux["nms_keep"][0], ref["nms_inds"][0]), "NMS indices differ from the oracle"
    assert torch.equal(model._last_aux["input_ids"], ref["input_ids"]), "spliced token ids differ from the oracle"
    a, b = out.logits.float().cpu(), ref["logits"]
    err = ((a - b).norm() / b.norm()).item()
    assert err < 1e-2, f"logits relative error {err}"
    print(f"smoke ok (Groma-7B width, reduced depth): logits {tuple(a.shape)} rel-L2 vs fp32 oracle {err:.2e}, "
          f"N={ref['pred_boxes'][0].shape[0]} regions, L={ref['input_ids'].shape[1]}")
    # the same invocation through the reference-precision build (operand pairs, libgroma_hip_ref.so), compared UNCHAINED: the oracle
    # runs its own fp32 ViT, as the reference computes the path in one pass (R: groma/model/groma.py:222-280,389-402)
    del model, out
    torch.cuda.empty_cache()
    mref = GromaModel.from_state_dict(cfg, sd, "cuda:0", precision="ref")
    mref.init_special_token_id(tk)
    torch.manual_seed(3)
    out = mref.forward(input_ids=ids.clone(), images=images, return_dict=True)
    torch.cuda.synchronize()
    torch.manual_seed(3)
    with torch.no_grad():
        ref = O.groma_forward(sd, cfg.to_dict(), tok, ids.clone(), images)
    aux = mref._last_aux
    a, b = out.logits.float().cpu(), ref["logits"]
    same_shape = a.shape == b.shape
    err = ((a - b).norm() / b.norm()).item() if same_shape else float("inf")
    topk_eq = torch.equal(aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"])
    nms_eq = torch.equal(aux["nms_keep"][0], ref["nms_inds"][0])
    ids_eq = torch.equal(aux["input_ids"], ref["input_ids"])
    assert ids_eq and same_shape and err < 1e-3, f"precision='ref' unchained: spliced ids equal {ids_eq}, logits relative error {err}"
    print(f"smoke ok (precision='ref', oracle UNCHAINED = its own fp32 ViT): logits rel-L2 {err:.2e} (north star 1e-3), top-300 ids equal "
          f"{topk_eq}, NMS ids equal {nms_eq}, spliced ids equal {ids_eq}")
    # third / fourth leg: the per-stage builds -- only the ViT on operand pairs, everything behind it on 16-bit operands.  Round 6's
    # benchmarked build is "hybrid-fp16" (IEEE half behind the ViT), round 5's was "hybrid" (bf16).  The index-valued results must equal the
    # UNCHAINED oracle's (same `ref` as above); the logits keep the 16-bit format's distance behind the ViT (stated tolerances)
    del mref, out
    torch.cuda.empty_cache()
    for prec, tol, what in (("hybrid-fp16", 2e-3, "the benchmarked build: ViT on operand pairs, rest IEEE half"),
                            ("hybrid", 1.5e-2, "ViT on operand pairs, rest bf16")):
        mh = GromaModel.from_state_dict(cfg, sd, "cuda:0", precision=prec)
        mh.init_special_token_id(tk)
        torch.manual_seed(3)
        out = mh.forward(input_ids=ids.clone(), images=images, return_dict=True)
        torch.cuda.synchronize()
        aux = mh._last_aux
        a = out.logits.float().cpu()
        topk_eq = torch.equal(aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"])
        nms_eq = torch.equal(aux["nms_keep"][0], ref["nms_inds"][0])
        sel_eq = torch.equal(aux["sel_idx"][0], ref["nms_inds"][0][ref["perms"][0]])
        ids_eq = torch.equal(aux["input_ids"], ref["input_ids"])
        vit_err = max(((h.float().cpu() - r).norm() / r.norm()).item() for h, r in zip(aux["hidden4"], ref["hidden_states"][-4:]))
        err = ((a - b).norm() / b.norm()).item() if a.shape == b.shape else float("inf")
        assert topk_eq and nms_eq and sel_eq and ids_eq, f"precision={prec!r} unchained: top-300 {topk_eq}, NMS {nms_eq}, selection {sel_eq}, spliced ids {ids_eq}"
        assert vit_err < 1e-5 and err < tol, f"precision={prec!r} unchained: ViT states {vit_err}, logits {err} (stated tolerance {tol})"
        print(f"smoke ok (precision={prec!r} = {what}; oracle UNCHAINED): top-300 ids equal {topk_eq}, NMS ids equal {nms_eq}, shuffled selection "
              f"equal {sel_eq}, spliced ids equal {ids_eq}; ViT states rel-L2 {vit_err:.2e}, logits rel-L2 {err:.2e} (stated tolerance {tol:g})")
        del mh, out, aux
        torch.cuda.empty_cache()
