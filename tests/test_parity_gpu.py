"""Stage-chained parity of the HIP path (through GromaModel -> C ABI) against the CPU oracle on the same seeded
inputs and the same reference-named state dict (tiny architecture so the fp32 oracle finishes in seconds).

Tolerances (stated per SURVEY 'Hard parts'): the device path computes the ViT, region encoder and LLM with bf16
operands / fp32 accumulation and fp32 residual streams, the proposer in fp32.  Against the fp32 oracle:
  * index-valued results (top-k proposal ids, NMS keep ids, shuffled box order, spliced token ids): bit-exact when the
    stage is fed identical inputs (stage chaining: the oracle consumes the device ViT states);
  * fp32 proposer tensors: 2e-4 relative L2;  * bf16 stages: 2e-2 relative L2 (measured ~3e-3);
  * logits: 2e-2 relative L2 and identical arg-max wherever the oracle's top-2 margin exceeds 4x the abs error."""
import pytest
import torch

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(dev):
    cfg, sd, tk = util.tiny_setup(seed=0)
    model = util.device_model(cfg, sd)
    from groma_amd import synth
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=1234)
    return cfg, sd, tk, model, images, ids


def test_vit_hidden_states(setup):
    cfg, sd, tk, model, images, ids = setup
    dev_h = model.vit.forward(images.cuda())
    ref_h = O.vit_forward(sd, cfg.to_dict(), images)[-4:]
    for a, b in zip(dev_h, ref_h):
        assert util.relerr(a, b) < 2e-2
    print("vit rel err", [util.relerr(a, b) for a, b in zip(dev_h, ref_h)])


def test_proposer_chained_exact_indices(setup):
    cfg, sd, tk, model, images, ids = setup
    dev_h = model.vit.forward(images.cuda())
    dbg = {}
    pred, scores, idx = model.proposer.forward(dev_h, debug=dbg)
    hs = tuple(h.cpu() for h in dev_h)
    det = O.ddetr_forward(sd, cfg.to_dict(), O.ddetr_inputs_from_hidden(hs))
    assert util.relerr(dbg["src"], det["src"]) < 2e-4
    assert util.relerr(dbg["memory"], det["memory"]) < 2e-4
    assert util.relerr(dbg["enc_class"], det["enc_class"]) < 2e-4
    # guard: exact index equality is only meaningful if the oracle's ranking has no near-ties
    srt = torch.sort(det["enc_class"], dim=1, descending=True)[0][:, : idx.shape[1] + 1]
    gap = (srt[:, :-1] - srt[:, 1:]).min().item()
    err = (dbg["enc_class"].cpu() - det["enc_class"]).abs().max().item()
    print("topk min gap", gap, "max abs err", err)
    if gap > 4 * err:
        assert torch.equal(idx.cpu().long(), det["topk_idx"])
        assert util.relerr(pred, det["pred_boxes"]) < 2e-4
        ref_scores = O.fuse_scores(det["logits_coco"], det["logits_sa1b"])
        assert util.relerr(scores, ref_scores) < 2e-4
    else:
        pytest.skip("seed has a near-tie in the two-stage ranking; exact-index check not meaningful")


def test_full_forward_chained(setup):
    cfg, sd, tk, model, images, ids = setup
    torch.manual_seed(77)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True)
    aux = model._last_aux
    dev_h = [model._ws.get(f"vit_h{i}", (2, model.vit.T, model.vit.D), torch.float32).cpu() for i in range(4)]
    torch.manual_seed(77)
    ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(dev_h))
    # index-valued: NMS keep, shuffled order, spliced ids
    for i in range(2):
        assert torch.equal(aux["nms_keep"][i], ref["nms_inds"][i]), "NMS keep indices differ"
        assert torch.equal(aux["sel_idx"][i], ref["nms_inds"][i][ref["perms"][i]]), "shuffled order differs"
        assert torch.allclose(out.hidden_states[1]["pred_boxes"][i].cpu(), ref["pred_boxes"][i], atol=1e-5)
    vis = out.hidden_states[1]
    e_img = util.relerr(vis["image_features"], ref["image_features"])
    e_reg = util.relerr(vis["region_features"], ref["region_features"])
    e_log = util.relerr(out.logits, ref["logits"])
    print("rel err image_features", e_img, "region_features", e_reg, "logits", e_log)
    assert e_img < 2e-2 and e_reg < 2e-2 and e_log < 2e-2
    # arg-max agreement where the oracle's margin is resolvable
    lg_d, lg_r = out.logits.float().cpu(), ref["logits"]
    abs_err = (lg_d - lg_r).abs().max().item()
    top2 = lg_r.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * abs_err
    assert clear.float().mean().item() > 0.5
    assert torch.equal(lg_d.argmax(-1)[clear], lg_r.argmax(-1)[clear])
    # region logits = last-position logits over <r0..r99>
    r0 = tk.box_idx_token_ids[0]
    assert util.relerr(lg_d[:, -1, r0:r0 + 100], lg_r[:, -1, r0:r0 + 100]) < 3e-2
    # KV cache view contract: pkv[0][0].shape == [bs, H, S, hd]
    pkv = out.past_key_values
    L = ref["input_ids"].shape[1]
    assert tuple(pkv[0][0].shape) == (2, cfg.llm_cfg.num_attention_heads, L, 128)
    assert util.relerr(pkv[0][0], ref["past"][0][0]) < 2e-2 and util.relerr(pkv[0][1], ref["past"][0][1]) < 2e-2


def test_forward_with_refer_and_ground_boxes(setup):
    """<refer_box>/<ground_box>/<refer_feat> rewriting (groma.py:283-309, a13) + ragged region counts + padding."""
    cfg, sd, tk, model, images, ids = setup
    ids = ids.clone()
    ids[0, 40], ids[0, 41], ids[0, 42] = tk.refer_box_token_id, tk.refer_feat_token_id, tk.ground_box_token_id
    ids[1, -5:] = tk.pad_token_id
    refer = [torch.tensor([[0.3, 0.3, 0.2, 0.2]]), torch.empty((0, 4))]
    ground = [torch.tensor([[0.7, 0.6, 0.25, 0.3]]), torch.empty((0, 4))]
    old = (model.config.max_region_num,)
    try:
        torch.manual_seed(5)
        ids_d = ids.clone()
        out = model.forward(input_ids=ids_d, images=images, refer_boxes=refer, ground_boxes=ground, return_dict=True)
        dev_h = [model._ws.get(f"vit_h{i}", (2, model.vit.T, model.vit.D), torch.float32).cpu() for i in range(4)]
        torch.manual_seed(5)
        ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, refer_boxes=refer,
                              ground_boxes=ground, hidden_states=tuple(dev_h))
    finally:
        (model.config.max_region_num,) = old
    assert torch.equal(model._last_aux["nms_keep"][0], ref["nms_inds"][0])
    assert ids_d[0, 40].item() in tk.box_idx_token_ids and ids_d[0, 42].item() in tk.box_idx_token_ids
    assert util.relerr(out.logits, ref["logits"]) < 2e-2
    assert out.logits.shape == ref["logits"].shape


def test_generate_matches_oracle_greedy(setup):
    cfg, sd, tk, model, images, ids = setup
    torch.manual_seed(9)
    g = model.generate(ids.clone(), images=images, use_cache=True, do_sample=False, max_new_tokens=3,
                       return_dict_in_generate=True, output_hidden_states=True, generation_config=model.generation_config)
    assert g.sequences.shape == (2, ids.shape[1] + 3)
    assert torch.equal(g.sequences[:, : ids.shape[1]].cpu(), ids)
    boxes = g.hidden_states[0][-1]['pred_boxes']
    assert len(boxes) == 2 and boxes[0].shape == (100, 4)
    dev_h = [model._ws.get(f"vit_h{i}", (2, model.vit.T, model.vit.D), torch.float32).cpu() for i in range(4)]
    torch.manual_seed(9)
    ref = O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, 3, eos_token_id=2,
                            hidden_states=tuple(dev_h))
    # token ids: exact unless the oracle's own top-2 margin at that step is inside the bf16 error band
    same = (g.sequences.cpu() == ref["sequences"])
    print("generated", g.sequences[:, -3:].tolist(), "oracle", ref["sequences"][:, -3:].tolist())
    assert same[:, : ids.shape[1]].all()
    # first generated token: must equal the oracle's wherever the oracle's own top-2 margin at the last prefill position
    # is resolvable under bf16 compute (abs logit error is ~1e-2 on these weights; see test_full_forward_chained)
    top2 = ref["prefill"]["logits"][:, -1].topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05
    assert same[:, ids.shape[1]][clear].all(), "first generated token differs from the oracle on a clear-margin row"


@pytest.mark.parametrize("eos", [None, "from_eager"])
def test_generate_graph_equals_eager(setup, eos):
    """The hipGraph-replayed decode loop (device-resident position / finished mask / output ids) must produce exactly
    the token ids of the eager per-kernel loop -- same kernels, same order, so bit-identical logits and arg-max --
    including HF's EOS handling (finished rows emit pad; stop when every row is finished)."""
    cfg, sd, tk, model, images, ids = setup
    gc = model.generation_config
    old = (gc.eos_token_id, getattr(gc, "pad_token_id", None), model.decode_graph)
    try:
        gc.eos_token_id = None
        model.decode_graph = False
        torch.manual_seed(9)
        free = model.generate(ids.clone(), images=images, max_new_tokens=12).cpu()
        if eos == "from_eager":  # pick row 0's 3rd generated token as EOS: row 0 stops early, row 1 may go on
            gc.eos_token_id = int(free[0, ids.shape[1] + 2])
            gc.pad_token_id = 0
        torch.manual_seed(9)
        eager = model.generate(ids.clone(), images=images, max_new_tokens=12).cpu()
        model.decode_graph = True
        for _ in range(2):  # second call replays the already-captured graph on a reused arena
            torch.manual_seed(9)
            out = model.generate(ids.clone(), images=images, max_new_tokens=12, return_dict_in_generate=True)
            assert torch.equal(out.sequences.cpu(), eager), (out.sequences[:, ids.shape[1]:].tolist(), eager[:, ids.shape[1]:].tolist())
            assert out.past_key_values.seq_len == out.past_key_values.seq_len  # arena stays readable
        if eos == "from_eager":
            assert eager.shape[1] <= ids.shape[1] + 12
            assert (eager[0, ids.shape[1] + 3:] == 0).all()  # finished row pads
    finally:
        gc.eos_token_id, gc.pad_token_id, model.decode_graph = old
