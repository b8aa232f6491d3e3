"""SURVEY 8(f) rows 3 and 4 on the MI355X:
  f3  a checkpoint directory in the reference's on-disk format (config.json + safetensors / .bin shards with the reference's
      parameter names) -> GromaModel.from_pretrained -> forward: bit-identical to the model built from the in-memory state
      dict, and within the parity tolerances of the CPU oracle run on the SAME files' tensors;
  f4  the REC evaluation loop of the reference (R: groma/eval/eval_rec.py:89-124) over the device model's generate()
      outputs: the `<r_k>` -> box lookup, IoU counters and the final metrics equal those computed from the oracle's greedy
      outputs, and equal a literal restatement of the reference loop."""
import pytest
import torch

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def test_from_pretrained_on_device_matches_state_dict_model_and_oracle(dev, tmp_path):
    from groma_amd import constants, synth
    from groma_amd.groma import GromaModel
    from safetensors.torch import save_file
    cfg, sd, tk = util.tiny_setup(seed=3)
    d = tmp_path / "ckpt"
    cfg.save_pretrained(d)
    keys = sorted(sd)
    third = len(keys) // 3
    for i in range(3):
        part = keys[i * third:(i + 1) * third] if i < 2 else keys[2 * third:]
        save_file({k: sd[k].contiguous() for k in part}, str(d / f"model-{i + 1:05d}-of-00003.safetensors"))
    model = GromaModel.from_pretrained(str(d)).cuda()   # eval_rec.py:69 form (no torch_dtype): the default bf16 operand build
    model.init_special_token_id(constants.SyntheticTokenizer())
    ref_model = util.device_model(cfg, sd)
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=99)
    outs = []
    for m in (model, ref_model):
        torch.manual_seed(4)
        outs.append(m.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True))
    assert torch.equal(outs[0].logits, outs[1].logits)  # same bytes in, same bytes out
    dev_h = [h.cpu() for h in model._last_aux["hidden4"]]
    torch.manual_seed(4)
    ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(dev_h))
    assert torch.equal(model._last_aux["nms_keep"][0], ref["nms_inds"][0])
    assert util.relerr(outs[0].logits, ref["logits"]) < 2e-2
    # an EXPLICIT torch_dtype=torch.float32 asks for the reference's fp32 arithmetic: the reference-precision build (operand pairs)
    m32 = GromaModel.from_pretrained(str(d), torch_dtype=torch.float32)
    m32.init_special_token_id(constants.SyntheticTokenizer())
    assert m32.precision == "ref" and model.precision == "bf16"
    torch.manual_seed(4)
    o32 = m32.forward(input_ids=ids.clone(), images=images, return_dict=True)
    torch.manual_seed(4)
    ref32 = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images,
                            hidden_states=tuple(h.cpu() for h in m32._last_aux["hidden4"]))   # (no gap-selected seed here: chained)
    assert torch.equal(m32._last_aux["nms_keep"][0], ref32["nms_inds"][0]) and torch.equal(m32._last_aux["input_ids"], ref32["input_ids"])
    assert util.relerr(o32.logits, ref32["logits"]) < 1e-4
    del m32
    # precision="hybrid" through from_pretrained: the build bench.py runs -- the ViT packed as operand pairs, the rest bf16; bitwise the
    # model built from the in-memory state dict, its index-valued results those of the UNCHAINED oracle given the pair ViT's states
    mh = GromaModel.from_pretrained(str(d), precision="hybrid")
    mh.init_special_token_id(constants.SyntheticTokenizer())
    assert (mh.precision, mh.vit_precision, mh.mode) == ("bf16", "ref", "hybrid")
    mh_ref = GromaModel.from_state_dict(cfg, sd, "cuda", precision="hybrid")
    mh_ref.init_special_token_id(constants.SyntheticTokenizer())
    torch.manual_seed(4)
    oh = mh.forward(input_ids=ids.clone(), images=images, return_dict=True)
    torch.manual_seed(4)
    assert torch.equal(oh.logits, mh_ref.forward(input_ids=ids.clone(), images=images, return_dict=True).logits)
    assert util.relerr(oh.logits, outs[0].logits) < 5e-1 and not torch.equal(mh._last_aux["hidden4"][-1], model._last_aux["hidden4"][-1])
    del mh, mh_ref
    # .bin shards load the same
    d2 = tmp_path / "bin"
    cfg.save_pretrained(d2)
    torch.save({k: sd[k] for k in keys[: len(keys) // 2]}, str(d2 / "pytorch_model-00001-of-00002.bin"))
    torch.save({k: sd[k] for k in keys[len(keys) // 2:]}, str(d2 / "pytorch_model-00002-of-00002.bin"))
    m2 = GromaModel.from_pretrained(str(d2))
    m2.init_special_token_id(constants.SyntheticTokenizer())
    torch.manual_seed(4)
    assert torch.equal(m2.forward(input_ids=ids.clone(), images=images, return_dict=True).logits, outs[0].logits)


def _reference_rec_loop(sequences, prompt_len, pred_boxes, gt_boxes, box_idx_token_ids, thr=0.5):
    """literal restatement of R: groma/eval/eval_rec.py:103-121 (one image per iteration)"""
    from groma_amd.evalkit import cxcywh_to_xyxy, pairwise_iou
    m_iou = hit = invalid = count = 0.0
    for i in range(sequences.shape[0]):
        count += 1
        output_ids = sequences[i, prompt_len:]
        box_inds = [box_idx_token_ids.index(int(t)) for t in output_ids if int(t) in box_idx_token_ids]
        box_inds = [k for k in box_inds if k < len(pred_boxes[i])]
        if len(box_inds) == 0:
            invalid += 1
            continue
        sel = pred_boxes[i][box_inds]
        ious = pairwise_iou(cxcywh_to_xyxy(sel), cxcywh_to_xyxy(gt_boxes[i])).max(dim=-1).values
        m_iou += float(ious[0])
        hit += 1.0 if float(ious[0]) > thr else 0.0
    return {"iou@0.5 accu": hit / count, "m_iou": m_iou / count, "missing percentage": invalid / count, "count": int(count)}


def test_rec_eval_loop_over_generate_outputs_equals_oracle(dev):
    from groma_amd import synth
    from groma_amd.evalkit import RecMeter
    cfg, sd, tk = util.tiny_setup(seed=0)
    sd = dict(sd)
    # random-init weights never emit <r_k>: boost the region-token rows of the extra head so that grounded answers appear
    w = sd["extra_lm_head.weight"].clone()
    n_new = w.shape[0]
    w[n_new - 100:] *= 40.0
    sd["extra_lm_head.weight"] = w
    model = util.device_model(cfg, sd)
    model.generation_config.eos_token_id = None
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=321)
    torch.manual_seed(12)
    g = model.generate(ids.clone(), images=images, max_new_tokens=3, return_dict_in_generate=True, output_hidden_states=True)
    dev_h = [h.cpu() for h in model._last_aux["hidden4"]]
    torch.manual_seed(12)
    ref = O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, 3, eos_token_id=-1, hidden_states=tuple(dev_h))
    P = ids.shape[1]
    new = g.sequences[:, P:].cpu()
    assert any(int(t) in tk.box_idx_token_ids for t in new.reshape(-1)), "the boosted head must produce grounded answers"
    util.assert_greedy_tokens_match(new, ref["sequences"][:, P:], ref["margins"], 0.05, "rec")
    boxes_d = [b.float().cpu() for b in g.hidden_states[0][-1]["pred_boxes"]]
    gt = [boxes_d[i][5:6].clone() * torch.tensor([1.0, 1.0, 1.1, 0.9]) for i in range(2)]  # ground truth near candidate 5
    md, mo = RecMeter(0.5), RecMeter(0.5)
    md.update(g.sequences.cpu(), P, boxes_d, gt, tk.box_idx_token_ids)
    mo.update(ref["sequences"], P, ref["pred_boxes"], gt, tk.box_idx_token_ids)
    sd_, so_ = md.summary(), mo.summary()
    if torch.equal(new, ref["sequences"][:, P:]):
        assert sd_["count"] == so_["count"] == 2
        for k in sd_:
            assert abs(sd_[k] - so_[k]) < 1e-5, (k, sd_, so_)
    lit = _reference_rec_loop(g.sequences.cpu(), P, boxes_d, gt, list(tk.box_idx_token_ids))
    for k in lit:
        assert abs(lit[k] - sd_[k]) < 1e-12, (k, lit, sd_)
    assert sd_["missing percentage"] < 1.0
