"""UNCHAINED end-to-end parity: the oracle runs its OWN fp32 ViT and everything behind it in one pass, exactly the reference's
computation (R: groma/model/groma.py:222-280 ViT -> mean-of-4 -> DDETR -> top-300 -> NMS; :317-402 splice -> LLaMA -> heads; eval
loads fp32 weights, groma/eval/eval_rec.py:69).  Every other index comparison in tests/ hands the oracle the DEVICE's ViT states
(stage chaining); this file is the half SURVEY 7 "Hard parts" asked for on top of that: an end-to-end check on committed seeds
with a minimum-gap assertion (tests/golden/select_e2e_seeds.py -> e2e_seeds.json).

 * precision="ref" (operand pairs, 3-pass contractions): top-300 ids, NMS ids, the shuffled selection and the spliced token ids
   are torch.equal to the oracle's on every committed seed, after ASSERTING that the device's class-logit error is below a
   quarter of the oracle's smallest adjacent gap; logits within 1e-4 at this depth (1e-3 at full depth:
   tests/test_fulldepth_parity_gpu.py).
 * precision="hybrid" / "hybrid-fp16" (round 5: ONLY the ViT on operand pairs, bridge / region encoder / LLaMA on bf16 / fp16 operands
   -- the build bench.py's headline runs): the same index equalities are ASSERTED, with the same gap guard; the ViT states are within
   1e-5 of the oracle's, the logits keep the 16-bit format's distance (asserted at the bounds of the chained tests).
 * bf16 / fp16 operands: what survives is MEASURED and printed (fraction of top-300 slots / set overlap, NMS ids, spliced ids),
   with the oracle's gap next to the device's logit error.  A 16-bit ViT perturbs the class logits by ~1e-3, ten times the
   gaps a random-init proposer leaves between neighbours, so equality is not expected there; only sanity bounds are asserted
   (>= 80 % of the top-300 set, every kept box a real proposal).  Numbers: profiles/r04_e2e_unchained.txt."""
import json
import os

import pytest
import torch

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rows(name):
    with open(os.path.join(HERE, "golden", "e2e_seeds.json")) as f:
        return json.load(f)[name]


_MODELS = {}


def _setup(name, precision):
    """one device model per (architecture, operand type) for the whole module; the state dict is shared with the oracle"""
    from tests.golden.select_e2e_seeds import e2e_cfg
    from groma_amd import constants, synth
    from groma_amd.groma import GromaModel
    if name not in _MODELS:
        cfg = e2e_cfg(name)
        _MODELS[name] = dict(cfg=cfg, sd=synth.make_state_dict(cfg, 0))
    ent = _MODELS[name]
    if precision not in ent:
        m = GromaModel.from_state_dict(ent["cfg"], ent["sd"], "cuda", precision=precision)
        m.init_special_token_id(constants.SyntheticTokenizer())
        ent[precision] = m
    return ent["cfg"], ent["sd"], ent[precision]


_ORACLE = {}


def _oracle(name, seed):
    """the fp32 oracle's unchained forward for this image seed (cached: three operand types are compared with it)"""
    from groma_amd import synth
    key = (name, seed)
    if key not in _ORACLE:
        cfg, sd = _MODELS[name]["cfg"], _MODELS[name]["sd"]
        tk = util.TokenIds()
        images, ids = synth.make_inputs(cfg, tk, 1, seed=seed)
        torch.manual_seed(seed)
        with torch.no_grad():
            ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images)   # hidden_states=None: its own ViT
        _ORACLE[key] = (images, ids, ref)
    return _ORACLE[key]


def _compare(name, precision, row):
    from tests.golden.select_proposer_seeds import min_gap
    cfg, sd, model = _setup(name, precision)
    seed = row["seed"]
    images, ids, ref = _oracle(name, seed)
    torch.manual_seed(seed)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True)
    aux = model._last_aux
    dbg = {}
    model.proposer.forward(aux["hidden4"], debug=dbg)
    o_cls = ref["det"]["enc_class"]
    Q = aux["topk_idx"].shape[1]
    gap = min_gap(o_cls, Q)
    err = (dbg["enc_class"].float().cpu() - o_cls).abs().max().item()
    d_ids, o_ids = aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"]
    r = dict(gap=gap, err=err, topk_equal=torch.equal(d_ids, o_ids), topk_slots=(d_ids == o_ids).float().mean().item(),
             topk_set=len(set(d_ids[0].tolist()) & set(o_ids[0].tolist())) / Q,
             nms_equal=torch.equal(aux["nms_keep"][0], ref["nms_inds"][0]),
             nms_set=len(set(aux["nms_keep"][0].tolist()) & set(ref["nms_inds"][0].tolist())) / max(1, ref["nms_inds"][0].numel()),
             sel_equal=torch.equal(aux["sel_idx"][0], ref["nms_inds"][0][ref["perms"][0]]),
             ids_equal=torch.equal(aux["input_ids"], ref["input_ids"]),
             vit=max(util.relerr(a, b) for a, b in zip(aux["hidden4"], ref["hidden_states"][-4:])),
             logits=util.relerr(out.logits, ref["logits"]) if aux["input_ids"].shape == ref["input_ids"].shape else float("nan"))
    print(f"[unchained {name} {precision} seed {seed}] oracle min gap {gap:.2e} (committed {row['min_gap']:.2e}) | device class-logit err {err:.2e} | "
          f"top-300: equal {r['topk_equal']}, slots {r['topk_slots']:.3f}, set {r['topk_set']:.3f} | NMS: equal {r['nms_equal']}, set {r['nms_set']:.3f} | "
          f"shuffled selection equal {r['sel_equal']} | spliced ids equal {r['ids_equal']} | ViT {r['vit']:.2e} | logits {r['logits']:.2e}")
    assert abs(gap - row["min_gap"]) <= 0.25 * row["min_gap"], "fixture drifted: re-run tests/golden/select_e2e_seeds.py"
    return r


@pytest.mark.parametrize("name", ["tiny", "width"])
def test_reference_precision_indices_bit_exact_unchained(dev, name):
    for row in _rows(name):
        r = _compare(name, "ref", row)
        assert r["gap"] > 4 * r["err"], "fixture no longer resolves the ranking for the reference-precision build"   # asserted, never skipped
        assert r["topk_equal"] and r["nms_equal"] and r["sel_equal"] and r["ids_equal"]
        assert r["vit"] < 1e-5 and r["logits"] < 1e-4


@pytest.mark.parametrize("precision", ["hybrid", "hybrid-fp16"])
@pytest.mark.parametrize("name", ["tiny", "width"])
def test_hybrid_precision_indices_bit_exact_unchained(dev, name, precision):
    """the FAST build holds configs[1]'s "box-index bit-exact vs ref" end to end: only the ViT feeds the fp32 proposer, so only the
    ViT runs on operand pairs (R: groma/model/groma.py:222-280: ViT -> mean-of-4 -> DDETR -> top-300 -> NMS in one fp32 pass)"""
    for row in _rows(name):
        r = _compare(name, precision, row)
        assert r["gap"] > 4 * r["err"], "fixture no longer resolves the ranking for the hybrid build"   # asserted, never skipped
        assert r["topk_equal"] and r["nms_equal"] and r["sel_equal"] and r["ids_equal"]
        assert r["vit"] < 1e-5
        assert r["logits"] < (1.5e-2 if precision == "hybrid" else 2e-3)   # the 16-bit format behind the ViT (chained tests: 7.8e-3 / 9.8e-4)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["tiny", "width"])
def test_16bit_operands_unchained_survival_is_measured(dev, name, precision):
    """reported, with sanity bounds: a 16-bit ViT legitimately reorders near-tied proposals (module docstring)"""
    for row in _rows(name):
        r = _compare(name, precision, row)
        assert r["topk_set"] >= 0.8 and r["nms_set"] >= 0.5
        assert r["vit"] < (2e-2 if precision == "bf16" else 3e-3)


def test_hybrid_generate_tokens_and_boxes_unchained(dev):
    """generate() of the benchmarked build against HF-greedy over the oracle running its OWN ViT (R: groma/eval/eval_rec.py:93-104 over
    groma/model/groma.py:176-200,376-402): the selected boxes the caller reads back (`hidden_states[0][-1]['pred_boxes']`) are the
    oracle's boxes in the oracle's order, and every greedy token equals the oracle's -- a mismatch is only accepted at a step whose
    oracle top-2 margin is inside the bf16 band (tests/util.assert_greedy_tokens_match).  The <r_k> rows of extra_lm_head are boosted
    as in tests/test_parity_gpu.py::gen_setup so that the margins are far outside that band."""
    from tests.golden.select_e2e_seeds import e2e_cfg
    from groma_amd import constants, synth
    from groma_amd.groma import GromaModel
    cfg = e2e_cfg("tiny")
    sd = dict(synth.make_state_dict(cfg, 0))
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0
    sd["extra_lm_head.weight"] = w
    model = GromaModel.from_state_dict(cfg, sd, "cuda", precision="hybrid")
    model.init_special_token_id(constants.SyntheticTokenizer())
    model.generation_config.eos_token_id = None
    tk = util.TokenIds()
    total = 0
    for row in _rows("tiny"):
        images, ids = synth.make_inputs(cfg, tk, 1, seed=row["seed"])
        torch.manual_seed(row["seed"])
        out = model.generate(ids.clone(), images=images, max_new_tokens=5, return_dict_in_generate=True, output_hidden_states=True)
        torch.manual_seed(row["seed"])
        with torch.no_grad():
            ref = O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, 5, eos_token_id=-1)   # hidden_states=None: its own ViT
        boxes = out.hidden_states[0][-1]["pred_boxes"][0].float().cpu()
        assert boxes.shape == ref["pred_boxes"][0].shape and torch.allclose(boxes, ref["pred_boxes"][0], atol=1e-5)
        P = ids.shape[1]
        total += util.assert_greedy_tokens_match(out.sequences[:, P:].cpu(), ref["sequences"][:, P:], ref["margins"], 0.05,
                                                 f"hybrid generate, seed {row['seed']}")
    assert total >= 10
