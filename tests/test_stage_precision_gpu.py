"""Per-stage operand types (round 6, groma_amd.groma.parse_precision): `precision="<base>+<stage>:<type>..."` runs ONE stage of the
path -- region encoder, bridge, LLaMA attention blocks, LLaMA MLP blocks, head -- on another operand storage than its neighbours.
The stages exchange fp32 tensors only (ViT states, region / image tokens, the residual stream, logits), so nothing is converted
between them; tests/diag/precision_ablation.py uses this to measure, at full depth, which stage's operand rounding owns the
logit distance from the reference's fp32 pass (R: groma/model/groma.py:389-402, groma/eval/eval_rec.py:69).

Asserted on the tiny architecture against the fp32 oracle running its OWN ViT (unchained):
 * every stage moved to operand pairs on top of "hybrid" brings the logits CLOSER to fp32 than "hybrid" alone, the index-valued
   results stay equal, and all five stages on pairs is the "ref" build up to its embedding tables (logits within 1e-4);
 * the oracle's per-stage rounding selection (oracle.groma_oracle.rounding(mode, skip=...)) predicts the device's distance for the
   same stage set -- the correspondence the ablation's oracle-side table rests on;
 * a mixed model prefills through replayed graphs (two libraries inside one captured launch sequence) with the eager bits, and
   decodes (general kernels: the weight streams are single-type) to the oracle's greedy tokens."""
import pytest
import torch

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

# device stage -> the oracle's rounding-point patterns it owns (tests/diag/precision_ablation.py COARSE)
ORACLE_STAGE = dict(region=("region",), bridge=("bridge",), attn=("llm.qkv", "llm.pv", "llm.o"), mlp=("llm.gateup", "llm.down"), head=("head",))
BEHIND = ("bridge", "region", "embed", "llm", "head")


@pytest.fixture(scope="module")
def world(dev):
    from tests.golden.select_e2e_seeds import e2e_cfg
    from groma_amd import synth
    cfg = e2e_cfg("tiny")
    sd = synth.make_state_dict(cfg, 0)
    tk = util.TokenIds()
    seed = 715                                      # (a committed unchained seed: tests/golden/e2e_seeds.json)
    images, ids = synth.make_inputs(cfg, tk, 1, seed=seed)
    torch.manual_seed(seed)
    with torch.no_grad():
        own = O.vit_forward(sd, cfg.to_dict(), images)
        ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(own))
    return dict(cfg=cfg, sd=sd, tk=tk, seed=seed, images=images, ids=ids, ref=ref, own=own)


def _model(world, precision):
    from groma_amd import constants
    from groma_amd.groma import GromaModel
    m = GromaModel.from_state_dict(world["cfg"], world["sd"], "cuda", precision=precision)
    m.init_special_token_id(constants.SyntheticTokenizer())
    return m


def _forward(world, m):
    torch.manual_seed(world["seed"])
    out = m.forward(input_ids=world["ids"].clone(), images=world["images"], return_dict=True)
    aux = m._last_aux
    ref = world["ref"]
    eq = (torch.equal(aux["topk_idx"].cpu().long(), ref["det"]["topk_idx"]) and torch.equal(aux["nms_keep"][0], ref["nms_inds"][0])
          and torch.equal(aux["input_ids"], ref["input_ids"]))
    return out.logits.float().cpu(), eq


@pytest.mark.parametrize("base", ["hybrid", "hybrid-fp16"])
def test_each_stage_on_pairs_moves_the_logits_towards_fp32(world, base):
    ref = world["ref"]["logits"]
    fmt = "bf16" if base == "hybrid" else "fp16"
    lg0, eq0 = _forward(world, _model(world, base))
    e0 = util.relerr(lg0, ref)
    assert eq0
    sel = lambda **kw: O.rounding(fmt, **kw)
    cd, tkd = world["cfg"].to_dict(), util.tok_dict(world["tk"])

    def oracle_dist(skip):
        torch.manual_seed(world["seed"])
        with torch.no_grad(), sel(only=BEHIND, skip=skip):
            r = O.groma_forward(world["sd"], cd, tkd, world["ids"].clone(), world["images"], hidden_states=tuple(world["own"]))
        return util.relerr(r["logits"], ref)
    o0 = oracle_dist(())
    print(f"[{base}] device {e0:.3e} | oracle with the same roundings {o0:.3e}")
    assert 0.5 * o0 < e0 < 2.0 * o0
    for stage, pats in ORACLE_STAGE.items():
        m = _model(world, f"{base}+{stage}:ref")
        assert m.stage_precision[stage] == "ref" and m.mode == f"{base}+{stage}:ref"
        lg, eq = _forward(world, m)
        e, o = util.relerr(lg, ref), oracle_dist(pats)
        print(f"[{base}+{stage}:ref] device {e:.3e} | oracle predicts {o:.3e} | gain over {base}: {e0 / e:.2f}x")
        assert eq and e < e0 * 1.02
        assert 0.5 * o < e < 2.0 * o, (stage, e, o)             # the oracle-side ablation predicts the device
        del m
    m = _model(world, base + "".join(f"+{s}:ref" for s in ORACLE_STAGE))
    lg, eq = _forward(world, m)
    e = util.relerr(lg, ref)
    print(f"[{m.mode}] device {e:.3e}")
    e_embed = util.relerr(oracle_logits_embed_only(world, fmt), ref)     # (the embedding tables are the one 16-bit operand left)
    print(f"    what the {fmt} embedding tables alone cost: {e_embed:.3e}")
    assert eq and e < 1e-4 + 2.0 * e_embed


def oracle_logits_embed_only(world, fmt):
    """the fp32 oracle with only the embedding tables rounded (what is left 16-bit when all five stages run on pairs)"""
    torch.manual_seed(world["seed"])
    with torch.no_grad(), O.rounding(fmt, only=("embed",)):
        r = O.groma_forward(world["sd"], world["cfg"].to_dict(), util.tok_dict(world["tk"]), world["ids"].clone(), world["images"],
                            hidden_states=tuple(world["own"]))
    return r["logits"]


def test_mixed_stack_prefill_graph_equals_eager_and_generates(world):
    """a LLaMA stack whose attention blocks run on pairs and whose MLP blocks on fp16: the captured prefill (kernels of two libraries in
    one graph) replays the eager bits; generate() decodes through the general kernels and matches HF-greedy over the oracle"""
    from groma_amd import engine
    m = _model(world, "hybrid-fp16+attn:ref")
    assert m.llm.prec == dict(attn="ref", mlp="fp16", head="fp16")
    engine.GraphPool.enabled = False
    try:
        eager, _ = _forward(world, m)
    finally:
        engine.GraphPool.enabled = True
    outs = [_forward(world, m)[0] for _ in range(4)]       # third sighting captures, fourth replays
    assert m.llm.graphs.captures >= 1 and m.llm.graphs.replays >= 1
    for o in outs:
        assert torch.equal(o, eager)
    # the KV cache belongs to the attention stage: pairs here (f32 values handed out through the legacy tuple view)
    torch.manual_seed(world["seed"])
    out = m.forward(input_ids=world["ids"].clone(), images=world["images"], return_dict=True, use_cache=True)
    k0 = out.past_key_values[0][0]
    # (pairs hold what the stage computed from its fp32 input; that input -- embeddings, image / region tokens -- carries the fp16
    #  stages' rounding, 3.4e-4 here, where the all-fp16 build's layer-0 K is at ~1e-3)
    assert k0.dtype == torch.float32 and util.relerr(k0, world["ref"]["past"][0][0]) < 6e-4
    # greedy tokens (boosted <r_k> rows as in tests/test_parity_gpu.py::gen_setup so that no step is a near-tie)
    sd = dict(world["sd"])
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0
    sd["extra_lm_head.weight"] = w
    from groma_amd import constants
    from groma_amd.groma import GromaModel
    g = GromaModel.from_state_dict(world["cfg"], sd, "cuda", precision="hybrid-fp16+mlp:ref")
    g.init_special_token_id(constants.SyntheticTokenizer())
    g.generation_config.eos_token_id = None
    torch.manual_seed(world["seed"])
    got = g.generate(world["ids"].clone(), images=world["images"], max_new_tokens=5, return_dict_in_generate=True)
    torch.manual_seed(world["seed"])
    with torch.no_grad():
        ref = O.greedy_generate(sd, world["cfg"].to_dict(), util.tok_dict(world["tk"]), world["ids"].clone(), world["images"], 5, eos_token_id=-1)
    P = world["ids"].shape[1]
    n = util.assert_greedy_tokens_match(got.sequences[:, P:].cpu(), ref["sequences"][:, P:], ref["margins"], 0.05, "mixed-stack generate")
    assert n == 5
