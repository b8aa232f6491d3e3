"""VERDICT r04 item 1, last bullet: would a 2-pass ViT (drop ONE cross term of hi.hi + hi.lo + lo.hi) keep the proposer's ranking?
Emulated on the CPU oracle, committed e2e seeds (tests/golden/e2e_seeds.json): dropping lo(A).hi(W) leaves the ACTIVATION operand of
every contraction at 11 mantissa bits while the weights keep 22 -- i.e. x -> f16(x) in front of every ViT linear and for one operand
of each attention product, everything else fp32.  Prints the proposer's class-logit error against the fp32 oracle next to the
smallest adjacent gap of the top-301 logits (the ranking survives only if error < gap / 4) and whether top-300 / NMS ids survive.
    python tests/diag/two_pass_vit_cpu.py [tiny|width]"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from groma_amd import synth
from oracle import groma_oracle as O
from tests import util
from tests.golden.select_e2e_seeds import e2e_cfg
from tests.golden.select_proposer_seeds import min_gap

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
torch.set_num_threads(min(32, os.cpu_count() or 8))
cfg = e2e_cfg(name)
sd = synth.make_state_dict(cfg, 0)
cd = cfg.to_dict()
tk = util.TokenIds()
h = lambda t: t.to(torch.float16).to(torch.float32)


def vit_two_pass(images):
    """oracle.vit_forward with the activation side of every contraction rounded to half (weights, accumulation, everything else fp32)"""
    orig_lin, orig_pv = O._lin16, O._softmax_pv
    O._lin16 = lambda x, sd_, nm, bias=True: F.linear(h(x), sd_[nm + ".weight"], sd_.get(nm + ".bias") if bias else None)
    O._softmax_pv = lambda scores, v: torch.softmax(scores, dim=-1, dtype=torch.float32) @ h(v)      # P.V: V at 11 bits, P full
    try:
        return O.vit_forward(sd, cd, images)   # (q @ k^T inside: both operands come out of 2-pass linears; k additionally rounded below is not
    finally:                                    #  modelled -- the errors reported are therefore a LOWER bound of the 2-pass build's)
        O._lin16, O._softmax_pv = orig_lin, orig_pv


for row in json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_seeds.json")))[name]:
    images, ids = synth.make_inputs(cfg, tk, 1, seed=row["seed"])
    with torch.no_grad():
        torch.manual_seed(row["seed"])
        ref = O.perceive(sd, cd, images)
        torch.manual_seed(row["seed"])
        two = O.perceive(sd, cd, images, hidden_states=vit_two_pass(images))
    Q = ref["det"]["topk_idx"].shape[1]
    gap = min_gap(ref["det"]["enc_class"], Q)
    err = (two["det"]["enc_class"] - ref["det"]["enc_class"]).abs().max().item()
    vit = max(util.relerr(a, b) for a, b in zip(two["hidden_states"][-4:], ref["hidden_states"][-4:]))
    slots = (two["det"]["topk_idx"] == ref["det"]["topk_idx"]).float().mean().item()
    nms = len(set(two["nms_inds"][0].tolist()) & set(ref["nms_inds"][0].tolist())) / max(1, ref["nms_inds"][0].numel())
    print(f"[2-pass ViT, {name}, seed {row['seed']}] ViT states rel-L2 {vit:.2e} | class-logit max abs err {err:.2e} vs min gap {gap:.2e} "
          f"(needs err < gap/4 = {gap / 4:.1e}) | top-300 slots equal {slots:.3f} | NMS ids kept {nms:.3f}", flush=True)
