"""Fixture selection for the UNCHAINED end-to-end index tests (tests/test_e2e_unchained_gpu.py): image seeds whose top-300
proposal ranking -- computed by the fp32 oracle from ITS OWN ViT states, as the reference does in one pass
(R: groma/model/groma.py:222-249, ddetr_transformer.py:546-568) -- has no near-tie.  Same rule as select_proposer_seeds.py:
scan seeds on the CPU oracle, keep those whose smallest adjacent gap among the top-301 sorted class logits is largest, commit
the seeds and gaps (e2e_seeds.json); the GPU test asserts that the device's max abs logit error is below gap / 4 before it
requires torch.equal on the indices (reference-precision build), and reports what survives for the 16-bit builds.

  python tests/golden/select_e2e_seeds.py            # rewrites tests/golden/e2e_seeds.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from groma_amd import config as gconfig, synth  # noqa: E402
from oracle import groma_oracle as O  # noqa: E402
from tests import util  # noqa: E402
from tests.golden.select_proposer_seeds import min_gap  # noqa: E402


def e2e_cfg(name):
    """"width": Groma-7B width at reduced depth (smoke()'s configuration, 6+6 DDETR); "tiny": the tiny architecture, 6+6 DDETR"""
    if name == "width":
        return gconfig.groma_7b_width(box_score_thres=0.0)
    return gconfig.groma_tiny(box_score_thres=0.0, ddetr_layers=6)


def main():
    out = {}
    tk = util.TokenIds()
    torch.set_num_threads(os.cpu_count() or 8)
    for name, n_scan, keep in (("tiny", 600, 3), ("width", 240, 3)):
        cfg = e2e_cfg(name)
        sd = synth.make_state_dict(cfg, 0, only=("perceiver.",))
        cd = cfg.to_dict()
        Q = cfg.perceiver_cfg.ddetr_cfg.two_stage_num_proposals
        rows = []
        with torch.no_grad():
            for seed in range(500, 500 + n_scan):
                images, _ = synth.make_inputs(cfg, tk, 1, seed=seed)
                hs = O.vit_forward(sd, cd, images)
                det = O.ddetr_forward(sd, cd, O.ddetr_inputs_from_hidden(hs))
                rows.append((min_gap(det["enc_class"], Q), seed))
        rows.sort(reverse=True)
        out[name] = [dict(seed=s, min_gap=g) for g, s in rows[:keep]]
        print(name, rows[:keep], "median gap", rows[len(rows) // 2][0], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "e2e_seeds.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
