"""Fixture selection for the bit-exact top-300 proposal-index test (groma/model/ddetr_transformer.py:546-568).

torch.topk over 1024 fp32 class logits is only a meaningful bit-exact target when the oracle's own ranking has no
near-ties: two logits closer than the fp32 evaluation error of EITHER implementation may legitimately swap.  This script
scans input seeds on the CPU oracle and keeps those whose smallest adjacent gap among the top-301 sorted logits is
largest; the chosen seeds and their gaps are committed in proposer_seeds.json, and the GPU test asserts (not skips) that
the device's max abs logit error is below gap/4 before requiring torch.equal on the indices.

  python tests/golden/select_proposer_seeds.py            # rewrites tests/golden/proposer_seeds.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from groma_amd import config as gconfig, synth  # noqa: E402
from oracle import groma_oracle as O  # noqa: E402


def proposer_cfg(width):
    """6+6-layer DDETR (the reference depth) behind a `width`-channel ViT state"""
    if width == 1024:
        return gconfig.groma_7b_width(box_score_thres=0.0)
    return gconfig.groma_tiny(box_score_thres=0.0, ddetr_layers=6)


def hidden_states(cfg, seed, bs=1):
    vc = cfg.perceiver_cfg.vis_encoder_cfg
    T = (cfg.image_size // vc.patch_size) ** 2 + 1
    g = torch.Generator().manual_seed(seed)
    return tuple(torch.randn((bs, T, vc.hidden_size), generator=g) for _ in range(4))


def min_gap(enc_class, k):
    srt = torch.sort(enc_class, dim=1, descending=True)[0][:, : k + 1]
    return (srt[:, :-1] - srt[:, 1:]).min().item()


def main():
    out = {}
    for width, n_scan, keep in ((256, 200, 4), (1024, 80, 3)):
        cfg = proposer_cfg(width)
        sd = {k: v for k, v in synth.make_state_dict(cfg, 0, only=("perceiver.input_proj", "perceiver.ddetr_transformer")).items()}
        cd = cfg.to_dict()
        rows = []
        with torch.no_grad():
            for seed in range(100, 100 + n_scan):
                det = O.ddetr_forward(sd, cd, O.ddetr_inputs_from_hidden(hidden_states(cfg, seed)))
                rows.append((min_gap(det["enc_class"], cfg.perceiver_cfg.ddetr_cfg.two_stage_num_proposals), seed))
        rows.sort(reverse=True)
        out[str(width)] = [dict(seed=s, min_gap=g) for g, s in rows[:keep]]
        print(width, rows[:keep], "median gap", rows[len(rows) // 2][0])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "proposer_seeds.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
