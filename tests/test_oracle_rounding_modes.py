"""The oracle's operand-rounding modes on CPU (no device): rounding("bf16") / rounding("fp16") place the 16-bit roundings where
the device library of that operand type holds 16-bit values, everything else stays fp32.  Index-valued results do not move with
the mode when the stages behind the ViT are fed the same ViT states; the distance from the fp32 oracle scales with the format's
unit roundoff (fp16 ~ 1/8 of bf16) -- the figure the device builds are compared with in tests/test_fp16_gpu.py."""
import torch

from oracle import groma_oracle as O
from tests import util


def test_rounding_modes_scale_with_the_format():
    cfg, sd, tk = util.tiny_setup(seed=0)
    from groma_amd import synth
    images, ids = synth.make_inputs(cfg, tk, bs=1, seed=1234)
    cd = cfg.to_dict()
    with torch.no_grad():
        hs = tuple(O.vit_forward(sd, cd, images)[-4:])
        out = {}
        for mode in (None, "bf16", "fp16"):
            torch.manual_seed(5)
            with O.rounding(mode):
                out[mode] = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images, hidden_states=hs)
    for mode in ("bf16", "fp16"):
        assert torch.equal(out[mode]["nms_inds"][0], out[None]["nms_inds"][0])      # the proposer is fp32 in every mode
        assert torch.equal(out[mode]["input_ids"], out[None]["input_ids"])
    e_bf = util.relerr(out["bf16"]["logits"], out[None]["logits"])
    e_h = util.relerr(out["fp16"]["logits"], out[None]["logits"])
    print(f"oracle logits vs fp32: bf16-rounded {e_bf:.2e}, fp16-rounded {e_h:.2e}")
    assert 0 < e_h < e_bf / 4 and e_bf < 2e-2 and e_h < 2e-3
    # modes nest and restore
    with O.rounding("fp16"):
        with O.rounding(None):
            assert O._ROUND[0] is None
        assert O._ROUND[0] == "fp16"
    assert O._ROUND[0] is None
