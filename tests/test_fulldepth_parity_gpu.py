"""Groma-7B at its real DEPTH with DISTINCT per-layer weights on the MI355X against the CPU oracle, fp32 and bf16-rounded
(tests/diag/fulldepth_parity.py: 24 ViT layers, 6+6 DDETR, 5 fusion rounds, 100 regions, 32 LLaMA layers, logits for all 582
positions; every parameter drawn per name on the device and served to the oracle through a lazy state dict).  ~1.5 minutes,
most of it the two oracle passes on the host cores.  R: groma/model/groma.py:202-427 end to end.

Asserted: index-valued results identical (top-300 ids, NMS ids, spliced ids); the device is no further from the fp32 oracle
than 1.5x what the bf16 format itself costs at this depth (bf16-rounded oracle <-> fp32 oracle), stage by stage; the
bf16-rounded oracle is the closer reference; arg-max identical on every clear-margin position.
Round 4: precision="ref" (operand pairs) is held to north_star's 1e-3 against an UNCHAINED fp32 oracle.  Round 6: the benchmarked build
("hybrid-fp16") unchained with its stated tolerance (this replaces round 4's chained all-fp16 run: same format distance, 3.3e-3), and
configs[4] (e4m3) at the real depth.  Measured numbers: profiles/r04_fulldepth_*.txt, r06_precision_ablation.txt, r06_fulldepth_fp8.txt."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _diag():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("fulldepth_parity", os.path.join(here, "diag", "fulldepth_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _format_gates(r, slack):
    """the device is no further from the fp32 oracle than `slack` x what the operand FORMAT itself costs at that depth, and the
    rounded oracle is the closer reference (or as close)"""
    for name in ("image_tokens", "region_tokens", "k0", "k31", "logits", "region_logits"):
        d32, d16, fmt = r[name]
        assert d32 <= slack * fmt, (name, r[name])
        assert d16 <= d32 * 1.05, (name, r[name])


# ABSOLUTE bounds (relative L2 against the fp32 oracle) at the real depth, next to the format-relative gates: what a caller of the
# bf16-behind-the-ViT builds can rely on whatever the format's own distance happens to be.  Measured: image
# tokens 3.4e-3, region tokens 5.2e-3, K cache layer 0 / 31 4.7e-3 / 2.6e-2, logits 2.6e-2, <r_k> logits 3.4e-2.
BF16_ABS = dict(image_tokens=5e-3, region_tokens=7e-3, k0=6e-3, k31=3.2e-2, logits=3.2e-2, region_logits=4.5e-2)


def _absolute_gates(r, bounds):
    for name, bound in bounds.items():
        assert r[name][0] <= bound, (name, r[name], bound)


def test_full_depth_distinct_weights_vs_both_oracles(dev):
    r = _diag().run()
    assert r["topk_equal"] and r["nms_equal"] and r["ids_equal"] and r["L"] == 582
    _format_gates(r, 1.5)
    _absolute_gates(r, BF16_ABS)
    for d32, d16, fmt in r["vit"][1:]:
        assert d32 <= 1.5 * fmt and d32 < 1e-2
    d32, d16, fmt = r["logits"]
    assert d32 <= 1.1 * fmt                            # 32 layers deep: the bf16 format's own distance (2.6e-2), within 10 %
    assert r["argmax_agree_clear"] == 1.0
    assert r["argmax_agree"] >= r["argmax_agree_bf16_oracle"] - 0.05


def test_full_depth_reference_precision_unchained(dev):
    """precision="ref" (operand pairs, 3-pass contractions) against ONE fp32 oracle pass that runs its own ViT: north_star's
    "logits within 1e-3 of reference" and configs[1]'s "box-index bit-exact vs ref" at the real depth, no stage chaining
    (R: groma/model/groma.py:222-280,389-402; eval loads fp32 weights, groma/eval/eval_rec.py:69)"""
    r = _diag().run(precision="ref")
    un = r["unchained"]
    assert r["topk_equal"] and r["nms_equal"] and r["ids_equal"] and r["L"] == 582
    assert un["topk_pos_equal"] == 1.0 and un["nms_equal"]   # (oracle min gap / device logit error are printed beside it)
    for d32, _, _ in r["vit"]:
        assert d32 < 1e-4
    for name in ("image_tokens", "region_tokens", "k0", "k31", "logits", "region_logits"):
        assert r[name][0] <= 1e-3, (name, r[name])    # the north-star tolerance, every stage, 24 + 32 layers deep
    assert r["argmax_agree"] >= 0.999


def test_full_depth_hybrid_precision_unchained(dev):
    """precision="hybrid" -- the build bench.py's headline runs (round 5): only the ViT on operand pairs, bridge / region encoder /
    LLaMA on bf16 operands -- against oracle passes that run their OWN fp32 ViT (no stage chaining): the index-valued results of the
    whole path equal the reference's at the real depth (configs[1] "box-index bit-exact vs ref", R: groma/model/groma.py:222-280),
    the ViT states are within 1e-4 of fp32, and everything behind the ViT keeps the bf16 format's own distance (R: :389-402)."""
    r = _diag().run(precision="hybrid")
    un = r["unchained"]
    assert r["topk_equal"] and r["nms_equal"] and r["ids_equal"] and r["L"] == 582
    assert un["topk_pos_equal"] == 1.0 and un["nms_equal"]
    for d32, _, _ in r["vit"]:
        assert d32 < 1e-4
    _format_gates(r, 1.5)
    _absolute_gates(r, BF16_ABS)      # the benchmarked build: absolute bounds, unchained
    d32, d16, fmt = r["logits"]
    assert d32 <= 1.1 * fmt
    assert r["argmax_agree_clear"] == 1.0


def test_full_depth_benchmarked_build_hybrid_fp16_unchained(dev):
    """precision="hybrid-fp16" -- the build bench.py's headline runs since round 6 (the ViT on operand pairs, bridge / region encoder /
    LLaMA on IEEE-half operands): against oracle passes that run their OWN fp32 ViT, at the real depth, the index-valued results equal the
    reference's and the logits are within the STATED tolerance 5e-3 (measured 3.3e-3 = the half format's own distance at this depth:
    profiles/r06_precision_ablation.txt shows no stage set under 2.2x the step that gets closer).  R: groma/model/groma.py:222-280,389-402."""
    r = _diag().run(precision="hybrid-fp16")
    un = r["unchained"]
    assert r["topk_equal"] and r["nms_equal"] and r["ids_equal"] and r["L"] == 582
    assert un["topk_pos_equal"] == 1.0 and un["nms_equal"]
    for d32, _, _ in r["vit"]:
        assert d32 < 1e-4
    _format_gates(r, 1.5)
    d32, d16, fmt = r["logits"]
    assert d32 <= 5e-3 and d32 <= 1.1 * fmt            # the stated tolerance of the headline build
    assert r["region_logits"][0] <= 7e-3 and r["k31"][0] <= 5e-3 and r["image_tokens"][0] <= 1e-3 and r["region_tokens"][0] <= 1.5e-3
    assert r["argmax_agree"] >= 0.98 and r["argmax_agree_clear"] == 1.0


def test_full_depth_fp8(dev):
    """BASELINE configs[4] at the real depth ("logits within stated tol vs bf16"): the e4m3 model ("hybrid" + fp8=True) against the bf16
    device path of the same weights, 32 LLaMA layers deep.  The stated tolerance is the e4m3 FORMAT's own distance at this depth -- the
    e4m3-rounded oracle is 3.27e-1 from the fp32 oracle on the logits, the device 3.26e-1 from both the bf16 device and the fp32 oracle
    (profiles/r06_fulldepth_fp8.txt, tests/diag/fulldepth_parity.py fp8: the oracle passes take two minutes and are not repeated here) -- not
    the one-layer 1e-1 of tests/test_fp8_width_gpu.py.  R: groma/model/groma.py:389-397 (the 32-layer stack)."""
    r = _diag().run_fp8(oracle=False)
    assert r["topk_equal"] and r["nms_equal"] and r["ids_equal"]          # the ViT / proposer are not e4m3: the index contract holds
    assert 1e-1 < r["logits"] <= 4e-1, r                                  # measured 3.27e-1 (the format's 3.27e-1)
    assert r["k31"] <= 4e-1 and r["region_tokens"] <= 8e-2 and r["region_logits"] <= 4.5e-1, r
    assert r["argmax"] >= 0.35, r                                         # (measured 0.46; the e4m3-rounded oracle agrees with fp32 at 0.44)
