"""BASELINE configs[4] (OCP e4m3 operands) at Groma-7B WIDTH against the CPU oracle's e4m3-rounded mode.

oracle.groma_oracle.rounding("e4m3") evaluates the DINOv2 / LLaMA linears with e4m3 operands formed exactly as the device
forms them (per-row dynamic activation scales amax/448 computed by the fused norm -> e4m3 kernel or the row quantiser,
per-output-channel weight scales, fp32 accumulation, dequantisation acc * w_scale[n] * a_scale[m] in the epilogue); everything
else (attention, bridge, the region encoder's 1x1 / round-0 convs and linears, residual streams) is the bf16-rounded oracle; round 4
added lm_head (per-row scales like every other norm -> GEMM) and the region encoder's 3x3 convs from fuse round 1 on plus the
per-ROI conv (static activation scale of the producing GroupNorm, oracle conv_act_scale).  The reference has no fp8 path
(R: groma/eval/run_groma.py:43-61 offers fp16 / 8-bit / 4-bit loading only): configs[4] is defined by BASELINE.json and
"logits within stated tol vs bf16" is stated here.

  * teacher-forced, kernel by kernel at the benchmark's shapes (ViT 1025 x {3072, 1024, 4096} x 1024 / 1025 x 1024 x 4096;
    LLaMA 582 x {12288, 4096, 22016} x 4096 / 582 x 4096 x 11008): quantisers bit-exact on their grid up to tie flips
    (tolerance 2e-3 on the de-quantised rows: one e4m3 step is 6-12 % of a value, so a single 1-ulp-fp32 tie flip per ~10^5
    elements already shows as 1e-4), GEMMs <= 1e-3 (bf16 out, measured 2.4-3.3e-4) / 2e-4 (fp32 out, measured 4.6-6.8e-5: the e4m3 matrix unit's
    own block accumulation, see TOL_F32OUT) on IDENTICAL quantised operands;
  * weights: the device's (w8, scale) equal the oracle's quantisation of the same matrices bit for bit;
  * chained: e4m3 logits vs the e4m3-rounded oracle (implementation error) and vs the bf16 device path / fp32 oracle (the
    format's own error: stated tolerance 1e-1 relative L2 at one LLaMA layer + head, 8e-2 on the ViT states after 3 layers)."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_QUANT = 1e-3, 2e-3
# fp32-output e4m3 GEMMs: the oracle accumulates the exact e4m3 products in fp32; v_mfma_f32_16x16x32_fp8_fp8 aligns the 32 products
# of a block to their largest exponent with a finite number of guard bits before adding (measured 4.6e-5 ... 6.8e-5 relative L2 at
# K = 1024 ... 11 008, round 2 measured 5e-5 against f64 at 2328x4096x4096): a property of the matrix unit, bounded here at 2e-4
TOL_F32OUT = 2e-4


@pytest.fixture(scope="module")
def f8(dev):
    from groma_amd import config as gconfig, constants, engine, synth
    from groma_amd.groma import GromaModel
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 8)))
    cfg = gconfig.groma_7b_width(box_score_thres=0.0, num_fuse=2)   # two fuse rounds: round 1 is the first e4m3 3x3 conv
    sd = synth.make_state_dict(cfg, 0)
    tk = util.TokenIds()
    m8 = GromaModel.from_state_dict(cfg, sd, device="cuda", fp8=True)
    m8.init_special_token_id(constants.SyntheticTokenizer())
    m8.capture_embeds = True
    images, ids = synth.make_inputs(cfg, tk, bs=1, seed=1234)
    torch.manual_seed(77)
    engine.TRACE = {}
    try:
        with torch.no_grad():
            out = m8.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True)
        torch.cuda.synchronize()
        trace = {k: v.float().cpu() for k, v in engine.TRACE.items()}
    finally:
        engine.TRACE = None
    aux = m8._last_aux
    d = dict(trace=trace, logits=out.logits.float().cpu(), embeds=aux["inputs_embeds"].cpu(),
             hidden4=[h.float().cpu() for h in aux["hidden4"]])
    return cfg, sd, tk, m8, images, ids, d


def _deq(t, tag):
    return t[tag + ".q8"] * t[tag + ".s8"][:, None]


def test_fp8_weights_match_oracle_quantisation(f8):
    """the load-time weight quantisation (groma_amd/weights.py q8, torch ops on the device) against the oracle's (same formula
    on the host): identical scales; identical e4m3 codes except where w / s lands within fp32 round-off of a rounding tie"""
    cfg, sd, tk, m8, images, ids, d = f8
    p = "llm.model.layers.0."
    v = "perceiver.vis_encoder.encoder.layer.0."
    for (w8, s), w in ((m8.llm.w["layers"][0]["wqkv"], torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)),
                       (m8.vit.w["layers"][0]["w1"], sd[v + "mlp.fc1.weight"])):
        ref_q, ref_s = O.quant_weight_e4m3(w)
        got = w8.float().cpu()
        assert util.relerr(s, ref_s) < 1e-6
        diff = (got != ref_q)
        frac = diff.float().mean().item()
        step = ((got - ref_q).abs() / ref_q.abs().clamp_min(2 ** -9))[diff].max().item() if diff.any() else 0.0
        print(f"[fp8 weights] {tuple(w.shape)}: codes differing from the oracle's {frac:.2e} of elements, largest relative step {step:.3f}")
        assert frac < 1e-3 and step <= 0.34   # neighbours on the e4m3 grid only (one step is <= 1/3 of the smaller magnitude incl. subnormals)
        assert util.relerr(got * s.cpu()[:, None], ref_q * ref_s[:, None]) < 2e-3
    assert m8.bridge["w0"].dtype == torch.bfloat16  # stays bf16
    # round 4: lm_head (+) extra_lm_head rows and the region encoder's 3x3 convs (fuse round >= 1, per-ROI conv)
    h8, hs = m8.llm.w["head8"]
    V0 = cfg.llm_cfg.vocab_size
    ref_q, ref_s = O.quant_weight_e4m3(torch.cat([sd["llm.lm_head.weight"], sd["extra_lm_head.weight"]], 0))
    V = ref_q.shape[0]
    assert util.relerr(hs[:V], ref_s) < 1e-6 and (h8.float().cpu()[:V] != ref_q).float().mean() < 1e-3
    assert (h8.float()[V:] == 0).all()   # padding rows of the head
    m = "region_encoder.mlvl_fuse."
    rw = m8.region.w
    assert rw["fp8"] and "w" in rw["fuse"][0] and "w8" in rw["fuse"][1] and "pconv_w8" in rw
    s_in = O.conv_act_scale(sd[m + "fuse_convs.0.gn.weight"], sd[m + "fuse_convs.0.gn.bias"])
    assert abs(rw["fuse"][1]["q_inv"] * s_in - 1.0) < 1e-6
    w = sd[m + "fuse_convs.1.conv.weight"]
    D = w.shape[0]
    sw = w.flatten(1).abs().amax(dim=1).clamp_min(1e-20) / 448.0
    wq = (w / sw[:, None, None, None]).to(O.E4M3).float()
    got = rw["fuse"][1]["w8"].float().cpu().view(D, 3, 3, D).permute(0, 3, 1, 2)   # the device's K order is (ky, kx, c)
    assert (got != wq).float().mean() < 1e-3 and util.relerr(rw["fuse"][1]["ws8"], sw * s_in) < 1e-6


def test_fp8_every_kernel_teacher_forced_at_width(f8):
    cfg, sd, tk, m8, images, ids, d = f8
    t = d["trace"]
    r, rel = O._r, util.relerr
    rows = []

    def chk(name, dev, ref, tol):
        e = rel(dev, ref)
        rows.append((name, e, tol))
        print(f"[fp8 kernel] {name:74s} rel-L2 {e:.2e}  (tol {tol:.0e})")

    def quant(name, tag, x):      # a quantiser: de-quantised rows and scales against the oracle's on the same input
        q, s = O.quant_rows_e4m3(x.reshape(-1, x.shape[-1]))
        chk(name, _deq(t, tag), q * s, TOL_QUANT)
        assert rel(t[tag + ".s8"], s[:, 0]) < 1e-6, name

    def gemm8(tag, w, b=None):    # the e4m3 GEMM on the DEVICE's quantised operand (nothing re-quantised on the host)
        wq, sw = O.quant_weight_e4m3(w)
        y = F.linear(t[tag + ".q8"], wq) * sw * t[tag + ".s8"][:, None]
        return y if b is None else y + b

    with torch.no_grad(), O.rounding("e4m3"):
        # ---------------- ViT layer 0: M = 1025, D = 1024
        p = "perceiver.vis_encoder.encoder.layer.0."
        a = p + "attention.attention."
        h = t["vit0.h_in"]
        quant("ViT LayerNorm -> e4m3 rows (norm_fp8)", "vit0.ln1", O._ln(h, sd, p + "norm1", 1e-6))
        wqkv = torch.cat([sd[a + n + ".weight"] for n in ("query", "key", "value")], 0)
        bqkv = torch.cat([sd[a + n + ".bias"] for n in ("query", "key", "value")], 0)
        chk("ViT QKV e4m3 GEMM 1025x3072x1024 + bias -> bf16", t["vit0.qkv"], r(gemm8("vit0.ln1", wqkv, bqkv)), TOL_BF16)
        quant("attention context bf16 -> e4m3 rows (quant_rows_fp8)", "vit0.ctx", t["vit0.ctx"])
        mid = h + sd[p + "layer_scale1.lambda1"] * gemm8("vit0.ctx", sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        chk("ViT o-proj e4m3 GEMM 1025x1024x1024 + bias + LayerScale + residual (f32)", t["vit0.mid"], mid, TOL_F32OUT)
        quant("ViT LayerNorm 2 -> e4m3 rows", "vit0.ln2", O._ln(t["vit0.mid"], sd, p + "norm2", 1e-6))
        chk("ViT fc1 e4m3 GEMM 1025x4096x1024 + bias + GELU -> bf16", t["vit0.fc1"],
            r(F.gelu(gemm8("vit0.ln2", sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))), TOL_BF16)
        quant("GELU output bf16 -> e4m3 rows", "vit0.fc1", t["vit0.fc1"])
        out = t["vit0.mid"] + sd[p + "layer_scale2.lambda1"] * gemm8("vit0.fc1", sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        chk("ViT fc2 e4m3 GEMM 1025x1024x4096 + bias + LayerScale + residual (f32)", t["vit0.out"], out, TOL_F32OUT)
        # ---------------- LLaMA layer 0: L = 582
        p = "llm.model.layers.0."
        h = t["llm0.h_in"]

        def rms(x, w):
            return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
        quant("RMSNorm -> e4m3 rows (norm_fp8)", "llm0.n1", rms(h, sd[p + "input_layernorm.weight"]))
        wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        chk("LLaMA QKV e4m3 GEMM 582x12288x4096 -> bf16", t["llm0.qkv"], r(gemm8("llm0.n1", wqkv)), TOL_BF16)
        quant("attention context bf16 -> e4m3 rows", "llm0.ctx", t["llm0.ctx"])
        chk("o-proj e4m3 GEMM 582x4096x4096 + residual (f32, in place)", t["llm0.h_attn"],
            h + gemm8("llm0.ctx", sd[p + "self_attn.o_proj.weight"]), TOL_F32OUT)
        quant("RMSNorm 2 -> e4m3 rows", "llm0.n2", rms(t["llm0.h_attn"], sd[p + "post_attention_layernorm.weight"]))
        act = F.silu(gemm8("llm0.n2", sd[p + "mlp.gate_proj.weight"])) * gemm8("llm0.n2", sd[p + "mlp.up_proj.weight"])
        chk("gate/up e4m3 GEMM 582x22016x4096 + SwiGLU epilogue -> bf16", t["llm0.act"], r(act), TOL_BF16)
        quant("SwiGLU output bf16 -> e4m3 rows", "llm0.act", t["llm0.act"])
        chk("down e4m3 GEMM 582x4096x11008 + residual (f32)", t["llm0.h_out"],
            t["llm0.h_attn"] + gemm8("llm0.act", sd[p + "mlp.down_proj.weight"]), TOL_F32OUT)
        quant("final RMSNorm -> e4m3 rows (the head's operand)", "llm.head_in", rms(t["llm0.h_out"], sd["llm.model.norm.weight"]))
        chk("lm_head (+) extra_lm_head e4m3 GEMM 582x32128x4096 (f32 logits)", d["logits"].view(582, -1),
            gemm8("llm.head_in", torch.cat([sd["llm.lm_head.weight"], sd["extra_lm_head.weight"]], 0)), TOL_F32OUT)
    bad = [(n, e, tol) for n, e, tol in rows if not e < tol]
    assert not bad, bad
    assert len(rows) >= 17


def test_fp8_region_convs_teacher_forced_at_width(f8):
    """round 4: the region encoder's e4m3 3x3 convs at the benchmark's channel width (1024 -> 1024, 9216- / 27 648-deep), each
    kernel against the oracle's arithmetic on the device's own inputs: the e4m3 map writer (fuse_shuffle), the implicit-GEMM conv on
    that map, the e4m3 RoIAlign tiles and the per-ROI conv on them"""
    cfg, sd, tk, m8, images, ids, d = f8
    t = d["trace"]
    from oracle import cref
    r, rel = O._r, util.relerr
    rows = []

    def chk(name, dev, ref, tol):
        e = rel(dev, ref)
        rows.append((name, e, tol))
        print(f"[fp8 kernel] {name:74s} rel-L2 {e:.2e}  (tol {tol:.0e})")

    def q8(x, s):
        return (x * (1.0 / s)).clamp(-448.0, 448.0).to(O.E4M3).float()

    m, ra = "region_encoder.mlvl_fuse.", "region_encoder.roi_align."
    C = 1024
    G = cfg.image_size // cfg.perceiver_cfg.vis_encoder_cfg.patch_size
    S = [4 * G, 2 * G, G]
    with torch.no_grad(), O.rounding("e4m3"):   # (r() rounds to bf16 only inside a rounding mode)
        s_in = O.conv_act_scale(sd[m + "fuse_convs.0.gn.weight"], sd[m + "fuse_convs.0.gn.bias"])
        w = sd[m + "fuse_convs.1.conv.weight"]
        sw = w.flatten(1).abs().amax(dim=1).clamp_min(1e-20) / 448.0
        wq = (w / sw[:, None, None, None]).to(O.E4M3).float()

        def act(l):   # relu(GroupNorm(conv_0)) of level l from the traced conv output and the traced GN coefficients, NCHW
            x = t[f"reg8.map{l}"].view(S[l], S[l], C)
            a, b = t[f"reg8.coef{l}"][0, 0], t[f"reg8.coef{l}"][0, 1]
            return F.relu(x * a + b).permute(2, 0, 1)[None]
        for l in (1, 2):   # (level 0 is the same code at 128 x 128: 155 GFLOP on the host, left out)
            top, dow = min(l + 1, 2), max(l - 1, 0)
            size = (S[l], S[l])
            fused = torch.cat([act(l)[:, : C // 2],
                               F.interpolate(act(top)[:, 3 * C // 4:], size=size, mode="bilinear", align_corners=True),
                               F.interpolate(act(dow)[:, C // 2: 3 * C // 4], size=size, mode="bilinear", align_corners=True)], 1)
            pad8 = t[f"reg8.pad{l}"].permute(0, 3, 1, 2)   # [1, C, S+2, S+2] e4m3 values
            assert (pad8[:, :, 0] == 0).all() and (pad8[:, :, -1] == 0).all() and (pad8[..., 0] == 0).all() and (pad8[..., -1] == 0).all()
            chk(f"fuse_shuffle -> e4m3 map, level {l} ({S[l]}x{S[l]}x1024, static scale)", pad8[:, :, 1:-1, 1:-1] * s_in,
                q8(fused, s_in) * s_in, TOL_QUANT)
            y = F.conv2d(pad8, wq) * (sw * s_in)[None, :, None, None]
            chk(f"e4m3 implicit-GEMM 3x3 conv, level {l}: {S[l] * S[l]}x1024x9216 -> bf16",
                t[f"reg8.conv{l}"].view(1, S[l], S[l], C).permute(0, 3, 1, 2), r(y), TOL_BF16)
        # RoIAlign tiles and the per-ROI conv
        nf = cfg.region_cfg.num_fuse
        s_last = O.conv_act_scale(sd[f"{m}fuse_convs.{nf - 1}.gn.weight"], sd[f"{m}fuse_convs.{nf - 1}.gn.bias"])
        rois, tiles = t["reg.rois"], t["reg.tiles"]            # [R, 5], [3, R, 16, 16, C] e4m3 values
        strides = [14 / 8, 14 / 4, 14 / 2]
        for l in range(3):
            f = t[f"reg8.feat{l}"].view(1, S[l], S[l], C).permute(0, 3, 1, 2).contiguous()
            rf = torch.from_numpy(cref.roi_align_avg(f.numpy(), rois.numpy(), (14, 14), 1.0 / strides[l], 2, True))
            chk(f"roi_align_pack -> e4m3 tiles, level {l}", tiles[l][:, 1:-1, 1:-1].permute(0, 3, 1, 2) * s_last,
                q8(rf, s_last) * s_last, TOL_QUANT)
        n = 8
        x = torch.cat([tiles[l][:n].permute(0, 3, 1, 2) for l in range(3)], 1)       # [n, 3C, 16, 16], zero border included
        wp = torch.cat([sd[f"{ra}pconvs.{l}.weight"] for l in range(3)], 1)
        swp = wp.flatten(1).abs().amax(dim=1).clamp_min(1e-20) / 448.0
        wpq = (wp / swp[:, None, None, None]).to(O.E4M3).float()
        bias = sum(sd[f"{ra}pconvs.{l}.bias"] for l in range(3))
        y = F.relu(F.conv2d(x, wpq) * (swp * s_last)[None, :, None, None] + bias[None, :, None, None])
        chk("e4m3 per-ROI conv (3 levels, 27 648-deep) + bias + ReLU -> bf16, first 8 ROIs",
            t["reg.pc"].view(-1, 14, 14, C)[:n].permute(0, 3, 1, 2), r(y), TOL_BF16)
    bad = [(n_, e, tol) for n_, e, tol in rows if not e < tol]
    assert not bad, bad
    assert len(rows) == 8


def test_fp8_chained_logits_and_vit_states_at_width(f8):
    """the chained numbers: implementation error (vs the e4m3-rounded oracle) and the format's error (vs bf16 / fp32)"""
    cfg, sd, tk, m8, images, ids, d = f8
    from groma_amd import constants
    from groma_amd.groma import GromaModel
    cd = cfg.to_dict()
    rel = util.relerr
    m16 = GromaModel.from_state_dict(cfg, sd, device="cuda")
    m16.init_special_token_id(constants.SyntheticTokenizer())
    L = d["embeds"].shape[1]
    emb = d["embeds"].cuda().reshape(L, -1)
    with torch.no_grad():
        lg = {}
        for name, m in (("bf16", m16), ("fp8", m8)):
            cache = m.llm.new_cache(1, L, emb.device)
            lg[name] = m.llm.forward(emb.clone(), 1, L, cache)[0].float().cpu().reshape(L, -1)
        h16 = [h.float().cpu() for h in m16.vit.forward(images.cuda())]
        ref = {}
        for mode in (None, "bf16", "e4m3"):
            with O.rounding(mode):
                hid, _ = O.llama_forward(sd, cd, d["embeds"], torch.ones((1, L)))
                ref[mode] = (O.lm_logits(sd, hid).reshape(L, -1), O.vit_forward(sd, cd, images)[-4:])
    e_impl = rel(lg["fp8"], ref["e4m3"][0])
    e_b16d = rel(lg["fp8"], lg["bf16"])
    e_f32 = rel(lg["fp8"], ref[None][0])
    o_fmt = rel(ref["e4m3"][0], ref["bf16"][0])
    print(f"[fp8 chained] LLaMA layer + head logits (same embeddings): device e4m3 vs e4m3-rounded oracle {e_impl:.3e} | vs bf16 device "
          f"path {e_b16d:.3e} | vs fp32 oracle {e_f32:.3e} | oracle e4m3 vs oracle bf16 (the format) {o_fmt:.3e}")
    # stated tolerance of configs[4]: logits within 1e-1 rel-L2 of the bf16 path (measured 8.4e-2 with a bf16 head, 9.27e-2 since the
    # head reads e4m3 operands too -- round 4; the oracle's own e4m3-vs-bf16 distance is 9.27e-2 as well)
    assert e_b16d < 1e-1 and e_f32 < 1e-1
    # the implementation is closer to its own oracle than the format is to bf16 (measured 4.7e-2: two e4m3 evaluations whose fp32
    # sums differ in the last bits re-quantise a few values one e4m3 step apart, and every later stage amplifies that)
    # -- tests/test_oracle_rounding_modes.py::test_a_rounded_evaluation_is_sensitive_to_its_last_input_bits shows the same ratio between the
    # e4m3-rounded oracle and ITSELF with its input perturbed by 1e-6: 3.6-4.7e-2 of a 1.0e-1 format distance (CPU, no device involved)
    assert e_impl < 0.6 * e_b16d
    assert abs(e_b16d - o_fmt) < 0.5 * o_fmt       # and the device's format error is the oracle's format error
    # the stated tolerance relative to what is MEASURED -- the device may exceed the e4m3 FORMAT's own distance
    # from bf16 (the oracle's, 9.27e-2 on this stack) by at most 5 %; the flat 1e-1 above is 1.08x the measured value
    assert e_b16d <= 1.05 * o_fmt
    vs = [rel(a, b) for a, b in zip(d["hidden4"], ref["e4m3"][1])]
    vb = [rel(a, b) for a, b in zip(d["hidden4"], h16)]
    print(f"[fp8 chained] ViT states (0..3 layers deep): vs e4m3-rounded oracle {[f'{x:.2e}' for x in vs]} | vs bf16 device {[f'{x:.2e}' for x in vb]}")
    assert max(vb) < 8e-2 and vs[0] < 1e-5   # 3 layers of e4m3 operands vs the bf16 path: measured 3.9e-2 / 5.1e-2 / 5.7e-2
    assert all(a < b for a, b in zip(vs[1:], vb[1:]))
