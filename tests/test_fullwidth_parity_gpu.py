"""Stage-wise parity at Groma-7B WIDTH (reduced depth) on the MI355X against the CPU oracle -- fp32 and bf16-rounded.

Every GEMM / conv / attention shape the benchmark runs is exercised with the oracle beside it (config.groma_7b_width):
  ViT D=1024 x 16 heads x 3 layers @ 1025 tokens; bridge 4096x4096; input 1x1 convs K=1026 and one 3x3 fusion round at
  C=1024 on 128^2/64^2/32^2 (K=9216) + GroupNorm; RoIAlign x3 -> per-ROI conv K=27 648 -> flatten_linear split-K
  200 704 -> updims, N=100 regions; one LLaMA layer 4096 / 11 008 / 32 heads at L=582 + the 32 114-wide head;
  full-depth 6+6 DDETR on the device ViT states.
R: groma/model/groma.py:218-402, groma/model/roi_align.py:150-193,274-327, groma/model/ddetr_transformer.py:484-609.

Two references, as SURVEY 'Hard parts' and BASELINE.md 3 prescribe:
  * fp32 oracle: what the reference's CPU path computes.  Tolerance 1e-2 relative L2 (bf16 operands: measured 2-6e-3).
  * bf16-rounded oracle (oracle.rounding("bf16")): the same restatement with operands rounded to bf16 exactly where the
    device holds bf16 -- isolates the implementation (accumulation order, exp/erf, indexing) from the number format.
    north_star's 1e-3 relative L2 is asserted PER KERNEL with every kernel fed the device's own input
    (test_every_kernel_teacher_forced: measured <= 4e-4, fp32-output kernels <= 2e-6), and per chained stage with the
    depth-dependent bounds below.  Why a chained stage cannot hold 1e-3 against ANY other implementation: a relative
    discrepancy d << 2^-8 in front of a bf16 rounding becomes ~sqrt(d * 2.8e-3) behind it (a fraction ~d/ulp of the
    elements flips by one ulp), so fp32 round-off (1e-7) grows 1.7e-5 -> 2.2e-4 -> 7.8e-4 -> 1.5e-3 -> ... over successive
    roundings and saturates at the format's own ~3e-3 -- the distance the fp32 oracle already shows.  Measured on MI355X:
    patch embedding 2e-7, bridge 2.6e-5 (2 roundings), one ViT layer 8.8e-4 (6 roundings), region tokens 2.5e-3
    (9 roundings), LLaMA layer + head 3.1e-3; fp32-oracle distance of the same tensors 2.3e-3 ... 6.2e-3.
Index-valued results (top-300 ids, NMS keep ids, shuffle order, spliced ids): bit-exact.
"""
import json
import math
import os

import pytest
import torch

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL_FP32, TOL_BF16 = 1e-2, 1e-3
TOL_KERNEL_F32OUT = 1e-5   # a kernel with fp32 output, identical inputs: accumulation-order noise only


@pytest.fixture(scope="module")
def fw(dev):
    from groma_amd import config as gconfig, synth
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 8)))
    cfg = gconfig.groma_7b_width(box_score_thres=0.0)
    sd = synth.make_state_dict(cfg, 0)
    tk = util.TokenIds()
    from groma_amd import engine
    model = util.device_model(cfg, sd)
    model.capture_embeds = True
    images, ids = synth.make_inputs(cfg, tk, bs=1, seed=1234)
    torch.manual_seed(77)
    engine.TRACE = {}
    try:
        with torch.no_grad():
            out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True, use_cache=True)
        torch.cuda.synchronize()
        trace = {k: v.float().cpu() for k, v in engine.TRACE.items()}
    finally:
        engine.TRACE = None
    aux = model._last_aux
    dev_out = dict(
        hidden4=[h.float().cpu() for h in aux["hidden4"]], logits=out.logits.float().cpu(),
        image_features=out.hidden_states[1]["image_features"].float().cpu(),
        region_features=out.hidden_states[1]["region_features"].float().cpu(),
        pred_boxes=[b.cpu() for b in out.hidden_states[1]["pred_boxes"]], embeds=aux["inputs_embeds"].cpu(),
        nms_keep=aux["nms_keep"], sel_idx=aux["sel_idx"], topk_idx=aux["topk_idx"].cpu().long(),
        all_boxes=aux["pred_boxes"].cpu(), scores=aux["scores"].cpu(), new_ids=aux["input_ids"],
        k0=out.past_key_values[0][0].float().cpu(), v0=out.past_key_values[0][1].float().cpu(), trace=trace)
    cd, tok = cfg.to_dict(), util.tok_dict(tk)
    refs = {}
    with torch.no_grad():
        for mode in (None, "bf16"):
            with O.rounding(mode):
                r = dict(vit=O.vit_forward(sd, cd, images)[-4:])
                torch.manual_seed(77)
                r["chain"] = O.groma_forward(sd, cd, tok, ids.clone(), images, hidden_states=tuple(dev_out["hidden4"]))
                hid, past = O.llama_forward(sd, cd, dev_out["embeds"], r["chain"]["attention_mask"])
                r["llm_logits"], r["past"] = O.lm_logits(sd, hid), past
            refs[mode] = r
    return cfg, sd, model, dev_out, refs


def _report(name, dev, refs, key):
    e32, e16 = util.relerr(dev, key(refs[None])), util.relerr(dev, key(refs["bf16"]))
    print(f"[fullwidth] {name}: rel-L2 vs fp32 oracle {e32:.3e}, vs bf16-rounded oracle {e16:.3e}")
    return e32, e16


def test_vit_states_fullwidth(fw):
    cfg, sd, model, d, refs = fw
    for i in range(4):  # hidden[-4] = embeddings (0 layers) ... hidden[-1] = after 3 layers
        e32, e16 = _report(f"vit hidden[-{4 - i}] ({i} layers deep)", d["hidden4"][i], refs, lambda r: r["vit"][i])
        assert e32 < TOL_FP32 and e16 < (1e-5 if i == 0 else i * TOL_BF16) and e16 < e32


def test_proposer_fullwidth_exact_indices(fw):
    """full-depth DDETR fed the device ViT states.  Ranking check that never skips: every device top-k slot must hold an
    element whose ORACLE logit equals the oracle's value at that rank within 2x the measured fp32 evaluation error --
    which is plain torch.equal wherever the oracle's neighbours are further apart than that (asserted for most slots)."""
    cfg, sd, model, d, refs = fw
    det = refs[None]["chain"]["det"]
    dbg = {}
    with torch.no_grad():
        pred, scores, idx = model.proposer.forward([h.cuda() for h in d["hidden4"]], debug=dbg)
    enc_d, enc_r = dbg["enc_class"].cpu(), det["enc_class"]
    err = (enc_d - enc_r).abs().max().item()
    assert util.relerr(dbg["memory"], det["memory"]) < 2e-4 and err < 1e-4
    srt = torch.sort(enc_r, dim=1, descending=True, stable=True)
    Q = idx.shape[1]
    got = enc_r.gather(1, idx.cpu().long())
    assert (got - srt[0][:, :Q]).abs().max().item() <= 2 * err + 1e-7, "device top-k is not a valid ranking of the oracle logits"
    gaps = (srt[0][:, :Q] - srt[0][:, 1:Q + 1])
    clear = (gaps > 4 * err) & (torch.cat([gaps[:, :1] * 0 + 1, gaps[:, :-1]], 1) > 4 * err)
    print(f"[fullwidth] proposer: max abs logit err {err:.2e}, min gap {gaps.min().item():.2e}, clear slots {int(clear.sum())}/{Q}")
    assert clear.float().mean() > 0.9
    assert torch.equal(idx.cpu().long()[clear], det["topk_idx"][clear])
    if bool(clear.all()):
        assert util.relerr(pred, det["pred_boxes"]) < 2e-4
        assert util.relerr(scores, O.fuse_scores(det["logits_coco"], det["logits_sa1b"])) < 2e-4


def test_bridge_and_region_tokens_fullwidth(fw):
    cfg, sd, model, d, refs = fw
    ch = refs[None]["chain"]
    # index-valued stages first: the region tokens are only comparable if the same boxes were selected in the same order
    assert torch.equal(d["nms_keep"][0], ch["nms_inds"][0]), "NMS keep ids differ"
    assert torch.equal(d["sel_idx"][0], ch["nms_inds"][0][ch["perms"][0]]), "shuffled order differs"
    assert torch.allclose(d["pred_boxes"][0], ch["pred_boxes"][0], atol=1e-5)
    assert torch.equal(d["new_ids"], ch["input_ids"]), "spliced token ids differ"
    assert d["region_features"].shape == (100, 4096) and d["new_ids"].shape[1] == 582
    e32, e16 = _report("image_features (s2d + bridge)", d["image_features"], refs, lambda r: r["chain"]["image_features"])
    assert e32 < TOL_FP32 and e16 < 1e-4
    e32, e16 = _report("region_features (fuse + RoIAlign + pconv + flatten + updims)", d["region_features"], refs,
                       lambda r: r["chain"]["region_features"])
    assert e32 < TOL_FP32 and e16 < 4 * TOL_BF16 and e16 < e32  # 9 chained bf16 roundings (see the module docstring)


def test_llama_layer_and_head_fullwidth(fw):
    cfg, sd, model, d, refs = fw
    # stage-isolated: the oracle consumes the device's own inputs_embeds
    e32, e16 = _report("LLaMA layer + 32114-wide head (device embeds)", d["logits"], refs, lambda r: r["llm_logits"])
    assert e32 < TOL_FP32 and e16 < 4.5 * TOL_BF16 and e16 < e32  # 8 chained roundings incl. a peaked soft-max
    # chained from the device ViT states through the oracle's own region encoder / bridge / embedding
    c32, c16 = _report("logits chained from the ViT states", d["logits"], refs, lambda r: r["chain"]["logits"])
    assert c32 < TOL_FP32 and c16 < 6 * TOL_BF16
    for name, dv, k in (("K", d["k0"], 0), ("V", d["v0"], 1)):
        e32, e16 = _report(f"layer-0 {name} cache", dv, refs, lambda r: r["past"][0][k])
        assert e32 < TOL_FP32 and e16 < TOL_BF16  # 2-3 roundings deep
    # arg-max: identical wherever the bf16-rounded oracle's top-2 margin exceeds 4x the abs error against it
    lr = refs["bf16"]["llm_logits"]
    abs_err = (d["logits"] - lr).abs().max().item()
    top2 = lr.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * abs_err
    print(f"[fullwidth] logits max abs err vs bf16-rounded oracle {abs_err:.3e} (max |logit| {lr.abs().max().item():.3f}); "
          f"clear-margin positions {clear.float().mean().item():.3f}")
    assert clear.float().mean().item() > 0.5  # random-init logits: many top-2 margins are inside the error band
    assert torch.equal(d["logits"].argmax(-1)[clear], lr.argmax(-1)[clear])
    r0 = model.box_idx_token_ids[0]
    assert util.relerr(d["logits"][:, -1, r0:r0 + 100], lr[:, -1, r0:r0 + 100]) < 6 * TOL_BF16  # the "region logits"


def test_every_kernel_teacher_forced(fw):
    """north_star's 1e-3, kernel by kernel at the benchmark's shapes: the oracle's version of ONE operation is applied to
    the tensor the device kernel actually consumed (engine.TRACE), so nothing is chained.  bf16-output kernels <= 1e-3
    (one rounding behind fp32-round-off-level differences: expected ~2e-5; attention, which rounds P and then the context:
    ~4e-4); fp32-output kernels <= 1e-5."""
    import torch.nn.functional as F
    cfg, sd, model, d, refs = fw
    t = d["trace"]
    r, rel = O._r, util.relerr
    rows = []

    def chk(name, dev, ref, tol):
        e = rel(dev, ref)
        rows.append((name, e, tol))
        print(f"[kernel] {name:58s} rel-L2 {e:.2e}  (tol {tol:.0e})")

    with torch.no_grad(), O.rounding("bf16"):
        # ---------------- ViT layer 0: M = 1025, D = 1024, 16 heads x 64
        p = "perceiver.vis_encoder.encoder.layer.0."
        h = t["vit0.h_in"]
        chk("ViT LayerNorm -> bf16", t["vit0.ln1"], r(O._ln(h, sd, p + "norm1", 1e-6)), TOL_BF16)
        x = t["vit0.ln1"]
        qkv = torch.cat([O._lin16(x, sd, p + "attention.attention." + n) for n in ("query", "key", "value")], -1)
        chk("ViT QKV GEMM 1025x3072x1024 + bias", t["vit0.qkv"], r(qkv), TOL_BF16)
        q, k, v = (u.view(1, 1025, 16, 64).transpose(1, 2) for u in t["vit0.qkv"].view(1, 1025, 3072).split(1024, -1))
        ctx = O._softmax_pv(q @ k.transpose(-1, -2) / 8.0, v).transpose(1, 2).reshape(1025, 1024)
        chk("ViT attention hd 64, 1025 keys, non-causal", t["vit0.ctx"], ctx, TOL_BF16)
        mid = h + sd[p + "layer_scale1.lambda1"] * O._lin16(t["vit0.ctx"].view(1, 1025, 1024), sd, p + "attention.output.dense")
        chk("ViT o-proj GEMM + bias + LayerScale + residual (f32 out)", t["vit0.mid"], mid, TOL_KERNEL_F32OUT)
        chk("ViT LayerNorm 2 -> bf16", t["vit0.ln2"], r(O._ln(t["vit0.mid"], sd, p + "norm2", 1e-6)), TOL_BF16)
        chk("ViT fc1 GEMM 1025x4096x1024 + bias + GELU(erf)", t["vit0.fc1"], r(F.gelu(O._lin16(t["vit0.ln2"], sd, p + "mlp.fc1"))), TOL_BF16)
        out = t["vit0.mid"] + sd[p + "layer_scale2.lambda1"] * O._lin16(t["vit0.fc1"], sd, p + "mlp.fc2").view(1, 1025, 1024)
        chk("ViT fc2 GEMM 1025x1024x4096 + LayerScale + residual (f32)", t["vit0.out"], out, TOL_KERNEL_F32OUT)
        # ---------------- bridge: M = 256, 4096 -> 4096 -> 4096
        chk("s2d pack (2x2 space-to-depth -> bf16)", t["bridge.s2d"].view(1, 256, 4096), r(O.s2d_image_features(d["hidden4"][-1])), TOL_BF16)
        chk("bridge.0 GEMM 256x4096x4096 + bias + GELU", t["bridge.mid"], r(F.gelu(O._lin16(t["bridge.s2d"], sd, "img_txt_bridge.0"))), TOL_BF16)
        chk("bridge.2 GEMM (f32 out)", d["image_features"].view(256, 4096), O._lin16(t["bridge.mid"], sd, "img_txt_bridge.2"), TOL_KERNEL_F32OUT)
        # ---------------- region pyramid, C = 1024, levels 128^2 / 64^2 / 32^2
        m = "region_encoder.mlvl_fuse."
        S = [128, 64, 32]
        maps = []
        for l in range(3):
            hl = d["hidden4"][1 + l][:, 1:].reshape(1, 32, 32, 1024).permute(0, 3, 1, 2)
            up = F.interpolate(hl, size=(S[l], S[l]), mode="bilinear", align_corners=True)
            yy, xx = torch.meshgrid(torch.linspace(-1, 1, S[l]), torch.linspace(-1, 1, S[l]), indexing="ij")
            up = torch.cat([up, xx.expand(1, 1, -1, -1), yy.expand(1, 1, -1, -1)], 1)
            a = t[f"reg.up{l}"].view(1, S[l], S[l], -1)[..., :1026].permute(0, 3, 1, 2)
            chk(f"level {l}: bilinear upsample to {S[l]}^2 + coord -> bf16", a, r(up), TOL_BF16)
            ref = F.conv2d(a, r(sd[f"{m}input_conv.{l}.weight"]), sd[f"{m}input_conv.{l}.bias"])
            got = t[f"reg.in{l}"].view(1, S[l], S[l], 1024).permute(0, 3, 1, 2)
            chk(f"level {l}: input 1x1 conv GEMM {S[l] * S[l]}x1024x1088", got, r(ref), TOL_BF16)
            maps.append(got)
        remain, shuffle = 512, 256
        for l in range(3):
            top, dow = min(l + 1, 2), max(l - 1, 0)
            ft = F.interpolate(maps[top][:, remain:][:, shuffle:], size=(S[l], S[l]), mode="bilinear", align_corners=True)
            fd = F.interpolate(maps[dow][:, remain:][:, :shuffle], size=(S[l], S[l]), mode="bilinear", align_corners=True)
            fused = torch.cat([maps[l][:, :remain], ft, fd], 1)
            pad = t[f"reg.pad{l}"].view(1, S[l] + 2, S[l] + 2, 1024)
            assert float(pad[:, 0].abs().max()) == 0 and float(pad[:, :, -1].abs().max()) == 0  # the zero border
            inner = pad[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
            chk(f"level {l}: channel shuffle + cross-level resize -> conv input", inner, r(fused), TOL_BF16)
            conv = F.conv2d(inner, r(sd[f"{m}fuse_convs.0.conv.weight"]), None, padding=1)
            got = t[f"reg.conv{l}"].view(1, S[l], S[l], 1024).permute(0, 3, 1, 2)
            chk(f"level {l}: 3x3 conv implicit GEMM {S[l] * S[l]}x1024x9216", got, r(conv), TOL_BF16)
            gn = F.relu(F.group_norm(got, cfg.region_cfg.gn_groups, sd[f"{m}fuse_convs.0.gn.weight"], sd[f"{m}fuse_convs.0.gn.bias"], 1e-5))
            chk(f"level {l}: GroupNorm(64) + ReLU -> feature map", t[f"reg.feat{l}"].permute(0, 3, 1, 2), r(gn), TOL_BF16)
        # ---------------- RoI extraction, N = 100
        from oracle import cref
        ra = "region_encoder.roi_align."
        rois, tiles = t["reg.rois"], t["reg.tiles"]
        acc = None
        for l in range(3):
            feat = t[f"reg.feat{l}"].permute(0, 3, 1, 2).contiguous()
            want = torch.from_numpy(cref.roi_align_avg(feat.numpy(), rois.numpy(), (14, 14), 1.0 / [14 / 8, 14 / 4, 14 / 2][l], 2, True))
            got = tiles[l][:, 1:-1, 1:-1].permute(0, 3, 1, 2)
            assert torch.equal(got, r(want)), f"RoIAlign level {l} is not bit-exact"
            y = F.conv2d(got, r(sd[f"{ra}pconvs.{l}.weight"]), sd[f"{ra}pconvs.{l}.bias"], padding=1)
            acc = y if acc is None else acc + y
        print("[kernel] RoIAlign + pack x3 levels (100 ROIs, negative widths)             bit-exact")
        pc = t["reg.pc"].view(100, 14, 14, 1024).permute(0, 3, 1, 2)
        chk("per-ROI 3x3 conv, 3 levels summed, K = 27648, + bias + ReLU", pc, r(F.relu(acc)), TOL_BF16)
        boxes = rois[:, 1:] / 448.0
        pe = F.layer_norm(F.relu(O._lin(boxes, sd, ra + "pos_embedd.0")), (cfg.region_cfg.pos_hidden,), sd[ra + "pos_embedd.2.weight"], sd[ra + "pos_embedd.2.bias"])
        pe = F.layer_norm(F.relu(O._lin(pe, sd, ra + "pos_embedd.3")), (cfg.region_cfg.mid_dim,), sd[ra + "pos_embedd.5.weight"], sd[ra + "pos_embedd.5.bias"])
        chk("box position MLP (fp32 GEMMs + LayerNorms)", t["reg.pe"], pe, 1e-4)
        fl = O._lin16(pc.flatten(1, -1), sd, ra + "flatten_linear") + t["reg.pe"]
        chk("flatten_linear split-K GEMM 100x1024x200704 + bias + pos", t["reg.fl"], r(fl), TOL_BF16)
        chk("updims GEMM 100x4096x1024 (f32 out)", t["reg.out"], O._lin16(t["reg.fl"], sd, ra + "updims"), TOL_KERNEL_F32OUT)
        # ---------------- LLaMA layer 0 + head: L = 582, 4096 / 11008 / 32 heads x 128, V = 32114
        p = "llm.model.layers.0."
        L = 582
        h = t["llm0.h_in"]
        assert torch.equal(h.view(1, L, 4096), d["embeds"])

        def rms(x, w):
            return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
        chk("RMSNorm -> bf16", t["llm0.n1"], r(rms(h, sd[p + "input_layernorm.weight"])), TOL_BF16)
        x = t["llm0.n1"]
        qkv = torch.cat([O._lin16(x, sd, p + f"self_attn.{n}_proj", False) for n in "qkv"], -1)
        chk("LLaMA QKV GEMM 582x12288x4096", t["llm0.qkv"], r(qkv), TOL_BF16)
        q, k, v = (u.view(1, L, 32, 128).transpose(1, 2) for u in t["llm0.qkv"].view(1, L, 12288).split(4096, -1))
        cos, sin = O.rope_tables(128, L)
        q, k = r(q * cos + O._rot_half(q) * sin), r(k * cos + O._rot_half(k) * sin)
        chk("RoPE + K cache write", d["k0"], k, TOL_BF16)
        chk("V^T cache write", d["v0"], v, 1e-6)
        fmin = torch.finfo(torch.float32).min
        att = torch.max(q @ k.transpose(2, 3) / math.sqrt(128) + torch.full((L, L), fmin).triu(1), torch.tensor(fmin))
        ctx = O._softmax_pv(att, v).transpose(1, 2).reshape(L, 4096)
        chk("LLaMA attention hd 128, causal, RoPE on q in registers", t["llm0.ctx"], ctx, TOL_BF16)
        chk("o-proj GEMM 582x4096x4096 + residual (f32, in place)", t["llm0.h_attn"], h + O._lin16(t["llm0.ctx"], sd, p + "self_attn.o_proj", False), TOL_KERNEL_F32OUT)
        chk("RMSNorm 2 -> bf16", t["llm0.n2"], r(rms(t["llm0.h_attn"], sd[p + "post_attention_layernorm.weight"])), TOL_BF16)
        x = t["llm0.n2"]
        act = F.silu(O._lin16(x, sd, p + "mlp.gate_proj", False)) * O._lin16(x, sd, p + "mlp.up_proj", False)
        chk("gate/up GEMM 582x22016x4096 + SwiGLU epilogue", t["llm0.act"], r(act), TOL_BF16)
        chk("down GEMM 582x4096x11008 + residual (f32)", t["llm0.h_out"], t["llm0.h_attn"] + O._lin16(t["llm0.act"], sd, p + "mlp.down_proj", False), TOL_KERNEL_F32OUT)
        chk("final RMSNorm -> bf16", t["llm.final_norm"], r(rms(t["llm0.h_out"], sd["llm.model.norm.weight"])), TOL_BF16)
        chk("lm_head (+) extra_lm_head GEMM 582x32128x4096 (f32 logits)", d["logits"].view(L, -1), O.lm_logits(sd, t["llm.final_norm"]), TOL_KERNEL_F32OUT)
    bad = [(n, e, tol) for n, e, tol in rows if not e < tol]
    assert not bad, bad
    assert len(rows) >= 40


# ---- a7: bit-exact top-300 proposal ids on committed seeds (selection rule: tests/golden/select_proposer_seeds.py) ----
def _seed_rows(width):
    with open(os.path.join(HERE, "golden", "proposer_seeds.json")) as f:
        return json.load(f)[str(width)]


def _exact_index_case(model, cfg, sd, seed, min_gap_committed):
    from tests.golden.select_proposer_seeds import hidden_states, min_gap
    hs = hidden_states(cfg, seed)
    with torch.no_grad():
        det = O.ddetr_forward(sd, cfg.to_dict(), O.ddetr_inputs_from_hidden(hs))
        dbg = {}
        pred, scores, idx = model.proposer.forward([h.cuda() for h in hs], debug=dbg)
    Q = idx.shape[1]
    gap = min_gap(det["enc_class"], Q)
    err = (dbg["enc_class"].cpu() - det["enc_class"]).abs().max().item()
    print(f"[a7] seed {seed}: oracle min adjacent gap {gap:.3e} (committed {min_gap_committed:.3e}), device max abs err {err:.3e}")
    assert abs(gap - min_gap_committed) <= 0.25 * min_gap_committed, "fixture drifted: re-run select_proposer_seeds.py"
    assert gap > 4 * err, "fixture no longer resolves the ranking: pick seeds with larger gaps"   # guard is asserted, never skipped
    assert torch.equal(idx.cpu().long(), det["topk_idx"])
    assert util.relerr(dbg["memory"], det["memory"]) < 2e-4
    assert util.relerr(pred, det["pred_boxes"]) < 2e-4
    assert util.relerr(scores, O.fuse_scores(det["logits_coco"], det["logits_sa1b"])) < 2e-4
    assert torch.allclose(dbg["ref0"].cpu(), det["init_reference"], atol=1e-5)


@pytest.mark.parametrize("row", _seed_rows(1024), ids=lambda r: f"seed{r['seed']}")
def test_top300_indices_bit_exact_fullwidth(fw, row):
    cfg, sd, model, d, refs = fw
    _exact_index_case(model, cfg, sd, row["seed"], row["min_gap"])


@pytest.fixture(scope="module")
def tiny66(dev):
    from tests.golden.select_proposer_seeds import proposer_cfg
    from groma_amd import synth
    cfg = proposer_cfg(256)
    sd = synth.make_state_dict(cfg, 0)
    return cfg, sd, util.device_model(cfg, sd)


@pytest.mark.parametrize("row", _seed_rows(256), ids=lambda r: f"seed{r['seed']}")
def test_top300_indices_bit_exact_tiny_6plus6(tiny66, row):
    cfg, sd, model = tiny66
    _exact_index_case(model, cfg, sd, row["seed"], row["min_gap"])
