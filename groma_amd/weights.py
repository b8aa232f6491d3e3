"""Reference-named parameters -> MI355X device layouts.

The on-disk / state-dict contract is the reference's (HF 4.32 parameter names); everything the
kernels want (bf16 [N,K] GEMM weights, fused QKV / gate-up rows, NHWC conv taps, padded K) is derived here once,
at load time.  torch ops in this file are load-time plumbing, never on the forward path.
"""
import math

import torch
import torch.nn.functional as F

from . import ops
from .ops import H16  # dtype of the active 16-bit operand type (the model being built sets ops.precision around packing)

F32 = torch.float32


def _ru(x, m):
    return (x + m - 1) // m * m


class Source:
    """get(name) -> fp32 tensor on `device`.  Backed by a CPU state dict or by the synthetic generator."""

    def __init__(self, get, device):
        self._get, self.device = get, device

    def __call__(self, name):
        return self._get(name).to(device=self.device, dtype=F32)

    @classmethod
    def from_state_dict(cls, sd, device):
        def get(name):
            if name in sd:
                return sd[name]
            # the reference registers bbox_embed / class_embed_* twice (decoder.* aliases): accept either
            alt = name.replace("ddetr_transformer.", "ddetr_transformer.decoder.")
            if alt in sd:
                return sd[alt]
            raise KeyError(f"missing parameter {name}")
        return cls(get, device)

    @classmethod
    def synthetic(cls, cfg, seed, device):
        from . import synth
        spec = {n: (s, k) for n, s, k in synth.param_spec(cfg)}
        names = {n: i for i, n in enumerate(spec)}

        def get(name):
            shape, kind = spec[name]
            g = torch.Generator(device=device).manual_seed(seed * 1000003 + names[name])
            return synth.materialize(shape, kind, g, device=device)
        return cls(get, device)


def bf(t):
    """f32 [.., K] -> the active operand storage: bf16 / fp16, or the (hi, lo) pair layout of precision "ref" (twice as wide;
    ops.split_pack).  A WEIGHT beyond the range of a half-based storage (|w| > 65504 in the fp16 build, > 131008 as a pair) raises
    ops.OperandOverflow at load time instead of being saturated silently (tests/test_operand_range.py)"""
    return ops.to_h16(t, on_overflow="raise", what="weight")


FP8 = torch.float8_e4m3fn


def q8(t):
    """per-output-channel OCP e4m3 quantisation of a [N,K] weight: (w8, scale[N]) with w ~= w8 * scale[:, None]"""
    t = t.float()
    s = (t.abs().amax(dim=1).clamp_min(1e-20) / 448.0).contiguous()
    return (t / s[:, None]).to(FP8).contiguous(), s


def wq(t, fp8):
    """GEMM weight in the compute format of the engine: bf16 / fp16 / operand pairs, or (e4m3, per-row scale)"""
    if fp8 and ops.SP() == 2:
        raise NotImplementedError("e4m3 operands and the reference-precision build are exclusive")
    return q8(t) if fp8 else (bf(t), None)


def pad_k(w, K):
    out = torch.zeros((w.shape[0], K), dtype=w.dtype, device=w.device)
    out[:, : w.shape[1]] = w
    return out


def vit_pos_embed_4_32(pos, grid):
    """HF 4.32 Dinov2Embeddings.interpolate_pos_encoding (bicubic, scale_factor (grid+0.1)/sqrt(N)).
    Input-independent, so it is evaluated once at load (fp32, CPU)."""
    pos = pos.detach().float().cpu()
    n_pos = pos.shape[1] - 1
    if n_pos == grid * grid:
        return pos[0]
    dim = pos.shape[-1]
    side = int(math.sqrt(n_pos))
    patch = pos[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
    sf = (grid + 0.1) / math.sqrt(n_pos)
    patch = F.interpolate(patch, scale_factor=(sf, sf), mode="bicubic", align_corners=False)
    if patch.shape[-1] != grid or patch.shape[-2] != grid:
        raise ValueError("position-embedding interpolation produced an unexpected grid")
    patch = patch.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat((pos[0, :1], patch), dim=0)


def pack_vit(W, cfg, fp8=False):
    vc = cfg.perceiver_cfg.vis_encoder_cfg
    D, P = vc.hidden_size, vc.patch_size
    grid = cfg.image_size // P
    v = "perceiver.vis_encoder."
    dev = W.device
    Kp = _ru(3 * P * P, 64)
    pw = W(v + "embeddings.patch_embeddings.projection.weight").reshape(D, -1)
    pos = vit_pos_embed_4_32(W(v + "embeddings.position_embeddings"), grid).to(dev)
    cls = W(v + "embeddings.cls_token").reshape(D)
    out = dict(fp8=fp8, grid=grid, Kpad=Kp, patch_w=bf(pad_k(pw, Kp)), patch_b=W(v + "embeddings.patch_embeddings.projection.bias"),
               cls_pos0=(cls + pos[0]).contiguous(), pos_patch=pos[1:].contiguous(), layers=[])
    for i in range(vc.num_hidden_layers):
        p = f"{v}encoder.layer.{i}."
        a = p + "attention.attention."
        out["layers"].append(dict(
            ln1_g=W(p + "norm1.weight"), ln1_b=W(p + "norm1.bias"),
            wqkv=wq(torch.cat([W(a + "query.weight"), W(a + "key.weight"), W(a + "value.weight")], 0), fp8),
            bqkv=torch.cat([W(a + "query.bias"), W(a + "key.bias"), W(a + "value.bias")], 0).contiguous(),
            wo=wq(W(p + "attention.output.dense.weight"), fp8), bo=W(p + "attention.output.dense.bias"),
            ls1=W(p + "layer_scale1.lambda1"),
            ln2_g=W(p + "norm2.weight"), ln2_b=W(p + "norm2.bias"),
            w1=wq(W(p + "mlp.fc1.weight"), fp8), b1=W(p + "mlp.fc1.bias"),
            w2=wq(W(p + "mlp.fc2.weight"), fp8), b2=W(p + "mlp.fc2.bias"),
            ls2=W(p + "layer_scale2.lambda1")))
    return out


def _sine_pos(h, w, d):
    """DeformableDetrSinePositionEmbedding(normalize=True) on an all-valid mask: input independent."""
    npf, eps, scale = d // 2, 1e-6, 2 * math.pi
    ones = torch.ones((1, h, w), dtype=F32)
    y, x = ones.cumsum(1), ones.cumsum(2)
    y = (y - 0.5) / (y[:, -1:, :] + eps) * scale
    x = (x - 0.5) / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=F32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    px, py = x[:, :, :, None] / dim_t, y[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).reshape(h * w, d)


def pack_ddetr(W, cfg):
    dc = cfg.perceiver_cfg.ddetr_cfg
    D = cfg.perceiver_cfg.vis_encoder_cfg.hidden_size
    d = dc.d_model
    g = cfg.image_size // cfg.perceiver_cfg.vis_encoder_cfg.patch_size
    dev = W.device
    t = "perceiver.ddetr_transformer."
    if dc.num_feature_levels != 1 or not dc.two_stage or not dc.with_box_refine:
        raise NotImplementedError("only the reference's single-level two-stage box-refine DDETR is built")

    def lin(name):
        return W(name + ".weight").contiguous(), W(name + ".bias").contiguous()

    def msda(p):
        ow, ob = lin(p + "sampling_offsets")
        aw, ab = lin(p + "attention_weights")
        vw, vb = lin(p + "value_proj")
        uw, ub = lin(p + "output_proj")
        return dict(offw_w=torch.cat([ow, aw], 0).contiguous(), offw_b=torch.cat([ob, ab], 0).contiguous(), v_w=vw, v_b=vb,
                    o_w=uw, o_b=ub)

    def ln(name):
        return W(name + ".weight"), W(name + ".bias")

    def mlp3(p):
        return [lin(f"{p}.layers.{k}") for k in range(3)]

    out = dict(g=g, d=d)
    out["proj_w"] = W("perceiver.input_proj.0.0.weight").reshape(d, D).contiguous()
    out["proj_b"] = W("perceiver.input_proj.0.0.bias")
    out["proj_ln"] = ln("perceiver.input_proj.0.1")
    out["pos"] = (_sine_pos(g, g, d).to(dev) + W(t + "level_embed")[0].view(1, -1)).contiguous()
    ry, rx = torch.meshgrid(torch.linspace(0.5, g - 0.5, g, dtype=F32), torch.linspace(0.5, g - 0.5, g, dtype=F32),
                            indexing="ij")
    out["enc_ref"] = torch.stack((rx.reshape(-1) / g, ry.reshape(-1) / g), -1).contiguous().to(dev)
    # two-stage proposals (ddetr_transformer.py:383-423): input independent
    gy, gx = torch.meshgrid(torch.linspace(0, g - 1, g, dtype=F32), torch.linspace(0, g - 1, g, dtype=F32), indexing="ij")
    grid = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([g, g], dtype=F32)
    prop = torch.cat((grid, torch.ones_like(grid) * 0.05), -1).view(-1, 4)
    if not ((prop > 0.01) & (prop < 0.99)).all():
        raise NotImplementedError("proposal validity mask is not all-true for this grid")
    out["proposals"] = torch.log(prop / (1 - prop)).contiguous().to(dev)
    out["enc"] = []
    for i in range(dc.encoder_layers):
        p = f"{t}encoder.layers.{i}."
        out["enc"].append(dict(att=msda(p + "self_attn."), ln1=ln(p + "self_attn_layer_norm"), fc1=lin(p + "fc1"),
                               fc2=lin(p + "fc2"), ln2=ln(p + "final_layer_norm")))
    out["dec"] = []
    for i in range(dc.decoder_layers):
        p = f"{t}decoder.layers.{i}."
        qw, qb = lin(p + "self_attn.q_proj")
        kw, kb = lin(p + "self_attn.k_proj")
        out["dec"].append(dict(qk_w=torch.cat([qw, kw], 0).contiguous(), qk_b=torch.cat([qb, kb], 0).contiguous(),
                               v=lin(p + "self_attn.v_proj"), o=lin(p + "self_attn.out_proj"),
                               ln1=ln(p + "self_attn_layer_norm"), att=msda(p + "encoder_attn."),
                               ln2=ln(p + "encoder_attn_layer_norm"), fc1=lin(p + "fc1"), fc2=lin(p + "fc2"),
                               ln3=ln(p + "final_layer_norm")))
    n = dc.decoder_layers
    out["enc_output"], out["enc_output_norm"] = lin(t + "enc_output"), ln(t + "enc_output_norm")
    out["pos_trans"], out["pos_trans_norm"] = lin(t + "pos_trans"), ln(t + "pos_trans_norm")
    out["class_enc"] = lin(t + "class_embed_enc")
    out["bbox_enc"] = mlp3(f"{t}bbox_embed.{n}")
    out["bbox_prev"] = mlp3(f"{t}bbox_embed.{n - 2}") if n > 1 else None
    out["bbox_last"] = mlp3(f"{t}bbox_embed.{n - 1}")
    out["class_coco"], out["class_sa1b"] = lin(f"{t}class_embed_coco.{n - 1}"), lin(f"{t}class_embed_sa1b.{n - 1}")
    out["target"] = W(t + "query_position_embeddings.weight").contiguous()
    return out


# e4m3 region-encoder convs (fp8 mode): the A operand of fuse round r >= 1 and of the per-ROI conv is relu(GroupNorm(conv)) of the
# round before, whose magnitude is bounded by the GroupNorm's own affine parameters: |gamma| * (a normalised deviation) + |beta|.
# CONV_ACT_SIGMAS normalised deviations are representable; beyond, the value saturates at the bound (e4m3 is a floating-point
# format, so a generous bound costs no relative precision -- only values below bound * 2^-15 flush).  The scale is therefore a
# constant of the weights: no reduction pass over the maps, nothing for the GEMM epilogue to look up (it is folded into w_scale),
# and the oracle restates it exactly (oracle/groma_oracle.py conv_act_scale).  Round 0 reads the raw 1x1-conv outputs of the ViT
# states, which have no such bound: it keeps 16-bit operands.
CONV_ACT_SIGMAS = 64.0
FP8_HEAD = True  # fp8 mode: lm_head (+) extra_lm_head as e4m3 rows (a21 is named by BASELINE configs[4])


def conv_act_scale(g, b):
    """static e4m3 scale of relu(GroupNorm(.)) activations with affine (g, b): value ~= e4m3 * scale"""
    return max(CONV_ACT_SIGMAS * float(g.abs().max()) + float(b.abs().max()), 1e-20) / 448.0


def pack_region(W, cfg, fp8=False):
    rc = cfg.region_cfg
    D = cfg.perceiver_cfg.vis_encoder_cfg.hidden_size
    m, ra = "region_encoder.mlvl_fuse.", "region_encoder.roi_align."
    Cp = _ru(D + 2, 64)
    fp8 = bool(fp8) and D % 128 == 0 and rc.num_fuse >= 1  # (the e4m3 conv gather needs whole 128-deep K-tiles per tap)
    if fp8 and ops.SP() == 2:
        raise NotImplementedError("e4m3 operands and the reference-precision build are exclusive")
    out = dict(Cpad=Cp, in_w=[], in_b=[], fuse=[], fp8=fp8)
    for l in range(rc.num_levels):
        out["in_w"].append(bf(pad_k(W(f"{m}input_conv.{l}.weight").reshape(D, D + 2), Cp)))
        out["in_b"].append(W(f"{m}input_conv.{l}.bias"))
    for r in range(rc.num_fuse):
        w = W(f"{m}fuse_convs.{r}.conv.weight")  # [D, D, 3, 3] -> [D, (ky,kx,c)]
        ent = dict(g=W(f"{m}fuse_convs.{r}.gn.weight"), b=W(f"{m}fuse_convs.{r}.gn.bias"))
        wk = w.permute(0, 2, 3, 1).reshape(D, 9 * D)
        if fp8 and r >= 1:  # e4m3 weights, per-output-channel scale x the static scale of this round's input maps
            s_in = conv_act_scale(out["fuse"][r - 1]["g"], out["fuse"][r - 1]["b"])
            w8, sw = q8(wk)
            ent.update(w8=w8, ws8=(sw * s_in).contiguous(), q_inv=1.0 / s_in)
        else:
            ent["w"] = bf(wk)
        out["fuse"].append(ent)
    pw = [W(f"{ra}pconvs.{l}.weight").permute(0, 2, 3, 1).reshape(D, 9 * D) for l in range(rc.num_levels)]
    if fp8:  # the RoIAlign tiles are bilinear means of the last round's relu(GroupNorm(.)) maps: same bound
        s_in = conv_act_scale(out["fuse"][-1]["g"], out["fuse"][-1]["b"])
        w8, sw = q8(torch.cat(pw, 1))
        out["pconv_w8"], out["pconv_ws8"], out["pconv_q_inv"] = w8, (sw * s_in).contiguous(), 1.0 / s_in
    else:
        out["pconv_w"] = bf(torch.cat(pw, 1))
    out["pconv_b"] = sum(W(f"{ra}pconvs.{l}.bias") for l in range(rc.num_levels)).contiguous()
    P2 = rc.roi_size ** 2
    fw = W(ra + "flatten_linear.weight")  # [mid, D*P2] with k = c*P2 + hw  ->  k' = hw*D + c
    out["flat_w"] = bf(fw.view(-1, D, P2).permute(0, 2, 1).reshape(-1, P2 * D))
    out["flat_b"] = W(ra + "flatten_linear.bias")
    out["pe0_w"] = pad_k(W(ra + "pos_embedd.0.weight"), 16).contiguous()
    out["pe0_b"] = W(ra + "pos_embedd.0.bias")
    out["pe_ln1"] = (W(ra + "pos_embedd.2.weight"), W(ra + "pos_embedd.2.bias"))
    out["pe3_w"], out["pe3_b"] = W(ra + "pos_embedd.3.weight").contiguous(), W(ra + "pos_embedd.3.bias")
    out["pe_ln2"] = (W(ra + "pos_embedd.5.weight"), W(ra + "pos_embedd.5.bias"))
    out["up_w"], out["up_b"] = bf(W(ra + "updims.weight")), W(ra + "updims.bias")
    return out


def pack_llm(W, cfg, fp8=False, prec=None):
    """prec: None (everything in the active operand type) or {"attn", "mlp", "head": type} -- the per-stage operand types of
    groma_amd.groma.parse_precision: each stage's weights are packed in ITS storage (the embedding tables stay in the active one)"""
    import contextlib
    lc = cfg.llm_cfg
    T, I = lc.hidden_size, lc.intermediate_size
    if prec is not None and fp8:
        raise NotImplementedError("per-stage operand types and e4m3 weights are exclusive")
    st = (lambda stage: ops.precision(prec[stage])) if prec is not None else (lambda stage: contextlib.nullcontext())
    out = dict(layers=[], fp8=fp8)
    out["embed"] = bf(W("llm.model.embed_tokens.weight"))
    out["new_embed"] = bf(W("new_input_embs.weight"))
    for i in range(lc.num_hidden_layers):
        p = f"llm.model.layers.{i}."
        gate, up = W(p + "mlp.gate_proj.weight"), W(p + "mlp.up_proj.weight")
        with st("attn"):
            ent = dict(n1=W(p + "input_layernorm.weight"),
                       wqkv=wq(torch.cat([W(p + "self_attn.q_proj.weight"), W(p + "self_attn.k_proj.weight"),
                                          W(p + "self_attn.v_proj.weight")], 0), fp8),
                       wo=wq(W(p + "self_attn.o_proj.weight"), fp8))
        with st("mlp"):
            ent.update(n2=W(p + "post_attention_layernorm.weight"),
                       wgu=wq(torch.stack([gate, up], 1).reshape(2 * I, T), fp8),  # interleaved rows: gate_0, up_0, gate_1, ...
                       wd=wq(W(p + "mlp.down_proj.weight"), fp8))
        out["layers"].append(ent)
    out["norm"] = W("llm.model.norm.weight")
    V = lc.vocab_size + cfg.num_new_token
    Vp = _ru(V, 128)
    with st("head"):
        head = torch.zeros((Vp, T * ops.SP()), dtype=H16(), device=W.device)
        head[: lc.vocab_size] = bf(W("llm.lm_head.weight"))
        head[lc.vocab_size: V] = bf(W("extra_lm_head.weight"))
    out["head"], out["V"], out["Vpad"] = head, V, Vp
    if fp8 and FP8_HEAD:  # BASELINE configs[4] names a21: lm_head (+) extra_lm_head as e4m3 rows with per-output-channel scales (padding rows: zeros)
        hf = torch.zeros((Vp, T), dtype=F32, device=W.device)
        hf[: lc.vocab_size] = W("llm.lm_head.weight").float()
        hf[lc.vocab_size: V] = W("extra_lm_head.weight").float()
        out["head8"] = q8(hf)
    hd = T // lc.num_attention_heads
    inv = 1.0 / (lc.rope_theta ** (torch.arange(0, hd, 2, dtype=F32) / hd))
    fr = torch.outer(torch.arange(lc.max_position_embeddings, dtype=F32), inv)
    out["cos"], out["sin"] = fr.cos().contiguous().to(W.device), fr.sin().contiguous().to(W.device)
    return out


def pack_bridge(W, cfg):
    return dict(w0=bf(W("img_txt_bridge.0.weight")), b0=W("img_txt_bridge.0.bias"), w2=bf(W("img_txt_bridge.2.weight")),
                b2=W("img_txt_bridge.2.bias"))
