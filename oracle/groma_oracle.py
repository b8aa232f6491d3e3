"""
oracle/groma_oracle.py -- TEST INFRASTRUCTURE ONLY.  CPU (PyTorch fp32) restatement of the reference algorithm of
Groma's localized-visual-tokenization forward path.  The product (groma_amd/) never imports this module; only
tests/, bench.py's `cpu_baseline` leg and __graft_entry__.smoke() do, and only as the checker.

PARITY PIN STATUS
  * nms / roi_align: pinned to the reference's golden vectors (tests/test_oracle_goldens.py) and, when built,
    to the reference's own CPU sources compiled from /root/reference (oracle/_ref).
  * MSDA: pinned by the reference's own recipe (mmcv/tests/test_ops/test_ms_deformable_attn.py:54-70) -- the
    grid_sample formulation of mmcv/mmcv/ops/multi_scale_deform_attn.py:93-150 is used verbatim here.
  * DINOv2 / Deformable-DETR layers / LLaMA arithmetic lives in the un-vendored `transformers==4.32.0`
    (pyproject.toml:19 of the reference) -- NOT under /root/reference, not installable here, and no reference test pins
    it.  It is restated from the published 4.32 algorithm and PINNED to the closest executable statement of it, the
    transformers 5.15 modules on disk, at 2e-5 (tests/test_oracle_vs_hf.py): Dinov2Model, LlamaModel (right padding,
    incremental decoding), DeformableDetrEncoderLayer / DecoderLayer / MultiscaleDeformableAttention /
    SinePositionEmbedding / get_reference_points / gen_encoder_output_proposals / get_proposal_pos_embed /
    MLPPredictionHead / inverse_sigmoid (4.32 <-> 5.15 renames mapped in the test).  T8 (4.32's position-table
    resize) is restated from the 4.32 formula: that one line is unpinned.

Every function cites the reference file:line it follows ("R:" = /root/reference/, "HF:" = transformers 4.32 semantics,
cross-checked in /usr/local/lib/python3.10/dist-packages/transformers/models/...).

State dict = the reference's own parameter names (what GromaModel.from_pretrained loads).
cfg = plain dict (groma_amd.config.GromaConfig.to_dict()).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cref

IGNORE_INDEX = -100

# ------------------------------------------------------------------------------------------------ operand rounding
# The "bf16-rounded oracle" (SURVEY 'Hard parts', BASELINE north_star "logits within 1e-3"): the same fp32 restatement,
# with GEMM operands rounded to bf16 at exactly the points where the MI355X path holds bf16 in HBM:
# weights of the ViT / bridge / region-encoder / LLaMA linears and convs, normalisation outputs, projection outputs,
# soft-max probabilities, attention context, activation outputs.  Accumulation, residual streams, soft-max statistics,
# normalisation statistics, biases and the whole DDETR proposer stay fp32 (as on the device).  With rounding off (the
# default) every _r() below is the identity, so the fp32 oracle and its goldens are untouched.
_ROUND = [None]
# Per-stage selection of the rounding points (round 6: the precision ablation, tests/diag/precision_ablation.py).  Every rounding
# point carries a key "<stage>.<kind>": the stage is the innermost `with _st(name):` block it is evaluated in --
#   vit | bridge | region.in / region.fuse / region.pconv / region.flat / region.up | embed |
#   llm.qkv (norm1 output, q/k/v weights, projection outputs incl. RoPE) | llm.pv (soft-max probabilities, context) |
#   llm.o | llm.gateup | llm.down (SwiGLU output, weights) | head
# -- and the kind is "a" (an activation operand), "w" (a weight) or "o" (a stored output).  rounding(mode, only={...}) rounds
# only the points whose key matches one of the patterns, rounding(mode, skip={...}) all but those; a pattern matches a key that
# equals it or that it prefixes at a dot ("llm" = every LLaMA point, "llm.down.w" = the down-projection weights alone).
# The device's per-stage operand types (groma_amd.groma.parse_precision) map onto it: a stage on operand pairs = its points skipped.
_ONLY, _SKIP, _STAGE = [None], [None], ["-"]


class rounding:
    """with rounding("bf16"): ...   -- evaluate the oracle with bf16-rounded operands (None = pure fp32).
    "fp16": the same rounding points with IEEE half (the fp16 operand build of the device library, libgroma_hip_f16.so).
    only / skip: restrict the rounding to (all but) the points matching these "<stage>[.<kind>]" patterns (see above)."""

    def __init__(self, mode, only=None, skip=None):
        assert mode in (None, "bf16", "e4m3", "fp16")
        assert (only is None and skip is None) or mode in ("bf16", "fp16"), "per-stage selection is defined for the 16-bit modes"
        self.mode, self.only, self.skip = mode, (None if only is None else tuple(only)), (None if skip is None else tuple(skip))

    def __enter__(self):
        self.prev = (_ROUND[0], _ONLY[0], _SKIP[0])
        _ROUND[0], _ONLY[0], _SKIP[0] = self.mode, self.only, self.skip

    def __exit__(self, *a):
        _ROUND[0], _ONLY[0], _SKIP[0] = self.prev


class _st:
    """with _st("llm.qkv"): ...  -- names the stage of the rounding points evaluated inside (innermost block wins)"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev, _STAGE[0] = _STAGE[0], self.name

    def __exit__(self, *a):
        _STAGE[0] = self.prev


def _in_stage(name):
    """decorator: the whole function body is the stage `name`"""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **kw):
            with _st(name):
                return fn(*a, **kw)
        return wrapped
    return deco


def _match(pats, key):
    return any(key == p or key.startswith(p + ".") for p in pats)


def rounds_here(kind="a"):
    """does a rounding point of this kind, in the current stage, round under the active mode / selection?"""
    if _ROUND[0] is None:
        return False
    if _ONLY[0] is None and _SKIP[0] is None:
        return True
    key = _STAGE[0] + "." + kind
    if _ONLY[0] is not None and not _match(_ONLY[0], key):
        return False
    return not (_SKIP[0] is not None and _match(_SKIP[0], key))


def _r(x, kind="a"):
    if not rounds_here(kind):
        return x
    return x.to(torch.float16 if _ROUND[0] == "fp16" else torch.bfloat16).to(torch.float32)


# "e4m3" mode (BASELINE configs[4]): everything of the bf16 mode, plus the DINOv2 / LLaMA linears evaluated with OCP e4m3
# operands exactly as the device's fp8 path forms them (groma_amd/csrc/fp8.hip, groma_amd/weights.py q8):
#   activations: per-ROW dynamic scale s = max(|x|, 1e-20) / 448, q = e4m3_rne(x * (1 / s))   (fp32 reciprocal, then multiply)
#   weights:     per-OUTPUT-CHANNEL scale s = max(|w|, 1e-20) / 448, q = e4m3_rne(w / s)      (quantised once at load)
#   product:     fp32 accumulation of the exact e4m3 products, then * w_scale[n] * a_scale[m], then the fp32 epilogue.
# A normalisation output is quantised straight from fp32 (the fused norm -> e4m3 kernel); a stored activation (attention
# context, GELU / SwiGLU output) is a bf16 tensor that is then row-quantised.  lm_head (+) extra_lm_head read the final RMSNorm
# the same way (round 4).  Bridge, patch embedding, the region encoder's 1x1 input convs, round-0 fuse conv and linears keep bf16.
# The region encoder's 3x3 convs from fuse round 1 on, and the per-ROI conv (round 4; groma_amd/weights.py pack_region):
#   activations: relu(GroupNorm(.)) maps / their RoIAlign tiles, STATIC scale s = (64 * max|gamma| + max|beta|) / 448 of the
#                GroupNorm that produced them (conv_act_scale), q = e4m3_rne(clamp(x * (1 / s), -448, 448))
#   weights:     per-OUTPUT-CHANNEL scale over the channel's (C, 3, 3) taps (the per-ROI conv: over its three levels' taps jointly)
#   product:     fp32 accumulation of the exact e4m3 products, then * (w_scale[n] * s)  (one fp32 factor, as the device folds it)
E4M3 = torch.float8_e4m3fn
CONV_ACT_SIGMAS = 64.0


def conv_act_scale(g, b):
    return max(CONV_ACT_SIGMAS * float(g.abs().max()) + float(b.abs().max()), 1e-20) / 448.0


def _conv8(x, w, s_in, b=None, **kw):
    """e4m3 3x3 conv of the device: x fp32 [bs, C, H, W] (the tensor the quantiser sees), w fp32 [N, C, 3, 3]"""
    xq = (x * (1.0 / s_in)).clamp(-448.0, 448.0).to(E4M3).to(torch.float32)
    sw = w.flatten(1).abs().amax(dim=1).clamp_min(1e-20) / 448.0
    wq = (w / sw[:, None, None, None]).to(E4M3).to(torch.float32)
    y = F.conv2d(xq, wq, None, **kw) * (sw * s_in)[None, :, None, None]
    return y if b is None else y + b[None, :, None, None]


def quant_rows_e4m3(x):
    """-> (q as fp32 values on the e4m3 grid, scale [rows, 1]) with x ~= q * scale"""
    s = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-20) / 448.0
    return (x * (1.0 / s)).to(E4M3).to(torch.float32), s


def quant_weight_e4m3(w):
    s = w.abs().amax(dim=1).clamp_min(1e-20) / 448.0
    return (w / s[:, None]).to(E4M3).to(torch.float32), s


def _lin8(x, w, b=None):
    """e4m3 GEMM of the device: x fp32 (already the tensor the quantiser sees) [.., K], w fp32 [N, K]"""
    xq, sx = quant_rows_e4m3(x)
    wq, sw = quant_weight_e4m3(w)
    y = F.linear(xq, wq) * sw * sx
    return y if b is None else y + b


def _linA(x, sd, name, bias=True, src="act"):
    """a DINOv2 / LLaMA linear: fp32 | bf16 operands | e4m3 operands, by the active mode.  src = "norm": x is a normalisation
    output (quantised from fp32 in e4m3 mode); "act": x is a stored bf16 activation."""
    if _ROUND[0] == "e4m3":
        return _lin8(x if src == "norm" else _r(x), sd[name + ".weight"], sd.get(name + ".bias") if bias else None)
    return _lin16(x, sd, name, bias)


def _lin16(x, sd, name, bias=True):
    """a Linear the device runs as a bf16 MFMA GEMM: bf16 operands, fp32 accumulate, fp32 bias"""
    return F.linear(_r(x), _r(sd[name + ".weight"], "w"), sd.get(name + ".bias") if bias else None)


def _conv16(x, w, b=None, **kw):
    return F.conv2d(_r(x), _r(w, "w"), b, **kw)


# ------------------------------------------------------------------------------------------------ helpers
def _lin(x, sd, name, bias=True):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias") if bias else None)


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def center_to_corners_format(b):
    # HF transformers.image_transforms.center_to_corners_format (used at R: groma/model/groma.py:268,287)
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def box_iou(b1, b2):
    # torchvision.ops.box_iou (R: groma/model/groma.py:286,299); torchvision 0.16 boxes.py: _box_inter_union
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def inverse_sigmoid(x, eps=1e-5):
    # HF modeling_deformable_detr.inverse_sigmoid
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# ------------------------------------------------------------------------------------------------ DINOv2 (a1)
def vit_pos_embed(sd, vc, grid, prefix="perceiver.vis_encoder."):
    """HF 4.32 Dinov2Embeddings.interpolate_pos_encoding: bicubic, scale_factor=(grid+0.1)/sqrt(N)  (SURVEY T8)."""
    pos = sd[prefix + "embeddings.position_embeddings"]
    n_pos = pos.shape[1] - 1
    if n_pos == grid * grid:
        return pos
    dim = pos.shape[-1]
    side = int(math.sqrt(n_pos))
    cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
    h = w = grid + 0.1
    patch_pos = patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2)
    patch_pos = F.interpolate(patch_pos, scale_factor=(h / math.sqrt(n_pos), w / math.sqrt(n_pos)), mode="bicubic",
                              align_corners=False)
    assert patch_pos.shape[-1] == grid and patch_pos.shape[-2] == grid
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)


ATT_KEY_TILE = 64  # keys per online-softmax step of the device's prefill attention kernel (csrc/attention.hip, KV)


def _softmax_pv(scores, v):
    """softmax(scores) @ v.  Rounded mode mirrors the flash kernel's number formats: keys are consumed in tiles of 64,
    the un-normalised exp(s - running_max) of a tile is rounded to bf16 for the P.V product (the running max being the
    maximum over the tiles seen SO FAR, so which value gets rounded depends on the tile order), earlier tiles are rescaled
    in fp32, the row sum is kept in fp32 from the un-rounded values, and the context is rounded to bf16 after the division.
    (With zero-mean V a different-but-equivalent rounding point -- e.g. exp(s - final_max) -- changes the context by
    ~1e-3 relative: the rounding errors of P do not average out, they random-walk like the signal.)"""
    if _ROUND[0] is None or not (rounds_here("a") or rounds_here("o")):
        # (float64 inputs stay float64: tests/diag/index_survival.py evaluates this restatement in double to apportion the fp32 noise)
        return torch.softmax(scores, dim=-1, dtype=torch.float64 if scores.dtype == torch.float64 else torch.float32) @ v
    S = scores.shape[-1]
    nt = (S + ATT_KEY_TILE - 1) // ATT_KEY_TILE
    pad = nt * ATT_KEY_TILE - S
    sp = F.pad(scores, (0, pad), value=torch.finfo(torch.float32).min) if pad else scores
    tile_max = sp.view(*sp.shape[:-1], nt, ATT_KEY_TILE).amax(dim=-1)
    m_run = torch.cummax(tile_max, dim=-1).values                      # running max after each tile
    m_key = m_run.repeat_interleave(ATT_KEY_TILE, dim=-1)[..., :S]     # the max each key's exp() was taken against
    m_fin = m_run[..., -1:]
    e = torch.exp(scores - m_key)
    w = torch.exp(m_key - m_fin)                                       # fp32 rescale of earlier tiles (product of alphas)
    return _r(((_r(e) * w) @ v) / (e * w).sum(dim=-1, keepdim=True), "o")


@_in_stage("vit")
def vit_forward(sd, cfg, images, prefix="perceiver.vis_encoder."):
    """HF Dinov2Model(images, output_hidden_states=True).hidden_states  (called at R: groma/model/groma.py:222).
    Returns the tuple (embeddings, layer_1, ..., layer_N); the final LayerNorm is never applied."""
    vc = cfg["perceiver_cfg"]["vis_encoder_cfg"]
    D, heads, P = vc["hidden_size"], vc["num_attention_heads"], vc["patch_size"]
    eps = vc["layer_norm_eps"]
    bs, _, S, _ = images.shape
    grid = S // P
    x = _conv16(images, sd[prefix + "embeddings.patch_embeddings.projection.weight"],
                sd[prefix + "embeddings.patch_embeddings.projection.bias"], stride=P)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd[prefix + "embeddings.cls_token"].expand(bs, -1, -1), x), dim=1)
    x = x + vit_pos_embed(sd, vc, grid, prefix)
    hidden = [x]
    hd = D // heads
    for i in range(vc["num_hidden_layers"]):
        p = f"{prefix}encoder.layer.{i}."
        y = _ln(x, sd, p + "norm1", eps)
        q = _r(_linA(y, sd, p + "attention.attention.query", src="norm"), "o").view(bs, -1, heads, hd).transpose(1, 2)
        k = _r(_linA(y, sd, p + "attention.attention.key", src="norm"), "o").view(bs, -1, heads, hd).transpose(1, 2)
        v = _r(_linA(y, sd, p + "attention.attention.value", src="norm"), "o").view(bs, -1, heads, hd).transpose(1, 2)
        ctx = _softmax_pv(q @ k.transpose(-1, -2) / math.sqrt(hd), v).transpose(1, 2).reshape(bs, -1, D)
        x = x + sd[p + "layer_scale1.lambda1"] * _linA(ctx, sd, p + "attention.output.dense")
        y = _ln(x, sd, p + "norm2", eps)
        y = _linA(F.gelu(_linA(y, sd, p + "mlp.fc1", src="norm")), sd, p + "mlp.fc2")
        x = x + sd[p + "layer_scale2.lambda1"] * y
        hidden.append(x)
    return tuple(hidden)


# ------------------------------------------------------------------------------------------------ DDETR (a5-a9)
def msda_pytorch(value, spatial_shapes, sampling_locations, attention_weights):
    """R: mmcv/mmcv/ops/multi_scale_deform_attn.py:93-150 (multi_scale_deformable_attn_pytorch)."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([H_ * W_ for H_, W_ in spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (H_, W_) in enumerate(spatial_shapes):
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, H_, W_)
        sampling_grid_l_ = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_l_ = F.grid_sample(value_l_, sampling_grid_l_, mode="bilinear", padding_mode="zeros",
                                          align_corners=False)
        sampling_value_list.append(sampling_value_l_)
    attention_weights = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries,
                                                                  num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1).view(
        bs, num_heads * embed_dims, num_queries)
    return output.transpose(1, 2).contiguous()


def _msda_module(sd, p, query, memory, reference_points, spatial_shapes, heads, n_points):
    """HF DeformableDetrMultiscaleDeformableAttention.forward (position embeddings already added to `query`)."""
    bs, nq, d = query.shape
    n_levels = len(spatial_shapes)
    value = _lin(memory, sd, p + "value_proj").view(bs, memory.shape[1], heads, d // heads)
    off = _lin(query, sd, p + "sampling_offsets").view(bs, nq, heads, n_levels, n_points, 2)
    aw = _lin(query, sd, p + "attention_weights").view(bs, nq, heads, n_levels * n_points)
    aw = F.softmax(aw, -1).view(bs, nq, heads, n_levels, n_points)
    if reference_points.shape[-1] == 2:
        norm = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / n_points * reference_points[:, :, None, :, None, 2:] * 0.5
    out = msda_pytorch(value, spatial_shapes, loc, aw)
    return _lin(out, sd, p + "output_proj")


def sine_position_embedding(bs, h, w, d_model):
    """HF DeformableDetrSinePositionEmbedding(d_model//2, normalize=True) on an all-valid mask
    (built at R: groma/model/ddetr_transformer.py:302, called :498)."""
    npf, temperature, scale, eps = d_model // 2, 10000, 2 * math.pi, 1e-6
    mask = torch.ones((bs, h, w), dtype=torch.float32)
    y_embed, x_embed = mask.cumsum(1), mask.cumsum(2)
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def gen_encoder_output_proposals(sd, p, enc_output, h, w):
    """R: groma/model/ddetr_transformer.py:383-430 with an all-valid mask, single level."""
    bs = enc_output.shape[0]
    grid_y, grid_x = torch.meshgrid(torch.linspace(0, h - 1, h, dtype=torch.float32),
                                    torch.linspace(0, w - 1, w, dtype=torch.float32), indexing="ij")
    grid = torch.cat([grid_x.unsqueeze(-1), grid_y.unsqueeze(-1)], -1)
    scale = torch.tensor([w, h], dtype=torch.float32).view(1, 1, 1, 2).expand(bs, -1, -1, -1)
    grid = (grid.unsqueeze(0).expand(bs, -1, -1, -1) + 0.5) / scale
    wh = torch.ones_like(grid) * 0.05
    proposals = torch.cat((grid, wh), -1).view(bs, -1, 4)
    valid = ((proposals > 0.01) & (proposals < 0.99)).all(-1, keepdim=True)
    proposals = torch.log(proposals / (1 - proposals))
    proposals = proposals.masked_fill(~valid, float("inf"))
    object_query = enc_output.masked_fill(~valid, float(0))
    object_query = F.layer_norm(_lin(object_query, sd, p + "enc_output"), (enc_output.shape[-1],),
                                sd[p + "enc_output_norm.weight"], sd[p + "enc_output_norm.bias"], 1e-5)
    return object_query, proposals


def get_proposal_pos_embed(proposals, num_pos_feats):
    """R: groma/model/ddetr_transformer.py:432-446."""
    temperature, scale = 10000, 2 * math.pi
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    proposals = proposals.sigmoid() * scale
    pos = proposals[:, :, :, None] / dim_t
    return torch.stack((pos[:, :, :, 0::2].sin(), pos[:, :, :, 1::2].cos()), dim=4).flatten(2)


def _mlp_head(x, sd, p, n=3):
    # HF DeformableDetrMLPPredictionHead
    for i in range(n):
        x = _lin(x, sd, f"{p}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def stable_topk(x, k):
    """torch.topk at R: ddetr_transformer.py:556 with the tie rule fixed to (value desc, index asc)."""
    return torch.sort(x, dim=1, descending=True, stable=True)[1][:, :k]


def ddetr_encoder_layer(sd, p, x, pos, ref, shapes, heads, n_points):
    """HF 4.32 DeformableDetrEncoderLayer (post-LN): x = LN(x + MSDA(x + pos, value = x)); x = LN(x + fc2(relu(fc1 x)))
    (instantiated at R: groma/model/ddetr_transformer.py:299-301 through HF DeformableDetrEncoder)."""
    y = _msda_module(sd, p + "self_attn.", x + pos, x, ref, shapes, heads, n_points)
    x = _ln(x + y, sd, p + "self_attn_layer_norm", 1e-5)
    y = _lin(F.relu(_lin(x, sd, p + "fc1")), sd, p + "fc2")
    return _ln(x + y, sd, p + "final_layer_norm", 1e-5)


def ddetr_decoder_layer(sd, p, hs, query_pos, memory, ref_in, shapes, heads, n_points):
    """HF 4.32 DeformableDetrDecoderLayer as called at R: groma/model/ddetr_transformer.py:136-145: MHA self-attention
    (q = k = hs + pos, v = hs, q scaled by hd^-0.5) -> LN -> MSDA cross-attention over `memory` -> LN -> FFN -> LN."""
    bs, _, d = hs.shape
    hd = d // heads
    qk_in = hs + query_pos
    q = (_lin(qk_in, sd, p + "self_attn.q_proj") * hd ** -0.5).view(bs, -1, heads, hd).transpose(1, 2)
    k = _lin(qk_in, sd, p + "self_attn.k_proj").view(bs, -1, heads, hd).transpose(1, 2)
    v = _lin(hs, sd, p + "self_attn.v_proj").view(bs, -1, heads, hd).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    y = _lin((att @ v).transpose(1, 2).reshape(bs, -1, d), sd, p + "self_attn.out_proj")
    hs = _ln(hs + y, sd, p + "self_attn_layer_norm", 1e-5)
    y = _msda_module(sd, p + "encoder_attn.", hs + query_pos, memory, ref_in, shapes, heads, n_points)
    hs = _ln(hs + y, sd, p + "encoder_attn_layer_norm", 1e-5)
    y = _lin(F.relu(_lin(hs, sd, p + "fc1")), sd, p + "fc2")
    return _ln(hs + y, sd, p + "final_layer_norm", 1e-5)


def encoder_reference_points(bs, h, w):
    """HF DeformableDetrEncoder.get_reference_points with valid_ratio = 1, one level: ((i + 0.5) / size) grid"""
    ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=torch.float32),
                                  torch.linspace(0.5, w - 0.5, w, dtype=torch.float32), indexing="ij")
    ref = torch.stack((ref_x.reshape(-1)[None] / w, ref_y.reshape(-1)[None] / h), -1)  # [1, hw, 2]
    return ref[:, :, None].expand(bs, -1, 1, -1)


def ddetr_forward(sd, cfg, ddetr_inputs, prefix="perceiver."):
    """input_proj + DeformableDetrTransformer.forward (R: groma/model/groma.py:243-246; ddetr.py:146-155;
    ddetr_transformer.py:484-609, 668-728).  ddetr_inputs: [bs, C, h, w] (mean of the last 4 ViT states)."""
    dc = cfg["perceiver_cfg"]["ddetr_cfg"]
    d, heads = dc["d_model"], dc["encoder_attention_heads"]
    bs, _, h, w = ddetr_inputs.shape
    # input_proj[0]: 1x1 conv + channel LayerNorm (R: ddetr.py:25-45,146-155)
    src = F.conv2d(ddetr_inputs, sd[prefix + "input_proj.0.0.weight"], sd[prefix + "input_proj.0.0.bias"])
    u = src.mean(1, keepdim=True)
    s = (src - u).pow(2).mean(1, keepdim=True)
    src = (src - u) / torch.sqrt(s + 1e-6)
    src = sd[prefix + "input_proj.0.1.weight"][:, None, None] * src + sd[prefix + "input_proj.0.1.bias"][:, None, None]
    t = prefix + "ddetr_transformer."
    pos = sine_position_embedding(bs, h, w, d).flatten(2).transpose(1, 2) + sd[t + "level_embed"][0].view(1, 1, -1)
    x = src.flatten(2).transpose(1, 2)
    shapes = [(h, w)]
    # encoder reference points (HF DeformableDetrEncoder.get_reference_points, valid_ratio = 1)
    enc_ref = encoder_reference_points(bs, h, w)
    for i in range(dc["encoder_layers"]):
        x = ddetr_encoder_layer(sd, f"{t}encoder.layers.{i}.", x, pos, enc_ref, shapes, heads, dc["encoder_n_points"])
    memory = x
    # two-stage proposals (R: ddetr_transformer.py:546-568)
    object_query, output_proposals = gen_encoder_output_proposals(sd, t, memory, h, w)
    enc_class = _lin(object_query, sd, t + "class_embed_enc")
    n_dec = dc["decoder_layers"]
    delta = _mlp_head(object_query, sd, f"{t}bbox_embed.{n_dec}")
    enc_coord_logits = delta + output_proposals
    topk = dc["two_stage_num_proposals"]
    topk_idx = stable_topk(enc_class[..., 0], topk)
    topk_coords_logits = torch.gather(enc_coord_logits, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4))
    reference_points = topk_coords_logits.sigmoid()
    pos_trans_out = _ln(_lin(get_proposal_pos_embed(topk_coords_logits, d // 2), sd, t + "pos_trans"), sd,
                        t + "pos_trans_norm", 1e-5)
    query_pos = pos_trans_out[..., :d]
    hs = sd[t + "query_position_embeddings.weight"].unsqueeze(0).expand(bs, -1, -1)
    # decoder (R: ddetr_transformer.py:107-172; HF 4.32 DeformableDetrDecoderLayer); refs never refined (T3)
    dheads = dc["decoder_attention_heads"]
    ref_in = reference_points[:, :, None]  # * valid_ratios (=1)
    inter, inter_ref = [], []
    for i in range(n_dec):
        hs = ddetr_decoder_layer(sd, f"{t}decoder.layers.{i}.", hs, query_pos, memory, ref_in, shapes, dheads,
                                 dc["decoder_n_points"])
        tmp = _mlp_head(hs, sd, f"{t}bbox_embed.{i}")
        new_ref = (tmp + inverse_sigmoid(reference_points)).sigmoid()
        inter.append(hs)
        inter_ref.append(new_ref)
    # heads, last level (R: ddetr_transformer.py:696-728)
    L = n_dec - 1
    reference = inverse_sigmoid(inter_ref[L - 1] if L > 0 else reference_points)
    pred_boxes = (_mlp_head(inter[L], sd, f"{t}bbox_embed.{L}") + reference).sigmoid()
    logits_coco = _lin(inter[L], sd, f"{t}class_embed_coco.{L}")
    logits_sa1b = _lin(inter[L], sd, f"{t}class_embed_sa1b.{L}")
    return dict(pred_boxes=pred_boxes, logits_coco=logits_coco, logits_sa1b=logits_sa1b, topk_idx=topk_idx,
                enc_class=enc_class[..., 0], memory=memory, init_reference=reference_points, last_hidden=inter[L],
                src=x if False else src.flatten(2).transpose(1, 2))


def fuse_scores(logits_coco, logits_sa1b):
    """R: groma/model/groma.py:247-249"""
    return logits_coco.squeeze(-1).sigmoid() ** 0.4 * logits_sa1b.squeeze(-1).sigmoid() ** 0.6


def select_regions(pred_boxes, scores_fused, refer_boxes, ground_boxes, nms_thres, box_score_thres, max_region_num):
    """R: groma/model/groma.py:252-280.  Consumes the CPU global RNG exactly like the reference (one
    torch.randperm(n) per image).  Returns (selected_boxes, nms_inds_per_image, rand_inds_per_image)."""
    bs = pred_boxes.shape[0]
    selected, all_inds, all_perm = [], [], []
    if refer_boxes is None:
        refer_boxes = [torch.empty((0, 4)) for _ in range(bs)]
    if ground_boxes is None:
        ground_boxes = [torch.empty((0, 4)) for _ in range(bs)]
    for i in range(bs):
        scores_refer = torch.ones(refer_boxes[i].shape[0])
        scores_ground = torch.ones(ground_boxes[i].shape[0]) * 0.2
        scores = torch.cat((scores_fused[i], scores_refer, scores_ground))
        input_boxes = torch.cat((pred_boxes[i], refer_boxes[i], ground_boxes[i]))
        inds = cref.nms(center_to_corners_format(input_boxes).numpy(), scores.numpy(), nms_thres, 0, box_score_thres,
                        max_region_num)
        nms_inds = torch.from_numpy(inds)
        perm = None
        if len(nms_inds) > 0:
            input_boxes = input_boxes[nms_inds]
            perm = torch.randperm(len(input_boxes))
            input_boxes = input_boxes[perm]
        else:
            max_ind = torch.max(scores, dim=0).indices
            input_boxes = input_boxes[max_ind: max_ind + 1]
        selected.append(input_boxes)
        all_inds.append(nms_inds)
        all_perm.append(perm)
    return selected, all_inds, all_perm


# ------------------------------------------------------------------------------------------------ region encoder (a14-a17)
def region_fuse(sd, cfg, mlvl_tokens, prefix="region_encoder."):
    """MLVLROIQueryModule.forward up to mlvl_fuse (R: groma/model/roi_align.py:215-228, 180-193, 150-178).
    mlvl_tokens: 3 x [bs, g*g, C] (ViT hidden states -3,-2,-1 without CLS).  Returns 3 NCHW maps."""
    rc = cfg["region_cfg"]
    bs, n, C = mlvl_tokens[0].shape
    g = int(math.sqrt(n))
    feats = [t.reshape(bs, g, g, C).permute(0, 3, 1, 2) for t in mlvl_tokens]
    nl = len(feats)
    to_shape = [(g * 2 ** lvl, g * 2 ** lvl) for lvl in range(nl)][::-1]
    feats = [F.interpolate(f, size=s, mode="bilinear", align_corners=True) for f, s in zip(feats, to_shape)]
    m = prefix + "mlvl_fuse."
    new = []
    for lvl, f in enumerate(feats):  # coord channels: x then y in [-1,1] (R: roi_align.py:118-126)
        H, W = f.shape[-2:]
        x_range, y_range = torch.linspace(-1, 1, W), torch.linspace(-1, 1, H)
        y, x = torch.meshgrid(y_range, x_range, indexing="ij")
        coord = torch.cat([x.expand(bs, 1, -1, -1), y.expand(bs, 1, -1, -1)], 1)
        f = torch.cat([f, coord], dim=1)
        with _st("region.in"):
            new.append(_r(_conv16(f, sd[f"{m}input_conv.{lvl}.weight"], sd[f"{m}input_conv.{lvl}.bias"]), "o"))
    inputs = new
    shuffle, remain = C // 4, C - 2 * (C // 4)
    for r in range(rc["num_fuse"]):
        fused = []
        for lvl in range(nl):
            top, dow = min(lvl + 1, nl - 1), max(lvl - 1, 0)
            tar = inputs[lvl]
            from_top = F.interpolate(inputs[top][:, remain:][:, shuffle:].to(torch.float32), size=tar.shape[-2:],
                                     mode="bilinear", align_corners=True)
            from_down = F.interpolate(inputs[dow][:, remain:][:, :shuffle].to(torch.float32), size=tar.shape[-2:],
                                      mode="bilinear", align_corners=True)
            fused.append(torch.cat([tar[:, :remain], from_top, from_down], dim=1))
        # mmcv ConvModule: conv(no bias) -> GN(groups) -> ReLU (R: mmcv/mmcv/cnn/bricks/conv_module.py:196-206)
        # (rounded mode: the conv output is stored as bf16 and the GN statistics are taken from the stored values; the
        #  normalised map is only rounded again where it is consumed -- as the next conv's / RoIAlign's bf16 input)
        wr = sd[f"{m}fuse_convs.{r}.conv.weight"]
        with _st("region.fuse"):
            if _ROUND[0] == "e4m3" and r >= 1 and C % 128 == 0:  # (the device's e4m3 conv gather needs C % 128 == 0: weights.pack_region)
                s_in = conv_act_scale(sd[f"{m}fuse_convs.{r - 1}.gn.weight"], sd[f"{m}fuse_convs.{r - 1}.gn.bias"])
                convs = [_conv8(x, wr, s_in, padding=1) for x in fused]
            else:
                convs = [_conv16(x, wr, None, padding=1) for x in fused]
            inputs = [F.relu(F.group_norm(_r(y, "o"), rc["gn_groups"], sd[f"{m}fuse_convs.{r}.gn.weight"],
                                          sd[f"{m}fuse_convs.{r}.gn.bias"], 1e-5)) for y in convs]
    with _st("region.fuse"):
        return [_r(x, "o") for x in inputs]


def roi_extract(sd, cfg, feats, rois_list, prefix="region_encoder.roi_align."):
    """MlvlRoIExtractor.forward (R: groma/model/roi_align.py:274-327).  feats: 3 NCHW maps; rois_list: per-image
    [N_i,4] normalised cxcywh.  ROIs are scaled by the image size only and read as x1y1x2y2 (T1); strides are
    [14/8,14/4,14/2] (T2, R: roi_align.py:204)."""
    rc = cfg["region_cfg"]
    img_size = cfg["image_size"]
    batched = torch.cat(rois_list, dim=0)
    pe = _lin(batched, sd, prefix + "pos_embedd.0")
    pe = F.layer_norm(F.relu(pe), (pe.shape[-1],), sd[prefix + "pos_embedd.2.weight"], sd[prefix + "pos_embedd.2.bias"])
    pe = _lin(pe, sd, prefix + "pos_embedd.3")
    pe = F.layer_norm(F.relu(pe), (pe.shape[-1],), sd[prefix + "pos_embedd.5.weight"], sd[prefix + "pos_embedd.5.bias"])
    rois = torch.cat([torch.cat([r.new_ones(len(r))[:, None] * i, r * img_size], dim=1)
                      for i, r in enumerate(rois_list)])
    strides = [14 / 8, 14 / 4, 14 / 2]
    P = rc["roi_size"]
    acc = None
    rfs = [torch.from_numpy(cref.roi_align_avg(f.float().contiguous().numpy(), rois.float().numpy(), (P, P),
                                               1.0 / strides[lvl], 2, True)) for lvl, f in enumerate(feats)]
    nf = rc["num_fuse"]
    with _st("region.pconv"):
        if _ROUND[0] == "e4m3" and nf >= 1 and feats[0].shape[1] % 128 == 0:  # one e4m3 conv over the three levels' taps (as the device)
            m = prefix.replace("roi_align.", "mlvl_fuse.")
            s_in = conv_act_scale(sd[f"{m}fuse_convs.{nf - 1}.gn.weight"], sd[f"{m}fuse_convs.{nf - 1}.gn.bias"])
            acc = _conv8(torch.cat(rfs, 1), torch.cat([sd[f"{prefix}pconvs.{l}.weight"] for l in range(len(rfs))], 1), s_in,
                         sum(sd[f"{prefix}pconvs.{l}.bias"] for l in range(len(rfs))), padding=1)
        else:
            for lvl, rf in enumerate(rfs):
                y = _conv16(rf, sd[f"{prefix}pconvs.{lvl}.weight"], sd[f"{prefix}pconvs.{lvl}.bias"], padding=1)
                acc = y if acc is None else acc + y
    x = F.relu(acc).flatten(1, -1)
    with _st("region.flat"):
        x = _lin16(x, sd, prefix + "flatten_linear")
    x = x + pe
    with _st("region.up"):
        x = _lin16(x, sd, prefix + "updims")
    return [x[rois[:, 0] == i] for i in range(len(rois_list))]


# ------------------------------------------------------------------------------------------------ LLaMA (a20-a22)
def rope_tables(hd, n, theta=10000.0):
    inv_freq = 1.0 / (theta ** (torch.arange(0, hd, 2).float() / hd))
    freqs = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def llama_forward(sd, cfg, inputs_embeds, attention_mask, past=None, prefix="llm.model.", layer_hook=None):
    """HF 4.32 LlamaModel.forward(inputs_embeds, attention_mask, past_key_values, use_cache=True) as called at
    R: groma/model/groma.py:389-397.  position_ids = arange(past, past+L); additive finfo.min masks; fp32 softmax.
    Returns (hidden [bs,L,D] after the final RMSNorm, new past = list of (k, v) [bs,H,S,hd])."""
    lc = cfg["llm_cfg"]
    D, H, eps = lc["hidden_size"], lc["num_attention_heads"], lc["rms_norm_eps"]
    hd = D // H
    bs, L, _ = inputs_embeds.shape
    past_len = past[0][0].shape[2] if past is not None else 0
    S = past_len + L
    cos, sin = rope_tables(hd, S, lc.get("rope_theta", 10000.0))
    cos, sin = cos[past_len:S][None, None], sin[past_len:S][None, None]
    fmin = torch.finfo(torch.float32).min
    mask = torch.zeros((bs, 1, L, S))
    if L > 1:
        causal = torch.full((L, L), fmin).triu(1)
        mask[:, :, :, past_len:] = causal
    if attention_mask is not None:
        pad = (1.0 - attention_mask[:, None, None, :].to(torch.float32)).bool()
        mask = mask + torch.zeros((bs, 1, L, S)).masked_fill(pad, fmin)

    def rms(x, w):
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + eps))

    h = inputs_embeds
    new_past = []
    for i in range(lc["num_hidden_layers"]):
        p = f"{prefix}layers.{i}."
        x = rms(h, sd[p + "input_layernorm.weight"])
        with _st("llm.qkv"):
            q = _r(_linA(x, sd, p + "self_attn.q_proj", False, src="norm"), "o").view(bs, L, H, hd).transpose(1, 2)
            k = _r(_linA(x, sd, p + "self_attn.k_proj", False, src="norm"), "o").view(bs, L, H, hd).transpose(1, 2)
            v = _r(_linA(x, sd, p + "self_attn.v_proj", False, src="norm"), "o").view(bs, L, H, hd).transpose(1, 2)
            q = _r(q * cos + _rot_half(q) * sin, "o")
            k = _r(k * cos + _rot_half(k) * sin, "o")
        if past is not None:
            k = torch.cat([past[i][0], k], dim=2)
            v = torch.cat([past[i][1], v], dim=2)
        new_past.append((k, v))
        att = q @ k.transpose(2, 3) / math.sqrt(hd) + mask
        att = torch.max(att, torch.tensor(fmin))
        with _st("llm.pv"):
            y = _softmax_pv(att, v).transpose(1, 2).reshape(bs, L, D)
        with _st("llm.o"):
            h = h + _linA(y, sd, p + "self_attn.o_proj", False)
        x = rms(h, sd[p + "post_attention_layernorm.weight"])
        with _st("llm.gateup"):
            x = F.silu(_linA(x, sd, p + "mlp.gate_proj", False, src="norm")) * _linA(x, sd, p + "mlp.up_proj", False, src="norm")
        with _st("llm.down"):
            x = _linA(x, sd, p + "mlp.down_proj", False)
        h = h + x
        if layer_hook is not None:
            layer_hook(i, h)
    return rms(h, sd[prefix + "norm.weight"]), new_past


@_in_stage("embed")
def get_input_embeddings(sd, input_ids):
    """R: groma/model/groma.py:165-174"""
    W0, W1 = _r(sd["llm.model.embed_tokens.weight"], "w"), _r(sd["new_input_embs.weight"], "w")  # bf16 tables on the device
    mask = input_ids >= W0.shape[0]
    ori = F.embedding(input_ids.masked_fill(mask, 0), W0)
    new = F.embedding((input_ids - W0.shape[0]).masked_fill(~mask, 0), W1)
    ori[mask] = new[mask]
    return ori


@_in_stage("head")
def lm_logits(sd, hidden):
    """R: groma/model/groma.py:399-402"""
    if _ROUND[0] == "e4m3":  # a21 in e4m3: `hidden` is the final RMSNorm output, quantised straight from fp32
        return torch.cat((_lin8(hidden, sd["llm.lm_head.weight"]), _lin8(hidden, sd["extra_lm_head.weight"])), dim=-1)
    return torch.cat((_lin16(hidden, sd, "llm.lm_head", False), _lin16(hidden, sd, "extra_lm_head", False)), dim=-1)


@_in_stage("bridge")
def bridge(sd, image_features):
    """img_txt_bridge: Linear -> GELU -> Linear (R: groma/model/groma.py:112-116, applied :361)"""
    return _lin16(F.gelu(_lin16(image_features, sd, "img_txt_bridge.0")), sd, "img_txt_bridge.2")


# ------------------------------------------------------------------------------------------------ glue (groma.py:202-427)
def s2d_image_features(last_hidden):
    """R: groma/model/groma.py:224-237"""
    f = last_hidden[:, 1:]
    bs, l, d = f.shape
    h = w = int(math.sqrt(l))
    f = f.reshape(bs, h, w, d)
    f = torch.cat([f[:, 0::2, 0::2, :], f[:, 1::2, 0::2, :], f[:, 0::2, 1::2, :], f[:, 1::2, 1::2, :]], dim=-1)
    return f.reshape(bs, l // 4, d * 4)


def ddetr_inputs_from_hidden(hidden_states):
    """R: groma/model/groma.py:240-242"""
    x = torch.mean(torch.stack(hidden_states[-4:]), dim=0)[:, 1:]
    bs, l, d = x.shape
    h = w = int(math.sqrt(l))
    return x.reshape(bs, h, w, d).permute(0, 3, 1, 2).contiguous()


def rewrite_box_tokens(input_ids, labels, refer_boxes, ground_boxes, selected_boxes, tok):
    """R: groma/model/groma.py:283-309 (mutates input_ids / labels in place like the reference)."""
    refer_box_inds = []
    box_ids = torch.tensor(tok["box_idx_token_ids"])
    for i in range(input_ids.shape[0]):
        if tok["refer_box_token_id"] in input_ids[i]:
            ious = box_iou(center_to_corners_format(refer_boxes[i]), center_to_corners_format(selected_boxes[i]))
            matched = torch.max(ious, dim=-1).indices
            refer_box_inds.append(matched)
            input_ids[i].masked_scatter_(input_ids[i] == tok["refer_box_token_id"], box_ids[matched])
        else:
            refer_box_inds.append([])
        if tok["ground_box_token_id"] in input_ids[i]:
            ious = box_iou(center_to_corners_format(ground_boxes[i]), center_to_corners_format(selected_boxes[i]))
            matched = torch.max(ious, dim=-1).indices
            mask = input_ids[i] == tok["ground_box_token_id"]
            input_ids[i].masked_scatter_(mask, box_ids[matched])
            if labels is not None:
                labels[i].masked_scatter_(mask, box_ids[matched])
    return refer_box_inds


def splice_placeholders(input_ids, num_image_tokens, num_region_tokens, tok):
    """R: groma/model/groma.py:317-357 -> (new_input_ids [bs,Lmax], attention_mask)"""
    new_ids = []
    for i in range(input_ids.shape[0]):
        ids = input_ids[i]
        assert tok["img_token_id"] in ids and tok["reg_token_id"] in ids
        img_pos = (ids == tok["img_token_id"]).nonzero(as_tuple=True)[0]
        reg_pos = (ids == tok["reg_token_id"]).nonzero(as_tuple=True)[0]
        pad_pos = (ids == tok["pad_token_id"]).nonzero(as_tuple=True)[0]
        pad_pos = pad_pos[0] if len(pad_pos) > 0 else len(ids)
        assert img_pos < reg_pos
        img_ph = torch.full((num_image_tokens,), tok["img_token_id"])
        reg_ph = torch.cat([torch.tensor([tok["box_idx_token_ids"][j], tok["reg_token_id"]])
                            for j in range(num_region_tokens[i])])
        new_ids.append(torch.cat((ids[:img_pos], img_ph, ids[img_pos + 1: reg_pos], reg_ph, ids[reg_pos + 1: pad_pos])))
    out = torch.nn.utils.rnn.pad_sequence(new_ids, batch_first=True, padding_value=tok["pad_token_id"])
    return out, out.ne(tok["pad_token_id"])


def perceive(sd, cfg, images, refer_boxes=None, ground_boxes=None, hidden_states=None):
    """Steps A-E of SURVEY §3.2.  `hidden_states` may be injected (stage-chained parity: feed the device ViT output)."""
    if hidden_states is None:
        hidden_states = vit_forward(sd, cfg, images)
    det = ddetr_forward(sd, cfg, ddetr_inputs_from_hidden(hidden_states))
    scores = fuse_scores(det["logits_coco"], det["logits_sa1b"])
    selected, nms_inds, perms = select_regions(det["pred_boxes"], scores, refer_boxes, ground_boxes, cfg["nms_thres"],
                                               cfg["box_score_thres"], cfg["max_region_num"])
    return dict(hidden_states=hidden_states, det=det, scores=scores, selected_boxes=selected, nms_inds=nms_inds,
                perms=perms)


def groma_forward(sd, cfg, tok, input_ids, images, refer_boxes=None, ground_boxes=None, hidden_states=None,
                  selected_boxes=None):
    """GromaModel.forward prefill (R: groma/model/groma.py:202-427, past_key_values=None) -> dict.
    Call torch.manual_seed(s) first: the path draws torch.randperm (T4)."""
    input_ids = input_ids.clone()
    if selected_boxes is None:
        per = perceive(sd, cfg, images, refer_boxes, ground_boxes, hidden_states)
        hidden_states, selected_boxes = per["hidden_states"], per["selected_boxes"]
    else:
        per = dict(hidden_states=hidden_states)
    bs = input_ids.shape[0]
    rb = refer_boxes if refer_boxes is not None else [torch.empty((0, 4)) for _ in range(bs)]
    gb = ground_boxes if ground_boxes is not None else [torch.empty((0, 4)) for _ in range(bs)]
    refer_box_inds = rewrite_box_tokens(input_ids, None, rb, gb, selected_boxes, tok)
    image_features = s2d_image_features(hidden_states[cfg["perceiver_cfg"].get("vis_output_layer", -1)])
    mlvl = [h[:, 1:] for h in hidden_states[-3:]]
    feats = region_fuse(sd, cfg, mlvl)
    region_features = roi_extract(sd, cfg, feats, selected_boxes)
    refer_region = [rf[ind] for rf, ind in zip(region_features, refer_box_inds)]
    new_ids, attention_mask = splice_placeholders(input_ids, image_features.shape[1], [x.shape[0] for x in region_features],
                                                  tok)
    embeds = get_input_embeddings(sd, new_ids)
    img = bridge(sd, image_features)
    reg = torch.cat(region_features)
    embeds.masked_scatter_((new_ids == tok["img_token_id"])[:, :, None], img)
    embeds.masked_scatter_((new_ids == tok["reg_token_id"])[:, :, None], reg)
    if any(len(r) for r in refer_region):
        embeds.masked_scatter_((new_ids == tok["refer_feat_token_id"])[:, :, None], torch.cat(refer_region))
    hidden, past = llama_forward(sd, cfg, embeds, attention_mask)
    logits = lm_logits(sd, hidden)
    out = dict(logits=logits, past=past, input_ids=new_ids, rewritten_ids=input_ids, attention_mask=attention_mask, inputs_embeds=embeds,
               pred_boxes=selected_boxes, image_features=img, region_features=reg, llm_hidden=hidden)
    out.update({k: v for k, v in per.items() if k not in out})
    return out


def groma_decode_step(sd, cfg, token_ids, past):
    """GromaModel.forward with past_key_values (R: groma/model/groma.py:376-402): all-ones mask over past+1 (T6)."""
    bs = token_ids.shape[0]
    S = past[0][0].shape[2] + 1
    embeds = get_input_embeddings(sd, token_ids.view(bs, 1))
    hidden, past = llama_forward(sd, cfg, embeds, torch.ones((bs, S)), past)
    return lm_logits(sd, hidden), past


def greedy_generate(sd, cfg, tok, input_ids, images, max_new_tokens, eos_token_id=2, **kw):
    """HF 4.32 GenerationMixin.greedy_search over GromaModel (R: groma/eval/eval_rec.py:93-104): next = argmax of the
    LAST position of the (right-padded) expanded sequence; finished rows emit pad.  Returns dict(sequences, pred_boxes)."""
    out = groma_forward(sd, cfg, tok, input_ids, images, **kw)
    seqs = input_ids.clone()
    past = out["past"]
    logits = out["logits"]
    unfinished = torch.ones(input_ids.shape[0], dtype=torch.long)
    margins = []  # per step: top-1 minus top-2 logit of every row (how resolvable the greedy choice is)
    for step in range(max_new_tokens):
        top2 = logits[:, -1, :].topk(2, dim=-1).values
        margins.append(top2[:, 0] - top2[:, 1])
        nxt = torch.argmax(logits[:, -1, :], dim=-1)
        nxt = nxt * unfinished + tok["pad_token_id"] * (1 - unfinished)
        seqs = torch.cat([seqs, nxt[:, None]], dim=-1)
        unfinished = unfinished.mul((nxt != eos_token_id).long())
        if unfinished.max() == 0 or step == max_new_tokens - 1:
            break
        logits, past = groma_decode_step(sd, cfg, nxt, past)
    return dict(sequences=seqs, pred_boxes=out["pred_boxes"], prefill=out, margins=torch.stack(margins, dim=1))
