"""Seeded synthetic weights and inputs (there is no network: no checkpoints, no datasets).

`param_spec(cfg)` enumerates every parameter of the reference model under the reference's own state-dict
names (HF 4.32 naming) with its shape and an init kind; `make_state_dict` materialises it on
the CPU in fp32 (shared by the oracle and the device model in the parity tests); GromaModel.init_synthetic
materialises the same spec directly on the GPU for the 7B benchmark configuration.
Init: N(0,0.02) for Linear/Conv weights, norm gains ~1, LayerScale ~1, N(0,1) level / query
embeddings, MSDA offset bias = the Deformable-DETR ring pattern; everything gets a little noise so no code
path multiplies by an exact 0 or 1."""
import math

import torch


def param_spec(cfg):
    vc, dc, lc = cfg.perceiver_cfg.vis_encoder_cfg, cfg.perceiver_cfg.ddetr_cfg, cfg.llm_cfg
    rc = cfg.region_cfg
    D = vc.hidden_size
    out = []

    def lin(name, n_out, n_in, bias=True, std=0.02):
        out.append((name + ".weight", (n_out, n_in), ("normal", std)))
        if bias:
            out.append((name + ".bias", (n_out,), ("normal", 0.02)))

    def norm(name, n):
        out.append((name + ".weight", (n,), ("gain", 0.1)))
        out.append((name + ".bias", (n,), ("normal", 0.1)))

    # ---- DINOv2 (HF Dinov2Model) ----
    v = "perceiver.vis_encoder."
    npos = (vc.image_size // vc.patch_size) ** 2 + 1
    out.append((v + "embeddings.cls_token", (1, 1, D), ("normal", 0.02)))
    out.append((v + "embeddings.mask_token", (1, D), ("zeros",)))
    out.append((v + "embeddings.position_embeddings", (1, npos, D), ("normal", 0.02)))
    out.append((v + "embeddings.patch_embeddings.projection.weight", (D, 3, vc.patch_size, vc.patch_size), ("normal", 0.02)))
    out.append((v + "embeddings.patch_embeddings.projection.bias", (D,), ("normal", 0.02)))
    for i in range(vc.num_hidden_layers):
        p = f"{v}encoder.layer.{i}."
        norm(p + "norm1", D)
        for n in ("query", "key", "value"):
            lin(p + "attention.attention." + n, D, D)
        lin(p + "attention.output.dense", D, D)
        out.append((p + "layer_scale1.lambda1", (D,), ("gain", 0.1)))
        norm(p + "norm2", D)
        lin(p + "mlp.fc1", D * vc.mlp_ratio, D)
        lin(p + "mlp.fc2", D, D * vc.mlp_ratio)
        out.append((p + "layer_scale2.lambda1", (D,), ("gain", 0.1)))
    norm(v + "layernorm", D)
    # ---- input_proj + DDETR ----
    d = dc.d_model
    out.append(("perceiver.input_proj.0.0.weight", (d, D, 1, 1), ("normal", 0.02)))
    out.append(("perceiver.input_proj.0.0.bias", (d,), ("normal", 0.02)))
    norm("perceiver.input_proj.0.1", d)
    t = "perceiver.ddetr_transformer."

    def msda(p, heads, pts):
        out.append((p + "sampling_offsets.weight", (heads * pts * 2, d), ("normal", 0.02)))
        out.append((p + "sampling_offsets.bias", (heads * pts * 2,), ("msda_grid", heads, pts)))
        lin(p + "attention_weights", heads * pts, d)
        lin(p + "value_proj", d, d, std=0.06)
        lin(p + "output_proj", d, d, std=0.06)

    for i in range(dc.encoder_layers):
        p = f"{t}encoder.layers.{i}."
        msda(p + "self_attn.", dc.encoder_attention_heads, dc.encoder_n_points)
        norm(p + "self_attn_layer_norm", d)
        lin(p + "fc1", dc.encoder_ffn_dim, d, std=0.06)
        lin(p + "fc2", d, dc.encoder_ffn_dim, std=0.04)
        norm(p + "final_layer_norm", d)
    for i in range(dc.decoder_layers):
        p = f"{t}decoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, d, d, std=0.06)
        norm(p + "self_attn_layer_norm", d)
        msda(p + "encoder_attn.", dc.decoder_attention_heads, dc.decoder_n_points)
        norm(p + "encoder_attn_layer_norm", d)
        lin(p + "fc1", dc.decoder_ffn_dim, d, std=0.06)
        lin(p + "fc2", d, dc.decoder_ffn_dim, std=0.04)
        norm(p + "final_layer_norm", d)
    out.append((t + "level_embed", (dc.num_feature_levels, d), ("normal", 1.0)))
    out.append((t + "query_position_embeddings.weight", (dc.num_queries, d), ("normal", 1.0)))
    lin(t + "enc_output", d, d, std=0.06)
    norm(t + "enc_output_norm", d)
    lin(t + "pos_trans", 2 * d, 2 * d, std=0.04)
    norm(t + "pos_trans_norm", 2 * d)
    out.append((t + "class_embed_enc.weight", (dc.num_labels, d), ("normal", 0.2)))
    out.append((t + "class_embed_enc.bias", (dc.num_labels,), ("const", -math.log(99.0))))
    for j in range(dc.decoder_layers):
        for head in ("class_embed_coco", "class_embed_sa1b"):
            out.append((f"{t}{head}.{j}.weight", (dc.num_labels, d), ("normal", 0.2)))
            out.append((f"{t}{head}.{j}.bias", (dc.num_labels,), ("const", -math.log(99.0))))
    for j in range(dc.decoder_layers + 1):
        lin(f"{t}bbox_embed.{j}.layers.0", d, d, std=0.06)
        lin(f"{t}bbox_embed.{j}.layers.1", d, d, std=0.06)
        lin(f"{t}bbox_embed.{j}.layers.2", 4, d, std=0.02)
    # ---- bridge, region encoder, extra vocabulary ----
    T = lc.hidden_size
    lin("img_txt_bridge.0", T, 4 * D)
    lin("img_txt_bridge.2", T, T)
    m = "region_encoder.mlvl_fuse."
    for l in range(rc.num_levels):
        out.append((f"{m}input_conv.{l}.weight", (D, D + 2, 1, 1), ("normal", 0.02)))
        out.append((f"{m}input_conv.{l}.bias", (D,), ("normal", 0.02)))
    for r in range(rc.num_fuse):
        out.append((f"{m}fuse_convs.{r}.conv.weight", (D, D, 3, 3), ("normal", 0.01)))
        norm(f"{m}fuse_convs.{r}.gn", D)
    ra = "region_encoder.roi_align."
    for l in range(rc.num_levels):
        out.append((f"{ra}pconvs.{l}.weight", (D, D, 3, 3), ("normal", 0.01)))
        out.append((f"{ra}pconvs.{l}.bias", (D,), ("normal", 0.02)))
    lin(ra + "pos_embedd.0", rc.pos_hidden, 4, std=0.5)
    norm(ra + "pos_embedd.2", rc.pos_hidden)
    lin(ra + "pos_embedd.3", rc.mid_dim, rc.pos_hidden, std=0.06)
    norm(ra + "pos_embedd.5", rc.mid_dim)
    lin(ra + "updims", T, rc.mid_dim)
    lin(ra + "flatten_linear", rc.mid_dim, D * rc.roi_size ** 2, std=0.01)
    out.append(("extra_lm_head.weight", (cfg.num_new_token, T), ("normal", 0.02)))
    out.append(("new_input_embs.weight", (cfg.num_new_token, T), ("normal", 0.02)))
    # ---- LLaMA ----
    out.append(("llm.model.embed_tokens.weight", (lc.vocab_size, T), ("normal", 0.02)))
    for i in range(lc.num_hidden_layers):
        p = f"llm.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(p + "self_attn." + n, T, T, bias=False)
        lin(p + "mlp.gate_proj", lc.intermediate_size, T, bias=False)
        lin(p + "mlp.up_proj", lc.intermediate_size, T, bias=False)
        lin(p + "mlp.down_proj", T, lc.intermediate_size, bias=False)
        out.append((p + "input_layernorm.weight", (T,), ("gain", 0.1)))
        out.append((p + "post_attention_layernorm.weight", (T,), ("gain", 0.1)))
    out.append(("llm.model.norm.weight", (T,), ("gain", 0.1)))
    out.append(("llm.lm_head.weight", (lc.vocab_size, T), ("normal", 0.02)))
    return out


def _msda_grid(heads, pts):
    thetas = torch.arange(heads, dtype=torch.float32) * (2.0 * math.pi / heads)
    g = torch.stack([thetas.cos(), thetas.sin()], -1)
    g = (g / g.abs().max(-1, keepdim=True)[0]).view(heads, 1, 2).repeat(1, pts, 1)
    for i in range(pts):
        g[:, i, :] *= i + 1
    return g.reshape(-1)


def materialize(shape, kind, gen, device="cpu"):
    k = kind[0]
    if k == "normal":
        return torch.randn(shape, generator=gen, device=device) * kind[1]
    if k == "gain":
        return 1.0 + torch.randn(shape, generator=gen, device=device) * kind[1]
    if k == "zeros":
        return torch.zeros(shape, device=device)
    if k == "const":
        return torch.full(shape, kind[1], device=device) + torch.randn(shape, generator=gen, device=device) * 0.02
    if k == "msda_grid":
        return _msda_grid(kind[1], kind[2]).to(device) + torch.randn(shape, generator=gen, device=device) * 0.02
    raise ValueError(kind)


def make_state_dict(cfg, seed=0, only=None):
    """only=(prefix, ...): keep just those parameters (all are still drawn, from the one seeded stream, so a subset
    equals the same entries of the full dict)."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape, kind in param_spec(cfg):
        t = materialize(shape, kind, gen)
        if only is None or name.startswith(tuple(only)):
            out[name] = t
    return out


def make_inputs(cfg, tok, bs, seed=1234, prompt_len=128, k1=20, k2=6):
    """SURVEY.md §8d synthetic inputs: randn images, 128-token prompt with one <image> and one <region>."""
    g = torch.Generator().manual_seed(seed)
    S = cfg.image_size
    images = torch.randn((bs, 3, S, S), generator=g)
    k3 = prompt_len - 3 - k1 - k2
    ids = []
    for _ in range(bs):
        r = lambda n: torch.randint(3, cfg.llm_cfg.vocab_size, (n,), generator=g)
        ids.append(torch.cat([torch.tensor([1]), r(k1), torch.tensor([tok.img_token_id]), r(k2),
                              torch.tensor([tok.reg_token_id]), r(k3)]))
    return images, torch.stack(ids)
