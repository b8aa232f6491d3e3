"""Cross-check of the oracle's restatement of the un-vendored `transformers==4.32` arithmetic against the transformers
modules installed here (5.15), on the sub-graphs whose math did not change between the two releases
(SURVEY.md §8c): DINOv2 blocks (position table at native size, so T8 is not involved) and LLaMA blocks."""
import pytest
import torch

from oracle import groma_oracle as O

tf = pytest.importorskip("transformers")


def test_dinov2_blocks_match_hf():
    c = tf.Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14,
                        mlp_ratio=4, layerscale_value=1.0)
    try:
        m = tf.Dinov2Model(c).eval()
    except Exception as e:  # pragma: no cover
        pytest.skip(f"HF Dinov2Model unavailable: {e}")
    torch.manual_seed(0)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05 + (1.0 if p.ndim == 1 and p.shape[0] == 128 and p.mean() > 0.5 else 0.0))
    sd = {"perceiver.vis_encoder." + k: v for k, v in m.state_dict().items()}
    cfg = {"perceiver_cfg": {"vis_encoder_cfg": dict(hidden_size=128, num_attention_heads=2, patch_size=14,
                                                     layer_norm_eps=c.layer_norm_eps, num_hidden_layers=3)}}
    x = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        ref = m(x, output_hidden_states=True).hidden_states
        got = O.vit_forward(sd, cfg, x)
    assert len(ref) == len(got) == 4
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, atol=2e-5, rtol=1e-4)


def test_llama_blocks_match_hf():
    lc = tf.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=2, vocab_size=100, rms_norm_eps=1e-5, rope_theta=10000.0,
                        attention_bias=False, mlp_bias=False)
    try:
        lc._attn_implementation = "eager"
        m = tf.LlamaModel(lc).eval()
    except Exception as e:  # pragma: no cover
        pytest.skip(f"HF LlamaModel unavailable: {e}")
    torch.manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05 + (1.0 if p.ndim == 1 else 0.0))
    sd = {"llm.model." + k: v for k, v in m.state_dict().items()}
    cfg = {"llm_cfg": dict(hidden_size=256, num_attention_heads=2, rms_norm_eps=1e-5, num_hidden_layers=2,
                           rope_theta=10000.0)}
    emb = torch.randn(2, 9, 256)
    mask = torch.ones(2, 9, dtype=torch.long)
    mask[1, 6:] = 0  # right padding
    with torch.no_grad():
        ref = m(inputs_embeds=emb, attention_mask=mask).last_hidden_state
        got, past = O.llama_forward(sd, cfg, emb, mask)
    assert torch.allclose(got[0], ref[0], atol=2e-5, rtol=1e-4)
    assert torch.allclose(got[1, :6], ref[1, :6], atol=2e-5, rtol=1e-4)  # valid rows of the padded sequence
    assert past[0][0].shape == (2, 2, 9, 128)
    # incremental decoding == full recomputation (KV cache semantics)
    with torch.no_grad():
        full, _ = O.llama_forward(sd, cfg, emb[:1], torch.ones(1, 9))
        pre, p0 = O.llama_forward(sd, cfg, emb[:1, :8], torch.ones(1, 8))
        inc, _ = O.llama_forward(sd, cfg, emb[:1, 8:], torch.ones(1, 9), p0)
    assert torch.allclose(inc[0, 0], full[0, 8], atol=2e-5, rtol=1e-4)


# ---- Deformable-DETR (a6-a8): the oracle's layers against the transformers-5.15 modules on disk ------------------
# HF renamed parameters between 4.32 (the reference's state-dict names, used by the oracle) and 5.15:
#   fc1 / fc2                    -> mlp.fc1 / mlp.fc2
#   self_attn.out_proj (decoder) -> self_attn.o_proj
# and the layer call arguments (position_embeddings -> spatial_position_embeddings / object_queries_position_embeddings,
# + spatial_shapes_list); the arithmetic is unchanged, which is what these tests establish.
def _ddetr_cfg(**kw):
    base = dict(d_model=64, encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=96, decoder_ffn_dim=96,
                encoder_n_points=4, decoder_n_points=4, num_feature_levels=1, dropout=0.0, attention_dropout=0.0,
                activation_dropout=0.0, activation_function="relu", two_stage=True, with_box_refine=True,
                num_queries=10, two_stage_num_proposals=10, disable_custom_kernels=True)
    base.update(kw)
    c = tf.DeformableDetrConfig(**base)
    c._attn_implementation = "eager"
    return c


def _randomize(mod, seed):
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.2)


def _to_4_32_names(sd, prefix):
    out = {}
    for k, v in sd.items():
        k = k.replace("mlp.fc1", "fc1").replace("mlp.fc2", "fc2").replace("self_attn.o_proj", "self_attn.out_proj")
        out[prefix + k] = v
    return out


def _hf_ddetr():
    try:
        from transformers.models.deformable_detr import modeling_deformable_detr as M
    except Exception as e:  # pragma: no cover
        pytest.skip(f"HF deformable_detr unavailable: {e}")
    return M


def test_ddetr_sine_position_embedding_matches_hf():
    M = _hf_ddetr()
    pe = M.DeformableDetrSinePositionEmbedding(128, normalize=True)
    ref = pe(torch.Size((2, 256, 8, 8)), "cpu", torch.float32, mask=torch.ones((2, 8, 8), dtype=torch.bool))
    got = O.sine_position_embedding(2, 8, 8, 256)
    assert torch.allclose(got, ref, atol=1e-6)
    ref32 = pe(torch.Size((1, 256, 32, 32)), "cpu", torch.float32, mask=torch.ones((1, 32, 32), dtype=torch.bool))
    assert torch.allclose(O.sine_position_embedding(1, 32, 32, 256), ref32, atol=1e-6)


def test_ddetr_msda_and_encoder_layer_match_hf():
    M = _hf_ddetr()
    c = _ddetr_cfg()
    layer = M.DeformableDetrEncoderLayer(c).eval()
    _randomize(layer, 3)
    sd = _to_4_32_names(layer.state_dict(), "L.")
    bs, h, w, d = 2, 6, 6, 64
    torch.manual_seed(4)
    x, pos = torch.randn(bs, h * w, d), torch.randn(bs, h * w, d)
    ref_pts = M.DeformableDetrEncoder.get_reference_points([(h, w)], torch.ones(bs, 1, 2), "cpu")
    assert torch.allclose(O.encoder_reference_points(bs, h, w), ref_pts, atol=1e-7)
    shapes = torch.tensor([[h, w]])
    with torch.no_grad():
        ref = layer(x, attention_mask=None, spatial_position_embeddings=pos, reference_points=ref_pts,
                    spatial_shapes=shapes, spatial_shapes_list=[(h, w)], level_start_index=torch.tensor([0]))
        got = O.ddetr_encoder_layer(sd, "L.", x, pos, ref_pts, [(h, w)], 2, 4)
        # the MSDA module alone (2-d reference points), incl. the 4.32 position-embedding add outside the module
        m_ref, _ = layer.self_attn(hidden_states=x, encoder_hidden_states=x, position_embeddings=pos,
                                   reference_points=ref_pts, spatial_shapes=shapes, spatial_shapes_list=[(h, w)],
                                   level_start_index=torch.tensor([0]))
        m_got = O._msda_module(sd, "L.self_attn.", x + pos, x, ref_pts, [(h, w)], 2, 4)
    assert torch.allclose(m_got, m_ref, atol=2e-5, rtol=1e-4)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4)


def test_ddetr_decoder_layer_matches_hf():
    M = _hf_ddetr()
    c = _ddetr_cfg()
    layer = M.DeformableDetrDecoderLayer(c).eval()
    _randomize(layer, 5)
    sd = _to_4_32_names(layer.state_dict(), "L.")
    bs, h, w, d, nq = 2, 6, 6, 64, 10
    torch.manual_seed(6)
    hs, qpos, memory = torch.randn(bs, nq, d), torch.randn(bs, nq, d), torch.randn(bs, h * w, d)
    ref4 = torch.rand(bs, nq, 1, 4) * 0.5 + 0.25  # 4-d references (cx, cy, w, h): the two-stage / box-refine form
    with torch.no_grad():
        ref = layer(hs, object_queries_position_embeddings=qpos, reference_points=ref4, spatial_shapes=torch.tensor([[h, w]]),
                    spatial_shapes_list=[(h, w)], level_start_index=torch.tensor([0]), encoder_hidden_states=memory,
                    encoder_attention_mask=None)
        got = O.ddetr_decoder_layer(sd, "L.", hs, qpos, memory, ref4, [(h, w)], 2, 4)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4)


def test_ddetr_two_stage_helpers_match_hf():
    """gen_encoder_output_proposals / get_proposal_pos_embed / inverse_sigmoid / the 3-layer box head: the reference's
    copies (R: groma/model/ddetr_transformer.py:383-446) are the HF methods, still present unchanged in 5.15."""
    from types import SimpleNamespace
    M = _hf_ddetr()
    d, h, w, bs = 64, 5, 5, 2
    torch.manual_seed(7)
    enc_output, norm = torch.nn.Linear(d, d), torch.nn.LayerNorm(d)
    head = M.DeformableDetrMLPPredictionHead(d, d, 4, 3)
    for m in (enc_output, norm, head):
        _randomize(m, 8)
    fake = SimpleNamespace(enc_output=enc_output, enc_output_norm=norm, config=SimpleNamespace(d_model=d))
    memory = torch.randn(bs, h * w, d)
    sd = {"T.enc_output.weight": enc_output.weight, "T.enc_output.bias": enc_output.bias,
          "T.enc_output_norm.weight": norm.weight, "T.enc_output_norm.bias": norm.bias}
    sd.update({"T.bbox." + k: v for k, v in head.state_dict().items()})
    with torch.no_grad():
        oq_ref, prop_ref = M.DeformableDetrModel.gen_encoder_output_proposals(
            fake, memory, torch.zeros((bs, h * w), dtype=torch.bool), [(h, w)])
        oq, prop = O.gen_encoder_output_proposals(sd, "T.", memory, h, w)
        assert torch.allclose(oq, oq_ref, atol=2e-5, rtol=1e-4)
        assert torch.equal(torch.isinf(prop), torch.isinf(prop_ref))
        fin = ~torch.isinf(prop)
        assert torch.allclose(prop[fin], prop_ref[fin], atol=1e-6)
        coords = torch.randn(bs, 7, 4)
        assert torch.allclose(O.get_proposal_pos_embed(coords, d // 2),
                              M.DeformableDetrModel.get_proposal_pos_embed(fake, coords), atol=1e-6)
        x = torch.rand(bs, 7, 4)
        assert torch.allclose(O.inverse_sigmoid(x), M.inverse_sigmoid(x), atol=1e-6)
        assert torch.allclose(O._mlp_head(memory, sd, "T.bbox"), head(memory), atol=2e-5, rtol=1e-4)
