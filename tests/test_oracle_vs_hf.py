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
