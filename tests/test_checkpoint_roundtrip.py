"""Checkpoint contract: a directory with config.json + shards holding the REFERENCE's
parameter names loads through GromaModel.from_pretrained and is repacked to the device layouts (CPU tensors here:
packing is load-time plumbing and needs no GPU)."""
import os

import pytest
import torch

from groma_amd import config as gconfig
from groma_amd import synth
from groma_amd.groma import GromaModel


def _tiny():
    cfg = gconfig.groma_tiny(box_score_thres=0.0)
    return cfg, synth.make_state_dict(cfg, 3)


def test_from_pretrained_safetensors_and_bin(tmp_path):
    cfg, sd = _tiny()
    # the reference registers the detection heads twice (decoder.* aliases): a real checkpoint carries both names
    extra = {k.replace("ddetr_transformer.", "ddetr_transformer.decoder."): v for k, v in sd.items()
             if ".bbox_embed." in k or ".class_embed_" in k}
    d1 = tmp_path / "st"
    cfg.save_pretrained(d1)
    from safetensors.torch import save_file
    keys = sorted(sd)
    half = len(keys) // 2
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(d1 / "model-00001-of-00002.safetensors"))
    save_file({**{k: sd[k].contiguous() for k in keys[half:]}, **{k: v.clone() for k, v in extra.items()}},
              str(d1 / "model-00002-of-00002.safetensors"))
    m1 = GromaModel.from_pretrained(str(d1), device="cpu")
    d2 = tmp_path / "bin"
    cfg.save_pretrained(d2)
    torch.save(sd, str(d2 / "pytorch_model.bin"))
    m2 = GromaModel.from_pretrained(str(d2), device="cpu")
    ref = GromaModel.from_state_dict(cfg, sd, device="cpu")
    for m in (m1, m2):
        assert m.config.to_dict() == cfg.to_dict()
        assert torch.equal(m.llm.w["head"], ref.llm.w["head"])
        assert torch.equal(m.llm.w["layers"][1]["wgu"][0], ref.llm.w["layers"][1]["wgu"][0])
        assert torch.equal(m.vit.w["layers"][0]["wqkv"][0], ref.vit.w["layers"][0]["wqkv"][0])
        assert torch.equal(m.region.w["flat_w"], ref.region.w["flat_w"])
        assert torch.equal(m.proposer.w["dec"][0]["qk_w"], ref.proposer.w["dec"][0]["qk_w"])


def test_packed_layouts_match_their_definitions():
    cfg, sd = _tiny()
    m = GromaModel.from_state_dict(cfg, sd, device="cpu")
    lc = cfg.llm_cfg
    # gate/up interleave: row 2i = gate_i, row 2i+1 = up_i
    wgu = m.llm.w["layers"][0]["wgu"][0].float()
    assert torch.equal(wgu[0::2], sd["llm.model.layers.0.mlp.gate_proj.weight"].bfloat16().float())
    assert torch.equal(wgu[1::2], sd["llm.model.layers.0.mlp.up_proj.weight"].bfloat16().float())
    # head = lm_head (+) extra_lm_head, zero padded to a multiple of 128 rows
    head = m.llm.w["head"].float()
    assert head.shape[0] % 128 == 0 and head.shape[0] >= cfg.vocab_size
    assert torch.equal(head[: lc.vocab_size], sd["llm.lm_head.weight"].bfloat16().float())
    assert torch.equal(head[lc.vocab_size: cfg.vocab_size], sd["extra_lm_head.weight"].bfloat16().float())
    assert head[cfg.vocab_size:].abs().max() == 0
    # flatten_linear: (c,h,w) -> (h,w,c)
    D, P2 = cfg.perceiver_cfg.vis_encoder_cfg.hidden_size, cfg.region_cfg.roi_size ** 2
    fw = sd["region_encoder.roi_align.flatten_linear.weight"]
    exp = fw.view(-1, D, P2).permute(0, 2, 1).reshape(-1, P2 * D).bfloat16().float()
    assert torch.equal(m.region.w["flat_w"].float(), exp)
    # 3x3 conv taps: [Cout, (ky,kx,c)]
    w = sd["region_encoder.mlvl_fuse.fuse_convs.0.conv.weight"]
    assert torch.equal(m.region.w["fuse"][0]["w"].float(), w.permute(0, 2, 3, 1).reshape(D, 9 * D).bfloat16().float())
    # DINOv2 position table resized once at load to the 32x32 grid (+CLS), T8
    assert m.vit.w["pos_patch"].shape == (1024, D) and m.vit.w["cls_pos0"].shape == (D,)


def test_unsupported_load_options_raise(tmp_path):
    cfg, _ = _tiny()
    cfg.save_pretrained(tmp_path)
    with pytest.raises(NotImplementedError):
        GromaModel.from_pretrained(str(tmp_path), load_in_8bit=True)
    with pytest.raises(FileNotFoundError):
        GromaModel.from_pretrained(str(tmp_path), device="cpu")
