"""Shared fixtures for the parity tests: tiny Groma config, seeded reference-named state dict, token table."""
import torch

from groma_amd import config as gconfig
from groma_amd import constants, synth


def tok_dict(model_like):
    return dict(pad_token_id=model_like.pad_token_id, img_token_id=model_like.img_token_id,
                reg_token_id=model_like.reg_token_id, refer_box_token_id=model_like.refer_box_token_id,
                refer_feat_token_id=model_like.refer_feat_token_id, ground_box_token_id=model_like.ground_box_token_id,
                box_idx_token_ids=list(model_like.box_idx_token_ids))


class TokenIds:
    """host-only holder of the special-token ids (what GromaModel.init_special_token_id sets)"""

    def __init__(self):
        tk = constants.SyntheticTokenizer()
        cv = tk.convert_tokens_to_ids
        self.pad_token_id = tk.pad_token_id
        self.img_token_id = cv([constants.DEFAULT_TOKENS['image']])[0]
        self.reg_token_id = cv([constants.DEFAULT_TOKENS['region']])[0]
        self.refer_box_token_id = cv([constants.DEFAULT_TOKENS['rbox']])[0]
        self.refer_feat_token_id = cv([constants.DEFAULT_TOKENS['rfeat']])[0]
        self.ground_box_token_id = cv([constants.DEFAULT_TOKENS['gbox']])[0]
        self.box_idx_token_ids = cv(constants.REGION_IDX_TOKENS)


def tiny_setup(seed=0, **cfg_kw):
    cfg = gconfig.groma_tiny(box_score_thres=0.0, **cfg_kw)
    sd = synth.make_state_dict(cfg, seed)
    return cfg, sd, TokenIds()


def device_model(cfg, sd, dev="cuda"):
    from groma_amd.groma import GromaModel
    m = GromaModel.from_state_dict(cfg, sd, dev)
    m.init_special_token_id(constants.SyntheticTokenizer())
    return m


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
