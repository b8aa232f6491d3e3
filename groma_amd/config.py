"""Configuration objects mirroring the reference's nested HF configs
(GromaConfig: groma/model/groma.py:31-83; CustomDDETRConfig: groma/model/ddetr.py:48-95; the HF 4.32
Dinov2Config / DeformableDetrConfig / LlamaConfig fields the path reads -- SURVEY.md §8 hyper-parameter list).
Field names are the reference's so a reference config.json loads unchanged."""
import copy
import json
import os


class _Cfg:
    _defaults = {}

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, copy.deepcopy(v))
        for k, v in kw.items():
            setattr(self, k, v)

    def to_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to_dict() if isinstance(v, _Cfg) else copy.deepcopy(v)
        return out

    def __repr__(self):
        return f"{type(self).__name__}({self.to_dict()})"


class Dinov2Config(_Cfg):  # DINOv2-L defaults
    _defaults = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                     image_size=518, layer_norm_eps=1e-6, layerscale_value=1.0, qkv_bias=True, num_channels=3)


class DeformableDetrConfig(_Cfg):  # scripts/det_pretrain.sh:12-19 + HF defaults
    _defaults = dict(d_model=256, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8,
                     decoder_attention_heads=8, encoder_ffn_dim=1024, decoder_ffn_dim=1024, num_feature_levels=1,
                     encoder_n_points=4, decoder_n_points=4, num_queries=300, two_stage_num_proposals=300,
                     two_stage=True, with_box_refine=True, num_labels=1, activation_function="relu")


class LlamaConfig(_Cfg):  # Vicuna-7B-v1.5
    _defaults = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                     rms_norm_eps=1e-5, vocab_size=32000, rope_theta=10000.0, max_position_embeddings=4096,
                     bos_token_id=1, eos_token_id=2, pad_token_id=0)


class RegionConfig(_Cfg):  # constants hard-coded in groma/model/roi_align.py:196-264
    _defaults = dict(num_fuse=5, gn_groups=64, roi_size=14, pos_hidden=256, mid_dim=1024, num_levels=3)


def _sub(cls, v):
    if v is None:
        return cls()
    if isinstance(v, dict):
        return cls(**v)
    if isinstance(v, cls):
        return v
    raise NotImplementedError(f"unsupported sub-config {type(v)}")


class CustomDDETRConfig(_Cfg):
    model_type = "ddetr"

    def __init__(self, vis_encoder_cfg=None, zs_weight_path=None, vis_output_layer=-1, ddetr_cfg=None, **kw):
        super().__init__(**kw)
        self.vis_encoder_cfg = _sub(Dinov2Config, vis_encoder_cfg)
        self.ddetr_cfg = _sub(DeformableDetrConfig, ddetr_cfg)
        self.zs_weight_path = zs_weight_path
        self.vis_output_layer = vis_output_layer


class GromaConfig(_Cfg):
    model_type = "groma"

    def __init__(self, llm_cfg=None, perceiver_cfg=None, num_new_token=0, nms_thres=0.6, box_score_thres=0.15,
                 max_region_num=100, region_cfg=None, image_size=448, **kw):
        super().__init__(**kw)
        self.perceiver_cfg = _sub(CustomDDETRConfig, perceiver_cfg)
        self.llm_cfg = _sub(LlamaConfig, llm_cfg)
        self.region_cfg = _sub(RegionConfig, region_cfg)
        self.nms_thres = nms_thres
        self.box_score_thres = box_score_thres
        self.max_region_num = max_region_num
        self.num_new_token = num_new_token
        self.image_size = image_size
        self.vocab_size = self.llm_cfg.vocab_size + num_new_token

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_pretrained(cls, path):
        """Reads the reference's config.json.  The reference writes the nested configs with `to_diff_dict()`
        (groma/model/groma.py:72-83, ddetr.py:84-95), i.e. keys equal to the *HF class defaults* are omitted -- so
        an absent key means the HF default (HF_DEFAULTS below), not this repo's Groma-7B default."""
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        pc = d.get("perceiver_cfg") or {}
        llm = _from_hf_dict(LlamaConfig, d.get("llm_cfg"), "llm_cfg")
        vis = _from_hf_dict(Dinov2Config, pc.get("vis_encoder_cfg"), "vis_encoder_cfg")
        det = _from_hf_dict(DeformableDetrConfig, pc.get("ddetr_cfg"), "ddetr_cfg")
        keep = {k: d[k] for k in ("num_new_token", "nms_thres", "box_score_thres", "max_region_num", "image_size")
                if k in d}
        per = dict(vis_encoder_cfg=vis, ddetr_cfg=det, vis_output_layer=pc.get("vis_output_layer", -1),
                   zs_weight_path=pc.get("zs_weight_path"))
        return cls(llm_cfg=llm, perceiver_cfg=per, region_cfg=d.get("region_cfg"), **keep)


# transformers==4.32.0 class defaults of the keys this path reads (what `to_diff_dict()` leaves out of config.json).
HF_DEFAULTS = {
    "llm_cfg": dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                    num_attention_heads=32, rms_norm_eps=1e-6, max_position_embeddings=2048, rope_theta=10000.0,
                    bos_token_id=1, eos_token_id=2, pad_token_id=None),
    "vis_encoder_cfg": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4, patch_size=16,
                            image_size=224, layer_norm_eps=1e-6, layerscale_value=1.0, qkv_bias=True, num_channels=3),
    "ddetr_cfg": dict(d_model=256, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8,
                      decoder_attention_heads=8, encoder_ffn_dim=1024, decoder_ffn_dim=1024, num_feature_levels=4,
                      encoder_n_points=4, decoder_n_points=4, num_queries=300, two_stage_num_proposals=300,
                      two_stage=False, with_box_refine=False, num_labels=2, activation_function="relu"),
}
# keys that change the arithmetic but have no implementation here: refuse instead of silently running another model
_UNSUPPORTED = {
    "llm_cfg": dict(hidden_act="silu", rope_scaling=None, attention_bias=False, pretraining_tp=1),
    "vis_encoder_cfg": dict(use_swiglu_ffn=False, hidden_act="gelu"),
    "ddetr_cfg": dict(position_embedding_type="sine", activation_function="relu"),
}


def _from_hf_dict(cls, sub, which):
    sub = dict(sub or {})
    for k, want in _UNSUPPORTED[which].items():
        if k in sub and sub[k] != want:
            raise NotImplementedError(f"{which}.{k}={sub[k]!r} is not implemented on the MI355X path (only {want!r})")
    if which == "llm_cfg":
        kv = sub.get("num_key_value_heads")
        heads = sub.get("num_attention_heads", HF_DEFAULTS[which]["num_attention_heads"])
        if kv is not None and kv != heads:
            raise NotImplementedError(f"grouped-query attention (num_key_value_heads={kv} != {heads}) is not implemented")
    if which == "ddetr_cfg" and "num_labels" not in sub and isinstance(sub.get("id2label"), dict):
        sub["num_labels"] = len(sub["id2label"])  # how PretrainedConfig serialises num_labels
    out = {}
    for k in cls._defaults:
        out[k] = sub[k] if k in sub else HF_DEFAULTS[which][k]
    return out


# ---- named configurations -------------------------------------------------------------------------------
def groma_7b(**kw):
    """The benchmark architecture: DINOv2-L + DDETR(300) + region encoder + Vicuna-7B, 114 new tokens."""
    return GromaConfig(num_new_token=114, **kw)


def groma_7b_width(vit_layers=3, num_fuse=1, llm_layers=1, **kw):
    """Groma-7B WIDTH (every GEMM / conv / attention shape of the benchmark: D 1024 x 16 heads, C 1024 pyramid,
    27 648-deep per-ROI conv, 200 704-deep flatten_linear, LLaMA 4096 / 11 008 / 32 heads, 32 114-wide head, full-depth
    6+6 DDETR) at reduced DEPTH, so the fp32 CPU oracle finishes in seconds: the full-width parity configuration."""
    return GromaConfig(llm_cfg=dict(num_hidden_layers=llm_layers),
                       perceiver_cfg=dict(vis_encoder_cfg=dict(num_hidden_layers=vit_layers)),
                       region_cfg=dict(num_fuse=num_fuse), num_new_token=114, **kw)


def groma_tiny(ddetr_layers=2, **kw):
    """Structurally identical, small enough for the fp32 CPU oracle to finish in seconds (parity tests)."""
    return GromaConfig(
        llm_cfg=dict(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4,
                     vocab_size=32000),
        perceiver_cfg=dict(vis_encoder_cfg=dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4),
                           ddetr_cfg=dict(encoder_layers=ddetr_layers, decoder_layers=ddetr_layers)),
        region_cfg=dict(num_fuse=2, mid_dim=256),
        num_new_token=114, **kw)
