"""the reference's OWN evaluation loops, executed unchanged against groma_amd.GromaModel.
(Named test_00_* so that it runs first: the reference builds DataLoaders with 4 worker processes, and forking them out of a
process that already holds every other test module's models took 220 s instead of 20 s.)

oracle/_ref/eval_rec.pyc and eval_lvis.pyc are R: groma/eval/eval_rec.py and groma/eval/eval_lvis.py byte-compiled where they lie
by oracle/build_ref.py (py_compile; git-ignored build outputs like libmmcv_ref.so -- no reference source is in the repository,
and /root/reference does not exist on the GPU box).  The modules are loaded with the reference's import list satisfied by stubs:

  groma.model.groma.GromaModel      -> groma_amd.groma.GromaModel (the thing under test; from_pretrained hands back the model)
  groma.constants                   -> groma_amd.constants
  groma.data.datasets.*             -> synthetic datasets that build the mmdet-style data_item and call the REFERENCE's
                                       RefCOCOTest.preprocess / LVISTest.preprocess / custom_collate_fn
  torchvision.ops, mmdet bbox transform, lvis.LVISEval, AutoTokenizer.from_pretrained -> minimal stand-ins

Then eval_model(args) of the reference runs as written -- DataLoader with 4 workers, DistributedSampler, model.generate(...,
max_new_tokens=3 / 10, return_dict_in_generate=True, output_hidden_states=True), the <r_k> -> pred_boxes lookup, box_iou,
the three torch.distributed.reduce calls over RCCL -- and what it prints (REC) / collects (LVIS) must equal
groma_amd.evalkit.RecMeter / lvis_results over the same generate() outputs."""
import importlib.machinery
import importlib.util
import os
import re
import sys
import types

import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref")


class _DC:  # mmcv DataContainer look-alike
    def __init__(self, data):
        self.data = data


class _Conv:
    roles = ("USER", "ASSISTANT")

    def get_prompt(self, conversations):
        return " ".join(f"{r}: {m}" for r, m in conversations)


class _Tok:
    """word-hash tokenizer with the Groma special tokens (same id table as groma_amd.constants.SyntheticTokenizer)"""
    model_max_length = 2048

    def __init__(self):
        from groma_amd import constants
        self._t = constants.SyntheticTokenizer()
        self.pad_token_id = self._t.pad_token_id
        self._special = [constants.DEFAULT_TOKENS[k] for k in ("image", "region", "boe", "eoe")]

    def convert_tokens_to_ids(self, toks):
        return self._t.convert_tokens_to_ids(toks)

    def __call__(self, prompt, return_tensors="pt", **kw):
        pat = "(" + "|".join(re.escape(s) for s in self._special) + ")"
        ids = [1]
        for piece in re.split(pat, prompt):
            if piece in self._special:
                ids.append(self._t.convert_tokens_to_ids([piece])[0])
            else:
                ids += [3 + (sum(ord(c) * (i + 7) for i, c in enumerate(w)) % 31000) for w in piece.split()]
        return types.SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.int64))


class _SynthBase:
    N = 5

    def __init__(self, ann_file=None, img_prefix=None, tokenizer=None, test_mode=True, conv_temp="llava"):
        self.tokenizer, self.conv_temp = tokenizer, _Conv()
        self.cat2label = {1000 + i: i for i in range(4)}
        self.CLASSES = ["traffic_light", "dog", "fire_hydrant", "person"]

    def __len__(self):
        return self.N

    def _image(self, i):
        g = torch.Generator().manual_seed(4000 + i)
        return torch.randn((3, 448, 448), generator=g)


class _RefCOCO(_SynthBase):   # stands in for groma.data.datasets.refcoco_rec.RefCOCO
    GT_OVERRIDE = {}   # item -> xyxy pixel box (the test plants ground truth on boxes the model is known to point at)

    def __getitem__(self, i):
        boxes = self.GT_OVERRIDE.get(i, torch.tensor([[40.0 + 10 * i, 60.0, 200.0 + 5 * i, 300.0]]))  # xyxy pixels
        item = dict(img=_DC(self._image(i)), gt_labels=[f"the {self.CLASSES[i % 4]} on the left"], gt_bboxes=_DC(boxes),
                    img_metas=_DC(dict(img_shape=(448, 448, 3))))
        return self.preprocess(item)   # the REFERENCE's RefCOCOTest.preprocess


class _LVISDet(_SynthBase):   # stands in for groma.data.datasets.lvis.LVISDet
    def _parse_ann_info(self, img_info, ann_info):
        return dict(ann_info)

    def __getitem__(self, i):
        item = dict(img=_DC(self._image(i)), gt_labels=_DC(torch.tensor([i % 4])), img_info=dict(id=77 + i),
                    img_metas=_DC(dict(ori_shape=(480 + i, 640, 3))))
        return self.preprocess(item)   # the REFERENCE's LVISTest.preprocess


class _LVISEval:   # lvis.LVISEval: keeps what eval_model collected (its `results` list is a local of the reference function)
    captured = None

    def __init__(self, ann_file, result_file, iou_type):
        _LVISEval.captured = list(sys._getframe(1).f_locals.get("results", []))   # (self is the reference's CustomLVISEval subclass)
        _LVISEval.invalid = sys._getframe(1).f_locals.get("invalid")

    def run(self):
        pass

    def print_results(self):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


@pytest.fixture(scope="module")
def ref_env(dev):
    if not os.path.exists(os.path.join(REF, "eval_rec.pyc")):
        pytest.skip("oracle/_ref/eval_rec.pyc not built (python oracle/build_ref.py /root/reference)")
    from groma_amd import constants, evalkit, synth
    from groma_amd.groma import GromaModel
    import transformers
    cfg, sd, tk = util.tiny_setup(seed=0)
    sd = dict(sd)
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0  # random-init weights never emit <r_k>: boost those rows so grounded answers appear
    sd["extra_lm_head.weight"] = w
    model = GromaModel.from_state_dict(cfg, sd, "cuda")

    class _GromaModel:   # groma.model.groma.GromaModel as the reference calls it: from_pretrained(name).cuda()
        @staticmethod
        def from_pretrained(name, **kw):
            return model

    def bbox_xyxy_to_cxcywh(b):
        x1, y1, x2, y2 = b.unbind(-1)
        return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)

    def normalize_box_coordinates(bbox, img_shape):  # R: groma/data/datasets/det_data.py:8-13
        h, w_ = img_shape[:2]
        return torch.clamp(bbox / torch.tensor([w_, h, w_, h], dtype=bbox.dtype), min=0.0, max=1.0)

    def box_convert(b, in_fmt, out_fmt):
        assert (in_fmt, out_fmt) == ("cxcywh", "xywh")
        return torch.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 2], b[:, 3]], -1)

    tv_ops = _mod("torchvision.ops", box_iou=evalkit.pairwise_iou, box_convert=box_convert)
    stubs = {
        "torchvision": _mod("torchvision", ops=tv_ops), "torchvision.ops": tv_ops,
        "mmdet": _mod("mmdet"), "mmdet.core": _mod("mmdet.core"), "mmdet.core.bbox": _mod("mmdet.core.bbox"),
        "mmdet.core.bbox.transforms": _mod("mmdet.core.bbox.transforms", bbox_xyxy_to_cxcywh=bbox_xyxy_to_cxcywh),
        "lvis": _mod("lvis", LVISEval=_LVISEval),
        "groma": _mod("groma"), "groma.utils": _mod("groma.utils", init_distributed_mode=lambda a: None, disable_torch_init=lambda: None),
        "groma.constants": constants, "groma.model": _mod("groma.model"), "groma.model.groma": _mod("groma.model.groma", GromaModel=_GromaModel),
        "groma.data": _mod("groma.data"), "groma.data.datasets": _mod("groma.data.datasets"),
        "groma.data.datasets.refcoco_rec": _mod("groma.data.datasets.refcoco_rec", RefCOCO=_RefCOCO,
                                                INSTRUCTIONS=["Locate {} in the image.", "Where is {}?"]),
        "groma.data.datasets.det_data": _mod("groma.data.datasets.det_data", normalize_box_coordinates=normalize_box_coordinates),
        "groma.data.datasets.lvis": _mod("groma.data.datasets.lvis", LVISDet=_LVISDet),
        "groma.data.datasets.coco": _mod("groma.data.datasets.coco", INSTRUCTIONS=["Find every {} : {}.", "Detect {} ({})."]),
    }
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    old_from_pretrained = transformers.AutoTokenizer.from_pretrained
    transformers.AutoTokenizer.from_pretrained = staticmethod(lambda *a, **k: _Tok())
    import torch.distributed as dist
    own_pg = not dist.is_initialized()
    if own_pg:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", rank=0, world_size=1)  # the reference reduces CUDA scalars: RCCL
    mods = {}
    for name in ("eval_rec", "eval_lvis"):
        loader = importlib.machinery.SourcelessFileLoader("ref_" + name, os.path.join(REF, name + ".pyc"))
        spec = importlib.util.spec_from_loader("ref_" + name, loader)
        m = importlib.util.module_from_spec(spec)
        sys.modules["ref_" + name] = m  # DataLoader workers pickle the dataset class by module name
        loader.exec_module(m)
        mods[name] = m
    yield model, tk, mods
    if own_pg:
        dist.destroy_process_group()
    transformers.AutoTokenizer.from_pretrained = old_from_pretrained
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_reference_eval_rec_loop_runs_unchanged_and_matches_recmeter(ref_env, capsys):
    from torch.utils.data import DataLoader, DistributedSampler
    from groma_amd import evalkit
    from groma_amd.evalkit import RecMeter
    model, tk, mods = ref_env
    ref = mods["eval_rec"]
    args = types.SimpleNamespace(model_name="synthetic-checkpoint", ann_file="synthetic/refcoco_val.json", img_prefix="", threshold=0.5,
                                 box_score_thres=0.0, batch_size_per_gpu=1, rank=0)

    def evalkit_loop():
        """the same loop through evalkit (identical dataset / sampler / loader construction, so the host RNG is consumed alike)"""
        ds = ref.RefCOCOTest(ann_file=args.ann_file, img_prefix="", tokenizer=_Tok(), test_mode=True, conv_temp="llava")
        dl = DataLoader(ds, batch_size=1, num_workers=4, sampler=DistributedSampler(ds, rank=0, shuffle=False), collate_fn=ref.custom_collate_fn)
        meter, first_boxes = RecMeter(0.5), []
        torch.manual_seed(2024)
        for input_ids, image, bboxes in dl:
            out = model.generate(input_ids.cuda(), images=image.cuda(), use_cache=True, do_sample=False, max_new_tokens=3,
                                 return_dict_in_generate=True, output_hidden_states=True, generation_config=model.generation_config)
            pb = out.hidden_states[0][-1]["pred_boxes"][0]
            meter.update(out.sequences.cpu(), input_ids.shape[1], [pb], [bboxes], model.box_idx_token_ids)
            g = evalkit.grounded_boxes(out.sequences[0, input_ids.shape[1]:].cpu(), pb.float().cpu(), model.box_idx_token_ids)
            first_boxes.append(g[0] if g.shape[0] else None)
        return meter.summary(), first_boxes, len(ds)

    # pass 0: where does the model point?  Plant the ground truth of items 0, 2, 4 on that box (the model never sees GT), so the
    # hit counter and the mean IoU are exercised with non-trivial values
    model.init_special_token_id(_Tok())
    model.config.box_score_thres = 0.0
    _RefCOCO.GT_OVERRIDE = {}
    _, first_boxes, _ = evalkit_loop()
    for i in (0, 2, 4):
        if first_boxes[i] is not None:
            cx, cy, w, h = (first_boxes[i] * 448.0).tolist()
            _RefCOCO.GT_OVERRIDE[i] = torch.tensor([[cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]])
    assert len(_RefCOCO.GT_OVERRIDE) >= 2
    torch.manual_seed(2024)
    ref.eval_model(args)     # R: groma/eval/eval_rec.py:65-131, as written
    printed = capsys.readouterr().out
    got = {k: float(v) for k, v in re.findall(r"(iou@0\.5 accu|m_iou|missing percentage): ([0-9.eE+-]+)", printed)}
    assert set(got) == {"iou@0.5 accu", "m_iou", "missing percentage"}, printed
    assert model.config.box_score_thres == 0.0 and model.box_idx_token_ids is not None   # the script configured the model
    s, _, n_items = evalkit_loop()
    _RefCOCO.GT_OVERRIDE = {}
    print("reference script printed", got, "| RecMeter", s)
    assert s["count"] == n_items == 5
    for k in got:
        assert abs(got[k] - s[k]) < 1e-6, (k, got, s)
    assert got["missing percentage"] < 1.0 and got["iou@0.5 accu"] >= 0.4 and got["m_iou"] > 0.3   # planted hits were found


def test_reference_eval_lvis_loop_runs_unchanged_and_matches_lvis_results(ref_env):
    from torch.utils.data import DataLoader, SequentialSampler
    from groma_amd import evalkit
    model, tk, mods = ref_env
    ref = mods["eval_lvis"]
    args = types.SimpleNamespace(model_name="synthetic-checkpoint", ann_file="synthetic/lvis_ground.json", img_prefix="", box_score_thres=0.0,
                                 batch_size_per_gpu=1, result_file="unused.json")
    _LVISEval.captured = None
    torch.manual_seed(99)
    ref.eval_model(args)     # R: groma/eval/eval_lvis.py:111-172, as written
    got = _LVISEval.captured
    assert got is not None
    ds = ref.LVISTest(ann_file=args.ann_file, img_prefix="", tokenizer=_Tok(), test_mode=True, conv_temp="llava")
    dl = DataLoader(ds, batch_size=1, num_workers=4, sampler=SequentialSampler(ds), collate_fn=ref.custom_collate_fn)
    label2cat = {v: k for k, v in ds.cat2label.items()}
    want = []
    torch.manual_seed(99)
    for input_ids, image, label, img_id, img_shape in dl:
        out = model.generate(input_ids.cuda(), images=image.cuda(), use_cache=True, do_sample=False, max_new_tokens=10,
                             return_dict_in_generate=True, output_hidden_states=True, generation_config=model.generation_config)
        want += evalkit.lvis_results(out.sequences.cpu(), input_ids.shape[1], [out.hidden_states[0][-1]["pred_boxes"][0]], [img_id], [label],
                                     [img_shape], model.box_idx_token_ids, label2cat)
    assert len(got) == len(want) > 0
    for a, b in zip(got, want):
        assert a["image_id"] == b["image_id"] and a["category_id"] == b["category_id"] and a["score"] == b["score"] == 1.0
        assert all(abs(x - y) < 1e-3 for x, y in zip(a["bbox"], b["bbox"]))
