"""Host-side logic of the drop-in boundary (no GPU): configs, token table, synthetic spec, placeholder splice,
KV-cache view contract, flop model, shard arithmetic."""
import os

import torch

from groma_amd import config as gconfig
from groma_amd import constants, synth
from groma_amd.dist import shard_range
from oracle import groma_oracle as O
from tests import util


def test_token_table_matches_survey():
    ids = constants.derived_token_ids()
    assert ids["[PAD]"] == 32000 and ids["<image>"] == 32008 and ids["<region>"] == 32009
    assert ids["<refer_box>"] == 32010 and ids["<ground_box>"] == 32011 and ids["<refer_feat>"] == 32012
    assert ids["<r0>"] == 32014 and ids["<r99>"] == 32113
    assert len(constants.SyntheticTokenizer()) == 32114


def test_config_roundtrip(tmp_path):
    cfg = gconfig.groma_7b(box_score_thres=0.0)
    assert cfg.vocab_size == 32114 and cfg.llm_cfg.hidden_size == 4096
    assert cfg.perceiver_cfg.ddetr_cfg.two_stage_num_proposals == 300
    cfg.save_pretrained(tmp_path)
    back = gconfig.GromaConfig.from_pretrained(tmp_path)
    assert back.to_dict() == cfg.to_dict()
    cfg.box_score_thres = 0.3  # mutable like the reference (eval_rec.py:71)
    assert cfg.box_score_thres == 0.3


def test_param_spec_uses_reference_names():
    names = {n for n, _, _ in synth.param_spec(gconfig.groma_tiny())}
    for n in ["perceiver.vis_encoder.embeddings.position_embeddings",
              "perceiver.vis_encoder.encoder.layer.0.attention.attention.query.weight",
              "perceiver.input_proj.0.0.weight", "perceiver.input_proj.0.1.bias",
              "perceiver.ddetr_transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
              "perceiver.ddetr_transformer.decoder.layers.1.self_attn.out_proj.weight",
              "perceiver.ddetr_transformer.decoder.layers.0.encoder_attn.value_proj.bias",
              "perceiver.ddetr_transformer.level_embed", "perceiver.ddetr_transformer.query_position_embeddings.weight",
              "perceiver.ddetr_transformer.class_embed_coco.1.weight", "perceiver.ddetr_transformer.bbox_embed.2.layers.2.bias",
              "llm.model.layers.1.mlp.gate_proj.weight", "llm.lm_head.weight", "img_txt_bridge.2.weight",
              "region_encoder.mlvl_fuse.input_conv.2.weight", "region_encoder.mlvl_fuse.fuse_convs.1.gn.weight",
              "region_encoder.roi_align.pconvs.0.bias", "region_encoder.roi_align.pos_embedd.5.weight",
              "region_encoder.roi_align.flatten_linear.weight", "extra_lm_head.weight", "new_input_embs.weight"]:
        assert n in names, n
    import math
    n7 = sum(math.prod(s) for _, s, _ in synth.param_spec(gconfig.groma_7b()))
    assert 7.3e9 < n7 < 7.5e9


def test_splice_matches_oracle_including_padding_and_ragged_regions():
    from groma_amd.groma import GromaModel
    cfg = gconfig.groma_tiny()
    m = GromaModel(cfg)  # no weights needed for host logic
    m.init_special_token_id(constants.SyntheticTokenizer())
    tk = util.TokenIds()
    _, ids = synth.make_inputs(cfg, tk, 3, seed=3, prompt_len=40, k1=5, k2=3)
    ids[1, -6:] = tk.pad_token_id
    n_reg = [100, 1, 37]
    got_ids, got_mask = m._splice(ids, 256, n_reg)
    exp_ids, exp_mask = O.splice_placeholders(ids, 256, n_reg, util.tok_dict(tk))
    assert torch.equal(got_ids, exp_ids) and torch.equal(got_mask, exp_mask)
    assert got_ids.shape[1] == 40 - 2 + 256 + 200
    import pytest
    bad = ids.clone()
    bad[0][bad[0] == tk.img_token_id] = 5
    with pytest.raises(AssertionError):
        m._splice(bad, 256, n_reg)


def test_prepare_inputs_for_generation_contract():
    from groma_amd.groma import GromaModel
    m = GromaModel(gconfig.groma_tiny())
    ids = torch.arange(12).view(2, 6)
    a = m.prepare_inputs_for_generation(ids, images="img", use_cache=True)
    assert a["input_ids"] is ids and a["images"] == "img" and a["past_key_values"] is None
    b = m.prepare_inputs_for_generation(ids, past_key_values=[1], use_cache=True)
    assert b["input_ids"].shape == (2, 1)
    import pytest
    with pytest.raises(RuntimeError, match="no weights"):
        m.forward(input_ids=ids)


def test_kv_cache_legacy_view():
    from groma_amd.engine import KVCache
    c = KVCache(2, 3, 4, 128, 64, "cpu")
    c.seq_len = 10
    assert tuple(c[0][0].shape) == (3, 4, 10, 128) and tuple(c[1][1].shape) == (3, 4, 10, 128)
    assert len(c) == 2 and bool(c)
    c.k[0][:, :, 5] = 1
    c.grow(128)
    assert c.smax == 128 and c.k[0].shape[2] == 128 and c.k[0][0, 0, 5, 0] == 1 and c.vt[0].shape[3] == 128


def test_flop_model_matches_survey_totals():
    import bench
    f = bench.flops_per_image(gconfig.groma_7b(), 100, 128)
    assert f["L"] == 582
    assert abs(f["vit"] / 1e9 - 723.6) < 8 and abs(f["llm"] / 1e9 - 7868.8) < 40
    assert abs(f["region"] / 1e9 - 3226.4) < 40 and abs(f["total"] / 1e9 - 11850) < 120


def test_shard_range_covers_batch():
    for B in (1, 7, 32):
        for W in (1, 2, 4, 8):
            spans = [shard_range(B, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))


def test_slot_table():
    from groma_amd.serving import SlotTable
    t = SlotTable(3)
    a, b, c = t.acquire("a"), t.acquire("b"), t.acquire("c")
    assert sorted([a, b, c]) == [0, 1, 2] and t.acquire("d") is None and t.n_free == 0
    t.release(b)
    assert t.n_free == 1 and t.active() == sorted([a, c])
    assert t.acquire("d") == b and t.owner[b] == "d"
    import pytest as _pt
    t.release(a)
    with _pt.raises(ValueError):
        t.release(a)


def test_prompt_templates_match_reference_goldens():
    """tests/golden/prompt_golden.json was produced by the reference's own conversation.py (make_prompt_golden.py)"""
    import json, os
    from groma_amd import prompt
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prompt_golden.json")))
    assert len(cases) >= 14
    for c in cases:
        if c["dialog"] == "run_groma":
            got = prompt.groma_query_prompt(c["query"], c["template"])
        elif c["template"] == "simple":
            got = prompt.render("simple", c["turns"])
        else:
            got = prompt.render(c["template"], [(r, tuple(m) if isinstance(m, list) else m) for r, m in c["turns"]])
        assert got == c["prompt"], (c["template"], c["dialog"])


def _ref_rec_loop(seqs, P, boxes, gts, tok_ids, thr):
    """the reference's per-image loop (groma/eval/eval_rec.py:103-121), restated literally as the checker"""
    import torch
    m_iou = hits = invalid = 0.0
    for i in range(seqs.shape[0]):
        pred = boxes[i]
        toks = [int(t) for t in seqs[i, P:] if int(t) in tok_ids]
        inds = [tok_ids.index(t) for t in toks]
        inds = [k for k in inds if k < len(pred)]
        if not inds:
            invalid += 1
            continue
        sel = pred[inds]
        def c2c(b):
            return torch.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], -1)
        a, g = c2c(sel), c2c(gts[i])
        iou = torch.zeros((a.shape[0], g.shape[0]))
        for x in range(a.shape[0]):
            for y in range(g.shape[0]):
                iw = max(0.0, float(min(a[x, 2], g[y, 2]) - max(a[x, 0], g[y, 0])))
                ih = max(0.0, float(min(a[x, 3], g[y, 3]) - max(a[x, 1], g[y, 1])))
                inter = iw * ih
                ua = float((a[x, 2] - a[x, 0]) * (a[x, 3] - a[x, 1]) + (g[y, 2] - g[y, 0]) * (g[y, 3] - g[y, 1])) - inter
                iou[x, y] = inter / ua
        best = iou.max(-1).values
        m_iou += float(best[0])
        hits += 1.0 if float(best[0]) > thr else 0.0
    return hits, m_iou, invalid


def test_rec_meter_matches_reference_loop():
    import torch
    from groma_amd import evalkit
    g = torch.Generator().manual_seed(0)
    tok_ids = list(range(32014, 32114))
    bs, P, new = 12, 9, 5
    seqs = torch.randint(3, 31000, (bs, P + new), generator=g)
    boxes, gts = [], []
    for i in range(bs):
        n = int(torch.randint(1, 30, (1,), generator=g))
        b = torch.rand((n, 4), generator=g) * 0.5 + 0.2
        boxes.append(b)
        gts.append(torch.cat([b[:1] + 0.02 * torch.randn((1, 4), generator=g), torch.rand((2, 4), generator=g)]))
        if i % 4 != 3:   # 3 of 4 answers contain <r_k> ids, some out of range for that image
            seqs[i, P + 1] = tok_ids[int(torch.randint(0, 40, (1,), generator=g))]
            seqs[i, P + 3] = tok_ids[0]
    m = evalkit.RecMeter(0.5)
    m.update(seqs[:7], P, boxes[:7], gts[:7], tok_ids)
    m.update(seqs[7:], P, boxes[7:], gts[7:], tok_ids)
    s = m.summary()
    hits, miou, invalid = _ref_rec_loop(seqs, P, boxes, gts, tok_ids, 0.5)
    assert s["count"] == bs and abs(s["iou@0.5 accu"] - hits / bs) < 1e-12 and abs(s["missing percentage"] - invalid / bs) < 1e-12
    assert abs(s["m_iou"] - miou / bs) < 1e-6 and invalid > 0 and hits > 0
    res = evalkit.lvis_results(seqs[:2], P, boxes[:2], [7, 8], [1, 2], [(480, 640), (100, 200)], tok_ids)
    assert all(set(r) == {"image_id", "category_id", "bbox", "score"} for r in res)


def test_gemm_plan_is_a_function_of_the_layer_shape_only():
    """ops.plan_splits: the split-K factor depends on (N, K) and the plan the caller picked -- never on M, so an image's
    arithmetic cannot depend on its batch mates.  "throughput" (default) never splits; "latency" splits exactly the GEMMs that
    leave the chip under-filled for one request (o-proj, down-proj, ViT fc2, bridge)."""
    from groma_amd import ops
    import inspect
    assert list(inspect.signature(ops.plan_splits).parameters) == ["N", "K", "plan"]  # no M anywhere
    llama = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32128, 4096)]  # (N, K): qkv, o, gate-up, down, head
    vit = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]                       # qkv, proj, fc1, fc2
    assert ops._PLAN[0] == "throughput"
    assert all(ops.plan_splits(N, K) == 1 for N, K in llama + vit)
    with ops.gemm_plan("latency"):
        assert [ops.plan_splits(N, K) for N, K in llama] == [1, 3, 1, 3, 1]
        assert [ops.plan_splits(N, K) for N, K in vit] == [1, 1, 1, 4]
        # operand-pair build (the hybrid ViT): a logical k-value is two physical ones and the rule counts physical K-steps (round 6:
        # one image per call 43.4 -> 44.3 img/s under the latency plan) -- still a function of (N, K) and the build only
        with ops.precision("ref"):
            assert ops.SP() == 2 and [ops.plan_splits(N, K) for N, K in vit] == [2, 2, 2, 8]
            assert all(ops.plan_splits(N, K, "throughput") == 1 for N, K in vit)
        assert [ops.plan_splits(N, K) for N, K in vit] == [1, 1, 1, 4]
        for N, K in [(4096, 4096), (1024, 4096), (4096, 11008)]:
            sp = ops.plan_splits(N, K)
            assert 1 < sp <= 8 and K // sp >= 1024                                       # at least 16 K-steps of 64 per split
        with ops.gemm_plan("throughput"):                                                # re-entrant
            assert ops.plan_splits(4096, 4096) == 1
        assert ops.plan_splits(4096, 4096) == 3
    assert ops._PLAN[0] == "throughput" and ops.plan_splits(4096, 4096, "latency") == 3
    import pytest
    with pytest.raises(ValueError):
        ops.gemm_plan("auto")


def test_precision_context_selects_library_and_dtype():
    """ops.precision(): the 16-bit operand type is a (library, torch dtype) pair switched together, re-entrant, restored on exit"""
    import pytest
    from groma_amd import _lib, config, ops
    from groma_amd.groma import GromaModel
    assert ops.H16() == torch.bfloat16 and _lib.load() is _lib.load("bf16")
    with ops.precision("fp16"):
        assert ops.H16() == torch.float16 and _lib.load() is _lib.load("fp16")
        with ops.precision("bf16"):
            assert ops.H16() == torch.bfloat16
        assert ops.H16() == torch.float16
        with pytest.raises(RuntimeError):
            with ops.precision("bf16"):
                raise RuntimeError("x")
        assert ops.H16() == torch.float16
    assert ops.H16() == torch.bfloat16
    with pytest.raises(ValueError):
        ops.precision("fp8")
    with pytest.raises(ValueError):
        GromaModel(config.groma_tiny(), precision="half")
    assert GromaModel(config.groma_tiny(), precision="fp16").precision == "fp16"
    # round 5: per-stage operand types.  "hybrid" = the ViT (the one 16-bit stage in front of the fp32 proposer) on operand pairs,
    # everything behind it on bf16 / fp16 / e4m3; `precision` stays the type serving / the KV cache are built for
    m = GromaModel(config.groma_tiny(), precision="hybrid")
    assert (m.precision, m.vit_precision, m.mode) == ("bf16", "ref", "hybrid")
    m = GromaModel(config.groma_tiny(), precision="hybrid-fp16")
    assert (m.precision, m.vit_precision, m.mode) == ("fp16", "ref", "hybrid-fp16")
    m = GromaModel(config.groma_tiny(), precision="hybrid", fp8=True)      # e4m3 behind a pair-operand ViT is a legal combination
    assert (m.precision, m.vit_precision, m.mode, m.fp8) == ("bf16", "ref", "hybrid+e4m3", True)
    assert GromaModel(config.groma_tiny(), precision="ref").mode == "ref" and GromaModel(config.groma_tiny()).mode == "bf16"
    assert GromaModel(config.groma_tiny(), precision="bf16", vit_precision="fp16").mode == "bf16+vit:fp16"
    with pytest.raises(ValueError):
        GromaModel(config.groma_tiny(), precision="ref", fp8=True)
    with pytest.raises(ValueError):
        GromaModel(config.groma_tiny(), precision="bf16", vit_precision="fp8")
    # round 6: the table covers every stage that exchanges fp32 tensors with its neighbours ("<base>+<stage>:<type>...")
    from groma_amd.groma import parse_precision, STAGES
    t, base = parse_precision("hybrid-fp16+attn:ref+head:ref")
    assert base == "fp16" and t == dict(vit="ref", region="fp16", bridge="fp16", attn="ref", mlp="fp16", head="ref") and tuple(t) == STAGES
    assert parse_precision("hybrid")[0] == dict(vit="ref", region="bf16", bridge="bf16", attn="bf16", mlp="bf16", head="bf16")
    assert parse_precision(dict(base="fp16", vit="ref", mlp="ref")) == (dict(vit="ref", region="fp16", bridge="fp16", attn="fp16", mlp="ref", head="fp16"), "fp16")
    m = GromaModel(config.groma_tiny(), precision="hybrid-fp16+mlp:ref")
    assert (m.precision, m.vit_precision, m.mode, m.stage_precision["mlp"]) == ("fp16", "ref", "hybrid-fp16+mlp:ref", "ref")
    assert GromaModel(config.groma_tiny(), precision="bf16+region:fp16").mode == "bf16+region:fp16"
    for bad in ("hybrid+llm:ref", "hybrid+attn:fp8", "hybrid+attn"):
        with pytest.raises(ValueError):
            GromaModel(config.groma_tiny(), precision=bad)
    with pytest.raises(ValueError):
        GromaModel(config.groma_tiny(), precision="hybrid+mlp:ref", fp8=True)      # pairs behind the ViT and e4m3 are exclusive
    with pytest.raises(ValueError):
        GromaModel(config.groma_tiny(), precision="hybrid+mlp:fp16", fp8=True)     # e4m3 takes one 16-bit type behind the ViT


def test_graph_pool_policy(monkeypatch):
    """engine.GraphPool without a GPU (the graph objects are faked): eager on the first two sightings of a key, capture + replay on
    the third, replay afterwards; the pool keeps `cap` graphs and an evicted key starts counting again; disabled / traced runs
    launch eagerly and leave the pool alone."""
    import contextlib
    import pytest
    import pytest
    from groma_amd import engine, ops

    log = []

    class FakeGraph:
        def __init__(self):
            self.fn = None

        def capture_begin(self, capture_error_mode="global"):
            assert capture_error_mode == "thread_local"   # a capture must not police other threads' streams
            log.append("capture")

        def capture_end(self):
            pass

        def replay(self):
            log.append("replay")
            if self.fn is not None:
                self.fn()

    class FakeStream:
        device = "fake"

        def __init__(self, device=None):
            pass

        def synchronize(self):
            log_sync.append("stream")

        def wait_stream(self, other):
            pass

    log_sync, reserved = [], [0]
    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: reserved[0])
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: log_sync.append("device"))
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    pool = engine.GraphPool(cap=2)
    runs = []

    def mk(tag):
        def launch():
            runs.append(tag)
        return launch

    # the fake "graph" records nothing, so give replay something to call: wrap run() the way the real capture would
    real_run = pool.run

    def run(key, launch):
        real_run(key, launch)
        g = pool._graphs.get(key)
        if g is not None and g.fn is None:
            g.fn = launch
    for i in range(5):
        run("a", mk("a"))
    # sightings 1, 2: eager; 3: captured (launch runs once under the fake capture) then replayed; 4, 5: replayed
    assert log == ["capture", "replay", "replay", "replay"] and pool.captures == 1 and pool.replays == 2
    assert log_sync == ["stream"]                  # a capture waits for the caller's stream only, never for the whole device
    assert runs.count("a") == 2 + 1 + 2          # (the real capture records instead of running; the first replay of the fake is empty)
    for k in ("b", "c"):
        for _ in range(3):
            run(k, mk(k))
    assert pool.captures == 3 and list(pool._graphs) == ["b", "c"]      # "a" was evicted (cap = 2, least recently used)
    log.clear()
    run("a", mk("a"))
    assert log == [] and "a" not in pool._graphs                         # counts from one again
    run("b", mk("b"))
    assert log == ["replay"]
    # disabled / traced: eager, no bookkeeping
    monkeypatch.setattr(engine.GraphPool, "enabled", False)
    n = len(runs)
    run("b", mk("b"))
    assert len(runs) == n + 1 and log == ["replay"]
    monkeypatch.setattr(engine.GraphPool, "enabled", True)
    monkeypatch.setattr(engine, "TRACE", {})
    run("b", mk("b"))
    assert log == ["replay"]
    monkeypatch.setattr(engine, "TRACE", None)
    # byte budget: what a capture reserved is measured around it; the least recently used graphs go when the total exceeds it
    small = engine.GraphPool(cap=8, byte_budget=100)
    for k, grow in (("x", 40), ("y", 40), ("z", 40)):
        for i in range(3):
            if i == 2:
                orig = reserved[0]
                monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, _c=[0], _o=orig, _g=grow, **kw: (_c.__setitem__(0, _c[0] + 1), _o + (_g if _c[0] > 1 else 0))[1])
            small.run(k, mk(k))
        reserved[0] += grow
        monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a, **k: reserved[0])
    assert list(small._graphs) == ["y", "z"] and sum(small._bytes.values()) == 80
    # warm(): capture at admission time, not on the third live request
    small.warm("w", mk("w"))
    assert "w" in small._graphs and small.captures == 4
    # first_sight(): a warm-up pass captures a new key at once (serving.ContinuousBatcher.warm_admission)
    fs = engine.GraphPool(cap=4)
    with engine.GraphPool.first_sight():
        fs.run("k", mk("k"))
    assert "k" in fs._graphs and fs.captures == 1 and not engine.GraphPool._first_sight[0]
    fs.run("k2", mk("k2"))
    assert "k2" not in fs._graphs                                         # outside the block: third sighting again
    # eager() (round 6): the settling pass of a warm-up launches eagerly and leaves no trace -- not even a sighting -- so the
    # first_sight() pass after it captures on the settled addresses and no never-replayable graph sits in the LRU pool
    n_runs = len(runs)
    with engine.GraphPool.eager():
        with engine.GraphPool.first_sight():                              # (eager wins over first_sight)
            fs.run("settle", mk("settle"))
        fs.run("settle", mk("settle"))
    assert len(runs) == n_runs + 2 and "settle" not in fs._graphs and "settle" not in fs._seen and not engine.GraphPool._eager[0]
    # drop(): graphs (and sighting counts) whose key bakes in memory that no longer exists are forgotten (serving._grow)
    fs.run("k2", mk("k2"))                                                # second sighting of k2: counted, not captured
    assert "k2" in fs._seen
    fs.drop(lambda key: key in ("k", "k2"))
    assert "k" not in fs._graphs and "k2" not in fs._seen and "k" not in fs._bytes
    # a launch that raises inside the capture: the capture is ended, the ORIGINAL error surfaces, nothing is kept
    def boom():
        raise RuntimeError("GR_EINVAL inside the capture")
    n_cap = small.captures
    with pytest.raises(RuntimeError, match="GR_EINVAL inside the capture"):
        small.warm("bad", boom)
    assert "bad" not in small._graphs and small.captures == n_cap
    small.warm("good", mk("good"))
    assert "good" in small._graphs
    # a key is everything baked into the launches: cache_addresses() changes when a cache grows
    class C:
        pass
    c = C()
    c.smax, c.k, c.vt = 64, [torch.zeros(4)], [torch.zeros(4)]
    a0 = engine.cache_addresses(c)
    assert engine.cache_addresses(c) is a0
    c._addr, c.smax = None, 128
    assert engine.cache_addresses(c) != a0


# ---------------------------------------------------------------------------------------- round 4: host logic of the new pieces
def test_split_pack_layout_matches_the_kernel_side_index_map():
    """ops.split_pack = the storage of libgroma_hip_ref.so (csrc/gr_common.h: sp_idx): logical element i sits at physical
    (i / 32) * 64 + i % 32 (hi) and 32 further (lo); hi = f16(x), lo = f16(x - hi); unsplit is exact in fp32"""
    from groma_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 96, generator=g) * 11
    s = ops.split_pack(x)
    assert s.dtype == torch.float16 and tuple(s.shape) == (3, 5, 192)
    flat, sf = x.reshape(-1), s.reshape(-1)
    for i in (0, 31, 32, 45, 95, 96, 96 * 7 + 33, flat.numel() - 1):     # the map is a function of the FLAT index (rows are multiples of 32)
        p = (i // 32) * 64 + i % 32
        hi = flat[i].to(torch.float16)
        assert sf[p] == hi and sf[p + 32] == (flat[i] - hi.float()).to(torch.float16)
    back = ops.unsplit(s)
    assert (back - x).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()
    assert torch.equal(ops.unsplit(ops.split_pack(back)), back)           # a pair is a fixed point
    import pytest
    with pytest.raises(ValueError):
        ops.split_pack(torch.zeros(4, 40))                                  # rows must be whole 32-element blocks
    big = torch.tensor([[1e6] + [0.0] * 31])
    assert torch.isfinite(ops.split_pack(big)).all()                        # saturates like the device-side conversions, never inf
    with ops.precision("fp16"):
        assert torch.isfinite(ops.to_h16(big)).all() and float(ops.to_h16(big)[0, 0]) == 65504.0
    with ops.precision("ref"):
        assert ops.SP() == 2 and ops.H16() == torch.float16 and tuple(ops.to_h16(x).shape) == (3, 5, 192)
    assert ops.SP() == 1 and ops.H16() == torch.bfloat16


def test_precision_and_plan_switches_are_per_thread():
    """two models of different precision / plan served from two threads must not see each other's switch"""
    import threading
    from groma_amd import ops
    seen, go, done = {}, threading.Event(), threading.Event()

    def worker():
        with ops.precision("ref"), ops.gemm_plan("latency"):
            seen["inside"] = (ops.SP(), ops.plan_splits(4096, 4096))
            go.set()
            done.wait(5)
        seen["after"] = (ops.SP(), ops.plan_splits(4096, 4096))
    t = threading.Thread(target=worker)
    t.start()
    assert go.wait(5)
    assert ops.SP() == 1 and ops.plan_splits(4096, 4096) == 1       # this thread still sees the defaults while the other is inside
    done.set()
    t.join()
    assert seen["inside"] == (2, 3) and seen["after"] == (1, 1)


def test_plan_workspace_sizes():
    from groma_amd import ops
    shapes = [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)]
    assert ops.plan_ws_elems(582, shapes) == 0                                        # throughput plan: nothing is split
    need = ops.plan_ws_elems(582, shapes, "latency")
    assert need == max(ops.plan_splits(N, K, "latency") * 582 * N for N, K in shapes if ops.plan_splits(N, K, "latency") > 1) > 0


def test_gemv_desc_mirrors_the_header(tmp_path):
    """ctypes GemvDesc / GemmDesc == gr_gemv_desc / gr_gemm_desc of include/groma_hip.h as gcc lays them out"""
    import ctypes, os, subprocess
    from groma_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", sizeof(gr_gemv_desc), '
                   'offsetof(gr_gemv_desc, epi), offsetof(gr_gemv_desc, q), offsetof(gr_gemv_desc, pos_dev), sizeof(gr_gemm_desc), '
                   'offsetof(gr_gemm_desc, a_parts));return 0;}\n' % os.path.join(root, "include", "groma_hip.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    D, G = _lib.GemvDesc, _lib.GemmDesc
    assert got == [ctypes.sizeof(D), D.epi.offset, D.q.offset, D.pos_dev.offset, ctypes.sizeof(G), G.a_parts.offset]


def test_bench_pins_host_threads_per_rank():
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    before_aff, before_thr = os.sched_getaffinity(0), torch.get_num_threads()
    try:
        assert b.pin_host_threads(0, 1) == before_thr and os.sched_getaffinity(0) == before_aff     # N = 1: untouched
        n = b.pin_host_threads(1, 2)
        mine = os.sched_getaffinity(0)
        cores = sorted(before_aff)
        assert mine == set(cores[len(cores) // 2: 2 * (len(cores) // 2)]) and n == min(len(mine), 16)
    finally:
        os.sched_setaffinity(0, before_aff)
        torch.set_num_threads(before_thr)


def test_e4m3_conv_scale_and_oracle_conv_agree_with_the_packer():
    """fp8 mode's region convs: the static activation scale is a formula of the GroupNorm's affine parameters that the product
    (groma_amd/weights.py) and the oracle (oracle/groma_oracle.py) each state for themselves -- they must be the same number -- and
    the oracle's e4m3 conv is what it says: fp32 accumulation of exact e4m3 products of clamped, statically scaled activations and
    per-output-channel scaled weights"""
    import torch.nn.functional as F
    from groma_amd import weights
    g = torch.Generator().manual_seed(3)
    gam, bet = torch.randn(64, generator=g), torch.randn(64, generator=g) * 0.1
    s = O.conv_act_scale(gam, bet)
    assert s == weights.conv_act_scale(gam, bet) and weights.CONV_ACT_SIGMAS == O.CONV_ACT_SIGMAS
    assert abs(s * 448.0 - (64.0 * float(gam.abs().max()) + float(bet.abs().max()))) < 1e-4
    x = torch.relu(torch.randn((2, 16, 9, 9), generator=g)) * 3.0
    x[0, 0, 0, 0] = 1e6                               # beyond the bound: saturates at 448 * s instead of becoming NaN
    w = torch.randn((8, 16, 3, 3), generator=g) * 0.05
    y = O._conv8(x, w, s, padding=1)
    assert torch.isfinite(y).all()
    xq = (x / s).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    sw = w.flatten(1).abs().amax(1) / 448.0
    wq = (w / sw[:, None, None, None]).to(torch.float8_e4m3fn).float()
    ref = F.conv2d(xq.double(), wq.double(), padding=1).float() * (sw * s)[None, :, None, None]
    assert util.relerr(y, ref) < 1e-6
    assert float(xq.max()) == 448.0                   # the clamp, not a NaN code
    # and the format's own distance on this input stays at the e4m3 level (3 mantissa bits on both operands)
    assert util.relerr(y, F.conv2d(x.clamp(max=448 * s), w, padding=1)) < 8e-2


def test_host_select_draws_like_the_reference_loop():
    """GromaModel._host_select (round 5: the general-path shuffle of propose(), also what `bench.py --host-glue` times): the same
    torch.randperm draws in image order as the reference's per-image loop (R: groma/model/groma.py:273-276, T4), the same global RNG
    state afterwards, and a flat gather list that addresses [image * nmax + index]"""
    from groma_amd.groma import GromaModel
    g = torch.Generator().manual_seed(3)
    keep = torch.stack([torch.randperm(300, generator=g)[:100] for _ in range(5)])
    n_keep = [100, 37, 100, 1, 64]
    torch.manual_seed(11)
    want = [keep[i].index_select(0, torch.randperm(n)) for i, n in enumerate(n_keep)]
    state = torch.get_rng_state()
    torch.manual_seed(11)
    sel, flat, img_of = GromaModel._host_select(keep, n_keep, 300)
    assert torch.equal(torch.get_rng_state(), state)
    assert all(torch.equal(a, b) for a, b in zip(sel, want))
    assert torch.equal(flat, torch.cat([w + 300 * i for i, w in enumerate(want)]))
    assert img_of.tolist() == sum(([i] * n for i, n in enumerate(n_keep)), [])
    # malformed prompts fail like the reference's tensor comparison does (RuntimeError, not a silent first-match)
    import pytest
    from groma_amd import config, constants
    m = GromaModel(config.groma_tiny())
    m.init_special_token_id(constants.SyntheticTokenizer())
    ids = torch.full((1, 12), 7, dtype=torch.int64)
    ids[0, 2], ids[0, 4], ids[0, 6] = m.img_token_id, m.img_token_id, m.reg_token_id
    with pytest.raises(RuntimeError):
        m._splice(ids, 4, [2])


def test_bench_extras_summary_shape():
    """bench.extras_summary: {line: [images/s, roofline fraction | None]} + the decode-step entries, robust to lines that failed"""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ex = {"forward_4_images_per_call": {"value": 75.2512, "roofline": {"frac": 0.49881}},
          "forward_1_image_per_call": {"value": 37.2},
          "generate_4_images_per_call": {"value": 24.2, "roofline": {"frac": 0.4986}, "decode_step": {"ms_per_token": 3.6109, "frac_of_8TBps": 0.4571}},
          "generate_fp8_4_images_per_call": {"value": 29.5, "roofline": {"frac": 0.294}, "decode_step": {"ms_per_token": 3.0582, "frac_of_8TBps": 0.2699}},
          "forward_ref": {"error": "RuntimeError: out of memory"}}
    s = bench.extras_summary(ex)
    assert s["forward_4_images_per_call"] == [75.25, 0.499] and s["forward_1_image_per_call"] == [37.2, None]
    assert s["decode_ms_per_token"] == [3.611, 0.457] and s["decode_fp8_ms_per_token"] == [3.058, 0.27]
    assert "forward_ref" not in s and bench.extras_summary({"error": "x"}) == {} and bench.extras_summary(None) == {}


def test_batcher_arena_growth_keeps_live_prefixes():
    """serving.ContinuousBatcher(grow_to=...): the context bound of a request and the re-allocation of the KV arena (host logic; the
    decode-step re-capture and the tokens are covered on the GPU: tests/test_serving_gpu.py::test_arena_grows_under_a_live_row)."""
    from types import SimpleNamespace
    import pytest
    from groma_amd import engine, ops
    from groma_amd.serving import ContinuousBatcher, Request

    class LLM:
        H, hd, T, w = 2, 32, 16, {"layers": [None, None]}
        fail_after = None       # new_cache raises once this many caches have been handed out (an allocation failure)

        def __init__(self):
            self.graphs, self.made = engine.GraphPool(), 0

        def new_cache(self, bs, smax, device):
            if self.fail_after is not None and self.made >= self.fail_after:
                raise RuntimeError("HIP out of memory (simulated)")
            self.made += 1
            return engine.KVCache(2, bs, self.H, self.hd, smax, device)
    for prec, sp in (("bf16", 1), ("ref", 2)):
        model = SimpleNamespace(device=torch.device("cpu"), fp8=False, precision=prec, llm=LLM(), vit=SimpleNamespace(G=8),
                                config=SimpleNamespace(max_region_num=5))
        b = ContinuousBatcher(model, max_rows=2, max_len=64, use_graph=False, grow_to=256)
        r = Request(0, torch.zeros(20, dtype=torch.int64), None, max_new_tokens=30, refer_boxes=torch.zeros(2, 4))
        assert b._bound(r) == 20 + 16 + 2 * (5 + 2) + 30
        old_k = [t.copy_(torch.randn(t.shape)).clone() for t in b.arena.k]
        old_v = [t.copy_(torch.randn(t.shape)).clone() for t in b.arena.vt]
        assert b.arena.k[0].shape[-1] == 32 * sp and b.arena.vt[0].shape[-1] == 64 * sp
        with ops.precision(prec):
            b._grow(100)                                   # max(2 x 64, 128) = 128
        assert (b.max_len, b.arena.smax, b.staging.smax) == (128, 128, 128)
        for l in range(2):
            assert torch.equal(b.arena.k[l][:, :, :64], old_k[l]) and torch.equal(b.arena.vt[l][..., : 64 * sp], old_v[l])
            if sp == 2:   # the VALUES the pairs stand for: the first 64 positions of V^T survive the move
                assert torch.equal(ops.unsplit(b.arena.vt[l])[..., :64], ops.unsplit(old_v[l]))
        with ops.precision(prec):
            b._grow(10_000)                                # capped by grow_to
            assert b.max_len == 256
            b._grow(10_000)                                # already there: nothing happens
        assert b.max_len == 256 and torch.equal(b.arena.k[1][:, :, :64], old_k[1])
    with pytest.raises(ValueError):
        ContinuousBatcher(model, max_rows=2, max_len=64, grow_to=100)
    # round 6: a growth that cannot be allocated leaves the batcher exactly as it was -- whichever of the two new caches
    # fails -- and the size is not retried for every over-long request of the queue; prefill graphs keyed on the old staging cache
    # are dropped from the LRU pool after a successful growth
    for fail_at in (2, 3):     # (the batcher's own two caches are #0 and #1: fail on the new staging cache, or on the new arena)
        model = SimpleNamespace(device=torch.device("cpu"), fp8=False, precision="bf16", llm=LLM(), vit=SimpleNamespace(G=8),
                                config=SimpleNamespace(max_region_num=5))
        b = ContinuousBatcher(model, max_rows=2, max_len=64, use_graph=False, grow_to=256)
        model.llm.fail_after = fail_at
        arena, staging = b.arena, b.staging
        snap = [t.copy_(torch.randn(t.shape)).clone() for t in b.arena.k]
        assert b._grow(100) is False
        assert b.arena is arena and b.staging is staging and (b.max_len, b.arena.smax, b.staging.smax) == (64, 64, 64)
        assert all(torch.equal(a, c) for a, c in zip(b.arena.k, snap)) and "simulated" in b.last_grow_error
        made = model.llm.made
        assert b._grow(100) is False and model.llm.made == made          # the failed size is not attempted again
        model.llm.fail_after = None
        assert b._grow(90) is False                                       # (same target size 128: still refused without an attempt)
    model.llm.fail_after = None
    del b._grow_failed_at
    stale_key = ("llm", 2, 70, 0, True, "throughput", False, 1) + engine.cache_addresses(b.staging)
    live_key = ("llm", 2, 70, 0, True, "throughput", False, 1, 64, 12345678)
    b.llm.graphs._graphs[stale_key], b.llm.graphs._graphs[live_key] = object(), object()
    assert b._grow(100) is True and b.max_len == 128
    assert stale_key not in b.llm.graphs._graphs and live_key in b.llm.graphs._graphs


def test_valid_ranking_rule_of_the_index_survival_scan():
    """tests/diag/index_survival.py::valid_ranking -- the property asserted on every unselected seed: the device's top-k order is a
    ranking of the ORACLE's values up to a tolerance (near-ties may swap, nothing else)"""
    import importlib.util, os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("index_survival", os.path.join(here, "diag", "index_survival.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    v = torch.tensor([5.0, 4.0, 3.9999, 3.0, 1.0, 0.5])
    ok = mod.valid_ranking
    assert ok(torch.tensor([0, 1, 2, 3]), v, 1e-3)
    assert ok(torch.tensor([0, 2, 1, 3]), v, 1e-3)             # the near-tie swapped
    assert not ok(torch.tensor([0, 2, 1, 3]), v, 1e-5)         # ... which a tighter tolerance rejects
    assert not ok(torch.tensor([0, 1, 3, 2]), v, 1e-3)         # a real inversion
    assert not ok(torch.tensor([0, 1, 2, 4]), v, 1e-3)         # a better candidate was left out
    assert ok(torch.tensor([0, 1, 3]), v, 1e-3) is False       # (2 left out although within the top 3)
    assert ok(torch.tensor([0, 2, 1]), v, 1e-3)
    assert ok(torch.arange(6), v, 0.0)
