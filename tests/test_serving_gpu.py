"""Continuous-batch decode loop + KV slot manager on the tiny architecture.

Properties pinned (all bit-exact on token ids):
  * max_rows = 1 batcher == GromaModel.generate(batch 1): the scheduler reproduces HF greedy_search semantics
    (first token from the prefill, EOS / max_new_tokens stopping) -- same kernels at the same row block;
  * a request's tokens do not depend on its neighbours nor on when it was admitted (rows are independent in every
    kernel of the step), including slot reuse after a request finishes;
  * graph-replayed and eager steps agree."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(dev):
    from groma_amd import synth
    cfg, sd, tk = util.tiny_setup(seed=0)
    model = util.device_model(cfg, sd)
    model.generation_config.eos_token_id = None
    reqs = []
    for i in range(5):
        images, ids = synth.make_inputs(cfg, tk, bs=1, seed=100 + i)
        reqs.append((ids[0], images[0], 6 + 3 * i, 500 + i))
    return cfg, model, reqs


def _solo_generate(model, ids, image, n, seed):
    torch.manual_seed(seed)
    out = model.generate(ids[None].clone(), images=image[None], max_new_tokens=n)
    return out[0, ids.shape[0]:].tolist()


def test_batcher_rows_match_oracle_greedy(setup):
    """ContinuousBatcher against the ORACLE (not against generate()): each request's tokens must equal HF greedy over the
    CPU restatement run at batch 1 on that request (R: groma/serve/model_worker.py:287-338), under the margin rule of
    tests/util.assert_greedy_tokens_match.  Requests are served two at a time, so rows share decode steps."""
    from groma_amd.serving import ContinuousBatcher
    from groma_amd import synth
    from oracle import groma_oracle as O
    cfg, model, reqs = setup
    cfg0, sd, tk = util.tiny_setup(seed=0)  # the same seeded state dict the device model was built from
    b = ContinuousBatcher(model, max_rows=2, max_len=1024)
    total = 0
    for ids, image, n, seed in reqs[:3]:
        n = min(n, 8)
        other = b.submit(*reqs[4][:2], max_new_tokens=20, seed=reqs[4][3])  # a neighbour decoding alongside
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        b.run_until_done()
        r = b.result(rid)
        assert r.error is None and r.done and len(r.tokens) == n
        # the oracle consumes the device ViT states of THIS request (stage chaining, as in test_parity_gpu)
        torch.manual_seed(seed)
        dev_h = tuple(h.cpu() for h in model.vit.forward(image[None].cuda().float()))
        torch.manual_seed(seed)
        ref = O.greedy_generate(sd, cfg0.to_dict(), util.tok_dict(tk), ids[None].clone(), image[None], n, eos_token_id=-1,
                                hidden_states=dev_h)
        assert torch.allclose(r.pred_boxes.cpu(), ref["pred_boxes"][0], atol=1e-5)
        P = ids.shape[0]
        total += util.assert_greedy_tokens_match(torch.tensor([r.tokens]), ref["sequences"][:, P:], ref["margins"], 0.05,
                                                 f"serving rid {rid}")
        b.result(other)
    assert total >= 8


def test_single_row_batcher_equals_generate(setup):
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    b = ContinuousBatcher(model, max_rows=1, max_len=1024)
    for ids, image, n, seed in reqs[:3]:
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        b.run_until_done()
        r = b.result(rid)
        assert r.error is None and r.done and len(r.tokens) == n
        assert r.tokens == _solo_generate(model, ids, image, n, seed)
        assert r.pred_boxes.shape[1] == 4


@pytest.mark.parametrize("use_graph", [True, False])
def test_rows_are_independent_of_batch_composition(setup, use_graph):
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    # reference: each request alone in a 4-row batcher
    solo = []
    for ids, image, n, seed in reqs:
        b = ContinuousBatcher(model, max_rows=4, max_len=1024, use_graph=use_graph)
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        b.run_until_done()
        solo.append(b.result(rid).tokens)
    # all at once: 5 requests on 4 rows -> the 5th waits for the first slot to free up (slot reuse)
    b = ContinuousBatcher(model, max_rows=4, max_len=1024, use_graph=use_graph)
    rids = [b.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs]
    res = b.run_until_done()
    for rid, want in zip(rids, solo):
        assert res[rid].tokens == want
    assert b.slots.n_free == 4
    # staggered arrival
    b = ContinuousBatcher(model, max_rows=4, max_len=1024, use_graph=use_graph)
    rids = []
    for ids, image, n, seed in reqs:
        rids.append(b.submit(ids, image, max_new_tokens=n, seed=seed))
        b.step(); b.step()
    res = b.run_until_done()
    for rid, want in zip(rids, solo):
        assert res[rid].tokens == want


def test_overlapped_admission_changes_no_token(setup):
    """overlap_admission=True (round 6): the admission prefill runs on a worker thread and a stream of its own while the live rows keep
    decoding; rows join at the first tick after their prefill.  Rows are independent, so every request's tokens equal the
    non-overlapped batcher's -- all at once (slot reuse, several admission waves) and with staggered arrival (admissions in the middle of
    live traffic); an over-long request inside a wave is refused without disturbing its companions; the scheduler drains and frees
    every slot.  (R: groma/serve/model_worker.py:287-338 serves one request at a time.)"""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    plain = ContinuousBatcher(model, max_rows=2, max_len=1024, overlap_admission=False)
    rids = [plain.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs]
    want = [plain.run_until_done()[r].tokens for r in rids]
    for rep in range(2):   # (twice: the second pass replays what the first one captured)
        b = ContinuousBatcher(model, max_rows=2, max_len=1024, overlap_admission=True)
        rids = [b.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs]
        big = b.submit(reqs[0][0], reqs[0][1], max_new_tokens=4000, seed=1)   # refused: prompt + 4000 > max_len
        res = b.run_until_done()
        assert [res[r].tokens for r in rids] == want
        assert all(res[r].error is None and res[r].done for r in rids)
        assert res[big].error is not None and res[big].done
        assert b.slots.n_free == 2 and b._job is None
    b = ContinuousBatcher(model, max_rows=4, max_len=1024, overlap_admission=True)
    rids = []
    ticks_with_prefill_in_flight = 0
    for ids, image, n, seed in reqs:
        rids.append(b.submit(ids, image, max_new_tokens=n, seed=seed))
        for _ in range(2):
            b.step()
            ticks_with_prefill_in_flight += int(b._job is not None and bool(b.slots.active()))
    res = b.run_until_done()
    assert [res[r].tokens for r in rids] == want
    assert b.slots.n_free == 4 and b._job is None
    with pytest.raises(ValueError):
        ContinuousBatcher(model, max_rows=2, max_len=1024, use_graph=False, overlap_admission=True)


@pytest.mark.parametrize("overlap", [False, True])
def test_a_failing_request_is_isolated_from_its_admission_wave(setup, overlap):
    """three requests admitted in ONE prefill, the middle one making the forward raise: the batched forward raises, the
    wave is re-run one by one, the offender is refused with its error and its companions' tokens are what they are alone (each
    survivor's KV has to end up in its own staging row although every batch-1 prefill lands in row 0)."""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    solo = []
    for ids, image, n, seed in (reqs[0], reqs[2]):
        b = ContinuousBatcher(model, max_rows=4, max_len=1024)
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        solo.append(b.run_until_done()[rid].tokens)
    orig = model.forward

    def poisoned(*a, **kw):   # (a request the model cannot serve, whatever the reason: here its seed)
        if 666 in (kw.get("_seeds") or []):
            raise RuntimeError("poisoned request")
        return orig(*a, **kw)
    model.forward = poisoned
    try:
        b = ContinuousBatcher(model, max_rows=4, max_len=1024, overlap_admission=overlap, admit_min=1)
        a = b.submit(reqs[0][0], reqs[0][1], max_new_tokens=reqs[0][2], seed=reqs[0][3])
        bad = b.submit(reqs[1][0], reqs[1][1], max_new_tokens=5, seed=666)
        c = b.submit(reqs[2][0], reqs[2][1], max_new_tokens=reqs[2][2], seed=reqs[2][3])
        res = b.run_until_done()
    finally:
        del model.forward
    assert res[bad].done and "poisoned" in res[bad].error and not res[bad].tokens
    assert res[a].error is None and res[c].error is None
    assert [res[a].tokens, res[c].tokens] == solo
    assert b.slots.n_free == 4


def test_admission_hold_policy(setup):
    """admit_min / admit_hold: while rows are live an admission waits until `admit_min` requests can share one prefill (or the queue
    holds fewer), at most `admit_hold` ticks; tokens do not depend on it"""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    want = []
    for ids, image, n, seed in reqs[:3]:
        b = ContinuousBatcher(model, max_rows=2, max_len=1024)
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        want.append(b.run_until_done()[rid].tokens)
    b = ContinuousBatcher(model, max_rows=2, max_len=1024, overlap_admission=False, admit_min=2, admit_hold=3)
    r0 = b.submit(reqs[0][0], reqs[0][1], max_new_tokens=reqs[0][2], seed=reqs[0][3])   # 6 tokens
    b.step()                                  # idle batcher: admitted at once although it is alone
    assert b.slots.n_free == 1
    r1 = b.submit(reqs[1][0], reqs[1][1], max_new_tokens=reqs[1][2], seed=reqs[1][3])
    b.step()
    assert b.slots.n_free == 0                # a queue of ONE is never held
    r2 = b.submit(reqs[2][0], reqs[2][1], max_new_tokens=reqs[2][2], seed=reqs[2][3])
    r3 = b.submit(reqs[0][0], reqs[0][1], max_new_tokens=4, seed=reqs[0][3])
    held = 0
    while b.live[r0].done is False:
        b.step()
    assert b.slots.n_free == 1 and len(b.queue) == 2      # r0 has left; r2 / r3 wait for a second slot ...
    for _ in range(3):
        b.step(); held += int(len(b.queue) == 2)
    assert held == 3
    b.step()
    assert len(b.queue) == 1                              # ... for admit_hold ticks, then one goes in alone
    res = b.run_until_done()
    assert [res[r0].tokens, res[r1].tokens, res[r2].tokens] == want and res[r3].tokens == want[0][:4]


def test_eos_stop_and_oversize_rejection(setup):
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    ids, image, n, seed = reqs[0]
    free = _solo_generate(model, ids, image, 8, seed)
    b = ContinuousBatcher(model, max_rows=2, max_len=1024)
    rid = b.submit(ids, image, max_new_tokens=8, seed=seed, eos_token_id=free[3])
    big = b.submit(ids, image, max_new_tokens=4000, seed=seed)  # prompt + 4000 > max_len: rejected, slot returned
    res = b.run_until_done()
    stop = free.index(free[3])
    assert res[rid].tokens == free[:stop + 1]
    assert res[big].error is not None and res[big].done
    assert b.slots.n_free == 2


def test_generate_stream_yields_growing_prefixes(setup):
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    ids, image, n, seed = reqs[1]
    b = ContinuousBatcher(model, max_rows=2, max_len=1024)
    other = b.submit(*reqs[2][:2], max_new_tokens=5, seed=reqs[2][3])  # decodes alongside the streamed request
    chunks = list(b.generate_stream(ids, image, max_new_tokens=7, stream_interval=2, seed=seed))
    assert chunks[-1] == _solo_generate(model, ids, image, 7, seed) or len(chunks[-1]) == 7
    assert all(chunks[i] == chunks[i + 1][:len(chunks[i])] for i in range(len(chunks) - 1))
    assert b.live[other].done


def test_sampled_rows_are_independent_of_batch_composition(setup):
    """temperature sampling in the batcher: a request's tokens are a function of its own (seed, positions) only -- the same
    alone, beside greedy neighbours, or admitted late; and a cold temperature equals the greedy tokens"""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    ids, image, n, seed = reqs[1]
    b = ContinuousBatcher(model, max_rows=4, max_len=1024)
    rid = b.submit(ids, image, max_new_tokens=9, seed=seed, temperature=0.8)
    b.run_until_done()
    alone = b.result(rid).tokens
    b = ContinuousBatcher(model, max_rows=4, max_len=1024)
    others = [b.submit(*reqs[i][:2], max_new_tokens=12, seed=reqs[i][3]) for i in (0, 2)]
    b.step(); b.step()
    rid = b.submit(ids, image, max_new_tokens=9, seed=seed, temperature=0.8)
    b.run_until_done()
    assert b.result(rid).tokens == alone
    greedy = [b.result(o).tokens for o in others]
    assert greedy[0] == _solo_generate(model, *reqs[0][:2], 12, reqs[0][3])
    b = ContinuousBatcher(model, max_rows=2, max_len=1024)
    rid = b.submit(ids, image, max_new_tokens=9, seed=seed, temperature=1e-6)
    b.run_until_done()
    assert b.result(rid).tokens == _solo_generate(model, ids, image, 9, seed)
    assert alone != b.live.get(rid, None)


def test_warm_admission_captures_before_live_traffic(setup):
    """ContinuousBatcher.warm_admission (round 5, ADVICE r04: GraphPool.warm had no caller): a throw-away admission prefill under
    engine.GraphPool.first_sight() captures the ViT / pyramid / prefill graphs of the shape up front; the first LIVE request of that
    shape is then a replay, and its tokens are exactly those of a batcher that was never warmed."""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    ids, image, n, seed = reqs[0]
    cold = ContinuousBatcher(model, max_rows=2, max_len=1024)
    rid = cold.submit(ids, image, max_new_tokens=n, seed=seed)
    want = cold.run_until_done()[rid].tokens
    for pool in (model.vit.graphs, model.llm.graphs, model.region.graphs):
        pool.clear()
    b = ContinuousBatcher(model, max_rows=2, max_len=1024)
    b.warm_admission(ids, image, rows=1)
    caps = (model.vit.graphs.captures, model.region.graphs.captures)
    reps = model.vit.graphs.replays
    rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
    got = b.run_until_done()[rid].tokens
    assert got == want
    assert model.vit.graphs.replays > reps                                             # the live request replayed the warmed graph
    assert (model.vit.graphs.captures, model.region.graphs.captures) == caps           # and captured nothing itself
    with pytest.raises(RuntimeError):
        b2 = ContinuousBatcher(model, max_rows=2, max_len=1024)
        b2.submit(ids, image, max_new_tokens=6, seed=seed)
        b2.step()
        b2.warm_admission(ids, image)                                                   # too late: a slot is taken


def test_batcher_over_a_hybrid_model_equals_its_generate(setup):
    """the continuous batcher over precision="hybrid" (the benchmarked build: pair-operand ViT inside the admission prefill, bf16 decode
    streams): a request's tokens equal generate() of the same model at batch 1, alone or beside a neighbour"""
    from groma_amd import constants
    from groma_amd.groma import GromaModel
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    cfg0, sd, tk = util.tiny_setup(seed=0)
    mh = GromaModel.from_state_dict(cfg0, sd, "cuda", precision="hybrid")
    mh.init_special_token_id(constants.SyntheticTokenizer())
    mh.generation_config.eos_token_id = None
    b = ContinuousBatcher(mh, max_rows=2, max_len=1024)
    rids = [b.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs[:3]]
    res = b.run_until_done()
    for rid, (ids, image, n, seed) in zip(rids, reqs[:3]):
        assert res[rid].error is None and res[rid].tokens == _solo_generate(mh, ids, image, n, seed)


@pytest.mark.parametrize("use_graph", [True, False])
def test_arena_grows_under_a_live_row(setup, use_graph):
    """grow_to: a request whose context bound exceeds max_len re-allocates the KV arena (live rows' prefixes copied, the step
    re-captured) instead of being rejected; the live row and the newcomer produce the tokens a large fixed arena gives them."""
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = setup
    (ids0, img0, _, s0), (ids1, img1, _, s1) = reqs[0], reqs[1]
    ref = ContinuousBatcher(model, max_rows=2, max_len=1024, use_graph=use_graph)
    a, b = ref.submit(ids0, img0, max_new_tokens=40, seed=s0), ref.submit(ids1, img1, max_new_tokens=120, seed=s1)
    want = ref.run_until_done()
    assert want[a].error is None and len(want[a].tokens) == 40 and len(want[b].tokens) == 120
    g = ContinuousBatcher(model, max_rows=2, max_len=640, use_graph=use_graph, grow_to=2048)
    ga = g.submit(ids0, img0, max_new_tokens=40, seed=s0)       # bound 128 + 256 + 2*100 + 40 = 624: fits
    for _ in range(5):
        g.step()
    assert g.max_len == 640 and not g.live[ga].done
    gb = g.submit(ids1, img1, max_new_tokens=120, seed=s1)      # bound 704 > 640: the arena doubles while row 0 is five tokens in
    got = g.run_until_done()
    assert g.max_len == 1280 and g.arena.smax == 1280 and g.staging.smax == 1280
    assert got[ga].error is None and got[gb].error is None
    assert got[ga].tokens == want[a].tokens and got[gb].tokens == want[b].tokens
    # a fixed arena (the default) still turns the long request away, and grow_to is validated
    f = ContinuousBatcher(model, max_rows=2, max_len=640, use_graph=use_graph)
    fb = f.submit(ids1, img1, max_new_tokens=300, seed=s1)      # (128 - 2 + 256 image tokens + 300 > 640 whatever the region count)
    assert f.run_until_done()[fb].error is not None and f.max_len == 640
    with pytest.raises(ValueError):
        ContinuousBatcher(model, max_rows=2, max_len=640, grow_to=512)


def test_batcher_over_an_e4m3_model(setup):
    """fp8=True models: the batcher's step runs the e4m3 decode kernels (per-row positions), one row == generate(batch 1), and a
    row's tokens do not depend on its neighbour (activations are quantised per row, weights per output row: nothing crosses rows)."""
    from groma_amd.groma import GromaModel
    from groma_amd.serving import ContinuousBatcher
    from groma_amd import constants
    cfg, model, reqs = setup
    cfg8, sd, tk = util.tiny_setup(seed=0)
    m8 = GromaModel.from_state_dict(cfg8, sd, device="cuda", fp8=True)
    m8.init_special_token_id(constants.SyntheticTokenizer())
    m8.generation_config.eos_token_id = None
    (ids0, img0, _, s0), (ids1, img1, _, s1) = reqs[0], reqs[1]
    solo = [_solo_generate(m8, ids0, img0, 10, s0), _solo_generate(m8, ids1, img1, 14, s1)]
    b = ContinuousBatcher(m8, max_rows=1, max_len=1024)
    r = b.submit(ids0, img0, max_new_tokens=10, seed=s0)
    assert b.run_until_done()[r].tokens == solo[0]
    alone = []
    for ids, img, n, s in ((ids0, img0, 10, s0), (ids1, img1, 14, s1)):
        b = ContinuousBatcher(m8, max_rows=2, max_len=1024)
        r = b.submit(ids, img, max_new_tokens=n, seed=s)
        alone.append(b.run_until_done()[r].tokens)
    b = ContinuousBatcher(m8, max_rows=2, max_len=1024)
    ra = b.submit(ids0, img0, max_new_tokens=10, seed=s0)
    b.step(), b.step()
    rb = b.submit(ids1, img1, max_new_tokens=14, seed=s1)
    got = b.run_until_done()
    assert got[ra].tokens == alone[0] and got[rb].tokens == alone[1]
    assert len(set(alone[0])) > 1 or len(set(alone[1])) > 1   # (not a constant stream)
    print("e4m3 batcher: rows=1 == generate; rows=2 alone == together;", alone[0][:5], solo[0][:5])
