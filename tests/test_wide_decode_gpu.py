"""Decode steps of 9..64 rows (round 6; SURVEY 8f rank 1, R: groma/serve/model_worker.py:287-338 serves one request at a time):
csrc/gemm_skinny.hip -- the weight stream on the matrix unit (gr_gemm_desc.tile = 3) -- and the step built on it
(engine.LlamaEngine._decode_forward_wide), through the continuous batcher.

 * the kernel against float64 of its stored operands, every epilogue the step uses (16-bit out, f32 out, fp32 residual in place,
   SwiGLU over interleaved rows) at M = 9 .. 64 and the LLaMA shapes' K (4096, 11008 = 43 slices: an odd tail of the 6-unrolled loop; 1408 and 64: a partial last slice);
 * a row's result is bitwise independent of its batch company (M = 64 rows vs the same rows at M = 9 .. 33), which is what lets a
   served request's tokens not depend on who shares its steps;
 * a 16-row decode step of the tiny model equals the general kernels' step (same roundings, another summation order);
 * ContinuousBatcher(max_rows = 16): every request's tokens equal its solo run in a 4-row batcher (the 8-row streams), graph and eager."""
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _f64(a, w):
    return a.double().cpu() @ w.double().cpu().t()


@pytest.mark.parametrize("M,N,K", [(9, 512, 4096), (16, 4096, 4096), (33, 1024, 11008), (64, 2048, 4096), (48, 22016, 4096), (17, 260, 512), (16, 512, 1408), (40, 256, 64)])
def test_skinny_gemm_epilogues(dev, M, N, K):
    from groma_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn((M, K), generator=g) * 0.5).to(dev).to(ops.H16())
    w = (torch.randn((N, K), generator=g) * 0.05).to(dev).to(ops.H16())
    ref = _f64(a, w)
    out = ops.gemm(a, w, out_f32=True, tile=3)
    assert util.relerr(out, ref) < 2e-6
    out16 = ops.gemm(a, w, tile=3)
    assert out16.dtype == ops.H16() and util.relerr(out16, ref) < 4e-3
    resid = torch.randn((M, N), generator=g).to(dev)
    r2 = resid.clone()
    ops.gemm(a, w, resid=r2, out=r2, out_f32=True, tile=3)                      # in place, as the residual stream is updated
    assert util.relerr(r2, ref + resid.double().cpu()) < 2e-6
    bias = torch.randn((N,), generator=g).to(dev)
    assert util.relerr(ops.gemm(a, w, bias=bias, out_f32=True, tile=3), ref + bias.double().cpu()) < 2e-6
    if N % 8 == 0:
        act = ops.gemm(a, w, act=3, tile=3)                                      # interleaved rows: even = gate, odd = up
        gate, up = ref[:, 0::2], ref[:, 1::2]
        assert act.shape == (M, N // 2) and util.relerr(act, torch.nn.functional.silu(gate) * up) < 4e-3
        assert torch.equal(act, ops.gemm(a, w, act=3, tile=128)) or util.relerr(act, ops.gemm(a, w, act=3, tile=128)) < 4e-3
    # the prefill kernels compute the same product (another summation order)
    assert util.relerr(out, ops.gemm(a, w, out_f32=True, tile=128)) < 2e-6
    assert torch.equal(out, ops.gemm(a, w, out_f32=True, tile=3))               # deterministic


def test_skinny_gemm_rows_independent_of_company_and_refusals(dev):
    from groma_amd import ops
    g = torch.Generator().manual_seed(5)
    K, N = 4096, 1536
    a = torch.randn((64, K), generator=g).to(dev).to(ops.H16())
    w = (torch.randn((N, K), generator=g) * 0.03).to(dev).to(ops.H16())
    # A row's sums are a function of the layer shape and of the row-BLOCK count (<= 16, <= 32, <= 64 rows: the K-ways / ring depths
    # are chosen per class, i.e. per batcher capacity), never of which other rows are there or where in the batch the row sits.
    for cap, ms in ((64, (49, 57, 64)), (32, (17, 25, 32)), (16, (9, 12, 16))):
        full = ops.gemm(a[:cap].contiguous(), w, out_f32=True, tile=3)
        for m in ms:
            assert torch.equal(ops.gemm(a[:m].contiguous(), w, out_f32=True, tile=3), full[:m]), (cap, m)
        h = cap // 2 + 1
        sub = torch.cat([a[cap - 6:cap], a[:h]]).contiguous()                    # the same rows at other positions, other company (same class)
        got = ops.gemm(sub, w, out_f32=True, tile=3)
        assert cap // 2 < sub.shape[0] <= cap and torch.equal(got[:6], full[cap - 6:cap]) and torch.equal(got[6:], full[:h]), cap
    full = ops.gemm(a, w, out_f32=True, tile=3)
    for m in (9, 17):                                                             # across classes: the same product, another summation order
        assert util.relerr(ops.gemm(a[:m].contiguous(), w, out_f32=True, tile=3), full[:m]) < 2e-6
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros((65, K), device=dev, dtype=ops.H16()), w, tile=3)   # M > 64
    with pytest.raises(RuntimeError):
        ops.gemm(a[:16, :336].contiguous(), w[:, :336].contiguous(), tile=3)     # K % 32 != 0 (K % 64 == 0 is the ABI's own rule)
    with ops.precision("ref"):
        with pytest.raises(RuntimeError):
            ops.gemm(ops.to_h16(torch.zeros((16, 256), device=dev)), ops.to_h16(torch.zeros((64, 256), device=dev)), tile=3)


@pytest.fixture(scope="module")
def tiny(dev):
    from groma_amd import synth
    cfg, sd, tk = util.tiny_setup(seed=0)
    model = util.device_model(cfg, sd)
    model.generation_config.eos_token_id = None
    reqs = []
    for i in range(12):
        images, ids = synth.make_inputs(cfg, tk, bs=1, seed=300 + i)
        reqs.append((ids[0], images[0], 5 + (i % 4) * 2, 900 + i))
    return cfg, model, reqs


def test_wide_step_equals_general_kernels(tiny):
    """one 16-row decode step on the matrix-unit weight stream against the same step on the general GEMM / attention kernels"""
    from groma_amd import engine
    cfg, model, reqs = tiny
    llm = model.llm
    dev = model.device
    bs = 16
    outs = {}
    for wide in (True, False):
        engine.WIDE_DECODE = wide
        try:
            with engine.ops.precision(model.precision):
                cache = llm.new_cache(bs, 128, dev)
                g = torch.Generator().manual_seed(3)
                for l in range(len(cache.k)):   # a synthetic 40-key prefix per row
                    cache.k[l][:, :, :40] = (torch.randn(cache.k[l][:, :, :40].shape, generator=g) * 0.3).to(dev).to(cache.k[l].dtype)
                    cache.vt[l][..., :40] = (torch.randn(cache.vt[l][..., :40].shape, generator=g) * 0.3).to(dev).to(cache.vt[l].dtype)
                cache.seq_len = 40
                h = (torch.randn((bs, llm.T), generator=g) * 0.5).to(dev)
                logits, _ = llm.forward(h, bs, 1, cache)
                outs[wide] = (logits.float().cpu().clone(), cache.k[0][:, :, 40].float().cpu().clone())
        finally:
            engine.WIDE_DECODE = True
    assert util.relerr(outs[True][0], outs[False][0]) < 5e-3     # (16-bit roundings of slightly different fp32 sums)
    assert util.relerr(outs[True][1], outs[False][1]) < 5e-3
    assert (outs[True][0].argmax(-1) == outs[False][0].argmax(-1)).float().mean() >= 0.9


@pytest.mark.parametrize("use_graph", [True, False])
def test_sixteen_row_batcher_rows_equal_their_solo_runs(tiny, use_graph):
    from groma_amd.serving import ContinuousBatcher
    cfg, model, reqs = tiny
    solo = []
    for ids, image, n, seed in reqs:            # reference: each request alone, 16-row batcher (same step kernels, other company)
        b = ContinuousBatcher(model, max_rows=16, max_len=1024, use_graph=use_graph)
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        b.run_until_done()
        solo.append(b.result(rid).tokens)
    b = ContinuousBatcher(model, max_rows=16, max_len=1024, use_graph=use_graph)
    rids = [b.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs]
    res = b.run_until_done()
    for rid, want in zip(rids, solo):
        assert res[rid].error is None and res[rid].tokens == want
    assert b.slots.n_free == 16
    # and against the 8-row streams (another kernel family: same roundings, another summation order) wherever no step is a near-tie
    b4 = ContinuousBatcher(model, max_rows=4, max_len=1024, use_graph=use_graph)
    same = 0
    for (ids, image, n, seed), want in zip(reqs[:4], solo[:4]):
        rid = b4.submit(ids, image, max_new_tokens=n, seed=seed)
        b4.run_until_done()
        same += int(b4.result(rid).tokens == want)
    print(f"requests whose tokens also equal the 8-row stream's: {same} / 4")
    assert same >= 2


@pytest.mark.parametrize("M,N,K", [(16, 4096, 4096), (33, 1024, 11008), (64, 2048, 4096), (12, 512, 1408 - 1408 % 128)])
def test_skinny_gemm_e4m3(dev, M, N, K):
    """csrc/gemm_skinny_fp8.hip against the e4m3 ping-pong GEMM on the SAME quantised operands (both dequantise acc * w_scale[n] * a_scale[m];
    they differ by the matrix unit's block accumulation order) and against float64 of the dequantised operands; rows independent of company"""
    from groma_amd import ops, weights
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn((M, K), generator=g) * 0.7).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.05).to(dev)
    x8, sx = ops.quant_rows_fp8(x)
    w8, sw = weights.q8(w)
    ref = (x8.float().double().cpu() * sx.double().cpu()[:, None]) @ (w8.float().double().cpu() * sw.double().cpu()[:, None]).t()
    out = ops.gemm(x8, w8, a_scale=sx, w_scale=sw, out_f32=True, tile=3)
    assert util.relerr(out, ref) < 2e-4
    assert util.relerr(out, ops.gemm(x8, w8, a_scale=sx, w_scale=sw, out_f32=True)) < 2e-4       # the 256 x 256 e4m3 kernel
    resid = torch.randn((M, N), generator=g).to(dev)
    r2 = resid.clone()
    ops.gemm(x8, w8, a_scale=sx, w_scale=sw, resid=r2, out=r2, out_f32=True, tile=3)
    assert util.relerr(r2, ref + resid.double().cpu()) < 2e-4
    if N % 8 == 0:
        act = ops.gemm(x8, w8, a_scale=sx, w_scale=sw, act=3, tile=3)
        assert util.relerr(act, torch.nn.functional.silu(ref[:, 0::2]) * ref[:, 1::2]) < 1e-2
    if M > 48:
        sub = ops.gemm(x8[:55].contiguous(), w8, a_scale=sx[:55].contiguous(), w_scale=sw, out_f32=True, tile=3)
        assert torch.equal(sub, out[:55])


def test_e4m3_model_wide_step_and_batcher(dev):
    """an fp8 = True model past 8 rows: the 12-row decode step on the e4m3 matrix-unit stream against the same step on the general e4m3
    kernels, and ContinuousBatcher(max_rows = 12) rows equal to their solo runs (quantisation is per row: nothing crosses rows)"""
    from groma_amd import constants, engine, synth
    from groma_amd.groma import GromaModel
    from groma_amd.serving import ContinuousBatcher
    cfg, sd, tk = util.tiny_setup(seed=0)
    model = GromaModel.from_state_dict(cfg, sd, "cuda", fp8=True)
    model.init_special_token_id(constants.SyntheticTokenizer())
    model.generation_config.eos_token_id = None
    llm = model.llm
    if llm.T % 128 or llm.I % 128:
        pytest.skip("tiny widths are not multiples of the e4m3 k-block")
    outs = {}
    bs = 12
    for wide in (True, False):
        engine.WIDE_DECODE = wide
        try:
            cache = llm.new_cache(bs, 128, model.device)
            g = torch.Generator().manual_seed(3)
            for l in range(len(cache.k)):
                cache.k[l][:, :, :40] = (torch.randn(cache.k[l][:, :, :40].shape, generator=g) * 0.3).to(model.device).to(cache.k[l].dtype)
                cache.vt[l][..., :40] = (torch.randn(cache.vt[l][..., :40].shape, generator=g) * 0.3).to(model.device).to(cache.vt[l].dtype)
            cache.seq_len = 40
            h = (torch.randn((bs, llm.T), generator=g) * 0.5).to(model.device)
            logits, _ = llm.forward(h, bs, 1, cache)
            outs[wide] = logits.float().cpu().clone()
        finally:
            engine.WIDE_DECODE = True
    e = util.relerr(outs[True], outs[False])
    print(f"e4m3 12-row step, matrix-unit stream vs general e4m3 kernels: {e:.2e}")
    assert e < 5e-2       # (e4m3 quantisation steps amplify last-bit differences of the fp32 sums: the chained bound of tests/test_fp8_gpu.py)
    reqs = []
    for i in range(6):
        images, ids = synth.make_inputs(cfg, tk, bs=1, seed=400 + i)
        reqs.append((ids[0], images[0], 5 + i % 3, 700 + i))
    solo = []
    for ids, image, n, seed in reqs:
        b = ContinuousBatcher(model, max_rows=12, max_len=1024)
        rid = b.submit(ids, image, max_new_tokens=n, seed=seed)
        b.run_until_done()
        solo.append(b.result(rid).tokens)
    b = ContinuousBatcher(model, max_rows=12, max_len=1024)
    rids = [b.submit(ids, image, max_new_tokens=n, seed=seed) for ids, image, n, seed in reqs]
    res = b.run_until_done()
    for rid, want in zip(rids, solo):
        assert res[rid].error is None and res[rid].tokens == want


@pytest.mark.parametrize("graph", [True, False])
def test_generate_twelve_rows_matches_oracle_greedy(dev, graph):
    """generate() with more than 8 rows decodes on the matrix-unit weight stream: every token of 12 rows against HF-greedy over the
    oracle (R: groma/eval/eval_rec.py:93-104 over groma/model/groma.py:176-200,376-402; the oracle consumes the device's ViT states,
    as in tests/test_parity_gpu.py; boosted <r_k> rows keep every step's margin far outside the 16-bit error band)"""
    from groma_amd import synth
    from oracle import groma_oracle as O
    cfg, sd, tk = util.tiny_setup(seed=0)
    sd = dict(sd)
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0
    sd["extra_lm_head.weight"] = w
    model = util.device_model(cfg, sd)
    model.generation_config.eos_token_id = None
    model.decode_graph = graph
    images, ids = synth.make_inputs(cfg, tk, bs=12, seed=1239)
    n = 4
    torch.manual_seed(9)
    g = model.generate(ids.clone(), images=images, max_new_tokens=n, return_dict_in_generate=True, output_hidden_states=True)
    dev_h = [model._ws.get(f"vit_h{i}", (12, model.vit.T, model.vit.D), torch.float32).cpu() for i in range(4)]
    torch.manual_seed(9)
    with torch.no_grad():
        ref = O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, n, eos_token_id=-1, hidden_states=tuple(dev_h))
    boxes = g.hidden_states[0][-1]["pred_boxes"]
    # 12 UNSELECTED images: on 7-9 % of ordinary images two fp32 evaluations of the proposer order a near-tie differently
    # (profiles/r06_index_survival.txt) and the row's regions -- hence its tokens -- are then not comparable; every other row must match
    same = [i for i in range(12) if boxes[i].shape == ref["pred_boxes"][i].shape and torch.allclose(boxes[i].cpu(), ref["pred_boxes"][i], atol=1e-5)]
    assert len(same) >= 8, same   # (expected 11: P(a row is a near-tie) ~ 8 %, and which rows are depends on the host BLAS's summation order)
    P = ids.shape[1]
    # (the 40x boost scales the logits' absolute 16-bit error as well -- ~0.1 on these unselected inputs, whose margins nobody chose:
    #  a step counts as resolvable from a margin of 0.5 on; tests/test_parity_gpu.py::gen_setup uses a seed scanned for margins >= 2)
    ncmp = util.assert_greedy_tokens_match(g.sequences[same, P:].cpu(), ref["sequences"][same, P:], ref["margins"][same], 0.5, "12-row generate")
    print("rows with the oracle's regions:", len(same), "of 12; tokens compared", ncmp, "of", len(same) * n, "min margins per row", ref["margins"][same].min(dim=1).values.tolist())
    assert ncmp >= len(same) * n * 0.6
