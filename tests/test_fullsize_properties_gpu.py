"""BASELINE.json full-size configuration (Groma-7B dims, random-init bf16) checked through size-independent properties
-- the fp32 CPU oracle cannot run a 7B model in test time, so at full size we assert what must hold for ANY weights:
  P1 determinism              same inputs + same host RNG seed -> bit-identical logits, boxes, ids
  P2 batch independence       an image's results do not depend on its batch mates: every kernel reduces over K in a fixed
                              order per output element and the split-K plan is a function of the layer shape (N, K) only
                              (ops.plan_splits), so this holds BITWISE under both plans -- "throughput" (never split) and
                              "latency" (fixed per-shape factors) -- across the 128/256 GEMM kernels; the two plans differ
                              from EACH OTHER at bf16-noise level
  P3 KV-cache consistency     logits of position L-1 from a full prefill == prefill of L-1 tokens + 1 decode step
                              (same weights, different kernels/shapes: bf16 tolerance)
  P4 contract at full size    N=100 regions/image at box_score_thres=0, L=582, logits [bs,582,32114], finite,
                              pred_boxes in (0,1), NMS keep ids strictly valid and unique
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(dev):
    from groma_amd import config, constants, synth
    from groma_amd.groma import GromaModel
    cfg = config.groma_7b(box_score_thres=0.0)
    m = GromaModel.from_synthetic(cfg, seed=0, device=dev)
    m.init_special_token_id(constants.SyntheticTokenizer())
    images, ids = synth.make_inputs(cfg, m, 3, seed=99)
    return m, images.to(dev), ids.to(dev)


def _fwd(m, images, ids, seed=5):
    torch.manual_seed(seed)
    out = m.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True)
    return out, m._last_aux


def test_contract_and_determinism(big):
    m, images, ids = big
    o1, a1 = _fwd(m, images, ids)
    l1 = o1.logits.clone()
    boxes1 = [b.clone() for b in o1.hidden_states[1]["pred_boxes"]]
    keep1 = [k.clone() for k in a1["nms_keep"]]
    assert tuple(l1.shape) == (3, 582, 32114) and torch.isfinite(l1).all()
    for b, k in zip(boxes1, keep1):
        assert tuple(b.shape) == (100, 4) and (b > 0).all() and (b < 1).all()
        assert len(k) == 100 and len(set(k.tolist())) == 100 and k.min() >= 0 and k.max() < 300
    o2, a2 = _fwd(m, images, ids)
    assert torch.equal(o2.logits, l1)
    assert all(torch.equal(x, y) for x, y in zip(a2["nms_keep"], keep1))
    assert all(torch.equal(x, y) for x, y in zip(o2.hidden_states[1]["pred_boxes"], boxes1))


def _batch_vs_single(m, images, ids, check):
    # the host RNG draws one randperm(100) per image in batch order: replay the same permutations per image
    torch.manual_seed(5)
    perms = [torch.randperm(100) for _ in range(3)]
    o_all, a_all = _fwd(m, images, ids, seed=5)
    la = o_all.logits.clone()
    keep_all = [k.clone() for k in a_all["nms_keep"]]
    for i in (0, 2):
        torch.manual_seed(5)
        for _ in range(i):
            torch.randperm(100)  # advance the RNG to image i's draw
        out = m.forward(input_ids=ids[i:i + 1].clone(), images=images[i:i + 1], return_dict=True)
        assert torch.equal(m._last_aux["nms_keep"][0], keep_all[i])
        assert torch.equal(m._last_aux["sel_idx"][0], keep_all[i][perms[i]])
        check(i, out.logits[0], la[i])


@pytest.mark.parametrize("plan", ["throughput", "latency"])
def test_batch_independence_bitwise(big, plan):
    """An image's logits, NMS ids and shuffled selection do not depend on its batch mates -- bit for bit, through all
    24 + 6 + 32 layers, under either GEMM plan (the plan is a function of the layer shape, never of the batch)."""
    m, images, ids = big
    old = m.gemm_plan
    m.gemm_plan = plan
    try:
        def check(i, single, batched):
            assert torch.equal(single, batched), f"image {i}: logits depend on batch mates under the {plan} plan"
        _batch_vs_single(m, images, ids, check)
    finally:
        m.gemm_plan = old


def test_batch_independence_bitwise_benchmarked_build_latency_plan(dev):
    """the benchmarked build (precision="hybrid-fp16": the ViT on operand pairs) under the latency plan -- in the pair build the plan
    counts PHYSICAL k-values (round 6), so the ViT's K = 1024 GEMMs are split in two and fc2 in eight: an image's logits and index
    results still do not depend on its batch mates, bit for bit (the factors are a function of (N, K) and the build, never of M)."""
    from groma_amd import config, constants, ops, synth
    from groma_amd.groma import GromaModel
    cfg = config.groma_7b(box_score_thres=0.0)
    m = GromaModel.from_synthetic(cfg, seed=0, device=dev, precision="hybrid-fp16")
    m.init_special_token_id(constants.SyntheticTokenizer())
    images, ids = synth.make_inputs(cfg, m, 3, seed=99)
    with ops.precision("ref"), ops.gemm_plan("latency"):
        assert [ops.plan_splits(N, K) for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))] == [2, 2, 2, 8]
    m.gemm_plan = "latency"

    def check(i, single, batched):
        assert torch.equal(single, batched), f"image {i}: logits depend on batch mates (hybrid-fp16, latency plan)"
    _batch_vs_single(m, images.to(dev), ids.to(dev), check)
    del m
    torch.cuda.empty_cache()


def test_plans_agree_to_bf16_noise(big):
    """"latency" splits the o-proj / down-proj / fc2 / bridge GEMMs along K: fp32 sums in another order, so the bf16 roundings
    behind them differ at noise level from the "throughput" plan.  Checked stage by stage on identical stage inputs: the ViT
    states agree to bf16 noise; the LLaMA stage, fed the same input embeddings, gives logits that agree to bf16 noise and pick
    the same token wherever the margin is clear of it.  (The region selection in between is discrete; with random-init weights
    the proposal scores are near-tied, so which boxes survive is decided by that noise.  Selection exactness is what the a7
    tests pin, on inputs whose score gaps are clear: tests/test_fullwidth_parity_gpu.py.)"""
    from groma_amd import ops
    m, images, ids = big
    m.capture_embeds = True
    try:
        o_all, a_all = _fwd(m, images[:1], ids[:1], seed=5)
    finally:
        m.capture_embeds = False
    la = o_all.logits.float().clone()[0]
    emb = a_all["inputs_embeds"].clone()
    h4 = [h.float().clone() for h in a_all["hidden4"]]
    L = emb.shape[1]
    m.gemm_plan = "latency"
    try:
        torch.manual_seed(5)
        m.forward(input_ids=ids[:1].clone(), images=images[:1], return_dict=True)
        for hs, hb in zip(m._last_aux["hidden4"], h4):
            assert ((hs.float() - hb).norm() / hb.norm()).item() < 1e-2
        cache = m.llm.new_cache(1, L, images.device)
        with ops.gemm_plan("latency"):
            logits, _ = m.llm.forward(emb[0].reshape(L, -1).clone(), 1, L, cache, kv_len=None, all_logits=True)
    finally:
        m.gemm_plan = "throughput"
    s, b = logits.float().reshape(L, -1), la
    rel = ((s - b).norm() / b.norm()).item()
    assert 0 < rel < 2e-2, rel      # different summation order (not identical), bf16-noise distance
    err = (s - b).abs().amax(-1)
    top2 = b.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert torch.equal(s.argmax(-1)[clear], b.argmax(-1)[clear])
    assert clear.float().mean().item() > 0.3


def test_kv_cache_step_matches_prefill(big):
    m, images, ids = big
    torch.manual_seed(11)
    full = m.forward(input_ids=ids[:1].clone(), images=images[:1], return_dict=True, use_cache=True)
    lg_full = full.logits[0, -1].float().clone()
    cache = full.past_key_values
    L = cache.seq_len
    # feed the last expanded token again as a decode step on a cache truncated by one position
    emb_ids = None
    torch.manual_seed(11)
    # rebuild the expanded ids on the host exactly as forward() did
    n_reg = [100]
    new_ids, _ = m._splice(ids[:1].cpu(), 256, n_reg)
    cache.seq_len = L - 1
    # the spliced last token is a text id (region placeholders sit before the prompt tail), so a plain embedding step
    step = m.forward(input_ids=new_ids[:, -1:].to(ids.device), past_key_values=cache, return_dict=True)
    lg_step = step.logits[0, -1].float()
    rel = ((lg_step - lg_full).norm() / lg_full.norm()).item()
    assert rel < 2e-2, rel
    assert lg_step.argmax().item() == lg_full.argmax().item() or (lg_full.topk(2).values.diff().abs().item() < 4 * (lg_step - lg_full).abs().max().item())
