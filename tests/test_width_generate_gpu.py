"""BASELINE configs[3] (greedy generate) at Groma-7B WIDTH against the CPU oracle -- the decode-side twin of
test_fullwidth_parity_gpu.py.  config.groma_7b_width: every shape of the 7B decode step (d 4096, 32 heads x 128, SwiGLU
11 008, 32 114-wide head) at reduced depth, so the fp32 oracle finishes in seconds.
R: groma/eval/eval_rec.py:93-104 (generate call), groma/model/groma.py:176-200,376-402 (decode branch, all-ones mask T6),
   groma/serve/model_worker.py:287-338 (serving loop).

  * every greedy token of generate() -- hipGraph replay and eager -- equals HF-greedy over the oracle, incl. a ragged
    right-padded batch (T6).  The region-token rows of extra_lm_head are boosted so the arg-max margins are far outside the
    bf16 error band (random-init logits are otherwise near-tied and every comparison would be vacuous);
  * the decode-step KERNELS at d = 4096, teacher-forced (engine.TRACE): fused residual-reduce + RMSNorm
    (in the QKV stream's prologue), QKV stream + RoPE + cache write (gemv_fused), single-query attention with the keys
    split over blocks (decode_attention nsplit > 1, merged inside the o-proj GEMV), gate/up GEMV + SwiGLU, down GEMV, head
    GEMV -- each against the oracle's version of that one operation on the tensor the kernel consumed;
  * serving rows at 7B width are bitwise independent of batch composition, under both GEMM plans."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import groma_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_F32OUT = 1e-3, 1e-5


@pytest.fixture(scope="module")
def gw(dev):
    from groma_amd import config as gconfig, synth
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 8)))
    cfg = gconfig.groma_7b_width(box_score_thres=0.0)
    sd = synth.make_state_dict(cfg, 0)
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0   # <r_k> rows: clear arg-max margins (tests/test_checkpoint_eval_gpu.py uses the same trick)
    sd["extra_lm_head.weight"] = w
    tk = util.TokenIds()
    model = util.device_model(cfg, sd)
    model.generation_config.eos_token_id = None
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=4242)
    return cfg, sd, tk, model, images, ids


def _oracle_generate(gw, ids, images, n, seed):
    cfg, sd, tk, model = gw[:4]
    bs = ids.shape[0]
    dev_h = [h.float().cpu() for h in model._last_aux["hidden4"]]
    assert dev_h[0].shape[0] == bs
    torch.manual_seed(seed)
    with torch.no_grad():
        return O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, n, eos_token_id=-1,
                                 hidden_states=tuple(dev_h))


def _min_margin(dev_first_logits, ref):
    """margin below which a token difference is a legitimate fork: 4x the measured abs error of the device's prefill
    last-position logits against the oracle's (never below 0.05)"""
    err = (dev_first_logits.float().cpu() - ref["prefill"]["logits"][:, -1]).abs().max().item()
    return max(0.05, 4 * err), err


@pytest.mark.parametrize("graph", [True, False])
def test_generate_tokens_match_oracle_at_width(gw, graph):
    cfg, sd, tk, model, images, ids = gw
    n = 8
    old = model.decode_graph
    try:
        model.decode_graph = graph
        torch.manual_seed(31)
        g = model.generate(ids.clone(), images=images, use_cache=True, do_sample=False, max_new_tokens=n,
                           return_dict_in_generate=True, output_hidden_states=True)
    finally:
        model.decode_graph = old
    ref = _oracle_generate(gw, ids, images, n, 31)
    assert torch.equal(model._last_aux["input_ids"], ref["prefill"]["input_ids"]) and ref["prefill"]["input_ids"].shape[1] == 582
    for i in range(2):
        assert torch.allclose(g.hidden_states[0][-1]["pred_boxes"][i].cpu(), ref["pred_boxes"][i], atol=1e-5)
    P = ids.shape[1]
    new = g.sequences[:, P:].cpu()
    torch.manual_seed(31)
    first = model.forward(input_ids=ids.clone(), images=images, return_dict=True).logits[:, -1]
    mm, err = _min_margin(first, ref)
    ncmp = util.assert_greedy_tokens_match(new, ref["sequences"][:, P:], ref["margins"], mm, f"width generate graph={graph}")
    print(f"[width generate graph={graph}] device {new.tolist()} oracle {ref['sequences'][:, P:].tolist()} compared {ncmp}; "
          f"min oracle margin {ref['margins'].min().item():.2f}, prefill logit abs err {err:.3e}, fork margin {mm:.3f}")
    assert ncmp == 2 * n, "every token of both rows must be resolvable with the boosted head"
    assert all(int(t) in tk.box_idx_token_ids for t in new.reshape(-1))


def test_generate_ragged_right_padded_batch_at_width(gw):
    """T6 at d = 4096: the shorter row's next token is the arg-max at its last PAD position and both rows decode at the same
    position with an all-ones mask -- reproduced, and equal to the oracle's tokens"""
    cfg, sd, tk, model, images, ids = gw
    ids = ids.clone()
    ids[1, -9:] = tk.pad_token_id
    n = 6
    torch.manual_seed(32)
    g = model.generate(ids.clone(), images=images, max_new_tokens=n, return_dict_in_generate=True)
    ref = _oracle_generate(gw, ids, images, n, 32)
    am = ref["prefill"]["attention_mask"]
    assert am[1].sum() < am[0].sum()
    P = ids.shape[1]
    new = g.sequences[:, P:].cpu()
    ncmp = util.assert_greedy_tokens_match(new, ref["sequences"][:, P:], ref["margins"], 0.25, "width ragged")
    print(f"[width ragged] device {new.tolist()} oracle {ref['sequences'][:, P:].tolist()} compared {ncmp} margins {ref['margins'].tolist()}")
    assert ncmp >= n


def test_decode_kernels_teacher_forced_at_width(gw):
    """one decode step at d = 4096 / 32 heads / L = 582 keys, kernel by kernel"""
    from groma_amd import engine
    cfg, sd, tk, model, images, ids = gw
    torch.manual_seed(33)
    out = model.forward(input_ids=ids.clone(), images=images, return_dict=True, use_cache=True)
    cache = out.past_key_values
    L = cache.seq_len
    tok = torch.tensor([[tk.box_idx_token_ids[3]], [1234]], dtype=torch.int64)  # one new-vocabulary id, one LLaMA id
    engine.TRACE = {}
    try:
        with torch.no_grad():
            step = model.forward(input_ids=tok.cuda(), past_key_values=cache, return_dict=True)
        torch.cuda.synchronize()
        nsplit = engine.TRACE.pop("dec0.nsplit")
        t = {k: v.float().cpu() for k, v in engine.TRACE.items()}
    finally:
        engine.TRACE = None
    assert cache.seq_len == L + 1 and nsplit > 1, "2 rows x 32 heads must take the key-sliced attention path"
    k_c, v_c = (x.float().cpu() for x in cache[0])          # [2, 32, L+1, 128] incl. the row this step wrote
    logits = step.logits[:, -1].float().cpu()
    rows, r, rel = [], O._r, util.relerr

    def chk(name, dev, ref, tol):
        e = rel(dev, ref)
        rows.append((name, e, tol))
        print(f"[decode kernel] {name:66s} rel-L2 {e:.2e}  (tol {tol:.0e})")

    def rms(x, w):
        return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))

    p = "llm.model.layers.0."
    with torch.no_grad(), O.rounding("bf16"):
        h = t["dec0.h_in"]
        chk("embedding gather (both tables) -> f32 residual row", h, O.get_input_embeddings(sd, tok).view(2, -1), 1e-6)
        # round 4: each weight stream builds its operand in its prologue (RMSNorm -> 16 bits) and feeds its consumer in its
        # epilogue, so the normed rows are never materialised: the oracle's rounded norm of the SAME fp32 rows stands in for them
        x = r(rms(h, sd[p + "input_layernorm.weight"]))
        q, k, v = (r(O._lin16(x, sd, p + f"self_attn.{n}_proj", False)).view(2, 32, 128) for n in "qkv")
        cos, sin = O.rope_tables(128, L + 1)
        cos, sin = cos[L], sin[L]
        q, k = r(q * cos + O._rot_half(q) * sin), r(k * cos + O._rot_half(k) * sin)
        chk("gemv_fused QKV 2x12288x4096 (RMSNorm prologue, RoPE + cache epilogue): q", t["dec0.q"].view(2, 32, 128), q, TOL_BF16)
        chk("                                              K cache row at position L", k_c[:, :, L], k, TOL_BF16)
        chk("                                              V^T cache column at position L", v_c[:, :, L], v, TOL_BF16)
        # single-query attention over the device's own cache (P stays fp32 in this kernel; context rounded once), merged
        # across the key slices inside the o-proj stream's prologue: compared through the o-proj + residual
        att = torch.softmax((t["dec0.q"].view(2, 32, 1, 128) @ k_c.transpose(2, 3)) / math.sqrt(128), dim=-1)
        ctx = r((att @ v_c).reshape(2, 4096))
        h_attn = h + O._lin16(ctx, sd, p + "self_attn.o_proj", False)
        chk(f"decode_attention ({nsplit} key slices) + gemv_fused o-proj (slice merge prologue, residual epilogue, f32)", t["dec0.h_attn"], h_attn, 1e-4)
        x = r(rms(t["dec0.h_attn"], sd[p + "post_attention_layernorm.weight"]))
        act = F.silu(O._lin16(x, sd, p + "mlp.gate_proj", False)) * O._lin16(x, sd, p + "mlp.up_proj", False)
        chk("gemv_fused gate/up 2x22016x4096 (RMSNorm prologue, SwiGLU epilogue)", t["dec0.act"], r(act), TOL_BF16)
        h_out = t["dec0.h_attn"] + O._lin16(t["dec0.act"], sd, p + "mlp.down_proj", False)
        chk("gemv_fused down 2x4096x11008 + residual (f32)", t["dec.h_out"], h_out, TOL_F32OUT)
        chk("gemv_fused head 2x32128x4096 (final RMSNorm prologue, f32 logits)", logits,
            O.lm_logits(sd, r(rms(t["dec.h_out"], sd["llm.model.norm.weight"]))), 2e-4)
    bad = [(n, e, tol) for n, e, tol in rows if not e < tol]
    assert not bad, bad
    # chained: the oracle's own decode step on the oracle's own prefill of the device embeddings (both rounding modes)
    model.capture_embeds = True
    try:
        torch.manual_seed(33)
        model.forward(input_ids=ids.clone(), images=images, return_dict=True)
        emb = model._last_aux["inputs_embeds"].cpu()
    finally:
        model.capture_embeds = False
    res = {}
    with torch.no_grad():
        for mode in (None, "bf16"):
            with O.rounding(mode):
                _, past = O.llama_forward(sd, cfg.to_dict(), emb, torch.ones((2, L)))
                res[mode] = O.groma_decode_step(sd, cfg.to_dict(), tok[:, 0], past)[0][:, -1]
    e32, e16 = rel(logits, res[None]), rel(logits, res["bf16"])
    print(f"[decode chained] prefill + 1 decode step logits: vs fp32 oracle {e32:.3e}, vs bf16-rounded oracle {e16:.3e}")
    assert e32 < 1.2e-2 and e16 < 6 * TOL_BF16 and e16 < e32  # prefill (8 roundings) + one more layer pass on top of it: measured 8.7e-3 / 5.0e-3
    assert torch.equal(logits.argmax(-1), res[None].argmax(-1))


@pytest.mark.parametrize("plan", ["throughput", "latency"])
def test_serving_rows_independent_of_batch_composition_at_width(gw, plan):
    """ContinuousBatcher at 7B width: a request's tokens alone == beside two neighbours == admitted late.  Admission batches of
    1, 2 and 3 requests give the prefill GEMMs different M; the split-K plan is a function of (N, K) only, so nothing changes."""
    from groma_amd import synth
    from groma_amd.serving import ContinuousBatcher
    cfg, sd, tk, model, images, ids = gw
    reqs = []
    for i in range(3):
        im, idd = synth.make_inputs(cfg, tk, bs=1, seed=700 + i)
        reqs.append((idd[0], im[0], 5 + i, 900 + i))
    old = model.gemm_plan
    model.gemm_plan = plan
    try:
        solo = []
        for idd, im, n, seed in reqs:
            b = ContinuousBatcher(model, max_rows=4, max_len=1024)
            rid = b.submit(idd, im, max_new_tokens=n, seed=seed)
            b.run_until_done()
            solo.append(b.result(rid).tokens)
        b = ContinuousBatcher(model, max_rows=4, max_len=1024)
        rids = [b.submit(idd, im, max_new_tokens=n, seed=seed) for idd, im, n, seed in reqs]   # one admission batch of 3
        res = b.run_until_done()
        assert [res[r].tokens for r in rids] == solo
        b = ContinuousBatcher(model, max_rows=4, max_len=1024)
        rids = []
        for idd, im, n, seed in reqs:                                                             # staggered admission
            rids.append(b.submit(idd, im, max_new_tokens=n, seed=seed))
            b.step()
        res = b.run_until_done()
        assert [res[r].tokens for r in rids] == solo
    finally:
        model.gemm_plan = old
    assert all(len(s) > 0 for s in solo)
