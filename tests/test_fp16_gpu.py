"""The fp16 operand build of the library (libgroma_hip_f16.so: the same sources and C ABI compiled with -DGR_F16, selected by
GromaModel(precision="fp16") / from_pretrained(torch_dtype=torch.float16)) -- the dtype the reference's own inference entry
points autocast to (R: groma/eval/run_groma.py:82, groma/serve/model_worker.py:256).

 * the per-kernel numerics tests of tests/test_kernels_gpu.py re-run with half operands through the f16 library;
 * half carries 3 more mantissa bits than bf16, so the same comparisons hold at ~8x tighter tolerances: asserted for the GEMM,
   attention and norm kernels, for the chained tiny forward (vs the fp32 oracle and vs the oracle with fp16 rounding points),
   and for the Groma-7B-width forward of smoke();
 * index-valued results (NMS ids, shuffled order, spliced ids) and every greedy token equal the oracle's;
 * a bf16 and an fp16 model live in one process (two libraries, two dtypes) without disturbing each other."""
import pytest
import torch

from oracle import groma_oracle as O
from tests import util
from tests import test_kernels_gpu as K

pytestmark = pytest.mark.gpu

TOL_F16_OUT = 6e-4   # rel-L2 of a half-rounded output (unit roundoff 4.9e-4, uniform -> ~2.8e-4 rms; bf16: 4e-3)


@pytest.fixture
def fp16(monkeypatch):
    """inside: ops.* launch from the f16 library and the kernel tests' `.bfloat16()` / `torch.bfloat16` operands become half"""
    from groma_amd import ops
    monkeypatch.setattr(torch.Tensor, "bfloat16", lambda self, *a, **k: self.half())
    monkeypatch.setattr(torch, "bfloat16", torch.float16)
    with ops.precision("fp16"):
        yield ops


def test_library_is_the_f16_build(fp16):
    from groma_amd import _lib
    lib = _lib.load()
    assert lib.gr_operand_type() == 1 and lib is _lib.load("fp16") and lib is not _lib.load("bf16")
    assert fp16.H16() == torch.float16


def test_kernel_suite_with_half_operands(dev, fp16):
    """the bf16 kernel tests, unchanged, on half operands (their references are built from the 16-bit inputs' exact values)"""
    for M, N, Kd in [(582, 4096, 1024), (100, 260, 192), (1025, 3072, 1024), (1, 128, 4096)]:
        for tile in (128, 256):
            K.test_gemm_plain(dev, M, N, Kd, tile)
    for shape in [(582, 4096, 11008), (512, 256, 1280), (300, 512, 1216)]:
        K.test_gemm_fp32_residual_large_shapes(dev, *shape)
    for tile in (128, 256, 192):
        K.test_gemm_epilogues(dev, tile)
        K.test_gemm_conv3x3(dev, 3, 14, 64, 64, 3, tile)
    K.test_gemm_decode_shape(dev, 4, 32128, 4096)
    K.test_gemm_decode_shape(dev, 3, 1024, 11008)
    for C in (256, 1024, 4096, 768):
        K.test_norms(dev, C)
    K.test_qkv_split(dev, 64, False)
    K.test_qkv_split(dev, 128, True)
    K.test_vit_packing(dev)
    K.test_gn_shuffle(dev)
    K.test_small_movers(dev)
    K.test_gemv_fused_operand_and_epilogue_modes(dev, 4, 4096, 4096)
    K.test_gemv_fused_operand_and_epilogue_modes(dev, 8, 512, 11008)
    K.test_gemv_fused_qkv_rope_matches_prefill_split(dev, 128, True, 4)
    K.test_gemv_fused_merges_attention_key_slices(dev)
    K.test_decode_attention(dev, 4, 32, 128, 583, "host", None)
    K.test_decode_attention(dev, 2, 8, 64, 70, "host", 1)
    K.test_attention_reads_q_in_place(dev, 2, 4, 128, 150, True, True)
    K.test_attention_reads_q_in_place(dev, 3, 2, 64, 70, False, False)


def test_attention_cases_with_half_operands(dev, fp16):
    import inspect
    marks = [m for m in K.test_attention.pytestmark if m.name == "parametrize"]
    names = [n.strip() for n in marks[0].args[0].split(",")]
    assert names == list(inspect.signature(K.test_attention).parameters)[1:]
    for case in marks[0].args[1]:
        K.test_attention(dev, *case)


def test_half_outputs_are_tighter_than_bf16(dev, fp16):
    ops = fp16
    a = K.rnd((777, 1024), dev, seed=1).half()
    w = K.rnd((640, 1024), dev, 0.05, seed=2).half()
    ref = a.float() @ w.float().t()
    for tile in (128, 256):
        assert K.relerr(ops.gemm(a, w, tile=tile), ref) < TOL_F16_OUT
        assert K.relerr(ops.gemm(a, w, out_f32=True, tile=tile), ref) < 1e-5
    x = K.rnd((300, 4096), dev, 2.0, seed=3)
    g = K.rnd((4096,), dev, 1.0, seed=4)
    ref = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * g
    assert K.relerr(ops.rmsnorm(x, g, 1e-6), ref) < TOL_F16_OUT
    # attention: causal hd 128 with RoPE-free packed operands
    B, H, L, hd = 2, 4, 200, 128
    q, k, v = (K.rnd((B, H, L, hd), dev, 1.0, seed=s).half() for s in (5, 6, 7))
    Sp = 256
    kp = torch.zeros((B, H, Sp, hd), dtype=torch.float16, device=dev)
    vt = torch.zeros((B, H, hd, Sp), dtype=torch.float16, device=dev)
    kp[:, :, :L], vt[..., :L] = k, v.transpose(2, 3)
    out = ops.attention(q, kp, vt, Skv=L, causal=True)
    ref = K._attn_ref(q.float(), k.float(), v.float(), True, 0, None)
    assert K.relerr(out, ref) < 1.5e-3   # the bf16 build is held to 1e-2 (test_attention)


def test_saturation_instead_of_inf(dev, fp16):
    """conversions to half saturate at +-65504 (a bf16 model never overflows; an fp16 one must not turn a large activation into inf)"""
    ops = fp16
    a = torch.full((128, 64), 200.0, device=dev).half()
    w = torch.full((128, 64), 200.0, device=dev).half()
    out = ops.gemm(a, w, tile=128)          # 64 * 200 * 200 = 2.56e6 > 65504
    assert torch.isfinite(out.float()).all() and float(out.float().max()) == 65504.0
    assert float(ops.gemm(a, w, out_f32=True, tile=128)[0, 0]) == 2.56e6


@pytest.fixture(scope="module")
def models(dev):
    cfg, sd, tk = util.tiny_setup(seed=0)
    sd = dict(sd)
    w = sd["extra_lm_head.weight"].clone()
    w[w.shape[0] - 100:] *= 40.0   # clear greedy margins (tests/test_parity_gpu.py::gen_setup)
    sd["extra_lm_head.weight"] = w
    from groma_amd import constants, synth
    from groma_amd.groma import GromaModel
    m16 = GromaModel.from_state_dict(cfg, sd, "cuda", precision="fp16")
    mbf = GromaModel.from_state_dict(cfg, sd, "cuda")
    for m in (m16, mbf):
        m.init_special_token_id(constants.SyntheticTokenizer())
    images, ids = synth.make_inputs(cfg, tk, bs=2, seed=1234)
    return cfg, sd, tk, m16, mbf, images, ids


def _fwd(m, ids, images, seed=77):
    torch.manual_seed(seed)
    out = m.forward(input_ids=ids.clone(), images=images, return_dict=True, output_hidden_states=True, use_cache=True)
    hs = tuple(m._ws.get(f"vit_h{i}", (ids.shape[0], m.vit.T, m.vit.D), torch.float32).cpu() for i in range(4))
    return out, hs


def test_tiny_forward_fp16_vs_oracles(models):
    cfg, sd, tk, m16, mbf, images, ids = models
    out, hs = _fwd(m16, ids, images)
    assert out.past_key_values[0][0].dtype == torch.float16
    aux = m16._last_aux
    torch.manual_seed(77)
    ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=hs)
    torch.manual_seed(77)
    with O.rounding("fp16"):
        ref16 = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=hs)
    for i in range(2):
        assert torch.equal(aux["nms_keep"][i], ref["nms_inds"][i])
        assert torch.equal(aux["sel_idx"][i], ref["nms_inds"][i][ref["perms"][i]])
    assert torch.equal(aux["input_ids"], ref["input_ids"])
    vis = out.hidden_states[1]
    e = dict(img=util.relerr(vis["image_features"], ref["image_features"]), reg=util.relerr(vis["region_features"], ref["region_features"]),
             logits=util.relerr(out.logits, ref["logits"]), logits_vs_fp16_oracle=util.relerr(out.logits, ref16["logits"]),
             kv=util.relerr(out.past_key_values[0][0], ref["past"][0][0]))
    # the same comparison for the bf16 model of the same weights
    out_b, hs_b = _fwd(mbf, ids, images)
    torch.manual_seed(77)
    ref_b = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, hidden_states=hs_b)
    e_b = util.relerr(out_b.logits, ref_b["logits"])
    print("fp16 rel-L2 vs fp32 oracle:", {k: f"{v:.2e}" for k, v in e.items()}, "| bf16 logits:", f"{e_b:.2e}")
    assert e["img"] < 2.5e-3 and e["reg"] < 2.5e-3 and e["logits"] < 2.5e-3 and e["kv"] < 2.5e-3   # bf16 bound: 2e-2
    assert e["logits_vs_fp16_oracle"] < 2.5e-3
    assert e["logits"] < 0.5 * e_b
    assert mbf.llm.w["head"].dtype == torch.bfloat16 and m16.llm.w["head"].dtype == torch.float16


@pytest.mark.parametrize("graph", [True, False])
def test_tiny_generate_fp16_matches_oracle_greedy(models, graph):
    cfg, sd, tk, m16, mbf, images, ids = models
    n = 6
    gc = m16.generation_config
    old = (gc.eos_token_id, m16.decode_graph)
    try:
        gc.eos_token_id, m16.decode_graph = None, graph
        torch.manual_seed(9)
        g = m16.generate(ids.clone(), images=images, max_new_tokens=n, return_dict_in_generate=True, output_hidden_states=True)
    finally:
        gc.eos_token_id, m16.decode_graph = old
    hs = tuple(m16._ws.get(f"vit_h{i}", (2, m16.vit.T, m16.vit.D), torch.float32).cpu() for i in range(4))
    torch.manual_seed(9)
    ref = O.greedy_generate(sd, cfg.to_dict(), util.tok_dict(tk), ids.clone(), images, n, eos_token_id=-1, hidden_states=hs)
    P = ids.shape[1]
    ncmp = util.assert_greedy_tokens_match(g.sequences[:, P:].cpu(), ref["sequences"][:, P:], ref["margins"], 0.01, "fp16 generate")
    print("fp16 generated", g.sequences[:, P:].tolist(), "oracle", ref["sequences"][:, P:].tolist(), "compared", ncmp)
    assert ncmp >= n


def test_width_forward_fp16_vs_fp32_oracle(dev):
    """smoke()'s check (Groma-7B width, reduced depth, one image) for the fp16 build: bf16 measures 7.8e-3 there"""
    from groma_amd import config, constants, synth
    from groma_amd.groma import GromaModel
    cfg = config.groma_7b_width(box_score_thres=0.0)
    sd = synth.make_state_dict(cfg, 0)
    m = GromaModel.from_state_dict(cfg, sd, "cuda", precision="fp16")
    m.init_special_token_id(constants.SyntheticTokenizer())
    images, ids = synth.make_inputs(cfg, m, bs=1, seed=1)
    torch.manual_seed(3)
    out = m.forward(input_ids=ids.clone(), images=images, return_dict=True)
    hs = tuple(h.float().cpu() for h in m._last_aux["hidden4"])
    torch.manual_seed(3)
    with torch.no_grad():
        ref = O.groma_forward(sd, cfg.to_dict(), util.tok_dict(m), ids.clone(), images, hidden_states=hs)
    assert torch.equal(m._last_aux["nms_keep"][0], ref["nms_inds"][0])
    assert torch.equal(m._last_aux["input_ids"], ref["input_ids"]) and ref["input_ids"].shape[1] == 582
    err = util.relerr(out.logits, ref["logits"])
    print(f"Groma-7B width, fp16 operands: logits rel-L2 vs fp32 oracle {err:.2e}")
    assert err < 2e-3
    amax = (out.logits.float().cpu().argmax(-1) == ref["logits"].argmax(-1)).float().mean().item()
    print(f"arg-max agreement over all {ref['logits'].shape[1]} positions: {amax:.4f}")
    assert amax > 0.97
