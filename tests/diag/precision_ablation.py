"""Which stage's operand rounding owns the logit error of the 16-bit builds -- full DEPTH, UNCHAINED.

The reference is one fp32 pass (R: groma/eval/eval_rec.py:69, groma/model/groma.py:389-402).  `precision="hybrid"` holds its
index-valued results; its logits sit at the bf16 format's distance (2.6e-2 at depth), "hybrid-fp16" at 3.3e-3, "ref" at 1.7e-5 for
2.5x the step.  This script measures, on Groma-7B at its real depth with distinct weights per layer (the set-up of
tests/diag/fulldepth_parity.py: 24 ViT layers, 6+6 DDETR, 5 fusion rounds, 100 regions, 32 LLaMA layers, 582 positions):

 A. ORACLE side (CPU, fine-grained): the fp32 oracle with the 16-bit roundings switched on for ONE group of rounding points at a
    time (oracle.groma_oracle.rounding(mode, only=...)): what that group alone costs the logits, and what is left when it alone is
    exact (skip=...).  The oracle's rounded modes reproduce the device's distance from fp32 to within a few per cent at every
    stage (tests/test_fulldepth_parity_gpu.py, ratio 1.00), so this is the device's error budget without building 16 libraries.
 B. DEVICE side (the real kernels, coarse): `precision="<base>+<stage>:ref..."` (groma_amd.groma.parse_precision) -- ONE stage at a
    time moved to operand pairs, logits against the same unchained fp32 oracle pass, arg-max agreement, index-valued results, and
    what the stage costs (14-image step time).

Output: profiles/r06_precision_ablation.txt.  ~25 minutes on the GPU box (mostly oracle passes on the host cores)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from groma_amd import config as gconfig, constants, synth
from groma_amd.groma import GromaModel
from oracle import groma_oracle as O
from tests import util
from fulldepth_parity import LazyDeviceStateDict

# oracle-side groups: name -> patterns of oracle.groma_oracle's "<stage>.<kind>" keys
FINE = [
    ("bridge (2 GEMMs)", ("bridge",)),
    ("region: 1x1 input convs", ("region.in",)),
    ("region: 5 fusion rounds (3x3 convs + stored maps)", ("region.fuse",)),
    ("region: per-ROI conv", ("region.pconv",)),
    ("region: flatten + updims linears", ("region.flat", "region.up")),
    ("embedding tables", ("embed",)),
    ("LLaMA: RMSNorm-1 output (QKV operand)", ("llm.qkv.a",)),
    ("LLaMA: QKV weights", ("llm.qkv.w",)),
    ("LLaMA: stored q / k / v incl. RoPE (QK^T, PV operands)", ("llm.qkv.o",)),
    ("LLaMA: soft-max probabilities + context", ("llm.pv",)),
    ("LLaMA: o-proj weights", ("llm.o.w",)),
    ("LLaMA: RMSNorm-2 output (gate/up operand)", ("llm.gateup.a",)),
    ("LLaMA: gate/up weights", ("llm.gateup.w",)),
    ("LLaMA: SwiGLU output (down operand)", ("llm.down.a",)),
    ("LLaMA: down weights", ("llm.down.w",)),
    ("lm_head (+) extra head (final norm output + weights)", ("head",)),
]
COARSE = [   # = the device's stages (groma_amd.groma.STAGES behind the ViT)
    ("region", ("region",)), ("bridge", ("bridge",)), ("embed", ("embed",)),
    ("attn  (qkv + pv + o)", ("llm.qkv", "llm.pv", "llm.o")), ("mlp   (gateup + down)", ("llm.gateup", "llm.down")),
    ("head", ("head",)),
    ("all LLaMA weights", ("llm.qkv.w", "llm.o.w", "llm.gateup.w", "llm.down.w", "head.w")),
    ("all LLaMA activations", ("llm.qkv.a", "llm.qkv.o", "llm.pv", "llm.o.a", "llm.gateup.a", "llm.down.a", "head.a")),
]
BEHIND = ("bridge", "region", "embed", "llm", "head")

DEVICE_SPECS = [
    "hybrid", "hybrid+attn:ref", "hybrid+mlp:ref",
    "hybrid-fp16", "hybrid-fp16+head:ref", "hybrid-fp16+bridge:ref", "hybrid-fp16+region:ref",
    "hybrid-fp16+attn:ref", "hybrid-fp16+mlp:ref", "hybrid-fp16+attn:ref+mlp:ref",
    "hybrid-fp16+attn:ref+mlp:ref+head:ref+region:ref+bridge:ref", "ref",
]


RK = [slice(32014, 32114)]   # columns of the <r0..r99> logits (run() sets it from the token table)


def _stats(lg, ref):
    err = (lg - ref).abs().max().item()
    top2 = ref.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4 * err
    agree = lg.argmax(-1) == ref.argmax(-1)
    return dict(rel=util.relerr(lg, ref), rel_region=util.relerr(lg[:, -1, RK[0]], ref[:, -1, RK[0]]), max_abs=err,
                argmax=agree.float().mean().item(), clear=clear.float().mean().item(),
                argmax_clear=agree[clear].float().mean().item() if clear.any() else float("nan"))


def run(out=sys.stdout, oracle_side=True, device_side=True, specs=None, bench_batch=14, seed=0, cfg=None, sd=None):
    """cfg / sd: another configuration + state dict (the CPU dry run of tests/test_oracle_rounding_modes.py uses the tiny one)"""
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    full = cfg if cfg is not None else gconfig.groma_7b(box_score_thres=0.0)
    dev = torch.device("cuda") if device_side or sd is None else None
    tk = util.TokenIds()
    RK[0] = slice(min(tk.box_idx_token_ids), max(tk.box_idx_token_ids) + 1)
    images, ids = synth.make_inputs(full, tk, 1, seed=1234)
    if sd is None:
        sd = LazyDeviceStateDict(full, seed, dev)
    cd = full.to_dict()

    def P(*a):
        print(*a, file=out, flush=True)
        if out is not sys.stdout:
            print(*a, flush=True)

    P("# precision ablation, Groma-7B at full depth (24 / 6+6 / 5 / 32), distinct weights per layer, one image, 582 positions, UNCHAINED")
    P("# reference = the fp32 oracle running its own fp32 ViT (R: groma/eval/eval_rec.py:69, groma/model/groma.py:222-280,389-402)")
    with torch.no_grad():
        t = time.time()
        own = O.vit_forward(sd, cd, images)
        torch.manual_seed(77)
        r32 = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(own))
        ref = r32["logits"]
        P(f"# fp32 oracle pass: {time.time() - t:.1f} s on {torch.get_num_threads()} host threads; L = {r32['input_ids'].shape[1]}, max |logit| {ref.abs().max().item():.2f}")

        def opass(mode, **sel):
            torch.manual_seed(77)
            with O.rounding(mode, **sel):
                r = O.groma_forward(sd, cd, util.tok_dict(tk), ids.clone(), images, hidden_states=tuple(own))
            assert torch.equal(r["input_ids"], r32["input_ids"])
            return _stats(r["logits"], ref)

        res = {"oracle": {}, "device": {}}
        if oracle_side:
            P("\n## A. oracle side: 16-bit roundings of ONE group of rounding points switched on (everything else fp32; the ViT is fp32 throughout = the hybrid builds)")
            for fmt in ("fp16", "bf16"):
                t = time.time()
                whole = opass(fmt, only=BEHIND)
                P(f"\n### {fmt}: every rounding point behind the ViT (= what \"hybrid{'-fp16' if fmt == 'fp16' else ''}\" rounds): logits rel-L2 {whole['rel']:.3e}, "
                  f"<r_k> logits {whole['rel_region']:.3e}, arg-max equal at {whole['argmax']:.3f} of positions   [{time.time() - t:.0f} s per pass]")
                P(f"{'group (only this one rounds)':64s} {'logits rel-L2':>13s} {'share of var.':>13s} {'arg-max':>8s}")
                rows = {}
                for name, pats in (FINE if fmt == "fp16" else []) + COARSE:
                    s = opass(fmt, only=pats)
                    rows[name] = s
                    P(f"{name:64s} {s['rel']:13.3e} {(s['rel'] / whole['rel']) ** 2:13.3f} {s['argmax']:8.3f}")
                res["oracle"][fmt] = dict(whole=whole, rows=rows)
                if fmt == "fp16":
                    P("(share of variance = (group / whole)^2; the fine groups' shares sum to "
                      f"{sum((rows[n]['rel'] / whole['rel']) ** 2 for n, _ in FINE):.2f})")
                    for name, pats in (("attn exact (pairs), rest fp16", ("llm.qkv", "llm.pv", "llm.o")),
                                       ("mlp exact, rest fp16", ("llm.gateup", "llm.down")),
                                       ("attn + mlp exact, rest fp16", ("llm",)),
                                       ("all weights exact (2-pass: activation pairs x fp16 ... upper bound of its gain)",
                                        ("llm.qkv.w", "llm.o.w", "llm.gateup.w", "llm.down.w", "head.w"))):
                        s = opass(fmt, only=BEHIND, skip=pats)
                        P(f"{'[skip] ' + name:64s} {s['rel']:13.3e} {'':13s} {s['argmax']:8.3f}")
                        res["oracle"][fmt]["skip " + name] = s
        if device_side:
            P("\n## B. device side: one stage at a time on operand pairs (groma_amd.groma.parse_precision), same inputs, same unchained fp32 oracle pass")
            P(f"{'precision':62s} {'logits rel-L2':>13s} {'<r_k> logits':>12s} {'arg-max':>8s} {'clear-margin':>12s} {'ids':>5s} {'ms/14 img':>10s} {'img/s':>7s}")
            im14, id14 = synth.make_inputs(full, tk, bench_batch, seed=1234)
            im14, id14 = im14.to(dev), id14.to(dev)
            for spec in (specs or DEVICE_SPECS):
                t = time.time()
                model = GromaModel.from_synthetic(full, seed=seed, device=dev, precision=spec)
                model.init_special_token_id(constants.SyntheticTokenizer())
                torch.manual_seed(77)
                o = model.forward(input_ids=ids.clone(), images=images, return_dict=True)
                aux = model._last_aux
                s = _stats(o.logits.float().cpu(), ref)
                ids_eq = (torch.equal(aux["topk_idx"].cpu().long(), r32["det"]["topk_idx"]) and torch.equal(aux["nms_keep"][0], r32["nms_inds"][0])
                          and torch.equal(aux["input_ids"], r32["input_ids"]))
                for _ in range(4):
                    model.forward(input_ids=id14, images=im14, return_dict=True)
                torch.cuda.synchronize()
                t0 = time.time()
                n = 5
                for _ in range(n):
                    model.forward(input_ids=id14, images=im14, return_dict=True)
                torch.cuda.synchronize()
                ms = (time.time() - t0) / n * 1e3
                P(f"{model.mode:62s} {s['rel']:13.3e} {s['rel_region']:12.3e} {s['argmax']:8.3f} {s['argmax_clear']:6.3f}@{s['clear']:5.3f} {str(ids_eq):>5s} {ms:10.1f} {bench_batch / ms * 1e3:7.1f}"
                  f"   [{time.time() - t:.0f} s]")
                res["device"][spec] = dict(s, ids_equal=ids_eq, ms=ms)
                del model, o, aux
                torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    a = sys.argv[1:]
    with open(os.path.join(ROOT, "gpurun_out", "r06_precision_ablation.txt"), "w") as f:
        run(out=f, oracle_side="--no-oracle" not in a, device_side="--no-device" not in a)
