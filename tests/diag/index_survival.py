"""Index parity on seeds NOBODY selected: the benchmarked build (`precision="hybrid"`: the ViT on operand pairs, the
proposer fp32) against the fp32 oracle running its OWN ViT, on CONSECUTIVE image seeds -- top-300 proposal ids
(R: groma/model/ddetr_transformer.py:546-559), NMS keep ids and the shuffled selection (R: groma/model/groma.py:266-276).

tests/test_e2e_unchained_gpu.py asserts torch.equal on the 3 seeds (of 600 / 240 scanned) whose oracle ranking has the largest
minimum gap.  Here nothing is chosen.  Per seed:
  gap   the oracle's smallest adjacent gap among its top-301 sorted class logits;
  err   max |device class logit - oracle class logit| over the 1024 proposals (two fp32 evaluations of the proposer: the device's
        v_mfma_f32 chains and the host BLAS sum in different orders);
  a seed RESOLVES when gap > 2 err -- then equal ids are a theorem, and the test asserts them; otherwise the two fp32 evaluations
  may legitimately order a near-tie differently (the reference's own CUDA and CPU paths would), and what is asserted instead is
  that the device's ranking is a VALID ranking of the oracle's logits within 2 err (every inversion is a near-tie) -- on EVERY seed.
Reported: fraction of seeds with all 300 ids equal / NMS ids equal / selection equal, fraction of equal slots, the distribution
of gap and err, the fraction that resolves at 2 err and at the 4 err the committed-seed tests use.  For the first `n64` seeds
the oracle is also evaluated in float64 ("truth" for the same fp32 parameters and pixels): how far each fp32 evaluation is
from it apportions err between the device and the oracle itself.

  python tests/diag/index_survival.py [n_seeds]   ->  gpurun_out/r06_index_survival.txt"""
import os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from groma_amd import constants, synth
from groma_amd.groma import GromaModel
from oracle import groma_oracle as O
from tests import util
from tests.golden.select_e2e_seeds import e2e_cfg

FIRST_SEED = 1000   # consecutive from here; nothing about these seeds was looked at before the script existed


def valid_ranking(dev_ids, o_cls, tol):
    """is `dev_ids` (a device top-k order) a top-k ranking of the oracle logits `o_cls` [S] up to `tol`: consecutive picks never
    ascend by more than tol, and nothing left out beats the last pick by more than tol"""
    v = o_cls[dev_ids]
    if v.numel() > 1 and (v[1:] - v[:-1]).max().item() > tol:
        return False
    rest = torch.ones_like(o_cls, dtype=torch.bool)
    rest[dev_ids] = False
    return (not rest.any()) or o_cls[rest].max().item() <= v.min().item() + tol


def oracle_f64(sd64, cd, images):
    torch.set_default_dtype(torch.float64)
    try:
        hs = O.vit_forward(sd64, cd, images.double())
        return O.ddetr_forward(sd64, cd, O.ddetr_inputs_from_hidden(hs))["enc_class"]
    finally:
        torch.set_default_dtype(torch.float32)


def run(name="tiny", n_seeds=100, precision="hybrid", n64=10, out=None):
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    cfg = e2e_cfg(name)
    sd = synth.make_state_dict(cfg, 0)
    cd = cfg.to_dict()
    tk = util.TokenIds()
    model = GromaModel.from_state_dict(cfg, sd, "cuda", precision=precision)
    model.init_special_token_id(constants.SyntheticTokenizer())
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items() if k.startswith("perceiver.")} if n64 else None
    Q = cfg.perceiver_cfg.ddetr_cfg.two_stage_num_proposals
    rows = []
    t0 = time.time()
    with torch.no_grad():
        for seed in range(FIRST_SEED, FIRST_SEED + n_seeds):
            images, ids = synth.make_inputs(cfg, tk, 1, seed=seed)
            torch.manual_seed(seed)
            per = O.perceive(sd, cd, images)                      # its own fp32 ViT -> proposer -> NMS -> randperm
            o_cls = per["det"]["enc_class"]
            torch.manual_seed(seed)
            hidden4, selected, aux = model.perceive(images)
            dbg = {}
            model.proposer.forward(hidden4, debug=dbg)
            d_cls = dbg["enc_class"].float().cpu()
            srt = torch.sort(o_cls, dim=1, descending=True)[0][:, : Q + 1]
            gaps = srt[0, :-1] - srt[0, 1:]
            d_ids, o_ids = aux["topk_idx"].cpu().long(), per["det"]["topk_idx"]
            err = (d_cls - o_cls).abs().max().item()
            r = dict(seed=seed, gap=gaps.min().item(), gap_med=gaps.median().item(), err=err,
                     err_rms=(d_cls - o_cls).pow(2).mean().sqrt().item(),
                     topk_equal=torch.equal(d_ids, o_ids), slots=(d_ids == o_ids).float().mean().item(),
                     set_overlap=len(set(d_ids[0].tolist()) & set(o_ids[0].tolist())) / Q,
                     valid=valid_ranking(d_ids[0], o_cls[0], 2 * err),
                     nms_equal=torch.equal(aux["nms_keep"][0], per["nms_inds"][0]),
                     sel_equal=per["perms"][0] is not None and torch.equal(aux["sel_idx"][0], per["nms_inds"][0][per["perms"][0]]),
                     boxes=torch.allclose(selected[0].float().cpu(), per["selected_boxes"][0], atol=1e-5) if selected[0].shape == per["selected_boxes"][0].shape else False)
            if len(rows) < n64:
                c64 = oracle_f64(sd64, cd, images)
                r.update(dev_vs_f64=(d_cls.double() - c64).abs().max().item(), orc_vs_f64=(o_cls.double() - c64).abs().max().item(),
                         dev_vs_f64_rms=(d_cls.double() - c64).pow(2).mean().sqrt().item(), orc_vs_f64_rms=(o_cls.double() - c64).pow(2).mean().sqrt().item())
            rows.append(r)
    n = len(rows)
    frac = lambda f: sum(1 for r in rows if f(r)) / n
    med = lambda k: statistics.median(r[k] for r in rows)
    s = dict(name=name, precision=precision, n=n, seconds=time.time() - t0,
             topk_all_equal=frac(lambda r: r["topk_equal"]), nms_equal=frac(lambda r: r["nms_equal"]), sel_equal=frac(lambda r: r["sel_equal"]),
             boxes_equal=frac(lambda r: r["boxes"]), slots_mean=sum(r["slots"] for r in rows) / n, set_overlap_mean=sum(r["set_overlap"] for r in rows) / n,
             valid_all=all(r["valid"] for r in rows), resolves_2err=frac(lambda r: r["gap"] > 2 * r["err"]), resolves_4err=frac(lambda r: r["gap"] > 4 * r["err"]),
             unresolvable_4err=frac(lambda r: r["gap"] < 4 * r["err"]), gap_median=med("gap"), gap_min=min(r["gap"] for r in rows), gap_max=max(r["gap"] for r in rows),
             adjacent_gap_median=med("gap_med"), err_median=med("err"), err_max=max(r["err"] for r in rows), err_rms_median=med("err_rms"),
             resolved_all_equal=all(r["topk_equal"] for r in rows if r["gap"] > 2 * r["err"]),
             nms_equal_given_topk=(sum(1 for r in rows if r["topk_equal"] and r["nms_equal"]) / max(1, sum(1 for r in rows if r["topk_equal"]))))
    r64 = [r for r in rows if "dev_vs_f64" in r]
    if r64:
        s.update(dev_vs_f64=statistics.median(r["dev_vs_f64"] for r in r64), orc_vs_f64=statistics.median(r["orc_vs_f64"] for r in r64),
                 dev_vs_f64_rms=statistics.median(r["dev_vs_f64_rms"] for r in r64), orc_vs_f64_rms=statistics.median(r["orc_vs_f64_rms"] for r in r64))

    def P(*a):
        print(*a, flush=True)
        if out is not None:
            print(*a, file=out, flush=True)
    P(f"\n## {name}, precision={precision!r}: {n} CONSECUTIVE image seeds {FIRST_SEED}..{FIRST_SEED + n - 1} (none selected), oracle runs its own fp32 ViT   [{s['seconds']:.0f} s]")
    P(f"all 300 top-k ids equal on {s['topk_all_equal']:.2f} of the seeds (mean fraction of equal slots {s['slots_mean']:.4f}, set overlap {s['set_overlap_mean']:.4f}); "
      f"NMS keep ids equal on {s['nms_equal']:.2f} ({s['nms_equal_given_topk']:.2f} of the seeds whose top-k ids are equal); shuffled selection equal on {s['sel_equal']:.2f}; "
      f"selected boxes equal (1e-5) on {s['boxes_equal']:.2f}")
    P(f"oracle min adjacent gap of the top-301 logits: median {s['gap_median']:.2e} (range {s['gap_min']:.1e} .. {s['gap_max']:.1e}; the committed fixtures: 1.3-2.0e-4); "
      f"median ADJACENT gap {s['adjacent_gap_median']:.2e}")
    P(f"device class-logit error vs the fp32 oracle: max-abs median {s['err_median']:.2e} (worst seed {s['err_max']:.2e}), rms median {s['err_rms_median']:.2e}")
    P(f"seeds that resolve: gap > 2 err on {s['resolves_2err']:.2f} (ids equal on every one of them: {s['resolved_all_equal']}), gap > 4 err on {s['resolves_4err']:.2f}; "
      f"unresolvable by the committed tests' rule (gap < 4 err): {s['unresolvable_4err']:.2f}")
    P(f"the device ranking is a valid ranking of the oracle's logits within 2 err on EVERY seed: {s['valid_all']}")
    if r64:
        P(f"apportioning err (first {len(r64)} seeds, the oracle re-evaluated in float64 on the same fp32 parameters): device vs f64 max-abs {s['dev_vs_f64']:.2e} "
          f"(rms {s['dev_vs_f64_rms']:.2e}), fp32 oracle vs f64 {s['orc_vs_f64']:.2e} (rms {s['orc_vs_f64_rms']:.2e})")
    worst = sorted(rows, key=lambda r: r["slots"])[:3]
    P("worst seeds: " + "; ".join(f"seed {r['seed']}: slots {r['slots']:.3f}, gap {r['gap']:.1e}, err {r['err']:.1e}, nms {r['nms_equal']}" for r in worst))
    del model
    torch.cuda.empty_cache()
    return s, rows


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_index_survival.txt"), "w") as f:
        print("# index parity on unselected seeds (tests/diag/index_survival.py)", file=f)
        for name in ("tiny", "width"):
            for precision in ("hybrid", "ref") if name == "tiny" else ("hybrid",):
                run(name, n, precision, out=f)
