"""Continuous batching throughput on Groma-7B (random init): R requests x T new tokens through max_rows slots.
   python tests/serve_bench.py [--fp8] [--rows 4,8,16,32] [--precision bf16] [--requests N] [--no-overlap] [--ragged] [--admit-min K] [--plan latency]
   (--ragged: 16..48 new tokens per request (mean 32) instead of 32 each, so slots free up -- and admissions happen -- in the middle of live traffic;
    --no-overlap: admission prefills between the decode ticks on the same stream, ContinuousBatcher(overlap_admission=False) -- the default
    since round 6 runs them on a worker thread + side stream while the live rows decode)
   (--fp8: e4m3 weights + activations, the batcher on the e4m3 decode streams;
    rows > 8: the matrix-unit weight streams, csrc/gemm_skinny.hip / gemm_skinny_fp8.hip -- round 6)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groma_amd import config, constants, synth
from groma_amd.groma import GromaModel
from groma_amd.serving import ContinuousBatcher


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


cfg = config.groma_7b(box_score_thres=0.0)
FP8 = '--fp8' in sys.argv
prec = arg('--precision', 'bf16')
m = GromaModel.from_synthetic(cfg, seed=0, device='cuda', fp8=FP8, precision=prec)
m.init_special_token_id(constants.SyntheticTokenizer())
m.generation_config.eos_token_id = None
m.gemm_plan = arg('--plan', 'throughput')   # 'latency': the under-filled GEMMs of small admission waves are split along K
T = 32
RAGGED = '--ragged' in sys.argv
rows_list = [int(x) for x in arg('--rows', '4,8,16,32').split(',')]
for rows in rows_list:
    R = int(arg('--requests', max(16, 2 * rows)))   # two admission waves per slot
    images, ids = synth.make_inputs(cfg, m, R, seed=5)
    images = images.cuda()
    b = ContinuousBatcher(m, max_rows=rows, max_len=1024, overlap_admission='--no-overlap' not in sys.argv, admit_min=int(arg('--admit-min', 2)))
    b.step()  # capture
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(R):
            b.submit(ids[i], images[i], max_new_tokens=(16 + (i * 7) % 33) if RAGGED else T, seed=i)
        res = b.run_until_done()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        steps, ntok = b.steps, sum(len(r.tokens) for r in res.values())
        for rid in list(res): b.result(rid)
    # one decode tick alone (all rows occupied is not needed: idle rows decode too)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        b.graph.replay()
    torch.cuda.synchronize(); tick = (time.perf_counter() - t) / 20
    print(("e4m3 " if FP8 else "") + ("" if b.overlap else "admission between ticks (--no-overlap) ") + f"{m.mode} max_rows={rows}" + (f" gemm_plan={m.gemm_plan}" if m.gemm_plan != "throughput" else "") + (f" admit_min={b.admit_min}" if b.admit_min != 2 else "") + f": {R} requests x {'16..48' if RAGGED else T} tokens in {dt*1e3:.0f} ms -> {R/dt:.1f} img/s, {ntok/dt:.0f} tok/s, "
          f"{steps} decode steps total; one decode tick {tick*1e3:.2f} ms = {rows/tick:.0f} tok/s at full occupancy", flush=True)
    del b
    torch.cuda.empty_cache()
