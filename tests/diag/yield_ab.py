"""A/B of engine.YIELD_PYRAMID (the region pyramid's GEMMs as one workgroup per tile vs persistent grids) on the benchmark step, one model,
settings alternating in one process:   python tests/diag/yield_ab.py [batch] [precision]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import config, constants, engine, synth
from groma_amd.groma import GromaModel

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 14
prec = sys.argv[2] if len(sys.argv) > 2 else "hybrid-fp16"
cfg = config.groma_7b(box_score_thres=0.0)
m = GromaModel.from_synthetic(cfg, seed=0, device="cuda", precision=prec)
m.init_special_token_id(constants.SyntheticTokenizer())
images, ids = synth.make_inputs(cfg, m, batch, seed=1234)
images, ids = images.cuda(), ids.cuda()


def run(n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        torch.manual_seed(1000 + i)
        m.forward(input_ids=ids, images=images, return_dict=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {True: [], False: []}
for flag in (True, False):
    engine.YIELD_PYRAMID = flag
    run(4)   # captures
for rep in range(4):
    for flag in (True, False):
        engine.YIELD_PYRAMID = flag
        res[flag].append(run(8))
for flag in (True, False):
    v = res[flag]
    print(f"{prec} batch {batch} YIELD_PYRAMID={flag}: {[round(x, 2) for x in v]} ms per step, median {sorted(v)[len(v) // 2]:.2f} -> {batch / sorted(v)[len(v) // 2] * 1e3:.2f} img/s")
