"""latency plan in the operand-pair build: split-K decided on physical K-steps (ops.PAIR_PLAN_PHYSICAL_K) vs logical ones -- the hybrid ViT of ONE image
and the whole one-image forward, one setting per process.   python tests/diag/pair_plan_ab.py 1|0"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import config, constants, ops, synth
from groma_amd.groma import GromaModel

cfg = config.groma_7b(box_score_thres=0.0)
m = GromaModel.from_synthetic(cfg, seed=0, device="cuda", precision="hybrid-fp16")
m.init_special_token_id(constants.SyntheticTokenizer())
m.gemm_plan = "latency"
images, ids = synth.make_inputs(cfg, m, 1, seed=1234)
images, ids = images.cuda(), ids.cuda()


def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        torch.manual_seed(1000 + i)
        m.forward(input_ids=ids, images=images, return_dict=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


flag = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"   # (one setting per process: the captured ViT graph is keyed on the plan's NAME)
ops.PAIR_PLAN_PHYSICAL_K = flag
run(6)
v = sorted(run(10) for _ in range(5))
with ops.precision("ref"), ops.gemm_plan("latency"):
    sp = [ops.plan_splits(N, K) for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))]
print(f"PAIR_PLAN_PHYSICAL_K={flag}: ViT splits qkv/proj/fc1/fc2 = {sp}; one image per call, latency plan: {[round(x, 2) for x in v]} ms -> {1e3 / v[2]:.2f} img/s")
