"""Continuous batching throughput on Groma-7B (random init): R requests x T new tokens through max_rows slots.
   python tests/serve_bench.py [--fp8]   (e4m3 weights + activations: the batcher on the e4m3 decode streams)"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from groma_amd import config, constants, synth
from groma_amd.groma import GromaModel
from groma_amd.serving import ContinuousBatcher
cfg = config.groma_7b(box_score_thres=0.0)
FP8 = '--fp8' in sys.argv
m = GromaModel.from_synthetic(cfg, seed=0, device='cuda', fp8=FP8)
m.init_special_token_id(constants.SyntheticTokenizer())
m.generation_config.eos_token_id = None
R, T = 16, 32
images, ids = synth.make_inputs(cfg, m, R, seed=5)
images = images.cuda()
for rows in (4, 8):
    b = ContinuousBatcher(m, max_rows=rows, max_len=1024)
    b.step()  # capture
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(R):
            b.submit(ids[i], images[i], max_new_tokens=T, seed=i)
        res = b.run_until_done()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        for rid in list(res): b.result(rid)
    print(("e4m3 " if FP8 else "") + f"max_rows={rows}: {R} requests x {T} tokens in {dt*1e3:.0f} ms -> {R/dt:.1f} img/s, {R*T/dt:.0f} tok/s, {b.steps} decode steps total", flush=True)
