"""BASELINE configs[3]: greedy generate at 4 images/GPU, eager launches vs the captured hipGraph step."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from groma_amd import config, constants, synth, ops
from groma_amd.groma import GromaModel
cfg = config.groma_7b(box_score_thres=0.0)
m = GromaModel.from_synthetic(cfg, seed=0, device='cuda')
m.init_special_token_id(constants.SyntheticTokenizer())
m.generation_config.eos_token_id = None  # random weights: never stop early
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
images, ids = synth.make_inputs(cfg, m, bs, seed=5)
images, ids = images.cuda(), ids.cuda()
res = {}
for graph in (False, True):
    m.decode_graph = graph
    for new in (1, 33):
        torch.manual_seed(0); a = m.generate(ids, images=images, max_new_tokens=new); torch.cuda.synchronize()
        t = time.perf_counter(); torch.manual_seed(0); b = m.generate(ids, images=images, max_new_tokens=new); torch.cuda.synchronize()
        res[(graph, new)] = (time.perf_counter() - t, b)
    dt1, dt33 = res[(graph, 1)][0], res[(graph, 33)][0]
    print(f"bs={bs} graph={graph}: prefill+1 {dt1*1e3:.1f} ms, +32 tokens {dt33*1e3:.1f} ms -> {(dt33-dt1)/32*1e3:.2f} ms/token, "
          f"generate(32 new) {bs/dt33:.2f} img/s", flush=True)
print("same tokens:", torch.equal(res[(False, 33)][1], res[(True, 33)][1]))
