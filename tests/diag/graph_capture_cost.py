"""What a GraphPool capture costs next to an eager pass and a replay (Groma-7B, one image): decides CAPTURE_AT / cap."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import config as gconfig, constants, engine, synth
from groma_amd.groma import GromaModel

cfg = gconfig.groma_7b()
m = GromaModel.from_synthetic(cfg, seed=0, device=torch.device("cuda"))
tk = constants.SyntheticTokenizer()
m.init_special_token_id(tk)
for bs in (1, 4):
    images, ids = synth.make_inputs(cfg, m, bs, seed=1, prompt_len=128)
    images, ids = images.cuda(), ids.cuda()
    ts = []
    for i in range(7):
        torch.manual_seed(3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.forward(input_ids=ids.clone(), images=images, return_dict=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"bs={bs}: ms per call (1-2 eager, 3 capture+replay, 4.. replay):", " ".join(f"{t:.1f}" for t in ts),
          "| captures vit/llm", m.vit.graphs.captures, m.llm.graphs.captures)
