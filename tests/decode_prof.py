"""decode-only kernel profile helper: prefill once, then N eager decode steps (run under rocprofv3 --kernel-trace --stats)"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from groma_amd import config, constants, synth
from groma_amd.groma import GromaModel
cfg = config.groma_7b(box_score_thres=0.0)
m = GromaModel.from_synthetic(cfg, seed=0, device='cuda')
m.init_special_token_id(constants.SyntheticTokenizer())
m.generation_config.eos_token_id = None
m.decode_graph = False
images, ids = synth.make_inputs(cfg, m, 4, seed=5)
torch.manual_seed(0)
m.generate(ids.cuda(), images=images.cuda(), max_new_tokens=65)
torch.cuda.synchronize()
