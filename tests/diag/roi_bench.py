"""roi_align_pack at the benchmark's shapes (14 images, 100 boxes each, three pyramid levels of 1024 channels), one build of the
library:  GROMA_HIP_LIB=<lib.so> python tests/diag/roi_bench.py [tag]   (boxes: the synthetic proposer's 0.05-wide boxes and a
second set 0.05-0.4 wide, both read as x1y1x2y2 the way the reference does -- T1)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
import torch
from groma_amd import ops
dev = "cuda"
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
bs, n, D, P = 14, 100, 1024, 14
S = [128, 64, 32]
g = torch.Generator(device=dev).manual_seed(0)
feats = [torch.randn((bs, s, s, D), device=dev, generator=g).relu().bfloat16() for s in S]
img = torch.arange(bs, device=dev).repeat_interleave(n).float()
for name, wlo, whi in (("0.05-wide boxes", 0.05, 0.05), ("0.05-0.4-wide boxes", 0.05, 0.4)):
    c = torch.rand((bs * n, 2), device=dev, generator=g)
    wh = wlo + (whi - wlo) * torch.rand((bs * n, 2), device=dev, generator=g)
    rois = torch.cat([img[:, None], torch.cat([c, wh], 1) * 448.0], 1).contiguous()
    tiles = torch.zeros((3, bs * n, P + 2, P + 2, D), device=dev, dtype=torch.bfloat16)
    def run():
        for l in range(3):
            ops.roi_align_pack(feats[l], rois, tiles[l], C=D, H=S[l], W=S[l], ph=P, pw=P, spatial_scale=1.0 / (1.75 * 2 ** l),
                               sampling_ratio=2, aligned=True, pad=1)
    for _ in range(3): run()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"[{tag}] {name}: three levels {statistics.median(ts):8.1f} us   checksum {float(tiles.float().sum()):.6e}")
