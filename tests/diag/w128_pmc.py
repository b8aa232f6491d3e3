"""A few launches of the w128 (tile=257) and ping-pong (tile=256) GEMM for a rocprofv3 --pmc pass (LDS bank conflicts etc.)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
dev = torch.device("cuda")
M, N, K = 8148, 4096, 4096
a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
o = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
for tile in (256, 257):
    for _ in range(3): ops.gemm(a, w, out=o, tile=tile)
torch.cuda.synchronize()
