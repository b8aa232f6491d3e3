import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
for B in (1, 4, 14):
    Q, heads = 300, 8
    qk = torch.randn((B * Q, 512), device="cuda"); v = torch.randn((B * Q, 256), device="cuda")
    for _ in range(3): ops.mha32(qk, v, B=B, Q=Q, heads=heads, scale=32 ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.mha32(qk, v, B=B, Q=Q, heads=heads, scale=32 ** -0.5)
    e1.record(); torch.cuda.synchronize()
    print(f"mha32 B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
