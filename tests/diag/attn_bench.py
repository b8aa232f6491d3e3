"""prefill attention at the benchmark's two shapes; GROMA_HIP_LIB=<build> for an A/B"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
import torch
from groma_amd import ops
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
for (B, H, L, hd, causal) in [(14, 32, 582, 128, True), (14, 16, 1025, 64, False), (4, 32, 582, 128, True), (1, 32, 582, 128, True)]:
    stride = (L + 63) // 64 * 64
    q = torch.randn((B, H, L, hd), device="cuda").bfloat16()
    k = torch.randn((B, H, stride, hd), device="cuda").bfloat16()
    vt = torch.randn((B, H, hd, stride), device="cuda").bfloat16()
    for _ in range(3): ops.attention(q, k, vt, Skv=L, causal=causal)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.attention(q, k, vt, Skv=L, causal=causal)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"[{tag}] attention B={B} hd={hd} L={L} causal={causal}: {statistics.median(ts):.1f} us", flush=True)
