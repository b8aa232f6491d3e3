"""A/B of epilogue forms of the 256x256 GEMM (GROMA_HIP_LIB=<variant build> selects the other library)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
from groma_amd import ops
dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("GROMA_HIP_LIB", "staged"))
def med(fn):
    for _ in range(3): fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return statistics.median(ts)
for M, N, K in [(8148, 12288, 4096), (8148, 4096, 4096), (8148, 4096, 11008), (14350, 3072, 1024), (14350, 1024, 1024)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    o16 = torch.empty((M, N), device=dev, dtype=torch.bfloat16); h = torch.randn((M, N), device=dev)
    ref = (a.float() @ w.float().t())
    t1 = med(lambda: ops.gemm(a, w, out=o16, tile=256))
    e1 = ((o16.float() - ref).norm() / ref.norm()).item()
    h0 = h.clone()
    ops.gemm(a, w, resid=h0, out=h0, out_f32=True, tile=256)
    e2 = ((h0 - (ref + h)).norm() / (ref + h).norm()).item()
    t2 = med(lambda: ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=256))
    print(f"{tag:14s} {M}x{N}x{K}: bf16-out {t1:7.1f} us (err {e1:.1e})   f32+resid {t2:7.1f} us (err {e2:.1e})", flush=True)
