"""shape-level timing of the 256x256 GEMM (run twice: default, and GROMA_HIP_LIB=<build_variant.py nopersist -DG256_NO_PERSIST>)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
from groma_amd import ops
dev = torch.device("cuda")
def once(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tag = os.path.basename(os.environ.get("GROMA_HIP_LIB", "persist"))
for M, N, K, kw in [(8148, 22016, 4096, dict(act=3)), (8148, 12288, 4096, {}), (8148, 4096, 11008, dict(res=1)), (8148, 4096, 4096, dict(res=1)),
                    (14350, 4096, 1024, dict(act=1)), (14350, 1024, 4096, {}), (14350, 3072, 1024, {}), (14350, 1024, 1024, {})]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    h = torch.randn((M, N), device=dev)
    if kw.get("res"):
        fn = lambda: ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=256)
    else:
        bias = torch.zeros((N,), device=dev)
        o = torch.empty((M, N // 2 if kw.get("act") == 3 else N), device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(a, w, out=o, act=kw.get("act", 0), bias=bias if kw.get("act") == 1 else None, tile=256)
    for _ in range(3): fn()
    t = statistics.median(once(fn) for _ in range(7))
    print(f"{tag} {M}x{N}x{K} {t:8.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF", flush=True)
