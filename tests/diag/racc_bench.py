"""A/B of epilogue forms on the LLaMA o-proj / down-proj shapes (bias=zeros forces the epilogue-side residual), interleaved
repetitions (median), plus a scan over M around the benchmark's 8148 rows."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
dev = torch.device("cuda")
def once(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def ab(cases, reps=7):
    for _, fn in cases:
        for _ in range(3): fn()
    res = {k: [] for k, _ in cases}
    for _ in range(reps):
        for k, fn in cases: res[k].append(once(fn))
    return {k: statistics.median(v) for k, v in res.items()}
for M, N, K in [(8148, 4096, 4096), (8148, 4096, 11008), (8192, 4096, 4096), (8192, 4096, 11008)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    h = torch.randn((M, N), device=dev); zb = torch.zeros((N,), device=dev)
    o16 = torch.empty((M, N), device=dev, dtype=torch.bfloat16); o32 = torch.empty((M, N), device=dev)
    cases = [("bf16 out", lambda: ops.gemm(a, w, out=o16, tile=256)),
             ("f32 out", lambda: ops.gemm(a, w, out=o32, out_f32=True, tile=256)),
             ("f32+resid epilogue in-place", lambda: ops.gemm(a, w, resid=h, out=h, bias=zb, out_f32=True, tile=256)),
             ("f32+resid racc in-place", lambda: ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=256))]
    r = ab(cases)
    print(f"{M}x{N}x{K}: " + "  ".join(f"{k}: {v:.1f}us ({2.0 * M * N * K / v / 1e6:.0f} TF)" for k, v in r.items()), flush=True)
print("M scan, N=12288 K=4096 bf16 out (QKV shape) and N=4096 K=4096:")
for M in (7936, 8064, 8128, 8148, 8160, 8176, 8192, 8208, 8448):
    a = torch.randn((M, 4096), device=dev).bfloat16()
    w1 = (torch.randn((12288, 4096), device=dev) * 0.02).bfloat16(); w2 = (torch.randn((4096, 4096), device=dev) * 0.02).bfloat16()
    o1 = torch.empty((M, 12288), device=dev, dtype=torch.bfloat16); o2 = torch.empty((M, 4096), device=dev, dtype=torch.bfloat16)
    r = ab([("qkv", lambda: ops.gemm(a, w1, out=o1, tile=256)), ("o", lambda: ops.gemm(a, w2, out=o2, tile=256))], reps=5)
    print(f"  M={M}: qkv {r['qkv']:.1f} us ({2.0 * M * 12288 * 4096 / r['qkv'] / 1e6:.0f} TF)   o {r['o']:.1f} us ({2.0 * M * 4096 * 4096 / r['o'] / 1e6:.0f} TF)", flush=True)
