"""One-wave-per-SIMD 256x256 GEMM (tile=257, gemm_bf16_w128.hip) against the ping-pong kernel (tile=256): bit equality of the
results and interleaved median timings on the shapes of the Groma-7B step."""
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
torch.manual_seed(0)
for M, N, K, kw in [(8148, 22016, 4096, dict(act=3)), (8148, 12288, 4096, {}), (8148, 4096, 11008, dict(res=1)), (8148, 4096, 4096, dict(res=1)),
                    (14350, 4096, 1024, dict(act=1)), (14350, 1024, 4096, dict(res=1)), (14350, 3072, 1024, {}), (14350, 1024, 1024, dict(res=1)),
                    (577, 768, 1024, {}), (300, 4096, 256, {})]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    h0 = torch.randn((M, N), device=dev)
    bias = torch.randn((N,), device=dev)
    def run(tile, timing):
        if kw.get("res"):
            h = h0.clone()
            if timing: return lambda: ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=tile)
            ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=tile); return h
        o = torch.empty((M, N // 2 if kw.get("act") == 3 else N), device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(a, w, out=o, act=kw.get("act", 0), bias=bias if kw.get("act") == 1 else None, tile=tile)
        if timing: return f
        f(); return o
    if os.environ.get("W128_TIME_ONLY"):
        f257 = run(257, True)
        for _ in range(3): f257()
        t7 = statistics.median(once(f257) for _ in range(5))
        print(f"{os.environ['W128_TIME_ONLY']:8s} {M}x{N}x{K} w128 {t7:8.1f} us {2.0 * M * N * K / t7 / 1e6:6.0f} TF", flush=True)
        continue
    r256, r257 = run(256, False), run(257, False)
    torch.cuda.synchronize()
    same = torch.equal(r256, r257)
    err = ((r256.float() - r257.float()).abs().max() / r256.float().abs().max()).item()  # 32x32x16 sums k in another order: not bit-equal
    f256, f257 = run(256, True), run(257, True)
    for _ in range(3): f256(); f257()
    t6, t7 = [], []
    for _ in range(7):
        t6.append(once(f256)); t7.append(once(f257))
    t6, t7 = statistics.median(t6), statistics.median(t7)
    fl = 2.0 * M * N * K / 1e6
    print(f"{M}x{N}x{K} {kw}: equal={same} rel_maxdiff={err:.3g}  pingpong {t6:8.1f} us {fl / t6:6.0f} TF   w128 {t7:8.1f} us {fl / t7:6.0f} TF   {t6 / t7:5.3f}x", flush=True)
