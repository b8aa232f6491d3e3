"""Is the 256x256 GEMM sensitive to WHERE its operands come from?  Same launch, operands either re-used every launch (A + W
<= 250 MB: resident in the 256 MB Infinity Cache after the first launch) or rotated through 8 distinct copies (2 GB: every
launch streams its operands from HBM).  FETCH_SIZE counts L2 -> fabric requests in both cases and cannot tell the two apart
(MI355X_MICROARCH.md 'Infinity Cache'); kernel time can."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
dev = torch.device("cuda")
def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K, act in [("gate-up", 8148, 22016, 4096, 3), ("qkv", 8148, 12288, 4096, 0), ("down", 8148, 4096, 11008, 0)]:
    As = [torch.randn((M, K), device=dev).bfloat16() for _ in range(8)]
    Ws = [(torch.randn((N, K), device=dev) * 0.02).bfloat16() for _ in range(8)]
    out = torch.empty((M, N // 2 if act == 3 else N), device=dev, dtype=torch.bfloat16)
    same = lambda i: ops.gemm(As[0], Ws[0], out=out, act=act, tile=256)
    rot = lambda i: ops.gemm(As[i % 8], Ws[i % 8], out=out, act=act, tile=256)
    for f in (same, rot):
        for i in range(8): f(i)
    ts = {"operands reused (Infinity-Cache warm)": [], "operands rotated over 8 copies (HBM cold)": []}
    for _ in range(5):
        ts["operands reused (Infinity-Cache warm)"].append(timed(same, 16))
        ts["operands rotated over 8 copies (HBM cold)"].append(timed(rot, 16))
    alg = (M * K + N * K) * 2 + out.numel() * 2
    for k, v in ts.items():
        t = statistics.median(v)
        print(f"{name} {M}x{N}x{K}: {k}: {t:.1f} us = {2.0 * M * N * K / t / 1e6:.0f} TF/s; algorithmic operand+result bytes {alg / 1e6:.0f} MB -> "
              f"{alg / t / 1e6:.2f} TB/s if all of it came from HBM")
