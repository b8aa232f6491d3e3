"""Decode-step weight streams at the LLaMA-7B shapes, M rows: the fused kernel (gr_gemv_fused: whole rows per workgroup, no
partials) against round 1-3's split-K slices + reduce kernel, interleaved medians.  GB/s = N*K*2 / time.
    python tests/diag/gemv_bench.py [M=4] [tag]      (GROMA_HIP_LIB=<variant.so> for an A/B of two builds)"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant
_variant.use_env()
from groma_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda")
T, I, V = 4096, 11008, 32128
shapes = [("qkv", 3 * T, T, "norm", "out"), ("o", T, T, "x", "resid"), ("gate_up", 2 * I, T, "norm", "swiglu"), ("down", T, I, "x", "resid"),
          ("head", V, T, "norm", "out")]
g = torch.Generator().manual_seed(0)
tot_new = tot_old = tot_bytes = 0.0
for name, N, K, xm, epi in shapes:
    # rotate through enough copies that every launch streams HBM-cold weights
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn((N, K), generator=g) * 0.02).bfloat16().to(dev) for _ in range(ncopy)]
    h = torch.randn((M, K), generator=g).to(dev)
    gam = torch.ones((K,), device=dev)
    x = torch.randn((M, K), generator=g).bfloat16().to(dev)
    out, res, act = torch.empty((M, N), device=dev), torch.zeros((M, N), device=dev), torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)

    def new(i):
        kw = dict(norm=(h, gam, 1e-5)) if xm == "norm" else dict(x=x)
        kw.update(dict(out=out) if epi == "out" else dict(resid=res) if epi == "resid" else dict(swiglu_out=act))
        ops.gemv_fused(ws[i % ncopy], M=M, **kw)

    def old(i):   # what the step used to launch for this matrix: (norm ->) slices -> reduce (+ epilogue)
        xx = ops.rmsnorm(h, gam, 1e-5) if xm == "norm" else x
        if epi == "swiglu":
            ops.gemm(xx, ws[i % ncopy], act=3, out=act, tile=1, splits=(K + 511) // 512, ws=ops._gemv_ws((K + 511) // 512, M, N, dev))
        else:
            ops.gemm(xx, ws[i % ncopy], out_f32=True, out=out, resid=res if epi == "resid" else None)
    times = {"new": [], "old": []}
    for rep in range(5):
        for nm, fn in (("new", new), ("old", old)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            n = 20
            for i in range(n):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            times[nm].append(a.elapsed_time(b) / n * 1e3)
    tn, to = statistics.median(times["new"]), statistics.median(times["old"])
    nb = N * K * 2
    tot_new += tn; tot_old += to; tot_bytes += nb
    print(f"[{tag}] M={M} {name:8s} {N:6d}x{K:6d}  fused {tn:7.1f} us = {nb / tn / 1e3:6.0f} GB/s | slices+reduce {to:7.1f} us = {nb / to / 1e3:6.0f} GB/s")
    del ws
print(f"[{tag}] M={M} per layer (4 matrices) fused {tot_new:.0f} us vs {tot_old:.0f} us incl. head; {tot_bytes / tot_new / 1e3:.0f} vs {tot_bytes / tot_old / 1e3:.0f} GB/s")
