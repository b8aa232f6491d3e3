"""Decode-step weight streams at the LLaMA-7B shapes, M rows: 16-bit weights vs e4m3 weights (gr_gemv_fused w8), HBM-cold, interleaved.
    python tests/diag/gemv_w8_bench.py [M=4] [tag]"""
import sys, os, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()   # GROMA_HIP_LIB=<build> for an A/B
from groma_amd import ops, weights

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda")
T, I, V = 4096, 11008, 32128
shapes = [("qkv", 3 * T, T, "norm", "out"), ("o", T, T, "x", "resid"), ("gate_up", 2 * I, T, "norm", "swiglu"), ("down", T, I, "x", "resid"),
          ("head", V, T, "norm", "out")]
g = torch.Generator().manual_seed(0)
tot = {"h16": 0.0, "w8": 0.0}
for name, N, K, xm, epi in shapes:
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)   # rotate through enough copies that every launch streams HBM-cold weights
    base = (torch.randn((N, K), generator=g) * 0.02).to(dev)
    w16 = [base.bfloat16().clone() for _ in range(ncopy)]
    q, sc = weights.q8(base)
    w8 = [q.clone() for _ in range(2 * ncopy)]
    h = torch.randn((M, K), generator=g).to(dev)
    gam = torch.ones((K,), device=dev)
    x = torch.randn((M, K), generator=g).bfloat16().to(dev)
    out, res, act = torch.empty((M, N), device=dev), torch.zeros((M, N), device=dev), torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)

    def run(kind, i):
        kw = dict(norm=(h, gam, 1e-5)) if xm == "norm" else dict(x=x)
        kw.update(dict(out=out) if epi == "out" else dict(resid=res) if epi == "resid" else dict(swiglu_out=act))
        if kind == "w8":
            ops.gemv_fused(w8[i % len(w8)], M=M, w_scale=sc, **kw)
        else:
            ops.gemv_fused(w16[i % ncopy], M=M, **kw)
    times = {"h16": [], "w8": []}
    for rep in range(5):
        for kind in ("h16", "w8"):
            for i in range(3):
                run(kind, i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            n = 20
            for i in range(n):
                run(kind, i)
            b.record()
            torch.cuda.synchronize()
            times[kind].append(a.elapsed_time(b) / n * 1e3)
    t16, t8 = statistics.median(times["h16"]), statistics.median(times["w8"])
    tot["h16"] += t16 * (1 if name == "head" else 32); tot["w8"] += t8 * (1 if name == "head" else 32)
    print(f"[{tag}] M={M} {name:8s} {N:6d}x{K:6d}  16-bit {t16:7.1f} us = {N * K * 2 / t16 / 1e3:6.0f} GB/s | e4m3 {t8:7.1f} us = {N * K / t8 / 1e3:6.0f} GB/s", flush=True)
    del w16, w8
print(f"[{tag}] M={M} streams per token (32 layers + head): 16-bit {tot['h16'] / 1e3:.2f} ms, e4m3 {tot['w8'] / 1e3:.2f} ms")
