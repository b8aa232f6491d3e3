"""the fp32-residual GEMMs of the benchmark step (o-proj, down-proj at M = 8148; ViT proj / fc2 at M = 14350) with the 256-row and the
192-row form of the ping-pong kernel forced, and the launcher's own choice:  python tests/diag/resid_tile_rows.py"""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import ops
dev = "cuda"
for name, M, N, K, kw in [("llama o-proj +res", 8148, 4096, 4096, {}), ("llama down +res", 8148, 4096, 11008, {}),
                          ("llama qkv", 8148, 12288, 4096, dict(plain=1)), ("llama gate-up", 8148, 22016, 4096, dict(plain=1, act=3)),
                          ("vit proj +res", 14350, 1024, 1024, dict(bias=1, scale=1)), ("vit fc2 +res", 14350, 1024, 4096, dict(bias=1, scale=1))]:
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    kws = {}
    if kw.get("bias"): kws["bias"] = torch.randn((N,), device=dev)
    if kw.get("scale"): kws["scale"] = torch.randn((N,), device=dev)
    if kw.get("act"): kws["act"] = kw["act"]
    h = None if kw.get("plain") else torch.randn((M, N), device=dev)
    ts = {256: [], 192: [], 0: []}
    for rnd in range(4):
        for t in ts:
            run = (lambda: ops.gemm(a, w, resid=h, out=h, out_f32=True, tile=t, **kws)) if h is not None else (lambda: ops.gemm(a, w, tile=t, **kws))
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            ts[t].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"[resid tile rows] {name:18s} {M}x{N}x{K}: 256 rows {statistics.median(ts[256]):7.1f} us | 192 rows {statistics.median(ts[192]):7.1f} us | launcher {statistics.median(ts[0]):7.1f} us", flush=True)
