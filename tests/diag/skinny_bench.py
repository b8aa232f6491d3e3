"""Per-shape timing of the matrix-unit weight stream (csrc/gemm_skinny.hip, tile 3) on the decode step's five weight shapes, HBM-cold
(the weights rotate through copies larger than the Infinity Cache), against bytes / 8 TB/s; K-ways forced through gr_diag_skinny_kw.
   python tests/diag/skinny_bench.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops, _lib

dev = torch.device("cuda")
lib = _lib.load()
lib.gr_diag_skinny_kw.argtypes, lib.gr_diag_skinny_kw.restype = [ctypes.c_int], ctypes.c_int
SHAPES = [("QKV", 12288, 4096, {}), ("o-proj", 4096, 4096, {"resid": True}), ("gate/up", 22016, 4096, {"act": 3}),
          ("down", 4096, 11008, {"resid": True}), ("head", 32128, 4096, {"out_f32": True})]
print(f"{'shape':10s} {'M':>3s} {'KW':>3s} {'us':>8s} {'TB/s':>6s} {'of 8':>6s}")
for name, N, K, epi in SHAPES:
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(ops.H16()) for _ in range(ncopy)]
    for M in (16, 32, 64):
        a = (torch.randn((M, K), device=dev) * 0.5).to(ops.H16())
        res = torch.zeros((M, N), device=dev)
        kws = [1, 2, 4] if M <= 32 else [1, 2]
        for kw in [0] + kws:
            lib.gr_diag_skinny_kw(kw)
            kwargs = dict(tile=3)
            if epi.get("resid"):
                kwargs.update(resid=res, out=res, out_f32=True)
            if epi.get("act"):
                kwargs.update(act=3)
            if epi.get("out_f32"):
                kwargs.update(out_f32=True)
            for i in range(3):
                ops.gemm(a, ws[i % ncopy], **kwargs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for i in range(n):
                ops.gemm(a, ws[i % ncopy], **kwargs)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            tb = N * K * 2 / (us * 1e-6) / 1e12
            print(f"{name:10s} {M:3d} {('auto' if kw == 0 else kw):>4} {us:8.1f} {tb:6.2f} {tb / 8:6.3f}", flush=True)
    del ws
    torch.cuda.empty_cache()
lib.gr_diag_skinny_kw(0)
# the 8-row fused stream on the same shapes, for scale
for name, N, K, epi in SHAPES[:2]:
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(ops.H16()) for _ in range(ncopy)]
    x = (torch.randn((8, K), device=dev) * 0.5).to(ops.H16())
    out = torch.zeros((8, N), device=dev)
    for i in range(3):
        ops.gemv_fused(ws[i % ncopy], M=8, x=x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ops.gemv_fused(ws[i % ncopy], M=8, x=x, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:10s}   8 fused {us:8.1f} {N * K * 2 / (us * 1e-6) / 1e12:6.2f}")
