"""Correctness of a variant build of the 256x256 GEMM before it is timed: a few shapes (edge tiles, 1-3 K-tiles, long K, fp32
residual, split-K) against an fp32 torch reference.  GROMA_HIP_LIB=<variant.so> python tests/diag/variant_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
import torch
from groma_amd import ops

torch.manual_seed(0)
worst = 0.0
for (M, N, K, kw) in [(300, 520, 64, {}), (256, 256, 128, {}), (777, 640, 192, {}), (1025, 1024, 4096, {}),
                      (2328, 4096, 11008, dict(resid=1)), (8148, 768, 1024, dict(f32=1)), (582, 4096, 4096, dict(splits=3))]:
    a = (torch.randn((M, K), device="cuda") * 0.5).bfloat16()
    w = (torch.randn((N, K), device="cuda") * 0.05).bfloat16()
    ref = a.float() @ w.float().t()
    if kw.get("resid"):
        h = torch.randn((M, N), device="cuda")
        ref = ref + h
        out = ops.gemm(a, w, resid=h, out=h.clone(), out_f32=True, tile=256)
    elif kw.get("f32"):
        out = ops.gemm(a, w, out_f32=True, tile=256)
    elif kw.get("splits"):
        out = ops.gemm(a, w, out_f32=True, tile=256, splits=kw["splits"])
    else:
        out = ops.gemm(a, w, tile=256)
    e = ((out.float() - ref).norm() / ref.norm()).item()
    tol = 4e-3 if out.dtype != torch.float32 else 2e-5
    worst = max(worst, e / tol)
    print(f"[check] {M}x{N}x{K} {kw}: rel err {e:.2e} (tol {tol:.0e}) {'OK' if e < tol else 'FAIL'}")
print("[check] ALL OK" if worst < 1 else "[check] FAILED")
