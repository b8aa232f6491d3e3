"""fp32 MFMA GEMM on the proposer's shapes (run with GROMA_HIP_LIB=<other build> for an A/B)"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
from groma_amd import ops
dev = torch.device("cuda")
tag = os.path.basename(os.environ.get("GROMA_HIP_LIB", "default"))
import ctypes
from groma_amd import _lib
lib = _lib.load()
lib.gr_diag_gemm_f32_tile.argtypes, lib.gr_diag_gemm_f32_tile.restype = [ctypes.c_int], ctypes.c_int
for force in (0, 128, 64):
  lib.gr_diag_gemm_f32_tile(force)
  tag = os.path.basename(os.environ.get("GROMA_HIP_LIB", "default")) + f" tile={force or 'auto'}"
  tot = 0.0
  for M, N, K, n in [(14336, 256, 1024, 1), (14336, 256, 256, 19), (14336, 96, 256, 6), (14336, 1024, 256, 6), (14336, 256, 1024, 6),
                     (4200, 512, 256, 6), (4200, 256, 256, 18), (4200, 96, 256, 6), (4200, 1024, 256, 6), (4200, 256, 1024, 6)]:
      a = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) * 0.05; b = torch.randn((N,), device=dev)
      fn = lambda: ops.gemm_f32(a, w, bias=b, act=2)
      for _ in range(3): fn()
      ts = []
      for _ in range(5):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(10): fn()
          e1.record(); torch.cuda.synchronize()
          ts.append(e0.elapsed_time(e1) / 10 * 1e3)
      t = statistics.median(ts)
      tot += t * n
      print(f"{tag} {M}x{N}x{K}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF/s  (x{n} per step)", flush=True)
  print(f"{tag} proposer GEMM time per step (these shapes): {tot / 1e3:.2f} ms")

lib.gr_diag_gemm_f32_tile(0)
