"""Yardstick, not product code: torch.matmul (hipBLASLt / rocBLAS behind PyTorch-ROCm) on the LLaMA / ViT GEMM shapes next to
gr_gemm_bf16 on the same box -- what a vendor-tuned plain GEMM reaches at this chip's sustained clock."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
dev = torch.device("cuda")
def med(fn, reps=7, n=10):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)
for M, N, K in [(8148, 22016, 4096), (8148, 12288, 4096), (8148, 4096, 11008), (8148, 4096, 4096), (8192, 8192, 8192),
                (14350, 4096, 1024), (14350, 1024, 4096), (14350, 3072, 1024)]:
    a = torch.randn((M, K), device=dev).bfloat16(); w = (torch.randn((N, K), device=dev) * 0.02).bfloat16()
    o = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t_lib = med(lambda: torch.matmul(a, w.t(), out=o))
    t_own = med(lambda: ops.gemm(a, w, out=o, tile=256))
    f = 2.0 * M * N * K / 1e6
    print(f"{M}x{N}x{K}: torch.matmul {t_lib:8.1f} us {f / t_lib:6.0f} TF/s | gr_gemm_bf16 {t_own:8.1f} us {f / t_own:6.0f} TF/s | ratio {t_lib / t_own:.3f}", flush=True)
