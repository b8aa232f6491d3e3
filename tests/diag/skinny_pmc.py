"""one eager launch sequence of the matrix-unit weight stream on the decode step's shapes at 32 rows (for rocprofv3 --pmc FETCH_SIZE / --stats):
   cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tests/diag/skinny_pmc.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops
dev = torch.device("cuda")
for N, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32128, 4096)):
    w = (torch.randn((N, K), device=dev) * 0.02).to(ops.H16())
    a = (torch.randn((32, K), device=dev) * 0.5).to(ops.H16())
    for _ in range(4):
        ops.gemm(a, w, out_f32=True, tile=3)
    torch.cuda.synchronize()
    print(N, K, "algorithmic MB", N * K * 2 / 1e6)
