"""one eager launch sequence of the 4-row decode streams (16-bit gemv_fused, e4m3 gemv_fp8) on the decode step's shapes, for rocprofv3 --pmc:
   cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tests/diag/gemv_pmc.py
   python tests/diag/gemv_pmc.py --summarise out      (per kernel: FETCH_SIZE x 2 (gfx950 correction) per launch against the weight bytes)"""
import csv, glob, os, sys
if "--summarise" in sys.argv:
    d = sys.argv[sys.argv.index("--summarise") + 1]
    tot = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "gemv_f" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                t = tot.setdefault(k, [0.0, 0])
                t[0] += float(r["Counter_Value"]); t[1] += 1
    alg = {"gemv_fused": 2.0, "gemv_fp8": 1.0}
    shapes = ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32128, 4096))
    for k, (v, n) in sorted(tot.items()):
        per = 2.0 * v * 1024 / n / 1e6
        b = [alg[a] for a in alg if a in k][0] * sum(N * K for N, K in shapes) / len(shapes) / 1e6
        print(f"{k:40s} {n:4d} launches: fetched {per:7.1f} MB per launch (FETCH_SIZE x 2), weights {b:7.1f} MB per launch on average -> x{per / b:.3f}")
    raise SystemExit
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from groma_amd import ops, weights
dev = torch.device("cuda")
M = 4
for N, K in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32128, 4096)):
    base = torch.randn((N, K), device=dev) * 0.02
    w16 = base.to(ops.H16())
    q, sc = weights.q8(base.cpu())
    q, sc = q.to(dev), sc.to(dev)
    x = (torch.randn((M, K), device=dev) * 0.5).to(ops.H16())
    out = torch.zeros((M, N), device=dev)
    for _ in range(4):
        ops.gemv_fused(w16, M=M, x=x, out=out)
        ops.gemv_fused(q, M=M, x=x, out=out, w_scale=sc)
    torch.cuda.synchronize()
