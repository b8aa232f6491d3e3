"""The hybrid model's ViT GEMMs on operand pairs (libgroma_hip_ref.so: gemm_pair_256_kernel / gemm_pair_kernel, 3 MFMA passes) at the
benchmark's M = 14 x 1025 rows with their real epilogues, every tile form forced and what the launcher's cost model picks:
    python tests/diag/pair_tile_rows.py"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from groma_amd import ops

dev = torch.device("cuda")
M = 14 * 1025
shapes = [("vit qkv", 3072, 1024, {}), ("vit proj+ls+res", 1024, 1024, {"resid": True}), ("vit fc1+gelu", 4096, 1024, {"act": 1}),
          ("vit fc2+ls+res", 1024, 4096, {"resid": True})]
g = torch.Generator().manual_seed(0)
tot = {256: 0.0, 192: 0.0, 128: 0.0, 0: 0.0}
with ops.precision("ref"):
    for name, N, K, kw in shapes:
        a = ops.to_h16(torch.randn((M, K), generator=g).to(dev))
        w = ops.to_h16((torch.randn((N, K), generator=g) * 0.02).to(dev))
        bias = torch.randn((N,), generator=g).to(dev)
        res = torch.randn((M, N), generator=g).to(dev) if kw.get("resid") else None
        ls = torch.rand((N,), generator=g).to(dev) if kw.get("resid") else None

        def run(tile):
            return ops.gemm(a, w, bias=bias, scale=ls, resid=res, out_f32=bool(kw.get("resid")), act=kw.get("act", 0), tile=tile)
        outs = {t: run(t) for t in (256, 192, 128)}
        for t in (192, 128):
            assert torch.equal(outs[t], outs[256]), (name, t)
        times = {}
        for rep in range(3):
            for t in (256, 192, 128, 0):
                for _ in range(3):
                    run(t)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(t)
                e1.record()
                torch.cuda.synchronize()
                times.setdefault(t, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        med = {t: statistics.median(v) for t, v in times.items()}
        for t in med:
            tot[t] += med[t]
        best = min((256, 192, 128), key=lambda t: med[t])
        fl = 2.0 * M * N * K
        print(f"[pair tile rows] {name:16s} M={M} N={N:5d} K={K:5d}  256: {med[256]:7.1f}  192: {med[192]:7.1f}  128x128: {med[128]:7.1f}  auto: {med[0]:7.1f} us"
              f"   best {best}: {3 * fl / med[best] / 1e6:5.0f} TF/s issued ({fl / med[best] / 1e6:4.0f} algorithmic), auto/best {med[0] / med[best]:.3f}", flush=True)
print("[pair tile rows] per ViT layer (us):", {k: round(v, 1) for k, v in tot.items()})
