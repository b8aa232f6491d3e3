"""Row-tile height of the ping-pong GEMM kernel (256 / 192 rows x 256 columns) and the 128x128 kernel, forced per launch, at
the LLaMA / ViT shapes of 1 and 4 images per call (M = 582 / 2328 / 1025 / 4100): the numbers the launcher's cost model
(gemm_bf16.hip: PP192_C / PP192_O, E128_SOLO) is fitted to, plus what the model picks (tile 0).  Bitwise equality of all
forms is asserted on the way.   python tests/diag/gemm_tile_rows.py"""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant
_variant.use_env()
from groma_amd import ops

dev = torch.device("cuda")
T, I = 4096, 11008
shapes = []
for M in (2328, 582):
    shapes += [("llm qkv", M, 3 * T, T, {}), ("llm o+res", M, T, T, {"resid": True}), ("llm gate-up", M, 2 * I, T, {"act": 3}),
               ("llm down+res", M, T, I, {"resid": True}), ("llm head", M, 32128, T, {"f32": True})]
for M in (4100, 1025):
    shapes += [("vit qkv", M, 3072, 1024, {}), ("vit proj+res", M, 1024, 1024, {"resid": True}), ("vit fc1", M, 4096, 1024, {"act": 1}),
               ("vit fc2+res", M, 1024, 4096, {"resid": True})]
g = torch.Generator().manual_seed(0)
for name, M, N, K, kw in shapes:
    a = torch.randn((M, K), generator=g).bfloat16().to(dev)
    w = (torch.randn((N, K), generator=g) * 0.02).bfloat16().to(dev)
    res = torch.randn((M, N), generator=g).to(dev) if kw.get("resid") else None
    f32 = bool(kw.get("resid") or kw.get("f32"))

    def run(tile):
        return ops.gemm(a, w, resid=res, out_f32=f32, act=kw.get("act", 0), tile=tile)
    outs = {t: run(t) for t in (256, 192, 128)}
    for t in (192, 128):
        assert torch.equal(outs[t], outs[256]), (name, M, t)
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
    best = min((256, 192, 128), key=lambda t: med[t])
    fl = 2.0 * M * N * K
    print(f"[tile rows] {name:13s} M={M:5d} N={N:6d} K={K:6d}  256: {med[256]:7.1f}  192: {med[192]:7.1f}  128x128: {med[128]:7.1f}  "
          f"auto: {med[0]:7.1f} us   best {best} = {fl / med[best] / 1e6:5.0f} TF/s, auto/best {med[0] / med[best]:.3f}")
