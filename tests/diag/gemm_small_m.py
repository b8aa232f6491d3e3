"""The GEMM shapes of ONE image per call (M = 582 LLaMA rows / 1025 ViT rows) on several builds of the library, interleaved in one
process: python tests/diag/gemm_small_m.py name=path.so ...   (the 128 x 128 kernel's stage count: gemm_bf16.hip G128_DEEP)"""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import _lib, ops

libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
opened = {n: _lib._open(os.path.abspath(p), 0) for n, p in libs}
dev = "cuda"
T, I = 4096, 11008
SHAPES = [("llama qkv", 582, 3 * T, T, 32, {}), ("llama o +res", 582, T, T, 32, dict(resid=1)), ("llama gate-up", 582, 2 * I, T, 32, dict(act=3)),
          ("llama down +res", 582, T, I, 32, dict(resid=1)), ("vit qkv", 1025, 3072, 1024, 24, dict(bias=1)), ("vit proj +res", 1025, 1024, 1024, 24, dict(resid=1, bias=1, scale=1)),
          ("vit fc1 gelu", 1025, 4096, 1024, 24, dict(act=1, bias=1)), ("vit fc2 +res", 1025, 1024, 4096, 24, dict(resid=1, bias=1, scale=1)),
          ("bridge fc", 256, 4096, 4096, 2, dict(bias=1))]
for plan in ("throughput", "latency"):
    tot = {n: 0.0 for n in opened}
    with ops.gemm_plan(plan):
        for name, M, N, K, calls, kw in SHAPES:
            ncopy = max(2, int(400e6 // (N * K * 2)) + 1)      # HBM-cold weights, as inside a forward
            ws = [(torch.randn((N, K), device=dev) * 0.05).bfloat16() for _ in range(ncopy)]
            a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
            kws = {}
            if kw.get("bias"): kws["bias"] = torch.randn((N,), device=dev)
            if kw.get("scale"): kws["scale"] = torch.randn((N,), device=dev)
            if kw.get("act"): kws["act"] = kw["act"]
            h0 = torch.randn((M, N), device=dev) if kw.get("resid") else None
            if h0 is not None: kws.update(resid=h0, out_f32=True)
            ts, outs = {n: [] for n in opened}, {}
            for rnd in range(4):
                for n, lib in opened.items():
                    _lib._lib = lib
                    for i in range(2):
                        o = ops.gemm(a, ws[i % ncopy], **kws)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(20):
                        ops.gemm(a, ws[i % ncopy], **kws)
                    e1.record(); torch.cuda.synchronize()
                    ts[n].append(e0.elapsed_time(e1) / 20 * 1e3)
                    if rnd == 0:
                        outs[n] = ops.gemm(a, ws[0], **kws).clone()
            base = next(iter(opened))
            line = f"[{plan:10s}] {name:16s} {M}x{N}x{K} splits {ops.plan_splits(N, K)}:"
            for n in opened:
                us = statistics.median(ts[n])
                tot[n] += us * calls
                line += f"  [{n}] {us:7.1f} us{'' if torch.equal(outs[n].view(torch.uint8), outs[base].view(torch.uint8)) else ' !!DIFFERS'}"
            print(line, flush=True)
            del ws
    print(f"[{plan}] weighted total per one-image forward (ms):", {n: round(v / 1e3, 2) for n, v in tot.items()}, flush=True)
