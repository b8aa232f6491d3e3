"""row quantiser (gr_quant_rows_fp8) at the e4m3 step's shapes on several builds, interleaved: python tests/diag/quant_variants.py name=path.so ..."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import _lib, ops
libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
opened = {n: _lib._open(os.path.abspath(p), 0) for n, p in libs}
tot = {n: 0.0 for n in opened}
for name, M, K, calls in [("llama ctx", 8148, 4096, 32), ("llama swiglu out", 8148, 11008, 32), ("vit ctx", 14350, 1024, 24), ("vit fc1 out", 14350, 4096, 24)]:
    x = (torch.randn((M, K), device="cuda") * 3).bfloat16()
    ts, outs = {n: [] for n in opened}, {}
    for rnd in range(4):
        for n, lib in opened.items():
            _lib._lib = lib
            for _ in range(2):
                q, s = ops.quant_rows_fp8(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.quant_rows_fp8(x)
            e1.record(); torch.cuda.synchronize()
            ts[n].append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[n] = (q.view(torch.uint8).clone(), s.clone())
    base = next(iter(opened))
    line = f"quant_rows {name:18s} {M}x{K}:"
    for n in opened:
        us = statistics.median(ts[n]); tot[n] += us * calls
        same = torch.equal(outs[n][0], outs[base][0]) and torch.equal(outs[n][1], outs[base][1])
        line += f"  [{n}] {us:7.1f} us = {M * K * 3 / us / 1e3:5.0f} GB/s{'' if same else ' !!DIFFERS'}"
    print(line, flush=True)
print("per e4m3 step (ms):", {n: round(v / 1e3, 2) for n, v in tot.items()})
