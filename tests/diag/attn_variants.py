"""One-box A/B of several BUILDS of the library on the prefill-attention shapes of the benchmark, in ONE process:
    python tests/diag/attn_variants.py [--ref] name=path.so [name=path.so ...]      (--ref: operand-pair builds, the hybrid ViT's kernel)
Each build is opened with ctypes and swapped in as the active library (measurement scaffolding: never used by tests/ or bench.py).
Rounds are interleaved (A B C A B C) so that clock drift of the box hits every build alike; the median over rounds is printed."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import _lib, ops

args = sys.argv[1:]
ref = "--ref" in args
libs = [a.split("=", 1) for a in args if "=" in a]
prec = "ref" if ref else "bf16"
opened = {n: _lib._open(os.path.abspath(p), 2 if ref else 0) for n, p in libs}
SHAPES = [(14, 16, 1025, 64, False)] if ref else [(14, 32, 582, 128, True), (14, 16, 1025, 64, False), (4, 32, 582, 128, True), (1, 32, 582, 128, True)]


def use(lib):
    if ref:
        _lib._lib_ref = lib
    else:
        _lib._lib = lib


res = {}
with ops.precision(prec):
    for (B, H, L, hd, causal) in SHAPES:
        stride = (L + 63) // 64 * 64
        q = ops.to_h16(torch.randn((B, H, L, hd), device="cuda"))
        k = ops.to_h16(torch.randn((B, H, stride, hd), device="cuda"))
        vt = ops.to_h16(torch.randn((B, H, hd, stride), device="cuda"))
        outs = {}
        for rnd in range(4):
            for n, lib in opened.items():
                use(lib)
                for _ in range(2):
                    o = ops.attention(q, k, vt, Skv=L, causal=causal)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.attention(q, k, vt, Skv=L, causal=causal)
                e1.record(); torch.cuda.synchronize()
                res.setdefault((B, H, L, hd), {}).setdefault(n, []).append(e0.elapsed_time(e1) / 20 * 1e3)
                outs[n] = o.float().cpu()
        base = next(iter(outs))
        for n in outs:
            same = torch.equal(outs[n], outs[base])
            ts = res[(B, H, L, hd)][n]
            print(f"attention B={B} H={H} L={L} hd={hd} causal={causal} [{n:>22s}]: median {statistics.median(ts):7.1f} us  (min {min(ts):.1f})  bitwise == {base}: {same}", flush=True)
