"""One-box A/B of several BUILDS of the library on the benchmark step's GEMM shapes (with their in-model epilogues), in ONE process:
    python tests/diag/gemm_variants.py [--ref] name=path.so [name=path.so ...]
--ref: operand-pair builds (libgroma_hip_ref.so variants) on the hybrid model's ViT shapes.  Rounds are interleaved (A B A B) so
clock drift hits every build alike; prints the median per shape and the weighted total per benchmark step; asserts bitwise equality."""
import os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from groma_amd import _lib, ops

args = sys.argv[1:]
ref = "--ref" in args
fp8 = "--fp8" in args   # e4m3 operands (gemm_fp8_256_kernel) on the LLaMA shapes, bf16-library variants
libs = [a.split("=", 1) for a in args if "=" in a]
prec = "ref" if ref else "bf16"
opened = {n: _lib._open(os.path.abspath(p), 2 if ref else 0) for n, p in libs}
dev = "cuda"
VIT = [("vit fc1 +bias+GELU", 14350, 4096, 1024, 24, dict(act=1, bias=1)),
       ("vit fc2 +bias+ls+resid f32", 14350, 1024, 4096, 24, dict(resid=1, bias=1, scale=1)),
       ("vit qkv +bias", 14350, 3072, 1024, 24, dict(bias=1)),
       ("vit proj +bias+ls+resid f32", 14350, 1024, 1024, 24, dict(resid=1, bias=1, scale=1))]
LLM = [("llama gate-up +SwiGLU", 8148, 22016, 4096, 32, dict(act=3)),
       ("llama qkv", 8148, 12288, 4096, 32, {}),
       ("llama down +resid f32", 8148, 4096, 11008, 32, dict(resid=1)),
       ("llama o-proj +resid f32", 8148, 4096, 4096, 32, dict(resid=1)),
       ("lm_head f32", 8148, 32128, 4096, 1, dict(f32=1))]
CONV = [("fuse conv 3x3 @128^2", 14, 128, 1024, 5), ("fuse conv 3x3 @64^2", 14, 64, 1024, 5)]
SHAPES = VIT if ref else (LLM if fp8 else LLM + VIT)


def use(lib):
    if ref:
        _lib._lib_ref = lib
    else:
        _lib._lib = lib


tot = {n: 0.0 for n in opened}
with ops.precision(prec):
    for name, M, N, K, calls, kw in SHAPES:
        a = ops.to_h16(torch.randn((M, K), device=dev) * 0.5)
        w = ops.to_h16(torch.randn((N, K), device=dev) * 0.05)
        kws = dict(tile=0)
        if fp8:
            from groma_amd import weights
            a, sa = ops.quant_rows_fp8(a)
            w, sw = weights.q8(w.float())
            kws.update(a_scale=sa, w_scale=sw)
        if kw.get("bias"): kws["bias"] = torch.randn((N,), device=dev)
        if kw.get("scale"): kws["scale"] = torch.randn((N,), device=dev)
        if kw.get("act"): kws["act"] = kw["act"]
        h0 = torch.randn((M, N), device=dev) if kw.get("resid") else None
        if kw.get("resid"): kws.update(out_f32=True)
        elif kw.get("f32"): kws.update(out_f32=True)
        ts, outs = {n: [] for n in opened}, {}
        for rnd in range(4):
            for n, lib in opened.items():
                use(lib)
                run = (lambda: ops.gemm(a, w, resid=h0, **kws)) if h0 is not None else (lambda: ops.gemm(a, w, **kws))
                for _ in range(2):
                    o = run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                ts[n].append(e0.elapsed_time(e1) / 10 * 1e3)
                if rnd == 0:
                    outs[n] = o
        base = next(iter(opened))
        line = f"{name:30s} {M}x{N}x{K}:"
        for n in opened:
            us = statistics.median(ts[n])
            tot[n] += us * calls
            line += f"  [{n}] {us:8.1f} us {(3 if ref else 1) * 2.0 * M * N * K / us / 1e6:5.0f} TF/s{'' if torch.equal(outs[n].view(torch.uint8), outs[base].view(torch.uint8)) else ' !!DIFFERS'}"
        print(line, flush=True)
        del a, w, h0, outs
    if not ref and not fp8:   # the region encoder's implicit-GEMM 3x3 convs
        for name, imgs, S, C, calls in CONV:
            pad = ops.to_h16(torch.randn((imgs, S + 2, S + 2, C), device=dev) * 0.5)
            w = ops.to_h16(torch.randn((C, 9 * C), device=dev) * 0.02)
            ts = {n: [] for n in opened}
            for rnd in range(4):
                for n, lib in opened.items():
                    use(lib)
                    for _ in range(2):
                        ops.gemm(pad, w, conv=(imgs, S, S, C, 0))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        ops.gemm(pad, w, conv=(imgs, S, S, C, 0))
                    e1.record(); torch.cuda.synchronize()
                    ts[n].append(e0.elapsed_time(e1) / 5 * 1e3)
            line = f"{name:30s} {imgs * S * S}x{C}x{9 * C}:"
            for n in opened:
                us = statistics.median(ts[n])
                tot[n] += us * calls
                line += f"  [{n}] {us:8.1f} us {2.0 * imgs * S * S * C * 9 * C / us / 1e6:5.0f} TF/s"
            print(line, flush=True)
print("weighted total of these launches per step (ms):", {n: round(v / 1e3, 2) for n, v in tot.items()})
