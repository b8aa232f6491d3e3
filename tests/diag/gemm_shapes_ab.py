"""The benchmark step's GEMM shapes WITH their in-model epilogues, timed on one build of the library.
usage: GROMA_HIP_LIB=<lib.so> python tests/diag/gemm_shapes_ab.py [tag]     (tests/diag/ab.sh alternates two builds on one box)
Prints one line per shape: median of 5 x 20-launch timings, and the weighted total per benchmark step."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
import torch
from groma_amd import ops
dev = "cuda"
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("GROMA_HIP_LIB", "default"))
# (name, M, N, K, calls per step, kwargs)
SHAPES = [("llama gate-up +SwiGLU", 8148, 22016, 4096, 32, dict(act=3)),
          ("llama qkv", 8148, 12288, 4096, 32, {}),
          ("llama down +resid f32", 8148, 4096, 11008, 32, dict(resid=1)),
          ("llama o-proj +resid f32", 8148, 4096, 4096, 32, dict(resid=1)),
          ("lm_head f32", 8148, 32128, 4096, 1, dict(f32=1)),
          ("vit fc1 +bias+GELU", 14350, 4096, 1024, 24, dict(act=1, bias=1)),
          ("vit fc2 +bias+ls+resid f32", 14350, 1024, 4096, 24, dict(resid=1, bias=1, scale=1)),
          ("vit qkv +bias", 14350, 3072, 1024, 24, dict(bias=1)),
          ("vit proj +bias+ls+resid f32", 14350, 1024, 1024, 24, dict(resid=1, bias=1, scale=1))]


def run(M, N, K, kw):
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) * 0.05).bfloat16()
    args = dict(tile=256)
    if kw.get("bias"): args["bias"] = torch.randn((N,), device=dev)
    if kw.get("scale"): args["scale"] = torch.randn((N,), device=dev)
    if kw.get("act"): args["act"] = kw["act"]
    if kw.get("resid"):
        h = torch.randn((M, N), device=dev)
        args.update(resid=h, out=h, out_f32=True)
    elif kw.get("f32"):
        args.update(out=torch.empty((M, N), device=dev), out_f32=True)
    else:
        args["out"] = torch.empty((M, N // 2 if kw.get("act") == 3 else N), device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, w, **args)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, **args)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return statistics.median(ts)


tot = 0.0
for name, M, N, K, calls, kw in SHAPES:
    us = run(M, N, K, kw)
    tot += us * calls
    print(f"[{tag}] {name:30s} {M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
print(f"[{tag}] weighted total of these launches per step: {tot / 1e3:.2f} ms")
