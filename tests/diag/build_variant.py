"""Build a named variant of the product library for a one-box A/B: the product objects, with gemm_bf16_256.hip (and whatever else
is listed) recompiled with extra -D flags.   python tests/diag/build_variant.py <name> -DFLAG [-DFLAG2 ...]  ->  tests/diag/<name>.so
(load it with GROMA_HIP_LIB=tests/diag/<name>.so in the tests/diag scripts; never used by tests/ or bench.py)"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from groma_amd.csrc import build as B

name, flags = sys.argv[1], sys.argv[2:]
# --src=a.hip,b.hip: the sources recompiled with the flags (default: the two GEMM files)
srcs = ("gemm_bf16_256.hip", "gemm_bf16.hip")
base = ""   # --base=ref | f16: a variant of libgroma_hip_ref.so / libgroma_hip_f16.so instead of the bf16 library
for f in list(flags):
    if f.startswith("--src="):
        srcs = tuple(f[6:].split(","))
        flags.remove(f)
    if f.startswith("--base="):
        base = "_" + f[7:]
        flags.remove(f)
base_defs = next(d for sfx, d, _ in B.VARIANTS if sfx == base)
B.build(verbose=False)
here = os.path.dirname(os.path.abspath(__file__))
tmp = tempfile.mkdtemp()
objs = []
for src, extra in B.SOURCES.items():
    o = os.path.join(B.HERE, src.replace(".hip", base + ".o"))
    if src in srcs:   # (gemm_bf16.hip includes the 256 kernel's launch path)
        o = os.path.join(tmp, src.replace(".hip", ".o"))
        subprocess.check_call(["hipcc"] + B.COMMON + base_defs + extra + flags + ["-c", os.path.join(B.HERE, src), "-o", o])
    objs.append(o)
out = os.path.join(here, name + ".so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
