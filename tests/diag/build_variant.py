"""Build the product library with extra -D flags into tests/diag/<name>.so (A/B experiments in ONE gpurun call via
GROMA_HIP_LIB=...).  usage: python tests/diag/build_variant.py name -DFOO=1 -DBAR=2"""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from groma_amd.csrc import build as B
name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".so")
tmp = tempfile.mkdtemp()
jobs, objs = [], []
for src, extra in B.SOURCES.items():
    o = os.path.join(tmp, src.replace(".hip", ".o"))
    objs.append(o)
    jobs.append(["hipcc"] + B.COMMON + extra + defs + ["-c", os.path.join(B.HERE, src), "-o", o])
with ThreadPoolExecutor(8) as ex:
    list(ex.map(subprocess.check_call, jobs))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
