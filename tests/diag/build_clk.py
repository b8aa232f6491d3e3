"""Diagnostic build of the product library with -DG256_CLK (block 0 of the 256x256 GEMM records shader/wall clocks):
tests/diag/libgroma_hip_clk.so.  Load it with GROMA_HIP_LIB=... ; never used by tests/bench."""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from groma_amd.csrc import build as B

F16 = "f16" in sys.argv[1:]   # python tests/diag/build_clk.py f16 -> the IEEE-half build of the same diagnostic library
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgroma_hip_clk_f16.so" if F16 else "libgroma_hip_clk.so")
tmp = tempfile.mkdtemp()
jobs, objs = [], []
for src, extra in B.SOURCES.items():
    o = os.path.join(tmp, src.replace(".hip", ".o"))
    objs.append(o)
    jobs.append(["hipcc"] + B.COMMON + extra + ["-DG256_CLK"] + (["-DGR_F16=1"] if F16 else []) + ["-c", os.path.join(B.HERE, src), "-o", o])
with ThreadPoolExecutor(8) as ex:
    list(ex.map(subprocess.check_call, jobs))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
