"""Build libgroma_hip.so, libgroma_hip_f16.so and libgroma_hip_ref.so (gfx950) in-tree with hipcc.  No cmake, no torch extension machinery:
each library is a plain C-ABI shared object (include/groma_hip.h) loaded through ctypes.  The two are the same sources and the
same ABI; they differ in the 16-bit operand type the kernels are compiled for (bfloat16 / IEEE half, -DGR_F16: gr_common.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = {
    "gemm_bf16.hip": [],
    "gemm_bf16_256.hip": [],
    "gemv_bf16.hip": [],
    "gemv_fused.hip": [],
    "gemv_fp8.hip": [],
    "gemm_fp8_256.hip": [],
    "gemm_skinny.hip": [],
    "gemm_skinny_fp8.hip": [],
    "fp8.hip": [],
    "gemm_f32.hip": [],
    "attention.hip": [],
    "decode.hip": [],
    "norm.hip": [],
    "pack.hip": [],
    "preprocess.hip": [],
    "ddetr.hip": [],
    # index-exact kernels: plain IEEE fp32 sequences, no FMA contraction (see oracle/roi_nms.c)
    "select.hip": ["-ffp-contract=off"],
    "roi_align.hip": ["-ffp-contract=off"],
}
LIB = os.path.join(HERE, "libgroma_hip.so")
LIB_F16 = os.path.join(HERE, "libgroma_hip_f16.so")
LIB_REF = os.path.join(HERE, "libgroma_hip_ref.so")
# _ref: the reference-precision build -- operands are (hi, lo) pairs of halves, every contraction is 3 MFMA passes (gr_common.h)
VARIANTS = [("", [], LIB), ("_f16", ["-DGR_F16=1"], LIB_F16), ("_ref", ["-DGR_F16=1", "-DGR_SPLIT=1"], LIB_REF)]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, "gr_common.h"), os.path.join(HERE, "gemm_common.h"), os.path.join(HERE, "gemv_args.h"), os.path.join(HERE, "gemm_bf16_256.hip"), os.path.join(HERE, "gemm_skinny.hip"), os.path.join(HERE, "..", "..", "include", "groma_hip.h"),
            os.path.abspath(__file__)]
    jobs, links = [], []
    for suffix, defs, lib in VARIANTS:
        objs, dirty = [], False
        for src, extra in SOURCES.items():
            s = os.path.join(HERE, src)
            o = os.path.join(HERE, src.replace(".hip", suffix + ".o"))
            objs.append(o)
            if force or _stale(o, [s] + hdrs):
                jobs.append(["hipcc"] + COMMON + defs + extra + ["-c", s, "-o", o])
                dirty = True
        links.append((lib, objs, dirty))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, 16, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    for lib, objs, dirty in links:
        if force or dirty or _stale(lib, objs):
            run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(LIB_F16)
    print(LIB_REF)
