"""Build oracle/_ref/libmmcv_ref.so from the reference's own CPU sources (TEST INFRASTRUCTURE ONLY).
Recipe: g++ directly on four reference files + our shim; no reference build system, nothing copied.
Usage: python oracle/build_ref.py /root/reference"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main(ref):
    import torch
    from torch.utils import cpp_extension as ce
    csrc = os.path.join(ref, "mmcv", "mmcv", "ops", "csrc")
    srcs = [("pytorch/nms.cpp", "disp_nms.o"), ("pytorch/cpu/nms.cpp", "cpu_nms.o"),
            ("pytorch/roi_align.cpp", "disp_roi_align.o"), ("pytorch/cpu/roi_align.cpp", "cpu_roi_align.o")]
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libmmcv_ref.so")
    inc = ["-I" + os.path.join(csrc, "common")] + ["-I" + p for p in ce.include_paths()]
    flags = ["-O2", "-fPIC", "-std=c++17", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-w"]
    objs = []
    for rel, oname in srcs:
        o = os.path.join(out_dir, oname)
        subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(csrc, rel), "-o", o])
        objs.append(o)
    shim_o = os.path.join(out_dir, "ref_shim.o")
    subprocess.check_call(["g++"] + flags + inc + ["-c", os.path.join(HERE, "ref_shim.cpp"), "-o", shim_o])
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    subprocess.check_call(["g++", "-shared", "-o", lib] + objs + [shim_o, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10",
                                                                 "-Wl,-rpath," + tlib])
    for o in objs + [shim_o]:
        os.remove(o)
    print(lib)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
