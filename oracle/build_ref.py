"""Build oracle/_ref/ from the reference's own sources where they lie (TEST INFRASTRUCTURE ONLY; outputs are git-ignored):
  * libmmcv_ref.so  -- g++ directly on four reference CPU files (nms, roi_align) + our shim; no reference build system;
  * eval_rec.pyc, eval_lvis.pyc -- the reference's evaluation scripts (groma/eval/eval_rec.py, eval_lvis.py) byte-compiled by
    py_compile, so that tests/test_00_reference_eval_scripts_gpu.py can execute the reference's OWN eval_model() loop -- unchanged --
    against groma_amd.GromaModel on the GPU box, where /root/reference does not exist.
Nothing is copied into the repository.  Usage: python oracle/build_ref.py /root/reference"""
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
    compile_eval_scripts(ref, out_dir)


def compile_eval_scripts(ref, out_dir):
    import py_compile
    for name in ("eval_rec", "eval_lvis"):
        src = os.path.join(ref, "groma", "eval", name + ".py")
        if os.path.exists(src):
            print(py_compile.compile(src, cfile=os.path.join(out_dir, name + ".pyc"), doraise=True))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
