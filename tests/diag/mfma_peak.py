"""Empirical MFMA ceiling + sustained shader clock on the box (diagnostic; see DESIGN.md 'what bounds the GEMM')."""
import ctypes, os, subprocess, sys
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libdiag.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.diag_mfma_peak.restype = ctypes.c_double
lib.diag_mfma_peak.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
sink = torch.zeros(4, device="cuda")
clk = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for shape in (16, 32):
    for threads in (256, 512):
        for dur in (200000, 200001):
            blocks = 256 * (512 // threads) * 2
            lib.diag_mfma_peak(shape, blocks, threads, 2000, sink.data_ptr(), clk.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fl = lib.diag_mfma_peak(shape, blocks, threads, dur, sink.data_ptr(), clk.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            c = clk.tolist()
            print(f"mfma {shape}: {blocks} blocks x {threads} thr, iters {dur} ({'random' if dur & 1 else 'constant'} operands): {ms:.2f} ms -> {fl / ms / 1e9:.0f} TFLOP/s; "
                  f"block0 shader clock {c[0] / max(c[1], 1) * 100:.0f} MHz")
