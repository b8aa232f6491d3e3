"""Empirical MFMA ceiling + sustained shader clock on the box (diagnostic; see DESIGN.md 'what bounds the GEMM').
Also: how many waves per SIMD does it take to keep the matrix pipe full?  (1 wave/SIMD = what a ping-pong phase has.)"""
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
    for threads, occ in ((256, 1), (256, 2), (512, 1), (256, 4)):
        for dur in (200000, 200001):
            blocks = 256 * occ
            lib.diag_mfma_peak(shape, blocks, threads, 2000, sink.data_ptr(), clk.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fl = lib.diag_mfma_peak(shape, blocks, threads, dur, sink.data_ptr(), clk.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            c = clk.tolist()
            mhz = c[0] / max(c[1], 1) * 100
            per = c[0] / (dur * (16 if shape == 16 else 8))
            print(f"mfma {shape}: {threads * occ // 256} wave(s)/SIMD ({blocks} x {threads}), {'random' if dur & 1 else 'constant'} operands: "
                  f"{fl / ms / 1e9:.0f} TFLOP/s at {mhz:.0f} MHz; {per:.1f} clk per MFMA per wave")
