"""Run with GROMA_HIP_LIB=tests/diag/libgroma_hip_clk.so: shader clock + per-tile phase times of the 256x256 GEMM.
GROMA_CLK_PRECISION=fp16 GROMA_HIP_LIB=tests/diag/libgroma_hip_clk_f16.so: the same for the IEEE-half build."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant
import torch
from groma_amd import ops, _lib
F16 = os.environ.get("GROMA_CLK_PRECISION") == "fp16"
if F16:
    _lib.LIB_PATH_F16, _lib.PRECISION[0] = os.path.abspath(os.environ["GROMA_HIP_LIB"]), "fp16"
else:
    _variant.use_env()

lib = _lib.load()
lib.gr_diag_clk.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 40)()
for (M, N, K) in [(256, 256, 2048), (256, 256, 8192), (256, 2048, 4096), (2048, 2048, 8192), (8192, 8192, 8192)]:
    a = (torch.randn((M, K), device="cuda") * 0.5).to(ops.H16())
    w = (torch.randn((N, K), device="cuda") * 0.5).to(ops.H16())
    out = torch.empty((M, N), device="cuda", dtype=ops.H16())
    for _ in range(3):
        ops.gemm(a, w, out=out, tile=256)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, w, out=out, tile=256)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    lib.gr_diag_clk(buf)
    c = list(buf)
    mhz = (c[2] - c[0]) / max(c[14] - c[12], 1) * 100
    loop_us = (c[13] - c[12]) / 100.0
    epi_us = (c[14] - c[13]) / 100.0
    marks = [c[1]] + [c[3 + i] for i in range(8)]
    ph = c[24:40]
    if ph[0]:
        segs = []
        for i in range(15):
            segs.append(f"{'mfma' if i % 2 == 0 else 'gap '}{ph[i + 1] - ph[i]}")
        print("   K-tiles 8-9, wave 0: " + " ".join(segs) + f" | K-tile period {ph[8] - ph[0]} clk")
    print("   epilogue clk: " + " ".join(f"{'stage' if i % 2 == 0 else 'store'}{i // 2}={marks[i + 1] - marks[i]}" for i in range(8)))
    ks = K // 64
    print(f"{M}x{N}x{K}: {ms * 1e3:.1f} us, {2.0 * M * N * K / ms / 1e9:.0f} TF | block0: shader clock {mhz:.0f} MHz, "
          f"prologue+loop {loop_us:.1f} us ({(c[1] - c[0]) / ks:.0f} clk per K-step; MFMA-bound floor 2048), epilogue {epi_us:.1f} us")
