"""Price list of one instruction placed between two MFMAs of a wave that owns its SIMD (see filler_price.hip)."""
import ctypes, os, subprocess, sys
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libfiller.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "filler_price.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.diag_filler_price.restype = ctypes.c_long
lib.diag_filler_price.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
KINDS = ["none", "ds_read_b128", "ds_write_b128", "global_load_dwordx4", "global_load_lds (DMA)", "v_add_u32", "s_nop", "ds_read_b128 + v_add_u32",
         "64-bit address add + global_load", "s_waitcnt vmcnt + ds_write_b128", "cycle of read, read, wait+write, global_load", "s_add_u32",
         "2 x ds_read_b128", "global_load saddr form", "DMA and s_add_u32 in alternate gaps", "global_load in every 4th gap", "ds_write_b128 in every 4th gap",
         "DMA in every 4th gap", "cycle of read, read, write, DMA", "cycle of read, read, write, -"]
SKIP = {8}  # faulted on the box (address nil) -- not needed for the list
g = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
sink = torch.zeros(4, device="cuda")
clk = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for shape in (32, 16):
    base = None
    for kind, name in enumerate(KINDS):
        if kind in SKIP: continue
        iters = 20000
        lib.diag_filler_price(shape, kind, 200, g.data_ptr(), sink.data_ptr(), clk.data_ptr(), st)
        n = lib.diag_filler_price(shape, kind, iters, g.data_ptr(), sink.data_ptr(), clk.data_ptr(), st)
        torch.cuda.synchronize()
        c = clk.tolist()
        per = c[0] / n
        mhz = c[0] / max(c[1], 1) * 100
        if kind == 0: base = per
        print(f"mfma {shape:2d}  {name:48s} {per:6.1f} clk/MFMA  (+{per - base:5.1f})  at {mhz:.0f} MHz", flush=True)
