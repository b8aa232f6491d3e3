"""How much VALU work fits in the shadow of an MFMA stream inside ONE wave (tests/diag/valu_shadow.hip): clocks per slot of
"MFMA + n fillers" against the bare MFMA stream and the bare filler stream, one wave per SIMD.
   python tests/diag/valu_shadow.py"""
import ctypes, os, subprocess
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libvalu_shadow.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "valu_shadow.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.diag_valu_shadow.restype = ctypes.c_long
lib.diag_valu_shadow.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3
KINDS = ["v_fma_f32", "v_exp_f32", "v_exp_f32 + v_fma_f32 + v_add_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32"]
sink = torch.zeros(4, device="cuda")
clk = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(kind, n, mode, blocks):
    lib.diag_valu_shadow(kind, n, mode, blocks, 200, sink.data_ptr(), clk.data_ptr(), st)
    slots = lib.diag_valu_shadow(kind, n, mode, blocks, 20000, sink.data_ptr(), clk.data_ptr(), st)
    torch.cuda.synchronize()
    c = clk.tolist()
    return c[0] / slots, c[0] / max(c[1], 1) * 100


for blocks, tag in ((256, "1 wave / SIMD"),):   # (one block per CU: a wave has its SIMD to itself)
    base, mhz = run(0, 1, 1, blocks)
    print(f"--- {tag}: bare MFMA stream {base:.1f} clk per MFMA at {mhz:.0f} MHz")
    for kind, name in enumerate(KINDS):
        for n in (1, 2, 3, 4, 6):
            if kind == 2 and n > 2:
                continue
            both, _ = run(kind, n, 0, blocks)
            alone, _ = run(kind, n, 2, blocks)
            print(f"{tag}  {n} x {name:36s}: MFMA + fillers {both:6.1f} clk/slot   fillers alone {alone:6.1f}   sum {base + alone:6.1f}   hidden {base + alone - both:6.1f}", flush=True)
