"""GROMA_HIP_LIB=tests/diag/libgroma_hip_clk.so: per-iteration segment clocks of the prefill attention kernel"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant; _variant.use_env()
import torch
from groma_amd import ops, _lib
lib = _lib.load()
lib.gr_diag_att_clk.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 64)()
for (B, H, L) in [(14, 32, 582), (1, 2, 582)]:
    hd, stride = 128, 640
    q = torch.randn((B, H, L, hd), device="cuda").bfloat16()
    k = torch.randn((B, H, stride, hd), device="cuda").bfloat16()
    vt = torch.randn((B, H, hd, stride), device="cuda").bfloat16()
    for _ in range(3):
        ops.attention(q, k, vt, Skv=L, causal=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention(q, k, vt, Skv=L, causal=True)
    e1.record(); torch.cuda.synchronize()
    lib.gr_diag_att_clk(buf)
    c = list(buf)[:20]
    names = ["wait+barrier", "issue+QK", "softmax", "PV", "loop"]
    print(f"B={B} H={H} L={L}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch; last q-block, wave 0, K-tiles 2..5 (clk):")
    for it in range(4):
        m = c[it * 5: it * 5 + 5]
        nxt = c[(it + 1) * 5] if it < 3 else None
        segs = [m[i + 1] - m[i] for i in range(4)] + ([nxt - m[4]] if nxt else [])
        print("   tile", it + 2, " ".join(f"{n}={v}" for n, v in zip(names, segs)), "| iteration", (nxt - m[0]) if nxt else "-")
