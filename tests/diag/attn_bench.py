import sys, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
for (B, H, L, hd, causal) in [(14, 32, 582, 128, True), (14, 16, 1025, 64, False)]:
    stride = (L + 63) // 64 * 64
    q = torch.randn((B, H, L, hd), device="cuda").bfloat16()
    k = torch.randn((B, H, stride, hd), device="cuda").bfloat16()
    vt = torch.randn((B, H, hd, stride), device="cuda").bfloat16()
    for _ in range(3): ops.attention(q, k, vt, Skv=L, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.attention(q, k, vt, Skv=L, causal=causal)
    e1.record(); torch.cuda.synchronize()
    print(f"hd={hd} L={L} causal={causal}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
