import sys, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev = 'cuda'
B, H, hd = 4, 32, 128
layers = 32
for S in (64, 256, 640, 1024, 2048):
    stride = (S + 63) // 64 * 64
    q = torch.randn((B, H, 1, hd), device=dev).bfloat16()
    ks = [torch.randn((B, H, stride, hd), device=dev).bfloat16() for _ in range(layers)]  # distinct caches: HBM-cold like a real step
    vs = [torch.randn((B, H, hd, stride), device=dev).bfloat16() for _ in range(layers)]
    out = torch.empty((B, H * hd), device=dev, dtype=torch.bfloat16)
    for ns in (1, 2, 4):
        for _ in range(2):
            for l in range(layers):
                ops.decode_attention(q, ks[l], vs[l], out, Smax=S, q_pos0=S - 1, nsplit=ns)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for l in range(layers):
                ops.decode_attention(q, ks[l], vs[l], out, Smax=S, q_pos0=S - 1, nsplit=ns)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * layers) * 1e3
        print(f"S={S} nsplit={ns}: {us:.1f} us/launch, KV {B*H*S*hd*4/1e6:.1f} MB -> {B*H*S*hd*4/us/1e6:.2f} TB/s", flush=True)
