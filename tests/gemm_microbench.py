import sys, torch, time
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev = 'cuda'
def bench(M,N,K,tile,conv=None,it=20):
    a = torch.randn((M,K), device=dev).bfloat16(); w = (torch.randn((N,K), device=dev)*0.05).bfloat16()
    out = torch.empty((M,N), dtype=torch.bfloat16, device=dev)
    for _ in range(3): ops.gemm(a,w,out=out,tile=tile)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(a,w,out=out,tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    return ms, 2.0*M*N*K/ms/1e9
shapes=[(4096,4096,4096),(8192,8192,8192),(2328,22016,4096),(2328,4096,11008),(2328,12288,4096),(2328,4096,4096),(4074,22016,4096),(4074,4096,11008),(4074,12288,4096),(4074,4096,4096),(4100,4096,1024),(4100,1024,4096),(7175,4096,1024),(7175,1024,4096),(7175,3072,1024),(65536,1024,9216)]
for s in shapes:
    r=[bench(*s,tile=t) for t in (128,256)]
    print(s, "128: %.1f us %.0f TF | 256: %.1f us %.0f TF" % (r[0][0]*1e3, r[0][1], r[1][0]*1e3, r[1][1]), flush=True)

print("---- fp8 (e4m3) 256x256 kernel")
def bench8(M,N,K,it=20):
    a = torch.randn((M,K), device=dev).bfloat16(); w = (torch.randn((N,K), device=dev)*0.05)
    a8, sa = ops.quant_rows_fp8(a)
    sw = w.abs().amax(-1)/448.0; w8 = (w/sw[:,None]).to(torch.float8_e4m3fn)
    out = torch.empty((M,N), dtype=torch.bfloat16, device=dev)
    for _ in range(3): ops.gemm(a8,w8,a_scale=sa,w_scale=sw,out=out)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(a8,w8,a_scale=sa,w_scale=sw,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    print((M,N,K), "fp8: %.1f us %.0f TF" % (ms*1e3, 2.0*M*N*K/ms/1e9), flush=True)
for s in [(8192,8192,8192),(8148,22016,4096),(8148,4096,11008),(8148,12288,4096),(8148,4096,4096)]: bench8(*s)
