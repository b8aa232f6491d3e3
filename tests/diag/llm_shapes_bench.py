import sys, os, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev='cuda'
def bench(M,N,K,it=20,**kw):
    a = (torch.randn((M,K), device=dev)*0.5).bfloat16(); w = (torch.randn((N,K), device=dev)*0.05).bfloat16()
    f32 = kw.get('out_f32', False)
    nout = N//2 if kw.get('act')==3 else N
    out = torch.empty((M,nout), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    if 'resid' in kw: kw['resid']=torch.randn((M,N),device=dev); out = kw['resid']
    for _ in range(3): ops.gemm(a,w,out=out,tile=256,**kw)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(a,w,out=out,tile=256,**kw)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    return "%.1f us %.0f TF" % (ms*1e3, 2.0*M*N*K/ms/1e9)
print("256x256 kernel on the four LLaMA GEMM shapes at 14 img/GPU")
print("qkv   ", bench(8148,12288,4096))
print("o     ", bench(8148,4096,4096,resid=1,out_f32=True))
print("gateup", bench(8148,22016,4096,act=3))
print("down  ", bench(8148,4096,11008,resid=1,out_f32=True))
