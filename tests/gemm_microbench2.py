import sys, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev = 'cuda'
def bench(M,N,K,tile,it=20,**kw):
    a = torch.randn((M,K), device=dev).bfloat16(); w = (torch.randn((N,K), device=dev)*0.05).bfloat16()
    if kw.get('act')==3: nout=N//2
    else: nout=N
    f32 = kw.get('out_f32', False)
    out = torch.empty((M,nout), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    if 'bias' in kw: kw['bias']=torch.randn((N,),device=dev)
    if 'scale' in kw: kw['scale']=torch.randn((N,),device=dev)
    if 'resid' in kw: kw['resid']=torch.randn((M,N),device=dev)
    for _ in range(3): ops.gemm(a,w,out=out,tile=tile,**kw)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.gemm(a,w,out=out,tile=tile,**kw)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    return "%.1f us %.0f TF" % (ms*1e3, 2.0*M*N*K/ms/1e9)
for (M,N,K) in [(14350,4096,1024),(14350,3072,1024),(14350,1024,1024),(14350,1024,4096)]:
    for tile in (128,256):
        print((M,N,K), tile, "plain", bench(M,N,K,tile), "| bias", bench(M,N,K,tile,bias=1), "| bias+gelu", bench(M,N,K,tile,bias=1,act=1),
              "| f32 bias scale resid", bench(M,N,K,tile,bias=1,scale=1,resid=1,out_f32=True), flush=True)
