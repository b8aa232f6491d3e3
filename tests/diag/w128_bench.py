import sys, os, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev='cuda'
def run(M,N,K,tile,**kw):
    torch.manual_seed(0)
    a = (torch.randn((M,K), device=dev)*0.5).bfloat16(); w = (torch.randn((N,K), device=dev)*0.05).bfloat16()
    f32 = kw.get('out_f32', False)
    nout = N//2 if kw.get('act')==3 else N
    kw2 = dict(kw)
    if 'resid' in kw2: kw2['resid']=torch.randn((M,N),device=dev)
    if 'bias' in kw2: kw2['bias']=torch.randn((N,),device=dev)
    out = torch.empty((M,nout), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    for _ in range(3): ops.gemm(a,w,out=out,tile=tile,**kw2)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(a,w,out=out,tile=tile,**kw2)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    return out, ms
for name,(M,N,K,kw) in {"qkv":(8148,12288,4096,{}), "o":(8148,4096,4096,dict(resid=1,out_f32=True)), "gateup":(8148,22016,4096,dict(act=3)),
                        "down":(8148,4096,11008,dict(resid=1,out_f32=True)), "vit_fc1":(14350,4096,1024,dict(bias=1,act=1)), "edge":(1000,1000,192,dict(bias=1))}.items():
    o1,t1 = run(M,N,K,256,**kw); o2,t2 = run(M,N,K,257,**kw)
    err = ((o1.float()-o2.float()).norm()/o1.float().norm()).item()
    print(f"{name:8s} pingpong {t1*1e3:8.1f} us {2.0*M*N*K/t1/1e9:6.0f} TF | w128 {t2*1e3:8.1f} us {2.0*M*N*K/t2/1e9:6.0f} TF | rel diff {err:.2e} equal {torch.equal(o1,o2)}", flush=True)
