import sys, torch
sys.path.insert(0, '/root/repo')
from groma_amd import ops
dev='cuda'
def bench(B,H,L,hd,causal,it=20):
    S=L; stride=(S+63)//64*64
    q=torch.randn((B,H,L,hd),device=dev).bfloat16(); k=torch.zeros((B,H,stride,hd),device=dev,dtype=torch.bfloat16); vt=torch.zeros((B,H,hd,stride),device=dev,dtype=torch.bfloat16)
    k[:,:,:S]=torch.randn((B,H,S,hd),device=dev).bfloat16(); vt[:,:,:,:S]=torch.randn((B,H,hd,S),device=dev).bfloat16()
    out=torch.empty((B*L,H*hd),device=dev,dtype=torch.bfloat16)
    for _ in range(3): ops.attention(q,k,vt,Skv=S,causal=causal,out=out)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): ops.attention(q,k,vt,Skv=S,causal=causal,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    fl=4.0*L*L*hd*H*B*(0.5 if causal else 1.0)
    print(f"B={B} H={H} L={L} hd={hd} causal={causal}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF (causal counted as half)", flush=True)
bench(7,32,582,128,True); bench(7,16,1025,64,False); bench(14,32,582,128,True); bench(14,16,1025,64,False)
