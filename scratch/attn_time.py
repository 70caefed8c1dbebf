import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
Hq,Hkv,D=64,8,64
d=torch.device('cuda')
for L,B in ((8192,1),(2048,4),(1024,8),(256,32)):
    T=L*B
    qkv=torch.randn((T,(Hq+2*Hkv)*D),device=d).half()
    q=qkv[:,:Hq*D].view(T,Hq,D); k=qkv[:,Hq*D:(Hq+Hkv)*D].view(T,Hkv,D); v=qkv[:,(Hq+Hkv)*D:].view(T,Hkv,D)
    starts=torch.arange(0,T+1,L,dtype=torch.int32,device=d)
    out=torch.empty((T,Hq,D),dtype=torch.float16,device=d)
    for w in (0,128):
        for _ in range(2): lib.attn_prefill(q,k,v,starts,L,scale=D**-0.5,window=w,out=out)
        torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): lib.attn_prefill(q,k,v,starts,L,scale=D**-0.5,window=w,out=out)
        b.record(); b.synchronize()
        us=a.elapsed_time(b)*1e3/5
        keys = (L*(L+1)/2) if w==0 else sum(min(i+1,w) for i in range(L))
        fl=4.0*keys*D*Hq*B
        print(f"L={L} B={B} window={w}: {us:.1f} us  {fl/us/1e6:.1f} TFLOP/s", flush=True)
