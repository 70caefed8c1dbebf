import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
H=2880; k=4
d=torch.device('cuda')
for T,E in ((8192,32),(8192,128),(2048,32),(1024,32),(1023,32),(512,32),(256,32),(64,32)):
    x=torch.randn((T,H),device=d).half(); wr=(torch.randn((E,H),device=d)*0.02).half()
    ids=torch.empty((T,k),dtype=torch.int32,device=d); w=torch.empty((T,k),dtype=torch.float32,device=d)
    for _ in range(3): lib.router_topk(x,wr,None,k,ids=ids,w=w)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): lib.router_topk(x,wr,None,k,ids=ids,w=w)
    b.record(); b.synchronize()
    print(f"router T={T} E={E}: {a.elapsed_time(b)*1e3/20:.1f} us", flush=True)
