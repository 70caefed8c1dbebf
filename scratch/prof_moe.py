import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
reps=int(sys.argv[2]) if len(sys.argv)>2 else 3
H=FF=2880; E=32; k=4; L=6
d=torch.device('cuda')
g=torch.Generator(device=d); g.manual_seed(1)
layers=[]
for l in range(L):
    layers.append(dict(
        w13_q=torch.randint(0,256,(E,FF*H),dtype=torch.uint8,device=d,generator=g),
        w13_a=(torch.randint(0,256,(E,2*FF*H//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03),
        w2_q=torch.randint(0,256,(E,H*FF//2),dtype=torch.uint8,device=d,generator=g),
        w2_a=(torch.randint(0,256,(E,H*FF//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03),
        wr=(torch.randn((E,H),device=d,generator=g)*0.02).half(),
        x=torch.randn((B,H),device=d,generator=g).half()))
ws=torch.empty(lib.moe_workspace_bytes(B,k,H,FF,E),dtype=torch.uint8,device=d)
out=torch.empty((B,H),dtype=torch.float16,device=d)
for r in range(reps):
    for lay in layers:
        ids,w=lib.router_topk(lay['x'],lay['wr'],None,k)
        lib.moe_nf4_forward(lay['x'],ids,w,lay['w13_q'],lay['w13_a'],lay['w2_q'],lay['w2_a'],E_local=E,e_first=0,H=H,FF=FF,workspace=ws,out=out)
torch.cuda.synchronize()
print("done")
