import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
T=int(sys.argv[1]) if len(sys.argv)>1 else 8192
reps=int(sys.argv[2]) if len(sys.argv)>2 else 2
H=FF=2880; E=32; k=4
d=torch.device('cuda')
g=torch.Generator(device=d); g.manual_seed(1)
lay=dict(
    w13_q=torch.randint(0,256,(E,FF*H),dtype=torch.uint8,device=d,generator=g),
    w13_a=(torch.randint(0,256,(E,2*FF*H//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03),
    w2_q=torch.randint(0,256,(E,H*FF//2),dtype=torch.uint8,device=d,generator=g),
    w2_a=(torch.randint(0,256,(E,H*FF//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03))
x=torch.randn((T,H),device=d,generator=g).half()
lg=(torch.randn((T,E),device=d,generator=g)).half()
ids,w=lib.topk_softmax(lg,k)
ws=torch.empty(lib.moe_workspace_bytes(T,k,H,FF,E),dtype=torch.uint8,device=d)
out=torch.empty((T,H),dtype=torch.float16,device=d)
for r in range(reps):
    lib.moe_nf4_forward(x,ids,w,lay['w13_q'],lay['w13_a'],lay['w2_q'],lay['w2_a'],E_local=E,e_first=0,H=H,FF=FF,workspace=ws,out=out,flags=lib.FLAG_FP16_COMBINE|lib.FLAG_FORCE_TILE)
torch.cuda.synchronize()
print("done")
