import sys; sys.path.insert(0,'.')
import torch, numpy as np
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
d=torch.device('cuda')
H=2880; E=32
wr=torch.zeros(E,H); wr[0]=torch.arange(H)//2; wr[1]=1.0
wh=wr.half().to(d)
js=[0,1,2,3,7,8,9,15,16,17,511,512,513,1000,2047,2560,2879]
x=torch.zeros(len(js),H)
for i,j in enumerate(js): x[i,j]=1
ids,w,lg=lib.router_topk(x.half().to(d),wh,None,4,want_logits=True)
print([(j,j//2,float(lg[i,0]),float(lg[i,1])) for i,j in enumerate(js)])
