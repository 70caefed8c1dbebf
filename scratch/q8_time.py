import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
H=FF=2880; E=32; k=4; L=12
d=torch.device('cuda')
g=torch.Generator(device=d); g.manual_seed(1)
layers=[]
for l in range(L):
    c13=torch.randint(0,256,(E,2*FF*H//64),dtype=torch.uint8,device=d,generator=g)
    c2=torch.randint(0,256,(E,H*FF//64),dtype=torch.uint8,device=d,generator=g)
    t13=torch.rand((E,256),device=d,generator=g)*0.03
    t2=torch.rand((E,256),device=d,generator=g)*0.03
    layers.append(dict(
        w13_q=torch.randint(0,256,(E,FF*H),dtype=torch.uint8,device=d,generator=g),
        w13_a=torch.gather(t13,1,c13.long()), c13=c13, t13=t13,
        w2_q=torch.randint(0,256,(E,H*FF//2),dtype=torch.uint8,device=d,generator=g),
        w2_a=torch.gather(t2,1,c2.long()), c2=c2, t2=t2,
        wr=(torch.randn((E,H),device=d,generator=g)*0.02).half(),
        x=torch.randn((B,H),device=d,generator=g).half(),
        ws=torch.empty(lib.moe_workspace_bytes(B,k,H,FF,E),dtype=torch.uint8,device=d)))
    layers[-1]['ids'],layers[-1]['w']=lib.router_topk(layers[-1]['x'],layers[-1]['wr'],None,k)
out=torch.empty((B,H),dtype=torch.float16,device=d)
F=lib.FLAG_FP16_COMBINE
def run(q8, flags):
    for lay in layers:
        a13=(lay['c13'],lay['t13']) if q8 else lay['w13_a']
        a2=(lay['c2'],lay['t2']) if q8 else lay['w2_a']
        lib.moe_nf4_forward(lay['x'],lay['ids'],lay['w'],lay['w13_q'],a13,lay['w2_q'],a2,E_local=E,e_first=0,H=H,FF=FF,workspace=lay['ws'],out=out,flags=flags)
for q8 in (False, True, False, True):
    for name, flags in (("all",F),("gate_up",F|lib.FLAG_SKIP_ROUTE|lib.FLAG_SKIP_DOWN|lib.FLAG_SKIP_COMBINE),("down",F|lib.FLAG_SKIP_ROUTE|lib.FLAG_SKIP_GATE_UP|lib.FLAG_SKIP_COMBINE)):
        run(q8, F); torch.cuda.synchronize()
        gr=torch.cuda.CUDAGraph()
        s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): run(q8, flags)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(gr): run(q8, flags)
        gr.replay(); torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): gr.replay()
        b.record(); b.synchronize()
        print("q8" if q8 else "f32", name, "%.2f us/launch" % (a.elapsed_time(b)*1e3/(10*L)))
