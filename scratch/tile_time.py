import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
H=FF=2880; E=32; k=4
d=torch.device('cuda')
g=torch.Generator(device=d); g.manual_seed(1)
L=2
layers=[]
for l in range(L):
    layers.append(dict(
        w13_q=torch.randint(0,256,(E,FF*H),dtype=torch.uint8,device=d,generator=g),
        w13_a=(torch.randint(0,256,(E,2*FF*H//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03),
        w2_q=torch.randint(0,256,(E,H*FF//2),dtype=torch.uint8,device=d,generator=g),
        w2_a=(torch.randint(0,256,(E,H*FF//64),dtype=torch.uint8,device=d,generator=g), torch.rand((E,256),device=d,generator=g)*0.03)))
F=lib.FLAG_FP16_COMBINE
for T in [int(a) for a in sys.argv[1:]] or [8192, 2048, 512, 128]:
    x=torch.randn((T,H),device=d,generator=g).half()
    lg=(torch.randn((T,E),device=d,generator=g)).half()
    ids,w=lib.topk_softmax(lg,k)
    ws=torch.empty(lib.moe_workspace_bytes(T,k,H,FF,E),dtype=torch.uint8,device=d)
    out=torch.empty((T,H),dtype=torch.float16,device=d)
    for kern,kf in (("tile",lib.FLAG_FORCE_TILE),("stream",lib.FLAG_FORCE_STREAM)):
        if kern=="stream" and T>2048: continue
        def run(flags):
            for lay in layers:
                lib.moe_nf4_forward(x,ids,w,lay['w13_q'],lay['w13_a'],lay['w2_q'],lay['w2_a'],E_local=E,e_first=0,H=H,FF=FF,workspace=ws,out=out,flags=flags|kf)
        run(F); torch.cuda.synchronize()
        for name, flags in (("all",F),("gate_up",F|lib.FLAG_SKIP_ROUTE|lib.FLAG_SKIP_DOWN|lib.FLAG_SKIP_COMBINE),("down",F|lib.FLAG_SKIP_ROUTE|lib.FLAG_SKIP_GATE_UP|lib.FLAG_SKIP_COMBINE)):
            run(flags); torch.cuda.synchronize()
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            reps=3
            a.record()
            for _ in range(reps): run(flags)
            b.record(); b.synchronize()
            us=a.elapsed_time(b)*1e3/(reps*L)
            fl={"all":3,"gate_up":2,"down":1}[name]*2*H*FF*T*k
            print(f"T={T} {kern} {name}: {us:.1f} us  {fl/us/1e6:.1f} TFLOP/s", flush=True)
