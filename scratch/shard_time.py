import sys, torch
sys.path.insert(0,'.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
H=FF=2880; E=32; k=4; EL=4; L=12
d=torch.device('cuda')
g=torch.Generator(device=d); g.manual_seed(1)
layers=[dict(
    w13_q=torch.randint(0,256,(EL,FF*H),dtype=torch.uint8,device=d,generator=g),
    w13_a=(torch.randint(0,256,(EL,2*FF*H//64),dtype=torch.uint8,device=d,generator=g), torch.rand((EL,256),device=d,generator=g)*0.03),
    w2_q=torch.randint(0,256,(EL,H*FF//2),dtype=torch.uint8,device=d,generator=g),
    w2_a=(torch.randint(0,256,(EL,H*FF//64),dtype=torch.uint8,device=d,generator=g), torch.rand((EL,256),device=d,generator=g)*0.03)) for _ in range(L)]
for T in (256,):
    x=torch.randn((T,H),device=d,generator=g).half()
    lg=torch.randn((T,E),device=d,generator=g).half()
    ids,w=lib.topk_softmax(lg,k)
    ws=torch.empty(lib.moe_workspace_bytes(T,k,H,FF,EL),dtype=torch.uint8,device=d)
    out=torch.empty((T,H),dtype=torch.float16,device=d)
    G=lib.FLAG_SKIP_DOWN|lib.FLAG_SKIP_COMBINE; D=lib.FLAG_SKIP_ROUTE|lib.FLAG_SKIP_GATE_UP|lib.FLAG_SKIP_COMBINE
    for name,kf in (("stream16",lib.FLAG_FORCE_STREAM),("stream32",lib.FLAG_FORCE_STREAM|lib.FLAG_STREAM_WIDE),("stream16 route+gate_up",lib.FLAG_FORCE_STREAM|G),("stream32 route+gate_up",lib.FLAG_FORCE_STREAM|lib.FLAG_STREAM_WIDE|G),
                    ("stream16 down",lib.FLAG_FORCE_STREAM|D),("stream32 down",lib.FLAG_FORCE_STREAM|lib.FLAG_STREAM_WIDE|D)):
        def run():
            for lay in layers:
                lib.moe_nf4_forward(x,ids,w,lay['w13_q'],lay['w13_a'],lay['w2_q'],lay['w2_a'],E_local=EL,e_first=8,H=H,FF=FF,workspace=ws,out=out,flags=1|kf)
        run(); torch.cuda.synchronize()
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); b.synchronize()
        print(f"shard of 4 experts, {T} token slots ({T*k*EL//E} local pairs): {name} {a.elapsed_time(b)*1e3/(5*L):.1f} us per layer", flush=True)
