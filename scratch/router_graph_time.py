"""Router launch time inside a HIP graph of 24 launches on distinct router matrices (as a decode step has them)."""
import sys, torch
sys.path.insert(0, '.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
H = 2880; k = 4; L = 24
d = torch.device('cuda')
big = torch.empty(512 << 20, dtype=torch.uint8, device=d)   # flushed through L2 between replays
for T, E in ((32, 32), (32, 128), (64, 128), (128, 128), (256, 128), (256, 32), (512, 32)):
    lay = [dict(x=torch.randn((T, H), device=d).half(), wr=(torch.randn((E, H), device=d) * 0.02).half(),
                ids=torch.empty((T, k), dtype=torch.int32, device=d), w=torch.empty((T, k), dtype=torch.float32, device=d)) for _ in range(L)]
    def run():
        for l in lay: lib.router_topk(l['x'], l['wr'], None, k, ids=l['ids'], w=l['w'])
    run(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(10):
        big.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        tot += a.elapsed_time(b)
    print(f"router T={T} E={E}: {tot * 1e3 / (10 * L):.2f} us per launch", flush=True)
