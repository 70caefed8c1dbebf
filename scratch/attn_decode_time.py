"""Decode attention: average launch time over a HIP graph of 24 back-to-back launches on distinct caches."""
import sys, torch
sys.path.insert(0, '.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
Hq, Hkv, D, BLOCK, LAYERS = 64, 8, 64, 16, 24
d = torch.device('cuda')
for B, ctx in ((1, 256), (32, 256), (1, 257), (32, 257), (32, 512), (4, 1024), (32, 1024), (32, 2048), (1, 8192), (256, 256)):
    nb = (ctx + BLOCK - 1) // BLOCK
    bt = torch.randperm(B * nb, device=d).to(torch.int32).reshape(B, nb).contiguous()
    cl = torch.full((B,), ctx, dtype=torch.int32, device=d)
    lay = [dict(q=torch.randn((B, Hq, D), device=d).half(), kc=torch.randn((B * nb, BLOCK, Hkv, D), device=d).half(),
                vc=torch.randn((B * nb, BLOCK, Hkv, D), device=d).half(), out=torch.empty((B, Hq, D), dtype=torch.float16, device=d))
           for _ in range(LAYERS)]
    need = lib.paged_attn_workspace_bytes(B, Hq, D, ctx)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=d)
    def run():
        for l in lay:
            lib.paged_attn_decode(l['q'], l['kc'], l['vc'], bt, cl, max_ctx=ctx, scale=D ** -0.5, out=l['out'], workspace=ws if need else None)
    run(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): g.replay()
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 1e3 / (10 * LAYERS)
    print(f"attn_decode B={B} ctx={ctx}: {us:.2f} us per launch  ({B * ctx * 2048 / us / 1e3:.0f} GB/s)", flush=True)
