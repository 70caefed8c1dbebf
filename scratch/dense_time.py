"""linear_f16 at decode sizes on the two gpt-oss projection shapes: microseconds per launch (graph of launches over 12 weight copies)."""
import sys, torch
sys.path.insert(0, '.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
d = torch.device('cuda')
g = torch.Generator(device=d); g.manual_seed(3)
for (N, K) in ((5120, 2880), (2880, 4096)):
    ws = [(torch.randn((N, K), device=d, generator=g) * 0.03).half() for _ in range(12)]
    for T in (1, 4, 5, 8, 16, 17, 32):
        x = torch.randn((T, K), device=d, generator=g).half()
        out = torch.empty((T, N), dtype=torch.float16, device=d)
        def run():
            for w in ws:
                lib.linear_f16(x, w, None, out=out)
        run(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s): run()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): run()
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): gr.replay()
        b.record(); b.synchronize()
        us = a.elapsed_time(b) * 1e3 / (5 * len(ws))
        print(f"N={N} K={K} T={T}: {us:.2f} us  {N * K * 2 / us / 1e6:.2f} TB/s")
