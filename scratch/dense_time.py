"""Time m355_linear_f16 over token counts for the q/k/v and o projection shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
lib = importlib.import_module("gpt_oss_vllm_amd.lib")
dev = torch.device("cuda:0")
for N, K in ((5120, 2880), (2880, 4096)):
    w = (torch.randn((N, K), device=dev) * 0.05).half()
    b = torch.randn((N,), device=dev).half()
    for T in ([int(a) for a in sys.argv[1:]] or (64, 128, 256, 512, 1024, 2048, 4096, 8192)):
        x = torch.randn((T, K), device=dev).half()
        out = torch.empty((T, N), device=dev, dtype=torch.float16)
        for _ in range(3):
            lib.linear_f16(x, w, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            lib.linear_f16(x, w, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"N={N} K={K} T={T}: {us:.1f} us  {2.0 * T * N * K / us / 1e6:.1f} TFLOP/s", flush=True)
