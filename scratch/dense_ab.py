import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
lib = importlib.import_module("gpt_oss_vllm_amd.lib")
dev = torch.device("cuda:0")
for N, K, T in ((5120, 2880, 8192), (2880, 4096, 8192), (2048, 2880, 8192), (4096, 4096, 8192), (5120, 2880, 2048), (2880, 4096, 2048)):
    w = (torch.randn((N, K), device=dev) * 0.05).half()
    x = torch.randn((T, K), device=dev).half()
    out = torch.empty((T, N), device=dev, dtype=torch.float16)
    for _ in range(3):
        lib.linear_f16(x, w, None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lib.linear_f16(x, w, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{os.environ.get('M355_LIB_PATH','main')[-12:]} N={N} K={K} T={T}: {us:.1f} us  {2.0 * T * N * K / us / 1e6:.1f} TFLOP/s", flush=True)
