"""What the chip does on plain streams of this kernel's read/write mix: torch fill (writes only), copy (1:1), and a 1:4 read:write mix."""
import torch
d = torch.device('cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e-3
N = 800 << 20
x = torch.empty(N, dtype=torch.float16, device=d)
y = torch.empty(N, dtype=torch.float16, device=d)
s = torch.empty(N // 4, dtype=torch.float16, device=d)
print("fill 1.6 GB: %.2f TB/s" % (2 * N / t(lambda: x.fill_(1.0)) / 1e12))
print("copy 1.6+1.6 GB: %.2f TB/s" % (4 * N / t(lambda: y.copy_(x)) / 1e12))
x8 = torch.empty(N // 4, dtype=torch.uint8, device=d)
print("u8 -> f16 convert (1 B in, 2 B out): %.2f TB/s" % (3 * (N // 4) / t(lambda: s.copy_(x8)) / 1e12))
