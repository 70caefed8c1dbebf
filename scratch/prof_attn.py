"""A few launches of the prompt attention kernel (8192 tokens, one sequence, full attention) for counter passes."""
import sys, torch
sys.path.insert(0, '.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
Hq, Hkv, D, L = 64, 8, 64, 8192
d = torch.device('cuda')
qkv = torch.randn((L, (Hq + 2 * Hkv) * D), device=d).half()
q = qkv[:, :Hq * D].view(L, Hq, D); k = qkv[:, Hq * D:(Hq + Hkv) * D].view(L, Hkv, D); v = qkv[:, (Hq + Hkv) * D:].view(L, Hkv, D)
starts = torch.tensor([0, L], dtype=torch.int32, device=d)
out = torch.empty((L, Hq, D), dtype=torch.float16, device=d)
for _ in range(3):
    lib.attn_prefill(q, k, v, starts, L, scale=D ** -0.5, window=0, out=out)
torch.cuda.synchronize()
