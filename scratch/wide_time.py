"""Streaming kernel with token tiles of 16 vs 32 vs the tile kernel: (local experts, token slots) as a rank of an N-way
expert shard sees them (N = 32 / EL, slots = N * 32), and unsharded mid-size batches."""
import sys, torch
sys.path.insert(0, '.')
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
H = FF = 2880; E = 32; k = 4; L = 8
d = torch.device('cuda')
g = torch.Generator(device=d); g.manual_seed(1)
def layers_for(EL):
    return [dict(
        w13_q=torch.randint(0, 256, (EL, FF * H), dtype=torch.uint8, device=d, generator=g),
        w13_a=(torch.randint(0, 256, (EL, 2 * FF * H // 64), dtype=torch.uint8, device=d, generator=g), torch.rand((EL, 256), device=d, generator=g) * 0.03),
        w2_q=torch.randint(0, 256, (EL, H * FF // 2), dtype=torch.uint8, device=d, generator=g),
        w2_a=(torch.randint(0, 256, (EL, H * FF // 64), dtype=torch.uint8, device=d, generator=g), torch.rand((EL, 256), device=d, generator=g) * 0.03)) for _ in range(L)]
cases = [(4, 256), (8, 128), (16, 64), (32, 48), (32, 64), (32, 96), (32, 128), (32, 192), (32, 256), (32, 320)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[1]), int(t)) for t in sys.argv[2:]]
cache = {}
for EL, T in cases:
    if EL not in cache:
        cache.clear(); torch.cuda.empty_cache(); cache[EL] = layers_for(EL)
    layers = cache[EL]
    x = torch.randn((T, H), device=d, generator=g).half()
    lg = torch.randn((T, E), device=d, generator=g).half()
    ids, w = lib.topk_softmax(lg, k)
    ws = torch.empty(lib.moe_workspace_bytes(T, k, H, FF, EL), dtype=torch.uint8, device=d)
    out = torch.empty((T, H), dtype=torch.float16, device=d)
    res = []
    outs = {}
    for name, kf in (("narrow", lib.FLAG_FORCE_STREAM), ("wide", lib.FLAG_FORCE_STREAM | lib.FLAG_STREAM_WIDE), ("tile", lib.FLAG_FORCE_TILE)):
        def run():
            for lay in layers:
                lib.moe_nf4_forward(x, ids, w, lay['w13_q'], lay['w13_a'], lay['w2_q'], lay['w2_a'], E_local=EL, e_first=0, H=H, FF=FF, workspace=ws, out=out, flags=1 | kf)
        run(); torch.cuda.synchronize()
        outs[name] = out.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); b.synchronize()
        res.append(f"{name} {a.elapsed_time(b) * 1e3 / (5 * L):.1f}")
    dmax = (outs["wide"].float() - outs["narrow"].float()).abs().max().item()
    ymax = outs["narrow"].float().abs().max().item()
    print(f"EL={EL} slots={T} ({T * k * EL // E} local pairs, {T * k // E} per expert): " + " | ".join(res) + f" us per layer; wide vs narrow max diff {dmax:.2e} of {ymax:.2e}", flush=True)
