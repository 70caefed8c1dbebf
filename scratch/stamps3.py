"""Prologue timeline of the gate_up launch as a decode step runs it (pairs bucketed inside; router inside for <= 8 tokens),
from a -DM355_EXP_STAMPS build (scratch/exp/stamps.so): python scratch/stamps3.py [batch]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ["M355_LIB_PATH"] = os.path.abspath("scratch/exp/stamps.so")
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = FF = 2880; E = 32; k = 4
d = torch.device('cuda')
g = torch.Generator(device=d); g.manual_seed(1)
L = 8
lays = []
for l in range(L):
    lays.append(dict(
        w13_q=torch.randint(0, 256, (E, FF * H), dtype=torch.uint8, device=d, generator=g),
        w13_a=(torch.randint(0, 256, (E, 2 * FF * H // 64), dtype=torch.uint8, device=d, generator=g), torch.rand((E, 256), device=d, generator=g) * 0.03),
        w2_q=torch.randint(0, 256, (E, H * FF // 2), dtype=torch.uint8, device=d, generator=g),
        w2_a=(torch.randint(0, 256, (E, H * FF // 64), dtype=torch.uint8, device=d, generator=g), torch.rand((E, 256), device=d, generator=g) * 0.03),
        wr=(torch.randn((E, H), device=d, generator=g) * 0.02).half(), x=torch.randn((B, H), device=d, generator=g).half(),
        ids=torch.empty((B, k), dtype=torch.int32, device=d), w=torch.empty((B, k), dtype=torch.float32, device=d),
        ws=torch.empty(lib.moe_workspace_bytes(B, k, H, FF, E), dtype=torch.uint8, device=d)))
out = torch.empty((B, H), dtype=torch.float16, device=d)
F = lib.FLAG_FP16_COMBINE | lib.FLAG_SKIP_DOWN | lib.FLAG_SKIP_COMBINE
def one(lay):
    if B <= 8:
        lib.moe_nf4_forward_routed(lay["x"], lay["wr"], None, k, lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"], E_local=E, e_first=0, H=H, FF=FF,
                                   ids=lay["ids"], w=lay["w"], workspace=lay["ws"], out=out, flags=F)
    else:
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"], E_local=E, e_first=0, H=H, FF=FF, workspace=lay["ws"], out=out, flags=F)
for lay in lays:
    lay["ids"], lay["w"] = lib.router_topk(lay["x"], lay["wr"], None, k)
for rep in range(3):
    for lay in lays:
        one(lay)
torch.cuda.synchronize()
buf = np.zeros(512 * 2 * 128, dtype=np.uint64)
assert lib.load().m355_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
buf = buf.reshape(512, 2, 128).astype(np.int64)
names = ["logits (router)", "routed", "range known", "first places", "codes requested", "loads issued", "code table built", "loop start",
         "item 0 blocks done", "item 0 barrier", "item 0 epilogue"]
rows = []
for wg in range(512):
    r = buf[wg, 0]
    n = int(r[2])
    if n <= 0 or n > 30 or r[3] == 0:
        continue
    t = [r[127], r[124], r[121], r[122], r[123], r[125], r[126], r[4], r[6], r[7], r[8]]
    rows.append((r[0], [x - r[0] for x in t], n, (r[3] - r[1]) * 10))
t0 = min(r[0] for r in rows)
print(f"gate_up as run, bs={B}: {len(rows)} workgroups, items per workgroup {np.mean([r[2] for r in rows]):.2f}; cycles from the workgroup's entry (median, p90)")
for i, nme in enumerate(names):
    v = np.array([r[1][i] for r in rows], dtype=np.float64)
    if (v <= 0).all():
        continue
    print(f"  {nme:22s} {np.median(v):8.0f} {np.percentile(v, 90):8.0f}")
print(f"  entry skew p90 {np.percentile([r[0] - t0 for r in rows], 90):.0f} cycles; loop wall time median {np.median([r[3] for r in rows]):.0f} ns")
