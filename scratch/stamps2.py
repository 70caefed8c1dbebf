"""Per-workgroup timeline of the streaming MoE GEMM from a -DM355_EXP_STAMPS build (scratch/exp/stamps.so):
python scratch/stamps2.py [batch] [gate_up|down]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
os.environ["M355_LIB_PATH"] = os.path.abspath("scratch/exp/stamps.so")
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "gate_up"
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
        ws=torch.empty(lib.moe_workspace_bytes(B, k, H, FF, E), dtype=torch.uint8, device=d)))
out = torch.empty((B, H), dtype=torch.float16, device=d)
F = lib.FLAG_FP16_COMBINE
for lay in lays:
    lay["ids"], lay["w"] = lib.router_topk(lay["x"], lay["wr"], None, k)
    lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"], E_local=E, e_first=0, H=H, FF=FF, workspace=lay["ws"], out=out)
torch.cuda.synchronize()
only = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_COMBINE | (lib.FLAG_SKIP_DOWN if which == "gate_up" else lib.FLAG_SKIP_GATE_UP)
for rep in range(3):
    for lay in lays:      # cycle the layers so the weights come from HBM; the stamps of the last launch are read
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"], E_local=E, e_first=0, H=H, FF=FF, workspace=lay["ws"], out=out, flags=only)
torch.cuda.synchronize()
buf = np.zeros(512 * 2 * 128, dtype=np.uint64)
rc = lib.load().m355_dbg_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0
buf = buf.reshape(512, 2, 128).astype(np.int64)
rows = []
for wg in range(512):
    for hw in range(2):
        r = buf[wg, hw]
        n = int(r[2])
        if n <= 0 or n > 30 or r[3] == 0:
            continue
        st = r[4:4 + 1 + 4 * n]
        clk = (st[-1] - st[0]) / max((r[3] - r[1]) * 10e-9, 1e-12) / 1e9      # cycles per ns over the item loop
        items = st[1:].reshape(n, 4)
        rows.append(dict(wg=wg, hw=hw, n=n, entry_to_loop=st[0] - r[0], blocks=items[:, 1] - items[:, 0], barrier=items[:, 2] - items[:, 1],
                         epi=items[:, 3] - items[:, 2], gap=np.concatenate([[items[0, 0] - st[0]], items[1:, 0] - items[:-1, 3]]),
                         total=st[-1] - r[0], clk=clk, t0=r[0],
                         pro=[r[124] - r[0], r[125] - r[124], r[126] - r[125], st[0] - r[126]]))
print(f"{which} bs={B}: {len(rows)} wave records")
t0min = min(r["t0"] for r in rows)
def stat(name, vals):
    v = np.array(vals, dtype=np.float64)
    print(f"  {name:28s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
stat("clock GHz x1000", [r["clk"] * 1000 for r in rows])
stat("entry skew (cycles)", [r["t0"] - t0min for r in rows])
stat("entry -> item loop", [r["entry_to_loop"] for r in rows])
stat("  entry -> routed", [r["pro"][0] for r in rows])
stat("  routed -> loads issued", [r["pro"][1] for r in rows])
stat("  -> code table built", [r["pro"][2] for r in rows])
stat("  -> scales ready, loop", [r["pro"][3] for r in rows])
stat("first item blocks", [r["blocks"][0] for r in rows])
stat("later items blocks", [b for r in rows for b in r["blocks"][1:]])
stat("barrier wait", [b for r in rows for b in r["barrier"]])
stat("epilogue", [b for r in rows for b in r["epi"]])
stat("between items", [b for r in rows for b in r["gap"]])
stat("items per wg", [r["n"] for r in rows])
stat("entry -> end (cycles)", [r["total"] for r in rows])
stat("end time rel. first entry", [r["t0"] - t0min + r["total"] for r in rows])
