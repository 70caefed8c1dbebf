"""Per-workgroup timeline of one gate_up launch (diagnostic build with M355_STAMPS)."""
import ctypes, importlib, sys, torch
import numpy as np
sys.path.insert(0, ".")
import bench
lib = importlib.import_module("gpt-oss-vllm_amd.lib")
ep_mod = importlib.import_module("gpt-oss-vllm_amd.ep")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "gate_up"
st = bench.DecodeStep(lib, ep_mod, B, 128, 6, 0, 1, dev)
L = lib.load()
buf = torch.zeros((1024, 32), dtype=torch.int64, device=dev)
F = lib.FLAG_FP16_COMBINE
for lay in st.L:
    t = torch.arange(B, device=dev).reshape(B, 1); k = torch.arange(4, device=dev).reshape(1, 4)
    lay["ids"].copy_(((4 * t + k) % 32).to(torch.int32)); lay["w"].fill_(0.25)
    ep = lay["ep"]
    lay["_args"] = dict(E_local=ep.epr, e_first=0, H=bench.H, FF=bench.FF, workspace=ep.workspace, out=lay["out"])
    lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *ep.weights, flags=F | lib.FLAG_SKIP_COMBINE, **lay["_args"])
torch.cuda.synchronize()
flg = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_COMBINE | (lib.FLAG_SKIP_DOWN if mode == "gate_up" else lib.FLAG_SKIP_GATE_UP)
for rep in range(3):
    for li, lay in enumerate(st.L):
        if rep == 2 and li == 3:
            assert L.m355_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=flg, **lay["_args"])
        if rep == 2 and li == 3:
            torch.cuda.synchronize()
            assert L.m355_debug_set_stamps(ctypes.c_void_p(0)) == 0
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
used = s[:, 0] > 0
s = s[used]
t0 = s[:, 0].min()
rel = lambda a: (a - t0) / 100.0   # us
print("workgroups %d" % used.sum())
print("start   : min %.2f med %.2f max %.2f us" % tuple(np.percentile(rel(s[:, 0]), [0, 50, 100])))
print("lut done: min %.2f med %.2f max %.2f (lut build med %.2f)" % (*np.percentile(rel(s[:, 1]), [0, 50, 100]), np.median(s[:, 1] - s[:, 0]) / 100))
print("loads issued: med %.2f | codes barrier: med %.2f" % (np.median(rel(s[:, 28])), np.median(rel(s[:, 29]))))
nit = (s[:, 2:28] > 0).sum(axis=1)
print("items per workgroup: min %d max %d" % (nit.min(), nit.max()))
prev = s[:, 1]
for j in range(nit.max()):
    col = s[:, 2 + j]; ok = col > 0
    d = (col[ok] - prev[ok]) / 100.0
    print("item %2d: n=%3d dur min %.2f med %.2f max %.2f | end med %.2f max %.2f" % (j, ok.sum(), d.min(), np.median(d), d.max(), np.median(rel(col[ok])), rel(col[ok]).max()))
    prev = np.where(ok, col, prev)
print("end     : min %.2f med %.2f max %.2f us" % tuple(np.percentile(rel(s[:, 31]), [0, 50, 100])))
if (s[:, 20] > 0).all():
    base = s[:, 6]   # end of item 4 (round 5 = sixth item)
    for a, b, name in ((6, 20, "item start -> before block 0"), (20, 21, "block 0 (incl. wait for W[0])"), (21, 22, "blocks 1-5"), (22, 23, "blocks 6-11"), (23, 7, "reduce+barrier+epilogue")):
        d = (s[:, b] - s[:, a]) / 100.0
        print("  %-34s min %.2f med %.2f max %.2f" % (name, d.min(), np.median(d), d.max()))
