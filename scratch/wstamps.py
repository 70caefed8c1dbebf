"""Per-wave timeline of one gate_up launch (diagnostic build with M355_STAMPS)."""
import ctypes, importlib, sys, torch
import numpy as np
sys.path.insert(0, ".")
import bench
lib = importlib.import_module("gpt-oss-vllm_amd.lib")
ep_mod = importlib.import_module("gpt-oss-vllm_amd.ep")
dev = torch.device("cuda:0")
B = 32
st = bench.DecodeStep(lib, ep_mod, B, 128, 6, 0, 1, dev)
L = lib.load()
buf = torch.zeros((256, 8, 32), dtype=torch.int64, device=dev)
F = lib.FLAG_FP16_COMBINE
for lay in st.L:
    t = torch.arange(B, device=dev).reshape(B, 1); k = torch.arange(4, device=dev).reshape(1, 4)
    lay["ids"].copy_(((4 * t + k) % 32).to(torch.int32)); lay["w"].fill_(0.25)
    ep = lay["ep"]
    lay["_args"] = dict(E_local=ep.epr, e_first=0, H=bench.H, FF=bench.FF, workspace=ep.workspace, out=lay["out"])
    lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *ep.weights, flags=F | lib.FLAG_SKIP_COMBINE, **lay["_args"])
torch.cuda.synchronize()
flg = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_COMBINE | lib.FLAG_SKIP_DOWN
for rep in range(3):
    for li, lay in enumerate(st.L):
        hit = rep == 2 and li == 3
        if hit:
            assert L.m355_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=flg, **lay["_args"])
        if hit:
            torch.cuda.synchronize()
            assert L.m355_debug_set_stamps(ctypes.c_void_p(0)) == 0
s = buf.cpu().numpy().astype(np.int64) / 100.0   # us
t0 = s[:, :, 0].min()
print("start med %.2f  lut done med %.2f" % (np.median(s[:, :, 0]) - t0, np.median(s[:, :, 1]) - t0))
for it in range(1, 12):
    arr, dep = s[:, :, 2 * it], s[:, :, 2 * it + 1]
    prev = s[:, :, 2 * it - 1] if it > 1 else s[:, :, 1]
    ok = arr[:, 0] > 0
    comp = (arr - prev)[ok]            # per wave: time from leaving the previous barrier to arriving at this one
    skew = (arr.max(axis=1) - arr.min(axis=1))[ok]
    wait = (dep - arr)[ok]
    print("item %2d: compute per wave med %.2f (q0 %.2f q1 %.2f q2 %.2f q3 %.2f) | arrival skew med %.2f max %.2f | barrier wait med %.2f" % (
        it, np.median(comp), *[np.median(comp[:, [q, q + 4]]) for q in range(4)], np.median(skew), skew.max(), np.median(wait)))
