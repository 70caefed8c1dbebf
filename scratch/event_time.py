"""How HIP events price one launch: a pair of events around every gate_up launch (eager) against the back-to-back
figure (one graph of 24 launches): python scratch/event_time.py [batch]"""
import sys, torch
sys.path.insert(0, '.')
import bench
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import lib, ep as ep_mod
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = torch.device('cuda', 0)
st = bench.DecodeStep(lib, ep_mod, B, 256, 24, 0, 1, d)
st.run_eager(); torch.cuda.synchronize()
r, act = bench.roofline_gate_up(st)
print("back to back (graph):", {k: round(v["avg_us"], 2) for k, v in r.items()})
F = lib.FLAG_FP16_COMBINE | lib.FLAG_SKIP_DOWN | lib.FLAG_SKIP_COMBINE
for rep in range(3):
    evs = []
    for lay in st.L:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=F, **lay["_args"])
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"event pair per launch: mean {sum(ts)/len(ts):.2f} us, median {ts[len(ts)//2]:.2f}, min {ts[0]:.2f}, max {ts[-1]:.2f}")
