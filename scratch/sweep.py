"""gate_up / down launch time against the number of active experts (fixed cost vs per-byte cost)."""
import importlib, sys, torch
sys.path.insert(0, ".")
import bench
lib = importlib.import_module("gpt-oss-vllm_amd.lib")
ep_mod = importlib.import_module("gpt-oss-vllm_amd.ep")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = bench.DecodeStep(lib, ep_mod, B, 128, 24, 0, 1, dev)
F = lib.FLAG_FP16_COMBINE
for n_act in (4, 8, 16, 24, 32):
    for lay in st.L:
        # token t, slot k -> expert (4t + k) mod n_act: every expert gets the same share
        t = torch.arange(B, device=dev).reshape(B, 1)
        k = torch.arange(4, device=dev).reshape(1, 4)
        lay["ids"].copy_(((4 * t + k) % n_act).to(torch.int32))
        lay["w"].fill_(0.25)
        ep = lay["ep"]
        lay["_args"] = dict(E_local=ep.epr, e_first=0, H=bench.H, FF=bench.FF, workspace=ep.workspace, out=lay["out"])
        lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *ep.weights, flags=F | lib.FLAG_SKIP_COMBINE, **lay["_args"])
    torch.cuda.synchronize()
    act = len(torch.unique(st.L[0]["ids"]))
    line = "B=%d active=%d" % (B, act)
    for name, flg, per in (("gate_up", F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_DOWN | lib.FLAG_SKIP_COMBINE, bench.GATE_UP_BYTES),
                           ("down", F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_GATE_UP | lib.FLAG_SKIP_COMBINE, bench.DOWN_BYTES)):
        def launches():
            for lay in st.L:
                lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=flg, **lay["_args"])
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            launches()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launches()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record(); b.synchronize()
        us = a.elapsed_time(b) * 1e3 / (5 * 24)
        line += " | %s %.1f us %.0f GB/s" % (name, us, act * per / us / 1e3)
        del g
    print(line, flush=True)
