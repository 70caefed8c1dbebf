"""CPU cost of enqueueing each call of the sharded decode step (one rank, RCCL initialised)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
import gpt_oss_vllm_amd
from gpt_oss_vllm_amd import ep as ep_mod, lib
lib.load()
st = bench.DecodeStep(lib, ep_mod, 32, 256, 4, 0, 1, dev, force_exchange=True)
st.run_eager(); torch.cuda.synchronize()
lay = st.L[0]; ep = lay["ep"]
calls = {
 "attn": lambda: lib.paged_attn_decode(lay["q"], lay["kc"], lay["vc"], st.block_tables, st.ctx_lens, max_ctx=256, scale=0.125, out=lay["attn_out"]),
 "router": lambda: lib.router_topk(lay["x"], lay["wr"], None, 4, ids=lay["ids"], w=lay["w"]),
 "pack": lambda: ep.pack(lay["x"], lay["ids"], lay["w"]),
 "a2a": ep.dispatch,
 "experts(unpack+moe)": ep.experts,
 "a2a_back": ep.combine,
 "reduce": lambda: ep.finish(lay["out"]),
 "empty_stream_query": lambda: torch.cuda.current_stream().cuda_stream,
}
for name, fn in calls.items():
    for _ in range(20): fn()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:24s} cpu {1e6*(t1-t0)/n:7.1f} us/call   incl. drain {1e6*(t2-t0)/n:7.1f} us/call", flush=True)

import time
for mode in ("plain", "prepared"):
    if mode == "prepared": st.prepare()
    for _ in range(5): st.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): st.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: step of 4 layers: host enqueue {1e6*(t1-t0)/20:.0f} us, with drain {1e6*(t2-t0)/20:.0f} us", flush=True)

# the same step with the all-to-all replaced by a copy on the launch stream (one rank: same bytes), to see what the
# collective's own stream and its event hops cost
for lay in st.L:
    lay["ep"].exchange = lambda recv, send: recv.copy_(send)
st.graph = None
st.prepare()
for _ in range(5): st.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): st.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"prepared, same-stream copy instead of all_to_all_single: host enqueue {1e6*(t1-t0)/20:.0f} us, with drain {1e6*(t2-t0)/20:.0f} us", flush=True)
pp = st.graph
names = ["k0","a2a","k1","a2a_back","k2"]
for i in range(5):
    isc, fn = pp[i]
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"piece {i} ({'collective' if isc else 'kernels'}): host {1e6*(t1-t0)/200:.1f} us", flush=True)
dist.destroy_process_group()
