#!/usr/bin/env python3
"""Decode-step benchmark of the NF4 MoE hot path on MI355X.

One "step" = one decode token for every sequence of the batch through the hot path of all 24 layers of a
gpt-oss-20b-shaped model: paged-attention decode, router top-k, fused NF4 MoE (gate/up GEMM, down GEMM,
combine).  Nothing outside that path runs in the timed region (no qkv/o projections, norms, lm_head).
Weights are synthetic (random NF4 codes and absmax, already in tile order), inputs are resident in HBM.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
      bench.py --gpus N --steps K --warmup W

With N > 1 the 32 experts of every layer are sharded over the ranks (expert parallelism, two all-to-alls per
layer); each rank keeps its own batch (attention and router are data-parallel), so the global batch is
N * batch-per-gpu ("weak" scaling).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = FF = 2880
E, TOPK, LAYERS = 32, 4, 24
HQ, HKV, D, BLOCK = 64, 8, 64, 16
GATE_UP_BYTES = (2 * FF * H) // 2 + (2 * FF * H) // 64      # packed codes + 1 B absmax per 64: 8,553,600
DOWN_BYTES = (H * FF) // 2 + (H * FF) // 64                  # 4,276,800
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16
GATE_UP_FLOPS_PER_PAIR = 2 * 2 * FF * H   # one (token, expert) pair through W13


STAMPED_SOURCES = ("gpt-oss-vllm_amd/csrc/moe.hip", "gpt-oss-vllm_amd/csrc/moe_tile.hip", "gpt-oss-vllm_amd/csrc/moe_shared.h")


def git_blob_hash(path):
    """The hash `git hash-object` gives the file (sha1 over "blob <size>\\0" + bytes): computed here because the GPU box
    receives the tree without .git."""
    import hashlib
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_stamps():
    return {rel: git_blob_hash(os.path.join(ROOT, rel)) for rel in STAMPED_SOURCES}


def _newest_counter_profile():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.basename(files[-1])
    except (OSError, ValueError):
        return None, None


def counters_stale(profile):
    """True when the kernel sources differ from the ones the counter profile was collected on (their git blob hashes are
    stamped into the profile by profiles/summarize.py), None when the profile carries no stamps."""
    stamps = (profile or {}).get("source_blobs")
    if not stamps:
        return None
    now = source_stamps()
    return any(now.get(rel) != h for rel, h in stamps.items())


def profiled_mfma_busy():
    """Fraction of the launch's cycles in which the matrix cores were busy, for the tile kernel's gate_up launch, from
    the newest committed counter profile (collected in separate rocprofv3 --pmc passes, not in this run)."""
    d, name = _newest_counter_profile()
    return (None, None) if d is None else (d.get("tile_gate_up_mfma_busy_frac"), name)


def profiled_traffic():
    """HBM bytes per gate_up launch from the newest committed counter profile (profiles/rNN_traffic.json, written by
    profiles/summarize.py from separate rocprofv3 --pmc passes): counters cannot be read from inside this run.
    Returns (bytes, file name, stale)."""
    d, name = _newest_counter_profile()
    return (None, None, None) if d is None else (d.get("gate_up_hbm_bytes_per_launch"), name, counters_stale(d))


class DecodeStep:
    """State and launch sequence of one decode step for `batch` sequences per rank."""

    def __init__(self, lib, ep_mod, batch, ctx, layers, rank, world, device, group=None, weights_from=None,
                 force_exchange=False, absmax="u8"):
        self.lib, self.batch, self.ctx, self.layers = lib, batch, ctx, layers
        self.world, self.rank, self.device = world, rank, device
        g = torch.Generator(device=device)
        g.manual_seed(355 + rank)
        e_local = E // world
        self.e_local, self.e_first = e_local, rank * e_local
        B = batch
        nblk_seq = (ctx + BLOCK - 1) // BLOCK
        nblk = B * nblk_seq
        self.block_tables = torch.randperm(nblk, device=device, generator=g).to(torch.int32).reshape(B, nblk_seq).contiguous()
        self.ctx_lens = torch.full((B,), ctx, dtype=torch.int32, device=device)
        self.L = []
        for li in range(layers):
            lay = {}
            if weights_from is not None:   # share the big tensors with another configuration
                for key in ("w13_q", "w13_a", "w2_q", "w2_a"):
                    lay[key] = weights_from.L[li][key]
            else:
                lay["w13_q"] = torch.randint(0, 256, (e_local, FF * H), dtype=torch.uint8, device=device, generator=g)
                lay["w2_q"] = torch.randint(0, 256, (e_local, H * FF // 2), dtype=torch.uint8, device=device, generator=g)
                if absmax == "u8":   # as the checkpoint stores it: a code byte per block + a 256-entry table per expert
                    lay["w13_a"] = (torch.randint(0, 256, (e_local, 2 * FF * H // 64), dtype=torch.uint8, device=device, generator=g),
                                    torch.rand((e_local, 256), dtype=torch.float32, device=device, generator=g) * 0.03)
                    lay["w2_a"] = (torch.randint(0, 256, (e_local, H * FF // 64), dtype=torch.uint8, device=device, generator=g),
                                   torch.rand((e_local, 256), dtype=torch.float32, device=device, generator=g) * 0.03)
                else:
                    lay["w13_a"] = torch.rand((e_local, 2 * FF * H // 64), dtype=torch.float32, device=device, generator=g) * 0.03
                    lay["w2_a"] = torch.rand((e_local, H * FF // 64), dtype=torch.float32, device=device, generator=g) * 0.03
            lay["wr"] = (torch.randn((E, H), device=device, generator=g) * 0.02).half()
            lay["x"] = torch.randn((B, H), device=device, generator=g).half()
            lay["q"] = torch.randn((B, HQ, D), device=device, generator=g).half()
            lay["kc"] = torch.randn((nblk, BLOCK, HKV, D), device=device, generator=g).half()
            lay["vc"] = torch.randn((nblk, BLOCK, HKV, D), device=device, generator=g).half()
            lay["ids"] = torch.empty((B, TOPK), dtype=torch.int32, device=device)
            lay["w"] = torch.empty((B, TOPK), dtype=torch.float32, device=device)
            lay["attn_out"] = torch.empty((B, HQ, D), dtype=torch.float16, device=device)
            lay["out"] = torch.empty((B, H), dtype=torch.float16, device=device)
            lay["ep"] = ep_mod.ExpertParallelMoE((lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"]), num_experts=E,
                                                 top_k=TOPK, hidden=H, ffn=FF, max_tokens=B, group=group, device=device,
                                                 force_exchange=force_exchange)
            self.L.append(lay)
        self.graph = None

    def pieces(self):
        """The launch sequence of a step as (is_collective, callable) pieces, adjacent kernel-only pieces merged: with
        experts sharded, a layer is [attention, router + pack] -> dispatch all-to-all -> [experts, reading the received
        rows in place] -> combine all-to-all -> [reduce], and the reduce joins the next layer's first piece."""
        lib = self.lib
        seq = []

        def add(is_coll, fn):
            if not is_coll and seq and not seq[-1][0]:
                prev = seq[-1][1]
                seq[-1] = (False, lambda prev=prev, fn=fn: (prev(), fn()))
            else:
                seq.append((is_coll, fn))

        for lay in self.L:
            ep = lay["ep"]
            add(False, lambda lay=lay: lib.paged_attn_decode(lay["q"], lay["kc"], lay["vc"], self.block_tables, self.ctx_lens,
                                                             max_ctx=self.ctx, scale=D ** -0.5, out=lay["attn_out"]))
            if ep.world == 1 and not ep.force_exchange:
                # router and experts in one library call (up to 24 tokens: logits launch + top-k inside gate_up)
                add(False, lambda lay=lay, ep=ep: lib.moe_nf4_forward_routed(
                    lay["x"], lay["wr"], None, TOPK, *ep.weights, E_local=ep.epr, e_first=0, H=H, FF=FF, ids=lay["ids"],
                    w=lay["w"], workspace=ep.workspace, out=lay["out"], flags=ep.flags))
            else:
                add(False, lambda lay=lay, ep=ep: ep.route_and_pack(lay["x"], lay["wr"], None, lay["ids"], lay["w"]))
                add(True, ep.dispatch)
                add(False, ep.experts)
                add(True, ep.combine)
                add(False, lambda lay=lay, ep=ep: ep.finish(lay["out"]))
        return seq

    def run_eager(self):
        for _, fn in self.pieces():
            fn()

    def _capture_one(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def _side_stream_pass(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run_eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def capture(self):
        """The whole step as one HIP graph (collectives included when experts are sharded)."""
        self._side_stream_pass()
        self.graph = [(False, self._capture_one(self.run_eager).replay)]

    def prepare(self):
        """For a step that is launched call by call (experts sharded: collectives between the kernels): every run of
        kernels between two collectives becomes one list of prepared library calls (arguments checked and converted
        once), the collectives stay ordinary torch.distributed calls."""
        self.graph = [(True, fn) if is_coll else (False, self.lib.prepare(fn)) for is_coll, fn in self.pieces()]

    def step(self):
        if self.graph is not None:
            for _, fn in self.graph:
                fn()
        else:
            self.run_eager()

    def active_experts_per_layer(self):
        """Distinct local experts hit per layer by the current routing (host sync)."""
        return [int(torch.unique(lay["ids"]).numel()) for lay in self.L]


def timed(stepper, steps, warmup, world, windows=1):
    """`windows` timed regions of exactly `steps` steps each, after `warmup` untimed steps; every region is bracketed by a
    barrier + synchronize on both sides and is the maximum over the ranks.  Returns the list of region durations."""
    import torch.distributed as dist
    for _ in range(warmup):
        stepper.step()
    out = []
    for _ in range(windows):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=stepper.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out.append(dt)
    return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def roofline_gate_up(stepper, reps=5):
    """Average duration of the streaming kernel's gate_up launches (HIP events on the launch stream, layers cycled so the
    weights come from HBM) against the algorithmic bytes of the experts each launch streams.  `gate_up` is the launch as
    the step runs it: the (token, k) pairs are bucketed inside it, and for up to 24 tokens it first picks the top-k from
    the router logits (which a launch of their own has left in the workspace: not part of this figure);
    `gate_up_given_routing` is the same launch handed finished routing lists (nothing in front of its first weight
    load), `down` the down launch alone (in the step, small batches also combine inside it)."""
    lib = stepper.lib
    F = lib.FLAG_FP16_COMBINE
    gate_up_as_run = F | lib.FLAG_SKIP_DOWN | lib.FLAG_SKIP_COMBINE
    only_gate_up = gate_up_as_run | lib.FLAG_SKIP_ROUTE
    only_down = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_GATE_UP | lib.FLAG_SKIP_COMBINE
    if stepper.world != 1:
        return None
    stepper.run_eager()
    torch.cuda.synchronize()
    active = stepper.active_experts_per_layer()
    res = {}
    topk_inside = stepper.batch <= 24     # the step's entry point then leaves the top-k to the gate_up launch
    for lay in stepper.L:   # every layer owns a workspace: leave its routing (and h for the down GEMM) in place
        ep = lay["ep"]
        lay["_args"] = dict(E_local=ep.epr, e_first=0, H=H, FF=FF, workspace=ep.workspace, out=lay["out"])
        if topk_inside:     # gate_up only: the router logits stay in the workspace (the down stage would overwrite them)
            lib.moe_nf4_forward_routed(lay["x"], lay["wr"], None, TOPK, *ep.weights, ids=lay["ids"], w=lay["w"],
                                       flags=gate_up_as_run, **lay["_args"])
        else:
            lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *ep.weights, flags=F | lib.FLAG_SKIP_COMBINE, **lay["_args"])
    torch.cuda.synchronize()

    def as_run(lay):
        if topk_inside:
            lib.moe_nf4_forward_routed(lay["x"], lay["wr"], None, TOPK, *lay["ep"].weights, ids=lay["ids"], w=lay["w"],
                                       flags=gate_up_as_run | lib.FLAG_LOGITS_READY, **lay["_args"])
        else:
            lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=gate_up_as_run, **lay["_args"])

    def given(flg):
        return lambda lay: lib.moe_nf4_forward(lay["x"], lay["ids"], lay["w"], *lay["ep"].weights, flags=flg, **lay["_args"])

    for name, one, per_expert in (("gate_up", as_run, GATE_UP_BYTES), ("gate_up_given_routing", given(only_gate_up), GATE_UP_BYTES),
                                  ("down", given(only_down), DOWN_BYTES)):
        def launches():
            for lay in stepper.L:
                one(lay)
        # one graph of the 24 back-to-back launches of this kernel, so host launch cost stays out of the events
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            launches()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            launches()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        b.synchronize()
        avg_us = a.elapsed_time(b) * 1e3 / (reps * len(stepper.L))
        bytes_per_launch = sum(active) / len(active) * per_expert
        res[name] = dict(avg_us=avg_us, bytes=bytes_per_launch, gbs=bytes_per_launch / (avg_us * 1e-6) / 1e9)
        del g
        # the launch alone: a pair of events around every launch (plain launches, so the GPU idles between them and the
        # figure holds no gap to a neighbour: what a kernel trace reports as the kernel's duration)
        pairs = []
        for _ in range(3):
            for lay in stepper.L:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                one(lay)
                b.record()
                pairs.append((a, b))
        torch.cuda.synchronize()
        us = [a.elapsed_time(b) * 1e3 for a, b in pairs]
        res[name]["alone_us"] = median(us)
        warm = us[len(stepper.L):]                      # (the first pass over the layers warms caches and clocks)
        res[name]["alone_mean_us"] = sum(warm) / len(warm)
        res[name]["alone_launches"] = len(warm)
        res[name]["alone_gbs"] = bytes_per_launch / (res[name]["alone_mean_us"] * 1e-6) / 1e9
        # the kernel's own duration: start / stop events carried by the launch itself (no marker cost, no gap): the figure
        # a kernel trace reports.  One launch at a time, 24 layers cycled, the first pass dropped.
        own = []
        for rep in range(3):
            for lay in stepper.L:
                lib.time_next_launch()
                one(lay)
                own.append(lib.timed_launch_us())
        own = own[len(stepper.L):]
        res[name]["kernel_us"] = sum(own) / len(own)
        res[name]["kernel_launches"] = len(own)
        res[name]["kernel_gbs"] = bytes_per_launch / (res[name]["kernel_us"] * 1e-6) / 1e9
    return res, active


def cpu_baseline():
    """The reference's CPU path for one MoE layer and one token (dequant of the 4 routed experts + fp16 matmuls),
    timed with the oracle port on this host's cores; scaled to 24 layers."""
    from oracle import ref
    ref.build()
    import numpy as np
    layer = ref.synth_layer(ref.stream_seed(9001), 4, H, FF)
    x = ref.normal_f16(ref.stream_seed(9002), (1, H))
    ids = np.arange(4, dtype=np.int32).reshape(1, 4)
    w = np.full((1, 4), 0.25, dtype=np.float32)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ref.moe_forward(x, ids, w, **layer, E_local=4, H=H, FF=FF)
        ts.append(time.perf_counter() - t0)
    t_stream = sorted(ts)[len(ts) // 2]
    w13 = np.stack([ref.nf4_dequant(layer["w13_q"][e], 2 * FF * H, layer["w13_absmax"][e]) for e in range(4)])
    w2 = np.stack([ref.nf4_dequant(layer["w2_q"][e], H * FF, layer["w2_absmax"][e]) for e in range(4)])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ref.moe_forward(x, ids, w, w13_f16=w13, w2_f16=w2, E_local=4, H=H, FF=FF)
        ts.append(time.perf_counter() - t0)
    t_res = sorted(ts)[len(ts) // 2]
    return {
        "value": 1.0 / (LAYERS * t_stream), "unit": "tokens/s", "cores": ref.threads(), "kind": "port",
        "sample": "1 MoE layer, 1 token, 4 routed experts: NF4->fp16 dequant of gate_up+down then fp16 matmuls "
                  "(streaming, as the reference does), median of 5, scaled x24 layers; attention not included",
        "ms_per_layer_streaming": t_stream * 1e3, "ms_per_layer_resident_fp16": t_res * 1e3,
        "resident_fp16_tokens_per_s": 1.0 / (LAYERS * t_res),
    }


def full_layer_leg(lib, main_step, batch, ctx, layers, device, steps, warmup):
    """The whole decoder stack at decode time (every layer: norm, q/k/v projection, rotary + cache write, attention
    with sinks, o projection, norm, router, NF4 experts with gpt-oss's own activation and biases), one HIP graph;
    embedding and lm_head are not part of it.  NF4 expert tensors are shared with the hot-path legs."""
    from gpt_oss_vllm_amd import layer as layer_mod
    g = torch.Generator(device=device)
    g.manual_seed(4242)
    rnd = lambda shape, std: (torch.randn(shape, device=device, generator=g) * std).half()
    nblk_seq = (ctx + BLOCK) // BLOCK + 1
    block_tables = torch.randperm(batch * nblk_seq, device=device, generator=g).to(torch.int32).reshape(batch, nblk_seq).contiguous()
    ctx_lens = torch.full((batch,), ctx + 1, dtype=torch.int32, device=device)      # the new token is position ctx
    positions = torch.full((batch,), ctx, dtype=torch.int32, device=device)
    slots = (block_tables[:, ctx // BLOCK].to(torch.int64) * BLOCK + ctx % BLOCK).contiguous()
    cos, sin = layer_mod.rope_tables(ctx + 2)
    cos, sin = cos.to(device), sin.to(device)
    stack = []
    for li in range(layers):
        src = main_step.L[li]
        w = dict(ln1=rnd((H,), 0.1) + 1, ln2=rnd((H,), 0.1) + 1, wqkv=rnd(((HQ + 2 * HKV) * D, H), 0.02), bqkv=rnd(((HQ + 2 * HKV) * D,), 0.02),
                 wo=rnd((H, HQ * D), 0.02), bo=rnd((H,), 0.02), sinks=rnd((HQ,), 1.0), router_w=src["wr"], router_b=rnd((E,), 0.02),
                 w13_q=src["w13_q"], w13_absmax=src["w13_a"], w2_q=src["w2_q"], w2_absmax=src["w2_a"],
                 b13=rnd((E, 2 * FF), 0.02), b2=rnd((E, H), 0.02))
        lyr = layer_mod.GptOssDecodeLayer(w, hidden=H, heads=HQ, kv_heads=HKV, head_dim=D, ffn=FF, experts=E, top_k=TOPK,
                                          max_tokens=batch, window=128 if li % 2 == 0 else 0, act_mode=lib.ACT_SWIGLU_OAI)
        # this leg's block tables address one block more per sequence than the hot-path legs' caches hold
        kc = rnd((batch * nblk_seq, BLOCK, HKV, D), 1.0)
        vc = rnd((batch * nblk_seq, BLOCK, HKV, D), 1.0)
        stack.append((lyr, kc, vc))
    x0 = rnd((batch, H), 1.0)
    layer_mod.validate_paging(block_tables, slots, ctx_lens, stack[0][1])     # host check before anything is launched

    def run():
        resid, delta = x0, None
        for i, (lyr, kc, vc) in enumerate(stack):
            # every layer but the last leaves its combine step to the next layer's first kernel
            resid, delta = lyr.forward(resid, delta, positions, slots, block_tables, ctx_lens, kc, vc, cos, sin, max_ctx=ctx + 1,
                                       defer_combine=i + 1 < len(stack))
        return resid, delta

    run()
    torch.cuda.synchronize()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        run()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dense_bytes = layers * 2 * (((HQ + 2 * HKV) * D) * H + H * HQ * D)
    return {"value": batch / dt, "unit": "tokens/s", "ms_per_step": dt * 1e3, "batch": batch, "ctx": ctx,
            "what": "24 whole decoder layers per step (dense q/k/v/o projections, rotary + KV write, attention with sinks and "
                    "alternating 128-token sliding window, norms, router, NF4 experts with swiglu-oai and biases), one HIP graph",
            "dense_projection_bytes_per_step": dense_bytes}


def full_layer_prompt_leg(lib, main_step, tokens, layers, device, reps=3):
    """The whole decoder stack over one prompt of `tokens` tokens (every layer: norm, q/k/v projection on the tiled dense
    kernel, rotary + cache write, prompt attention with sinks, o projection, norm, router, NF4 experts with gpt-oss's
    activation and biases), one HIP graph."""
    from gpt_oss_vllm_amd import layer as layer_mod
    g = torch.Generator(device=device)
    g.manual_seed(4343)
    rnd = lambda shape, std: (torch.randn(shape, device=device, generator=g) * std).half()
    T = tokens
    nblk = (T + BLOCK - 1) // BLOCK
    kc = torch.zeros((nblk, BLOCK, HKV, D), dtype=torch.float16, device=device)      # one cache, reused by the layers (synthetic)
    vc = torch.zeros_like(kc)
    positions = torch.arange(T, dtype=torch.int32, device=device)
    slots = torch.arange(T, dtype=torch.int64, device=device)
    starts = torch.tensor([0, T], dtype=torch.int32, device=device)
    cos, sin = layer_mod.rope_tables(T)
    cos, sin = cos.to(device), sin.to(device)
    stack = []
    for li in range(layers):
        src = main_step.L[li]
        w = dict(ln1=rnd((H,), 0.1) + 1, ln2=rnd((H,), 0.1) + 1, wqkv=rnd(((HQ + 2 * HKV) * D, H), 0.02), bqkv=rnd(((HQ + 2 * HKV) * D,), 0.02),
                 wo=rnd((H, HQ * D), 0.02), bo=rnd((H,), 0.02), sinks=rnd((HQ,), 1.0), router_w=src["wr"], router_b=rnd((E,), 0.02),
                 w13_q=src["w13_q"], w13_absmax=src["w13_a"], w2_q=src["w2_q"], w2_absmax=src["w2_a"],
                 b13=rnd((E, 2 * FF), 0.02), b2=rnd((E, H), 0.02))
        stack.append(layer_mod.GptOssDecodeLayer(w, hidden=H, heads=HQ, kv_heads=HKV, head_dim=D, ffn=FF, experts=E, top_k=TOPK,
                                                 max_tokens=1, window=128 if li % 2 == 0 else 0, act_mode=lib.ACT_SWIGLU_OAI))
    x0 = rnd((T, H), 1.0)

    def run():
        resid, delta = x0, None
        for lyr in stack:
            resid, delta = lyr.prefill(resid, delta, positions, slots, starts, T, kc, vc, cos, sin)
        return resid, delta

    run()
    torch.cuda.synchronize()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        run()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dense_flops = 2.0 * T * H * ((HQ + 2 * HKV) * D + HQ * D)
    return {"value": T / dt, "unit": "tokens/s", "tokens": T, "ms_per_layer": dt / layers * 1e3,
            "what": f"{layers} whole decoder layers over one {T}-token prompt, one HIP graph",
            "dense_projection_tflop_per_layer": dense_flops / 1e12}


def prefill_leg(lib, main_step, device, tokens, reps=3):
    """The same experts fed a prefill-sized batch (configs[4] of BASELINE.json is an 8k-token prefill): per layer router top-k
    + fused NF4 MoE, where the library picks its kernel for many tokens per expert (256-token tiles, weights
    dequantised once per tile, MFMA-bound), preceded by the prompt attention of the same tokens.  The MFMA figure
    times the gate_up launches alone (one HIP graph of all layers, HIP events on the launch stream)."""
    g = torch.Generator(device=device)
    g.manual_seed(8192)
    T = tokens
    x = torch.randn((T, H), device=device, generator=g).half()
    ids = torch.empty((T, TOPK), dtype=torch.int32, device=device)
    w = torch.empty((T, TOPK), dtype=torch.float32, device=device)
    out = torch.empty((T, H), dtype=torch.float16, device=device)
    ws = torch.empty(lib.moe_workspace_bytes(T, TOPK, H, FF, E), dtype=torch.uint8, device=device)
    F = lib.FLAG_FP16_COMBINE

    # prompt attention of the same tokens: one packed sequence, gpt-oss's alternation of 128-token sliding-window and
    # full-attention layers, q / k / v as slices of one fused projection output (shared by the layers: synthetic)
    qkv = torch.randn((T, (HQ + 2 * HKV) * D), device=device, generator=g).half()
    q_, k_, v_ = (qkv[:, :HQ * D].view(T, HQ, D), qkv[:, HQ * D:(HQ + HKV) * D].view(T, HKV, D),
                  qkv[:, (HQ + HKV) * D:].view(T, HKV, D))
    attn_out = torch.empty((T, HQ, D), dtype=torch.float16, device=device)
    starts = torch.tensor([0, T], dtype=torch.int32, device=device)

    def run(flags, route=True, attention=False):
        for li, lay in enumerate(main_step.L):
            if attention:
                lib.attn_prefill(q_, k_, v_, starts, T, scale=D ** -0.5, window=128 if li % 2 == 0 else 0, out=attn_out)
            if route:
                lib.router_topk(x, lay["wr"], None, TOPK, ids=ids, w=w)
            lib.moe_nf4_forward(x, ids, w, lay["w13_q"], lay["w13_a"], lay["w2_q"], lay["w2_a"], E_local=E, e_first=0,
                                H=H, FF=FF, workspace=ws, out=out, flags=flags)

    def graph_time(fn):
        fn()
        torch.cuda.synchronize()
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fn()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        gr.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            gr.replay()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e-3 / reps

    t_all = graph_time(lambda: run(F, attention=True))
    t_moe = graph_time(lambda: run(F))
    only_gate_up = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_DOWN | lib.FLAG_SKIP_COMBINE
    only_down = F | lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_GATE_UP | lib.FLAG_SKIP_COMBINE
    t_gu = graph_time(lambda: run(only_gate_up, route=False)) / len(main_step.L)
    t_dn = graph_time(lambda: run(only_down, route=False)) / len(main_step.L)
    fl_gu = float(T) * TOPK * GATE_UP_FLOPS_PER_PAIR
    return {
        "value": T / t_all, "unit": "tokens/s", "tokens": T, "ms_per_layer": t_all / len(main_step.L) * 1e3,
        "ms_per_layer_router_and_moe": t_moe / len(main_step.L) * 1e3,
        "what": f"{T}-token prompt (one sequence) through prompt attention (sliding window 128 on even layers, full on odd), "
                f"router top-k and fused NF4 MoE of {len(main_step.L)} layers ({E} experts, top-4: about {T * TOPK // E} "
                f"tokens per expert), one HIP graph",
        "roofline": {"bound": "mfma", "kernel": "moe_tile_gemm_kernel<gate_up> (NF4 dequant fused into the grouped GEMM, 256-token tiles)",
                     "achieved": fl_gu / t_gu / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fl_gu / t_gu / 1e12 / MFMA_PEAK_TFLOPS, "avg_us": t_gu * 1e6, "flops_per_launch": fl_gu,
                     "mfma_busy_frac_counters": profiled_mfma_busy()[0], "mfma_busy_source": profiled_mfma_busy()[1],
                     "mfma_busy_stale": counters_stale(_newest_counter_profile()[0]),
                     "down_kernel": {"achieved": fl_gu / 2 / t_dn / 1e12, "avg_us": t_dn * 1e6,
                                     "frac": fl_gu / 2 / t_dn / 1e12 / MFMA_PEAK_TFLOPS}},
    }


def graph_seconds(fn, reps=5):
    """fn() (library launches on the current stream) captured in one HIP graph; seconds per replay, HIP events."""
    fn()
    torch.cuda.synchronize()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        fn()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return median(ts)


def rank_of_n_leg(lib, main_step, n_ranks, ctx, device):
    """What ONE rank of an n_ranks-way expert-parallel job launches per layer, collectives excluded, timed on this GPU:
    the rank keeps E / n_ranks experts, routes its own `batch` tokens, and runs its experts over the n_ranks * batch token
    slots of the receive buffer (filled here with the rows n_ranks peers with different tokens would have sent it).  Two
    launch sequences: the separate kernels (router, pack, unpack, experts, reduce) and the fused forms (router + pack,
    experts on the rows in place, reduce inside the residual add + RMSNorm).  No collective runs and no other rank
    exists: this prices the kernels of a rank, not the scaling of the job."""
    B, N = main_step.batch, n_ranks
    epr = E // N
    g = torch.Generator(device=device)
    g.manual_seed(8)
    rb = lib.ep_row_bytes(H, TOPK)
    NT = N * B
    flags = lib.FLAG_FP16_COMBINE | lib.kernel_choice_flags(NT, TOPK, E)
    ws = torch.empty(max(256, lib.moe_workspace_bytes(NT, TOPK, H, FF, epr)), dtype=torch.uint8, device=device)
    send = torch.zeros(N * B * rb, dtype=torch.uint8, device=device)
    x_all = torch.zeros((NT, H), dtype=torch.float16, device=device)
    ids_all = torch.empty((NT, TOPK), dtype=torch.int32, device=device)
    w_all = torch.empty((NT, TOPK), dtype=torch.float32, device=device)
    partial = torch.empty((NT, H), dtype=torch.float16, device=device)
    delta = torch.empty((B, H), dtype=torch.float16, device=device)
    resid = torch.randn((B, H), device=device, generator=g).half()
    resid_out = torch.empty_like(resid)
    normed = torch.empty_like(resid)
    norm_w = torch.ones(H, dtype=torch.float16, device=device)
    layers = []
    for lay in main_step.L:
        part = lambda t: (t[0][:epr], t[1][:epr]) if isinstance(t, tuple) else t[:epr]
        recv = torch.zeros(N * B * rb, dtype=torch.uint8, device=device)
        for s_ in range(N):   # peer s_: its own tokens, the same router; this rank is rank 0 and receives chunk 0
            xs = torch.randn((B, H), device=device, generator=g).half()
            lib.router_topk_ep_pack(xs, lay["wr"], None, TOPK, N, epr, send, ids=lay["ids"], w=lay["w"])
            recv.view(N, B * rb)[s_] = send.view(N, B * rb)[0]
        back = (torch.randn((N, B, H), device=device, generator=g) * 0.1).half()
        layers.append(dict(lay=lay, recv=recv, back=back,
                           weights=(lay["w13_q"][:epr], part(lay["w13_a"]), lay["w2_q"][:epr], part(lay["w2_a"]))))
    hit = float(sum((ly["recv"].view(NT, rb)[:, 2 * H:2 * H + 4].contiguous().view(torch.int32) >= 0).sum().item()
                    for ly in layers)) / len(layers)

    def attention(lay):
        lib.paged_attn_decode(lay["q"], lay["kc"], lay["vc"], main_step.block_tables, main_step.ctx_lens, max_ctx=ctx,
                              scale=D ** -0.5, out=lay["attn_out"])

    def separate(with_norm):
        for ly in layers:
            lay = ly["lay"]
            attention(lay)
            lib.router_topk(lay["x"], lay["wr"], None, TOPK, ids=lay["ids"], w=lay["w"])
            lib.ep_pack(lay["x"], lay["ids"], lay["w"], N, epr, send)
            lib.ep_unpack(ly["recv"], NT, TOPK, H, x_all, ids_all, w_all)
            lib.moe_nf4_forward(x_all, ids_all, w_all, *ly["weights"], E_local=epr, e_first=0, H=H, FF=FF, workspace=ws,
                                out=partial, flags=flags)
            lib.ep_reduce(ly["back"], N, B, H, delta)
            if with_norm:
                lib.add_rmsnorm(resid, delta, norm_w, 1e-5, resid_out=resid_out, out=normed)

    def fused(with_norm):
        for ly in layers:
            lay = ly["lay"]
            attention(lay)
            lib.router_topk_ep_pack(lay["x"], lay["wr"], None, TOPK, N, epr, send, ids=lay["ids"], w=lay["w"])
            lib.moe_nf4_forward_ep_rows(ly["recv"], NT, TOPK, *ly["weights"], E_local=epr, e_first=0, H=H, FF=FF,
                                        workspace=ws, out=partial, flags=flags)
            if with_norm:
                lib.ep_reduce_add_rmsnorm(ly["back"], N, resid, norm_w, 1e-5, resid_out=resid_out, out=normed)
            else:
                lib.ep_reduce(ly["back"], N, B, H, delta)

    def only(fn):
        def run():
            for ly in layers:
                fn(ly)
        return graph_seconds(run) / len(layers) * 1e6

    us = lambda fn: graph_seconds(fn) / len(layers) * 1e6
    return {
        "what": f"kernels of one rank of {N} per layer at bs={B} per rank ({epr} local experts, {NT} token slots of which "
                f"{hit:.0f} carry a token for this rank), one HIP graph of {len(layers)} layers, collectives excluded; "
                "no scaling curve is measured here",
        "us_per_layer": us(lambda: fused(False)), "us_per_layer_separate_kernels": us(lambda: separate(False)),
        "us_per_layer_with_add_rmsnorm": us(lambda: fused(True)),
        "us_per_layer_with_add_rmsnorm_separate_kernels": us(lambda: separate(True)),
        "pieces_us": {
            "attention": only(lambda ly: attention(ly["lay"])),
            "router_and_pack": only(lambda ly: lib.router_topk_ep_pack(ly["lay"]["x"], ly["lay"]["wr"], None, TOPK, N, epr, send,
                                                                      ids=ly["lay"]["ids"], w=ly["lay"]["w"])),
            "experts_on_rows": only(lambda ly: lib.moe_nf4_forward_ep_rows(ly["recv"], NT, TOPK, *ly["weights"], E_local=epr,
                                                                           e_first=0, H=H, FF=FF, workspace=ws, out=partial,
                                                                           flags=flags)),
            "reduce": only(lambda ly: lib.ep_reduce(ly["back"], N, B, H, delta)),
            "reduce_add_rmsnorm": only(lambda ly: lib.ep_reduce_add_rmsnorm(ly["back"], N, resid, norm_w, 1e-5,
                                                                            resid_out=resid_out, out=normed)),
        },
    }


def dequant_leg(lib, main_step, device, reps=5):
    """The standalone NF4 -> fp16 dequantiser (m355_nf4_dequant, what the reference's dequant_unsloth_expert_cpu computes)
    over the packed bytes of one layer's experts: GB/s against the algorithmic 0.515625 B read + 2 B written per
    weight (the kernel is handed fp32 absmax, 0.0625 B per weight instead of 0.015625; priced at the algorithmic figure)."""
    lay = main_step.L[0]
    q = torch.cat([lay["w13_q"].reshape(-1), lay["w2_q"].reshape(-1)])
    n = q.numel() * 2
    g = torch.Generator(device=device)
    g.manual_seed(64)
    am = torch.rand(n // 64, dtype=torch.float32, device=device, generator=g) * 0.03
    out = torch.empty(n, dtype=torch.float16, device=device)
    lib.nf4_dequant(q, am, 64, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.nf4_dequant(q, am, 64, out=out)
    b.record()
    b.synchronize()
    t = a.elapsed_time(b) * 1e-3 / reps
    alg = n * 2.515625
    return {"kernel": "nf4_dequant_kernel", "weights": n, "avg_us": t * 1e6, "algorithmic_bytes": alg,
            "achieved": alg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / HBM_PEAK_GBS,
            "what": f"one layer's {lay['w13_q'].shape[0]} experts (gate_up + down), checkpoint-order bytes, fp32 absmax per 64"}


def load_leg(device):
    """Checkpoint -> GPU-resident tile-order weights for one full-size layer (32 experts): a synthetic shard in the
    Unsloth key schema is written to a temporary directory, then read through the plug-in's weight_loader and
    process_weights_after_loading (un-double-quantise absmax, upload packed blocks, repack on the GPU)."""
    import tempfile
    import numpy as np
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint, plugin
    rng = np.random.default_rng(355)
    nmap = torch.linspace(-1, 1, 256)
    tensors = {}
    for e in range(E):
        for proj, n in (("gate_up", 2 * FF * H), ("down", H * FF)):
            key = checkpoint.expert_weight_key(0, e, proj)
            tensors[key] = torch.from_numpy(rng.integers(0, 256, n // 2, dtype=np.uint8)).reshape(-1, 1)
            tensors[key + ".absmax"] = torch.from_numpy(rng.integers(0, 256, n // 64, dtype=np.uint8))
            tensors[key + ".nested_absmax"] = torch.full(((n // 64 + 255) // 256,), 0.03)
            tensors[key + ".nested_quant_map"] = nmap.clone()
    with tempfile.TemporaryDirectory() as d:
        shard = os.path.join(d, "model-00001-of-00001.safetensors")
        save_file(tensors, shard)
        del tensors
        method = plugin.Mi355Nf4MoEConfig().get_quant_method(None, "model.layers.0.mlp.experts")
        layer = torch.nn.Module()
        method.create_weights(layer, E, H, FF, torch.float16, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        checkpoint.load_layer(method, layer, [shard], 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"seconds_per_layer": dt, "seconds_24_layers": dt * LAYERS,
            "what": "one 32-expert layer from a local safetensors shard (page cache warm) to repacked GPU tensors"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed regions of --steps steps each; the reported figure is their median")
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=256)
    ap.add_argument("--shape", choices=("20b", "120b"), default="20b",
                    help="20b: 32 experts, 24 layers (BASELINE configs[1..3]); 120b: 128 experts, 36 layers (the shape of configs[4])")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--eager", action="store_true", help="do not capture the step in a HIP graph")
    ap.add_argument("--absmax", choices=("u8", "fp32"), default="u8",
                    help="absmax storage: u8 = the checkpoint's code byte per block + a 256-entry table per expert (default), "
                         "fp32 = one un-double-quantised fp32 per block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bs1", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="one-rank rehearsal of the multi-GPU path: initialise RCCL and run the token exchange even at --gpus 1")
    ap.add_argument("--graph-dist", action="store_true", help="capture the step in a HIP graph even when it holds collectives")
    ap.add_argument("--prefill-tokens", type=int, default=8192,
                    help="also run a prefill-sized batch through the MoE path (0 = skip): the MFMA-bound leg")
    ap.add_argument("--load-leg", action="store_true", help="also time checkpoint -> GPU weights for one layer")
    ap.add_argument("--full-layer-leg", action="store_true",
                    help="also time the whole decoder stack over one --prefill-tokens prompt (the decode form of that leg is in the default line)")
    ap.add_argument("--no-full-layer", action="store_true", help="skip the whole-decoder-stack decode leg")
    ap.add_argument("--no-dequant-leg", action="store_true")
    ap.add_argument("--no-rank-of-8", action="store_true",
                    help="skip the leg that times the kernels one rank of an 8-way expert-parallel job launches per layer")
    ap.add_argument("--legs", default="all",
                    help="comma-separated legs to run beside the headline measurement (bs1, roofline, prefill, cpu_baseline, "
                         "dequant, full, rank8, or none): profiling runs take one leg per trace so that a kernel row is one workload")
    a = ap.parse_args()
    if a.legs != "all":
        legs = set(a.legs.split(",")) - {"none", ""}
        unknown = legs - {"bs1", "roofline", "prefill", "cpu_baseline", "dequant", "full", "rank8"}
        if unknown:
            raise SystemExit(f"--legs: unknown leg(s) {sorted(unknown)}")
        a.no_bs1 |= "bs1" not in legs
        a.no_roofline |= "roofline" not in legs
        a.no_cpu_baseline |= "cpu_baseline" not in legs
        a.no_dequant_leg |= "dequant" not in legs
        a.no_full_layer |= "full" not in legs
        a.no_rank_of_8 |= "rank8" not in legs
        if "prefill" not in legs:
            a.prefill_tokens = 0
    global E, LAYERS
    if a.shape == "120b":
        E, LAYERS = 128, 36
    if a.layers is None:
        a.layers = LAYERS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    import gpt_oss_vllm_amd  # noqa: F401
    from gpt_oss_vllm_amd import ep as ep_mod
    from gpt_oss_vllm_amd import lib
    lib.load()

    main_step = DecodeStep(lib, ep_mod, a.batch_per_gpu, a.ctx, a.layers, rank, world, device, force_exchange=a.force_dist,
                           absmax=a.absmax)
    main_step.run_eager()          # first call sets kernel attributes; must precede graph capture
    torch.cuda.synchronize()
    sharded = world > 1 or a.force_dist
    # With experts sharded the step is launched call by call: --graph-dist captures it whole, collectives included, which
    # has only been exercised on one rank.  (One graph per run of kernels between two collectives was measured on one
    # rank and is slower than plain launches: 3.98 against 3.74 ms per step, a graph launch costs more than the three or
    # four kernel launches it replaces.)
    use_graph = (not sharded or a.graph_dist) and not a.eager

    def capture(stepper):
        if use_graph:
            stepper.capture()
        elif sharded and not a.eager:
            stepper.prepare()

    capture(main_step)
    dts = timed(main_step, a.steps, a.warmup, world, windows=a.windows)
    dt = median(dts)      # the headline figure is the median window; every window is in `windows_ms_per_step`
    tokens = a.batch_per_gpu * world
    result = {
        "metric": f"decode tokens/sec gpt-oss-{a.shape} NF4 (hot path: paged-attn + router + fused NF4 MoE, {a.layers} layers)",
        "value": tokens * a.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "windows": a.windows,
        "windows_ms_per_step": [round(d / a.steps * 1e3, 4) for d in dts],
        "spread_frac": (max(dts) - min(dts)) / dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {
            "workload": (f"gpt-oss-{a.shape}-shaped NF4 decode, bs={a.batch_per_gpu} per GPU (global {tokens}), ctx {a.ctx}, "
                         f"{a.layers} layers, H=FF=2880, E={E} top-4, 64q/8kv heads d=64; "
                         + (("1 GPU: BASELINE configs[2]" if a.shape == "20b" else "1 GPU, the model shape of BASELINE configs[4]") if world == 1 else f"experts sharded over {world} GPUs (EP, all-to-all): BASELINE configs[3] shape")),
            "batch_per_gpu": a.batch_per_gpu, "global_batch": tokens, "seq_len": a.ctx,
            "parallelism": "single" if world == 1 else f"ep{world}+dp{world}", "hip_graph": use_graph,
            "host_calls": "hip graph" if use_graph else ("plain" if a.eager or not sharded else "prepared"),
            "absmax_storage": a.absmax,
            "attention_parity": "unpinned: the reference's decode attention lives in flash-attn 2 and it holds no fixture; "
                                "the kernel is checked against an fp32 softmax restatement and transformers' eager attention",
        },
    }

    if sharded:
        # what the collective library saw, so a line from an N-GPU run shows that N ranks really took part
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:                                       # the version query is informative only
            rccl = f"unknown ({type(exc).__name__})"
        result["config"]["rccl_ranks"] = dist.get_world_size()
        result["config"]["rccl_version"] = rccl
        result["config"]["exchange"] = ("all_to_all_single of full-capacity chunks: every rank ships "
                                        f"{a.batch_per_gpu} token rows to every peer (dispatch and return), empty rows included")

    def attempt(key, fn):
        """A leg beside the headline measurement: if it fails, the line is still printed and carries the failure in the
        leg's place (the traceback goes to stderr)."""
        try:
            result[key] = fn()
        except Exception as exc:
            import traceback
            traceback.print_exc()
            result[key] = {"error": f"{type(exc).__name__}: {exc}"}

    def bs1_leg():
        # configs[1] of BASELINE.json: bs=1 greedy decode, seq_len 256 (per GPU when N > 1)
        main_step.graph = None
        bs1 = DecodeStep(lib, ep_mod, 1, a.ctx, a.layers, rank, world, device, weights_from=main_step,
                         force_exchange=a.force_dist)
        bs1.run_eager()
        torch.cuda.synchronize()
        capture(bs1)
        dts1 = timed(bs1, a.steps, a.warmup, world, windows=a.windows)
        dt1 = median(dts1)
        out = {"value": world * a.steps / dt1, "unit": "tokens/s", "ms_per_step": dt1 / a.steps * 1e3,
               "windows_ms_per_step": [round(d / a.steps * 1e3, 4) for d in dts1],
               "workload": f"bs=1 per GPU (global {world}), ctx {a.ctx}: BASELINE configs[1]" if world == 1
               else f"bs=1 per GPU (global {world}), ctx {a.ctx}, EP{world}"}
        if rank == 0 and world == 1 and not a.no_roofline:
            bs1.graph = None
            r1, act1 = roofline_gate_up(bs1)
            out["roofline_gate_up"] = {"kernel": "moe_gemm_stream_kernel<gate_up, route = top-k + inline>: the launch as the step runs it (top-k of the "
                                                 "finished router logits and bucketing of the pairs inside)",
                                       "achieved": r1["gate_up"]["kernel_gbs"], "frac": r1["gate_up"]["kernel_gbs"] / HBM_PEAK_GBS,
                                       "avg_us": r1["gate_up"]["kernel_us"], "event_pair_us": r1["gate_up"]["alone_mean_us"],
                                       "back_to_back_us": r1["gate_up"]["avg_us"],
                                       "given_routing": {"avg_us": r1["gate_up_given_routing"]["kernel_us"],
                                                         "frac": r1["gate_up_given_routing"]["kernel_gbs"] / HBM_PEAK_GBS},
                                       "down_gbs": r1["down"]["kernel_gbs"], "down_avg_us": r1["down"]["kernel_us"],
                                       "down_back_to_back_us": r1["down"]["avg_us"]}
        return out

    def roofline_leg():
        r, active = roofline_gate_up(main_step)
        traffic, traffic_src, stale = profiled_traffic()
        tflops = a.batch_per_gpu * TOPK * GATE_UP_FLOPS_PER_PAIR / (r["gate_up"]["kernel_us"] * 1e-6) / 1e12
        return {
            "bound": "hbm", "kernel": "moe_gemm_stream_kernel<gate_up, route = inline> (NF4 dequant fused into the grouped GEMM; "
                                      "the launch as the step runs it, pairs bucketed inside)",
            "achieved": r["gate_up"]["kernel_gbs"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": r["gate_up"]["kernel_gbs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "traffic_stale": stale,
            "avg_us": r["gate_up"]["kernel_us"], "algorithmic_bytes_per_launch": r["gate_up"]["bytes"],
            "avg_us_is": f"mean over {r['gate_up']['kernel_launches']} launches (24 layers cycled, so the weights come from HBM) of the "
                         "kernel's own duration: HIP start / stop events carried by the launch itself (hipExtLaunchKernelGGL, on "
                         "the launch stream), which is what the kernel trace under profiles/ reports for it",
            "event_pair_around_launch": {"avg_us": r["gate_up"]["alone_mean_us"], "median_us": r["gate_up"]["alone_us"],
                                         "frac": r["gate_up"]["alone_gbs"] / HBM_PEAK_GBS,
                                         "what": "hipEventRecord before and after each plain launch: the kernel plus the markers' own cost"},
            "back_to_back": {"avg_us": r["gate_up"]["avg_us"], "frac": r["gate_up"]["gbs"] / HBM_PEAK_GBS,
                             "what": "24 launches back to back in one HIP graph, events around the graph: the launch plus the gap "
                                     "to the dependent launch behind it, i.e. what the launch costs the step"},
            "given_routing": {"avg_us": r["gate_up_given_routing"]["kernel_us"], "frac": r["gate_up_given_routing"]["kernel_gbs"] / HBM_PEAK_GBS,
                              "back_to_back_us": r["gate_up_given_routing"]["avg_us"]},
            "mean_active_experts": sum(active) / len(active),
            "down_kernel": {"achieved": r["down"]["kernel_gbs"], "avg_us": r["down"]["kernel_us"],
                            "frac": r["down"]["kernel_gbs"] / HBM_PEAK_GBS, "back_to_back_us": r["down"]["avg_us"]},
            # the same launch priced against the matrix cores: decode at 4 tokens per expert is far left of the ridge
            "mfma": {"achieved": tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_PEAK_TFLOPS},
        }

    if not a.no_bs1:
        attempt("bs1", bs1_leg)
    single = rank == 0 and world == 1
    if single and not a.no_roofline:
        attempt("roofline", roofline_leg)
    if single and a.prefill_tokens > 0:
        attempt("prefill", lambda: prefill_leg(lib, main_step, device, a.prefill_tokens))
    if single and not a.no_cpu_baseline:
        attempt("cpu_baseline", cpu_baseline)
    if single and a.full_layer_leg and a.prefill_tokens > 0:
        attempt("full_decoder_stack_prompt", lambda: full_layer_prompt_leg(lib, main_step, a.prefill_tokens, a.layers, device))
    if single and not a.no_dequant_leg:
        attempt("dequant", lambda: dequant_leg(lib, main_step, device))
    if single and not a.no_full_layer:
        attempt("full_decoder_stack", lambda: [full_layer_leg(lib, main_step, b, a.ctx, a.layers, device, a.steps, a.warmup)
                                               for b in (a.batch_per_gpu, 1)])
    if single and a.load_leg:
        attempt("load", lambda: load_leg(device))
    if single and a.shape == "20b" and not a.no_rank_of_8:
        attempt("ep_rank_of_8", lambda: rank_of_n_leg(lib, main_step, 8, a.ctx, device))

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1 or a.force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
