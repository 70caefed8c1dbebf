#!/usr/bin/env python3
"""Condense a rocprofv3 run of bench.py (gpurun_out/prof_rNN/) into profiles/rNN_*.  Usage: summarize.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = f"gpurun_out/prof_{tag}"
# (label, directory, what the run was)
LEGS = [
    ("hot path bs=32", "hot32", "`bench.py --steps 10 --warmup 3 --legs roofline`: the headline step (attention, router, gate_up, down, combine) "
                                "and the roofline leg's launches of the same gate_up / down kernels on the same inputs"),
    ("hot path bs=1", "hot1", "the same with `--batch-per-gpu 1` (router_logits in front, gate_up picks the top-k, the down launch combines)"),
    ("prefill 8192 tokens", "prefill", "`--legs prefill`: prompt attention, tiled router, routing, tile kernel, combine (+ a few bs=32 headline steps)"),
    ("whole decoder stack", "full", "`--legs full`: 24 whole decoder layers at bs=32 and bs=1 (+ a few bs=32 headline steps)"),
    ("dequant / rank of 8", "side", "`--legs dequant,rank8`: the standalone dequantiser and the kernels of one rank of an 8-way shard (+ a few bs=32 headline steps)"),
]
for _, d, _ in LEGS:
    for f in glob.glob(f"{src}/trace_{d}/*/*kernel_stats.csv"):
        shutil.copy(f, f"profiles/{tag}_{d}_kernel_stats.csv")
    if os.path.exists(f"{src}/bench_line_{d}.json"):
        shutil.copy(f"{src}/bench_line_{d}.json", f"profiles/{tag}_bench_line_{d}.json")

import re


def short_name(mangled):
    """_ZN4m355<len><name>[I<template args>E]... -> name[<args>] with the MoE GEMM kernels' arguments spelled out."""
    m = re.match(r"_ZN4m355(\d+)", mangled)
    if not m:
        return mangled.split("(")[0].replace("m355::", "")
    ln = int(m.group(1))
    name = mangled[m.end():m.end() + ln]
    rest = mangled[m.end() + ln:]
    args = []
    if rest.startswith("I"):
        args = re.findall(r"Li(\d+)E", rest.split("EvPK")[0].split("EvP")[0])
    name = name.replace("_kernel", "")
    stage = {"0": "gate_up", "1": "down"}
    if name == "moe_gemm_stream" and len(args) >= 3:
        return f"moe_gemm_stream<{stage.get(args[0], args[0])},pairs={args[1]},route={('listed', 'inline', 'topk+inline')[int(args[2])]}>"
    if name in ("moe_tile_gemm", "moe_gemm") and args:
        return f"{name}<{stage.get(args[0], args[0])}>"
    return name + (f"<{','.join(args)}>" if args else "")


groups = collections.defaultdict(list)
for leg, d, _ in LEGS:
    for tr_file in glob.glob(f"{src}/trace_{d}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(tr_file)):
            n = r["Kernel_Name"]
            if "m355" not in n:
                continue
            key = (leg, short_name(n), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]),
                   int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
            groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

tile_pmc = collections.defaultdict(list)
mfma_busy = {}
for f in glob.glob(f"{src}/pmc_tile_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "moe_tile" in r["Kernel_Name"]:
            tile_pmc[("gate_up" if "ILi0E" in r["Kernel_Name"] else "down", r["Counter_Name"])].append(float(r["Counter_Value"]))

pmc = collections.defaultdict(list)
for f in glob.glob(f"{src}/pmc_rd/*/*counter_collection.csv") + glob.glob(f"{src}/pmc_fetch/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "moe_gemm" in r["Kernel_Name"]:
            pmc[("gate_up" if "ILi0E" in r["Kernel_Name"] else "down", r["Counter_Name"])].append(float(r["Counter_Value"]))

lines = {d: json.load(open(f"{src}/bench_line_{d}.json")) for _, d, _ in LEGS if os.path.exists(f"{src}/bench_line_{d}.json")}
with open(f"profiles/{tag}_summary.md", "w") as out:
    out.write(f"# {tag}: rocprofv3 --kernel-trace --stats of bench.py, one run per leg\n\n")
    for leg, d, what in LEGS:
        rows = sorted((k, v) for k, v in groups.items() if k[0] == leg)
        if not rows:
            continue
        out.write(f"## {leg}\n\n{what}\n\n")
        out.write("| kernel | workgroups | threads | launches | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for k, v in rows:
            out.write(f"| {k[1]} | {k[2]} | {k[3]} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} |\n")
        plain_file = f"{src}/bench_line_plain{d[3:]}.json" if d in ("hot32", "hot1") else None
        if plain_file and os.path.exists(plain_file):
            line = json.load(open(plain_file))
            shutil.copy(plain_file, f"profiles/{tag}_bench_line_plain{d[3:]}.json")
            r = line.get("roofline")
            gu = [v for k, v in rows if k[1].startswith("moe_gemm_stream<gate_up") and "listed" not in k[1]]
            if isinstance(r, dict) and "avg_us" in r and gu:
                v = max(gu, key=len)
                avg = sum(v) / len(v) / 1e3
                b2b = r.get("back_to_back", {}).get("avg_us")
                out.write(f"\nbench.py's own figure for the gate_up launch as the step runs it, from a run of the same command without the "
                          f"profiler (`roofline.avg_us`: the kernel's own duration from start / stop events carried by each launch): "
                          f"{r['avg_us']:.2f} us, frac {r['frac']:.3f}; the trace row above: {avg:.2f} us over {len(v)} "
                          f"launches ({(r['avg_us'] / avg - 1) * 100:+.1f} %)."
                          + (f"  24 launches back to back in one graph (launch + gap to the next): {b2b:.2f} us." if b2b else "")
                          + f"  ms_per_step of that run: {line['ms_per_step']:.3f}.\n")
        out.write("\n")
    for _, d, _ in LEGS:
        if d in lines:
            out.write(f"## bench.py line of the `{d}` run\n\n```\n" + json.dumps(lines[d], indent=1) + "\n```\n\n")
    if pmc:
        out.write("\n## HBM traffic of the MoE GEMM kernels (separate --pmc passes over scratch/prof_moe.py, bs=32, per launch)\n\n")
        out.write("| kernel | counter | launches | mean |\n|---|---|---|---|\n")
        for k, v in sorted(pmc.items()):
            out.write(f"| {k[0]} | {k[1]} | {len(v)} | {sum(v) / len(v):.6g} |\n")
        out.write("\nTCC_EA0_RDREQ counts 128-byte requests here (no 32-byte ones); FETCH_SIZE is in KiB and reads one half of a "
                  "wide coalesced stream on gfx950, so bytes = 2 x FETCH_SIZE x 1024 = RDREQ x 128.\n")
if tile_pmc:
    with open(f"profiles/{tag}_summary.md", "a") as out:
        out.write("\n## Matrix-core counters of the tile kernel (separate --pmc passes over scratch/prof_tile.py: 8192 tokens, 32 experts, per launch)\n\n")
        out.write("| kernel | counter | launches | mean |\n|---|---|---|---|\n")
        for k, v in sorted(tile_pmc.items()):
            out.write(f"| {k[0]} | {k[1]} | {len(v)} | {sum(v) / len(v):.6g} |\n")
        for kern in ("gate_up", "down"):
            busy, act = tile_pmc.get((kern, "SQ_VALU_MFMA_BUSY_CYCLES")), tile_pmc.get((kern, "GRBM_GUI_ACTIVE"))
            if busy and act:
                cyc = sum(act) / len(act) / 8          # the counter sums the 8 XCDs
                mfma_busy[kern] = sum(busy) / len(busy) / (1024 * cyc)
                out.write(f"\n{kern}: matrix cores busy {sum(busy) / len(busy) / (1024 * cyc):.3f} of the cycles of the launch "
                          f"(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 = {cyc:.4g} cycles)).\n")
import hashlib
import os


def blob(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


stamped = ("gpt-oss-vllm_amd/csrc/moe.hip", "gpt-oss-vllm_amd/csrc/moe_tile.hip", "gpt-oss-vllm_amd/csrc/moe_shared.h")
# the sources the counters were taken on travel with the run (profile_round.sh copies their hashes); fall back to the tree
stamp_file = f"{src}/source_blobs.json"
source_blobs = json.load(open(stamp_file)) if os.path.exists(stamp_file) else {p: blob(p) for p in stamped}
rd = pmc.get(("gate_up", "TCC_EA0_RDREQ_sum"))
rd_down = pmc.get(("down", "TCC_EA0_RDREQ_sum"))
if rd:
    with open(f"profiles/{tag}_traffic.json", "w") as out:
        json.dump({"gate_up_hbm_bytes_per_launch": sum(rd) / len(rd) * 128,
                   "down_hbm_bytes_per_launch": (sum(rd_down) / len(rd_down) * 128) if rd_down else None,
                   "source_blobs": source_blobs,
                   "source": "TCC_EA0_RDREQ_sum x 128 B, rocprofv3 --pmc pass over scratch/prof_moe.py (bs=32)",
                   "tile_gate_up_mfma_busy_frac": mfma_busy.get("gate_up"), "tile_down_mfma_busy_frac": mfma_busy.get("down"),
                   "mfma_busy_source": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), rocprofv3 --pmc passes "
                                       "over scratch/prof_tile.py (8192 tokens, 32 experts)"}, out, indent=1)
print(open(f"profiles/{tag}_summary.md").read()[:3000])
