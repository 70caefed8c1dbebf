#!/usr/bin/env python3
"""Generate tests/golden/*.json|*.npy by running the reference's own Python on seeded inputs.

Runs ONLY in the authoring container (it imports /root/reference/patches); the committed outputs
are data: for each dequant case the sha256 of the reference's fp16 output plus 4096 sampled
(position, bits) pairs, and for each forward case the reference's full fp16 output.

The reference's modules import bitsandbytes / vllm at module top although the functions used here
(dequant_unsloth_expert_cpu, NF4_QUANT_MAP, UnslothMoEWrapper.forward) never touch them; both
packages are absent from this image, so empty placeholder modules satisfy the import statements.
No reference source text is copied or stored.

Usage: python tests/golden/make_golden.py [--only dequant|forward]
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

REF_DIR = "/root/reference/patches"


def _placeholder_modules():
    bnb = types.ModuleType("bitsandbytes")
    bnb_nn = types.ModuleType("bitsandbytes.nn")
    bnb_fn = types.ModuleType("bitsandbytes.functional")

    class Linear4bit(torch.nn.Module):  # name only; never instantiated here
        pass

    class Params4bit:  # name only
        pass

    class QuantState:  # name only
        pass

    bnb_nn.Linear4bit, bnb_nn.Params4bit, bnb_fn.QuantState = Linear4bit, Params4bit, QuantState
    bnb.nn, bnb.functional = bnb_nn, bnb_fn
    vllm = types.ModuleType("vllm")
    vllm_logger = types.ModuleType("vllm.logger")
    import logging
    vllm_logger.init_logger = logging.getLogger
    vllm.logger = vllm_logger
    for name, mod in [("bitsandbytes", bnb), ("bitsandbytes.nn", bnb_nn), ("bitsandbytes.functional", bnb_fn),
                      ("vllm", vllm), ("vllm.logger", vllm_logger)]:
        sys.modules.setdefault(name, mod)


def _write_case_file(path, case, layer=3, expert=5):
    """Write one tensor in the Unsloth key schema (streaming_unsloth_bnb.py:118-146)."""
    from safetensors.torch import save_file
    proj = "gate_up_projs" if case["proj"] == "gate_up" else "down_projs"
    key = f"model.layers.{layer}.mlp.experts.{proj}.{expert}.weight"
    t = {key: torch.from_numpy(case["q"].copy()).reshape(-1, 1)}
    if case["absmax"] is not None:
        t[key + ".absmax"] = torch.from_numpy(np.ascontiguousarray(case["absmax"]).copy())
    if case["nested_absmax"] is not None:
        t[key + ".nested_absmax"] = torch.from_numpy(case["nested_absmax"].copy())
        t[key + ".nested_quant_map"] = torch.from_numpy(case["nested_map"].copy())
    save_file(t, path)


def make_dequant(out_path):
    import streaming_unsloth_bnb as sub
    res = {"nf4_table_bits": [int(v) for v in sub.NF4_QUANT_MAP.numpy().view(np.uint32)], "cases": {}}
    with tempfile.TemporaryDirectory() as td:
        empty = os.path.join(td, "empty.safetensors")
        from safetensors.torch import save_file
        save_file({"unrelated": torch.zeros(1)}, empty)
        for name in ref.GOLDEN_DEQUANT_CASES:
            case = ref.golden_dequant_case(name)
            f = os.path.join(td, name + ".safetensors")
            _write_case_file(f, case)
            # first file lacks the key: the reference moves on to the next one (:131-135)
            out = sub.dequant_unsloth_expert_cpu([empty, f], 3, 5, case["proj"])
            assert out is not None and out.dtype == torch.float16
            bits = out.numpy().reshape(-1).view(np.uint16)
            pos = ref.sample_positions(ref.stream_seed(*[ord(c) for c in name]), bits.size)
            res["cases"][name] = {
                "shape": list(out.shape),
                "sha256": hashlib.sha256(bits.tobytes()).hexdigest(),
                "sample_pos": [int(p) for p in pos[:256]],
                "sample_bits": [int(b) for b in bits[pos[:256]]],
                "sample_sha256": hashlib.sha256(bits[pos].tobytes()).hexdigest(),
            }
            os.remove(f)
            print(name, res["cases"][name]["sha256"][:16], flush=True)
        # a key that no file holds -> None (:230); a wrong layer index likewise
        res["missing_returns_none"] = sub.dequant_unsloth_expert_cpu([empty], 0, 0, "down") is None
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)


def make_forward(out_dir):
    import unsloth_to_vllm_converter as conv
    meta = {}
    for name in ref.GOLDEN_FORWARD_CASES:
        c = ref.golden_forward_case(name)
        H, FF, E, k, T = c["H"], c["FF"], c["E"], c["k"], c["T"]
        w13, w2 = [], []
        for e in range(E):
            # expert weights = the dequantizer's fp16 output (the oracle's, itself pinned bit-exact to
            # the reference by the dequant cases), carried by plain fp16 nn.Linear modules
            a = torch.from_numpy(ref.nf4_dequant(c["w13_q"][e], 2 * FF * H, c["w13_absmax"][e]).reshape(2 * FF, H))
            b = torch.from_numpy(ref.nf4_dequant(c["w2_q"][e], H * FF, c["w2_absmax"][e]).reshape(H, FF))
            l13 = torch.nn.Linear(H, 2 * FF, bias=False, dtype=torch.float16)
            l2 = torch.nn.Linear(FF, H, bias=False, dtype=torch.float16)
            with torch.no_grad():
                l13.weight.copy_(a)
                l2.weight.copy_(b)
            w13.append(l13)
            w2.append(l2)
        wrapper = conv.UnslothMoEWrapper({"w13": w13, "w2": w2}, num_experts=E)
        x = torch.from_numpy(c["x"]).reshape(1, T, H)
        lg = torch.from_numpy(c["logits"]).reshape(1, T, E)
        with torch.no_grad():
            out = wrapper(x, lg, top_k=k)
            tw, ti = torch.topk(lg, k, dim=-1)
            tw = torch.nn.functional.softmax(tw, dim=-1)
        out = out.reshape(T, H).numpy()
        np.save(os.path.join(out_dir, name + ".out.npy"), out)
        np.save(os.path.join(out_dir, name + ".topk_ids.npy"), ti.reshape(T, k).numpy().astype(np.int32))
        np.save(os.path.join(out_dir, name + ".topk_w.npy"), tw.reshape(T, k).numpy())
        meta[name] = {"H": H, "FF": FF, "E": E, "k": k, "T": T, "absmax_out": float(np.abs(out.astype(np.float32)).max())}
        print(name, meta[name], flush=True)
    with open(os.path.join(out_dir, "forward_golden.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["dequant", "forward"])
    a = ap.parse_args()
    _placeholder_modules()
    sys.path.insert(0, REF_DIR)
    torch.manual_seed(0)
    if a.only in (None, "dequant"):
        make_dequant(os.path.join(HERE, "dequant_golden.json"))
    if a.only in (None, "forward"):
        make_forward(HERE)
