#!/usr/bin/env python3
"""Time the reference's own CPU path next to the oracle port, in the authoring container only (it imports
/root/reference/patches like make_golden.py does): one MoE layer, one token, 4 routed experts =
dequant_unsloth_expert_cpu for gate_up and down of each expert, then the fp16 matmuls of
UnslothMoEWrapper.forward's inner loop.  Confirms that the port timed by bench.py's cpu_baseline leg on the GPU
box (where the reference cannot travel) is representative of the reference itself.

Usage: python tests/golden/time_reference.py
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import ref  # noqa: E402

H = FF = 2880


def main():
    mg._placeholder_modules()
    sys.path.insert(0, mg.REF_DIR)
    import streaming_unsloth_bnb as sub
    ref.build()
    from safetensors.torch import save_file
    nm = ref.nested_quant_map()
    tensors, layer = {}, dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
    for e in range(4):
        for proj, n, tag in (("gate_up_projs", 2 * FF * H, "w13"), ("down_projs", H * FF, "w2")):
            q, a8, na, _ = ref.synth_expert_tensor(ref.stream_seed(9001, e, n), n)
            key = f"model.layers.0.mlp.experts.{proj}.{e}.weight"
            tensors[key] = torch.from_numpy(q.copy()).reshape(-1, 1)
            tensors[key + ".absmax"] = torch.from_numpy(a8.copy())
            tensors[key + ".nested_absmax"] = torch.from_numpy(na.copy())
            tensors[key + ".nested_quant_map"] = torch.from_numpy(nm.copy())
            layer[tag + "_q"].append(q)
            layer[tag + "_absmax"].append(ref.resolve_absmax(a8, nm, na))
    layer = {k: np.stack(v) for k, v in layer.items()}
    x = ref.normal_f16(ref.stream_seed(9002), (1, H))
    xt = torch.from_numpy(x)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "layer0.safetensors")
        save_file(tensors, f)

        def reference_layer():
            out = torch.zeros(H, dtype=torch.float16)
            for e in range(4):
                w13 = sub.dequant_unsloth_expert_cpu([f], 0, e, "gate_up")
                w2 = sub.dequant_unsloth_expert_cpu([f], 0, e, "down")
                gu = torch.nn.functional.linear(xt, w13)
                gate, up = gu.chunk(2, dim=-1)
                out += torch.tensor(0.25, dtype=torch.float16) * torch.nn.functional.linear(torch.nn.functional.silu(gate) * up, w2)[0]
            return out
        ids = np.arange(4, dtype=np.int32).reshape(1, 4)
        w = np.full((1, 4), 0.25, dtype=np.float32)

        def port_layer():
            return ref.moe_forward(x, ids, w, **layer, E_local=4, H=H, FF=FF)

        res = {}
        for name, fn in (("reference", reference_layer), ("port", port_layer)):
            fn()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                o = fn()
                ts.append(time.perf_counter() - t0)
            res[name] = (sorted(ts)[2], np.asarray(o, dtype=np.float32).reshape(-1))
        d = np.abs(res["reference"][1] - res["port"][1]).max()
        print(f"threads: torch {torch.get_num_threads()}, port {ref.threads()}")
        print(f"reference: {res['reference'][0] * 1e3:.1f} ms per layer and token   port: {res['port'][0] * 1e3:.1f} ms   "
              f"max |out difference| {d:.3g} (max |out| {np.abs(res['reference'][1]).max():.3g})")


if __name__ == "__main__":
    main()
