"""The checkpoint reader's "true_bnb" layout (tensors read as bitsandbytes wrote them) and shard discovery.

bitsandbytes is not installed and the reference never reads a tensor this way, so there is no golden vector:
parity unpinned.  The checks are round trips through a numpy restatement of bitsandbytes' published NF4
quantiser (oracle/ref.py) and agreement between the torch host code and that restatement.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref as oracle_ref


def _bnb_expert(seed, n):
    w = (oracle_ref.normal_f16(seed, (n,)).astype(np.float32) * 0.02)
    q, a8, na, nm, off = oracle_ref.bnb_quantize_nf4(w)
    state = {"quant_type": "nf4", "blocksize": 64, "dtype": "float16", "shape": [n],
             "nested_blocksize": 256, "nested_dtype": "float32", "nested_offset": off}
    blob = torch.tensor(list(json.dumps(state).encode()), dtype=torch.uint8)
    return w, q, a8, na, nm, off, blob


def test_quantise_dequantise_round_trip():
    n = 64 * 700
    w, q, a8, na, nm, off, _ = _bnb_expert(5, n)
    am = oracle_ref.bnb_resolve_absmax(a8, nm, na, off)
    d = oracle_ref.bnb_nf4_dequant(q, n, am).astype(np.float32).reshape(-1)
    true_am = np.abs(w.reshape(-1, 64)).max(axis=1)
    assert np.abs(am - true_am).max() <= 0.02 * true_am.max()          # 8-bit nested code
    assert (np.abs(d - w).reshape(-1, 64).max(axis=1) <= 0.17 * true_am + 1e-6).all()   # half the widest NF4 gap + absmax error
    # the same bytes read the reference's way (low nibble first) give a different tensor: the modes are not aliases
    other = oracle_ref.nf4_dequant_numpy(q, n, am).astype(np.float32).reshape(-1)
    assert np.abs(other - w).max() > 0.3 * true_am.max()


def test_host_resolution_matches_restatement():
    from gpt_oss_vllm_amd import plugin
    n = 64 * 600      # 600 blocks: three nested blocks, the last one partial
    _, q, a8, na, nm, off, blob = _bnb_expert(6, n)
    got = plugin.resolve_absmax_bnb(torch.from_numpy(a8), torch.from_numpy(na), torch.from_numpy(nm), off)
    want = oracle_ref.bnb_resolve_absmax(a8, nm, na, off)
    assert np.array_equal(got.numpy().view(np.uint32), want.view(np.uint32))
    st = plugin.parse_quant_state(blob)
    assert st["blocksize"] == 64 and st["nested_offset"] == pytest.approx(off)
    assert plugin.parse_quant_state(None) == {}
    with pytest.raises(ValueError):
        plugin.parse_quant_state(torch.tensor([255, 254], dtype=torch.uint8))
    with pytest.raises(ValueError):
        plugin.resolve_absmax_bnb(torch.from_numpy(a8), torch.from_numpy(na[:1]), torch.from_numpy(nm), off)
    with pytest.raises(ValueError):
        plugin.Mi355Nf4MoEConfig(layout="something else")
    assert plugin.Mi355Nf4MoEConfig.from_config({"layout": "true_bnb"}).layout == "true_bnb"
    assert plugin.Mi355Nf4MoEConfig.from_config({}).layout == "reference"


def test_discover_shards(tmp_path):
    from gpt_oss_vllm_amd import checkpoint
    assert checkpoint.discover_shards(str(tmp_path)) == []
    snap = tmp_path / "models--unsloth--gpt-oss-20b-unsloth-bnb-4bit" / "snapshots" / "abc"
    snap.mkdir(parents=True)
    for name in ("model-00002-of-00003.safetensors", "model-00001-of-00003.safetensors", "config.json"):
        (snap / name).write_bytes(b"")
    found = checkpoint.discover_shards(str(tmp_path))
    assert [os.path.basename(f) for f in found] == ["model-00001-of-00003.safetensors", "model-00002-of-00003.safetensors"]


@pytest.mark.gpu
def test_true_bnb_layer_end_to_end(tmp_path):
    """bitsandbytes-layout shard -> plug-in (layout true_bnb) -> GPU forward, against the oracle forward on the
    weights the numpy restatement dequantises from the same bytes."""
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint, plugin
    cuda = torch.device("cuda:0")
    H, FF, E, k, T = 256, 192, 4, 2, 7
    tensors, w13, w2 = {}, [], []
    for e in range(E):
        for proj, n, dst, shape in (("gate_up", 2 * FF * H, w13, (2 * FF, H)), ("down", H * FF, w2, (H, FF))):
            _, q, a8, na, nm, off, blob = _bnb_expert(oracle_ref.stream_seed(31, e, n), n)
            key = checkpoint.expert_weight_key(0, e, proj)
            tensors[key] = torch.from_numpy(q.copy()).reshape(-1, 1)
            tensors[key + ".absmax"] = torch.from_numpy(a8.copy())
            tensors[key + ".nested_absmax"] = torch.from_numpy(na.copy())
            tensors[key + ".nested_quant_map"] = torch.from_numpy(nm.copy())
            tensors[key + ".quant_map"] = torch.tensor(plugin.NF4_CODE, dtype=torch.float32)
            tensors[key + ".quant_state.bitsandbytes__nf4"] = blob
            am = oracle_ref.bnb_resolve_absmax(a8, nm, na, off)
            dst.append(oracle_ref.bnb_nf4_dequant(q, n, am).reshape(shape))
    shard = str(tmp_path / "model-00001-of-00001.safetensors")
    save_file(tensors, shard)
    method = plugin.Mi355Nf4MoEConfig(layout="true_bnb").get_quant_method(None, "model.layers.0.mlp.experts")
    layer = torch.nn.Module()
    method.create_weights(layer, E, H, FF, torch.float16, device=cuda)
    checkpoint.load_layer(method, layer, [shard], 0)
    x = oracle_ref.normal_f16(oracle_ref.stream_seed(31, 100), (T, H))
    lg = oracle_ref.normal_f16(oracle_ref.stream_seed(31, 101), (T, E), 1.5)
    out = method.apply(layer, torch.from_numpy(x).to(cuda), torch.from_numpy(lg).to(cuda), top_k=k, renormalize=True)
    ids, w, _ = oracle_ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    want = oracle_ref.moe_forward(x, ids, w, w13_f16=np.stack(w13), w2_f16=np.stack(w2), E_local=E, H=H, FF=FF)
    got = out.cpu().numpy().astype(np.float32)
    assert np.abs(got - want.astype(np.float32)).max() <= 2e-3 * np.abs(want.astype(np.float32)).max() + 1e-5
    # a codebook other than NF4 is refused rather than silently mis-read
    tensors[checkpoint.expert_weight_key(0, 0, "down") + ".quant_map"] = torch.linspace(-1, 1, 16)
    save_file(tensors, shard)
    layer2 = torch.nn.Module()
    method.create_weights(layer2, E, H, FF, torch.float16, device=cuda)
    with pytest.raises(ValueError, match="quant_map"):
        checkpoint.load_layer(method, layer2, [shard], 0)
