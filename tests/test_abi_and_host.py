"""CPU: the C-ABI library loads and exports what include/m355.h declares; host-side plug-in logic."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(m355):
    header = open(os.path.join(ROOT, "include", "m355.h")).read()
    declared = set(re.findall(r"M355_API\s+[\w\s\*]+?\b(m355_\w+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(m355.EXPORTS)
    L = m355.load()
    for sym in declared:
        assert hasattr(L, sym), sym
    assert m355.version() >= 100
    assert m355.ep_row_bytes(2880, 4) == 5792
    assert m355.moe_workspace_bytes(32, 4, 2880, 2880, 32) > 32 * 4 * 2880 * 2 * 2


def test_no_cpu_fallback(m355):
    q = torch.zeros(64, dtype=torch.uint8)
    a = torch.ones(2, dtype=torch.float32)
    with pytest.raises(m355.Mi355LibraryError, match="GPU tensor"):
        m355.nf4_dequant(q, a)
    from gpt_oss_vllm_amd import plugin
    cfg = plugin.Mi355Nf4MoEConfig()
    m = cfg.get_quant_method(None, "model.layers.0.mlp.experts")
    layer = torch.nn.Module()
    m.create_weights(layer, 2, 64, 64, torch.float16)
    with pytest.raises(m355.Mi355LibraryError):
        m.process_weights_after_loading(layer)     # CPU layer: refuse instead of computing on the host


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    import gpt_oss_vllm_amd  # noqa: F401
    from gpt_oss_vllm_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.Mi355LibraryError, match="missing"):
        lib.load()


def test_config_surface_mirrors_reference():
    from gpt_oss_vllm_amd import plugin
    cfg = plugin.Mi355Nf4MoEConfig.from_config({})
    assert cfg.get_name() == "mi355_nf4_moe"
    assert cfg.weight_bits == 4 and cfg.use_nf4 and cfg.compute_dtype == torch.float16
    assert cfg.get_supported_act_dtypes() == [torch.float16, torch.bfloat16]
    assert cfg.get_min_capability() == 70
    assert cfg.get_config_filenames() == [] and cfg.get_scaled_act_names() == []
    assert cfg.get_quant_method(None, "model.layers.3.attn.qkv") is None          # attention stays fp16
    assert cfg.get_quant_method(None, "model.layers.3.mlp.router") is None        # router stays fp16
    assert isinstance(cfg.get_quant_method(None, "model.layers.3.mlp.experts"), plugin.Mi355Nf4MoEMethod)
    # the reference's own prefix rule (patches/bnb_moe_4bit.py:46) as an option: experts AND a projection name
    strict = plugin.Mi355Nf4MoEConfig.from_config({"prefix_filter": "reference"})
    assert strict.get_quant_method(None, "model.layers.3.mlp.experts") is None
    assert strict.get_quant_method(None, "model.layers.3.attn.qkv") is None
    for tail in ("w13_weight", "w2_weight", "gate_up_proj", "down_proj"):
        assert isinstance(strict.get_quant_method(None, "model.layers.3.mlp.experts." + tail), plugin.Mi355Nf4MoEMethod)
    with pytest.raises(ValueError):
        plugin.Mi355Nf4MoEConfig(prefix_filter="all")
    assert plugin.is_moe_weight("model.layers.0.mlp.experts.w13_weight")
    assert plugin.is_moe_weight("model.layers.0.mlp.experts.gate_up_projs.3.weight")
    assert not plugin.is_moe_weight("model.layers.0.attn.qkv.weight")


def test_weight_loader_hook_contract():
    from gpt_oss_vllm_amd import plugin
    m = plugin.Mi355Nf4MoEMethod(plugin.Mi355Nf4MoEConfig())
    layer = torch.nn.Module()
    m.create_weights(layer, 4, 64, 64, torch.float16)
    assert tuple(layer.w13_qweight.shape) == (4, 64 * 64) and layer.w13_qweight.dtype == torch.uint8
    assert tuple(layer.w2_absmax.shape) == (4, 64)
    t = torch.zeros(8, dtype=torch.uint8)
    name = "model.layers.7.mlp.experts.gate_up_projs.3.weight"
    assert m.weight_loader(layer, None, t, name, None, None, return_success=True) is True
    assert m.weight_loader(layer, None, t, name + ".absmax", None, None) is None
    # the loader attached to a parameter is called the way vLLM's model loaders call FusedMoE.weight_loader:
    # param.weight_loader(param, loaded_weight, weight_name, shard_id=.., expert_id=..[, return_success=..])  (patch_9:19)
    p13 = layer.w13_qweight
    assert p13.weight_loader(p13, t, "model.layers.7.mlp.experts.down_projs.1.weight.nested_absmax", shard_id="w2",
                             expert_id=1, return_success=True) is True
    assert p13.weight_loader(p13, t, "model.layers.7.mlp.experts.down_projs.1.weight", "w2", 1) is None
    assert layer._m355_raw[(1, "down_projs")]["weight"] is t
    with pytest.raises(TypeError):
        p13.weight_loader(t, "model.layers.7.mlp.experts.down_projs.1.weight", None, None)     # param left out
    assert set(layer._m355_raw) == {(3, "gate_up_projs"), (1, "down_projs")}
    assert set(layer._m355_raw[(3, "gate_up_projs")]) == {"weight", "absmax"}
    assert "nested_absmax" in layer._m355_raw[(1, "down_projs")]
    with pytest.raises(ValueError):
        m.weight_loader(layer, None, t, "model.layers.7.mlp.experts.w13_weight", None, None)
    with pytest.raises(RuntimeError):
        m.apply(layer, torch.zeros(1, 64), torch.zeros(1, 4), top_k=2)


@pytest.mark.parametrize("name", ["down_nested_mean", "down_nested_scalar", "down_nested_elementwise",
                                  "down_absmax_f32", "down_absmax_u8_plain", "down_no_absmax"])
def test_host_absmax_resolution_equals_oracle(name, ref):
    from gpt_oss_vllm_amd import plugin
    c = ref.golden_dequant_case(name)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    got = plugin.resolve_absmax(t(c["absmax"]), t(c["nested_absmax"]), t(c["nested_map"])).numpy().astype(np.float32)
    want = ref.resolve_absmax(c["absmax"], c["nested_map"], c["nested_absmax"])
    assert np.array_equal(got.view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))
    # code byte + 256-entry table: only for the scalar / mean branches, and then the same bits entry by entry
    ct = plugin.resolve_absmax_table(t(c["absmax"]), t(c["nested_absmax"]), t(c["nested_map"]))
    assert (ct is not None) == (name in ("down_nested_mean", "down_nested_scalar"))
    if ct is not None:
        codes, table = ct
        assert codes.dtype == torch.uint8 and tuple(table.shape) == (256,)
        assert np.array_equal(table[codes.long()].numpy().view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


@pytest.mark.parametrize("name", ["down_nested_mean", "down_nested_scalar", "down_nested_elementwise", "down_absmax_f32",
                                  "down_absmax_u8_plain", "down_no_absmax", "gate_up_absmax_single", "down_absmax_bs128",
                                  "down_ragged_plus1", "down_ragged_bs128", "down_ragged_short", "down_nested_mean_u005",
                                  "gate_up_nested_mean_u005"])
def test_host_per_block_absmax_covers_every_reference_branch(name, ref):
    """The product keeps one fp32 scale per 64 weights.  plugin.resolve_absmax + plugin.absmax_per_block must yield, for
    every branch of the reference's dequantiser (other block sizes, a single absmax, none, counts that do not divide the
    weight count: patches/streaming_unsloth_bnb.py:148-218), scales with which a plain 64-block dequantisation gives the
    reference's output bit for bit (sha256 of the committed golden)."""
    import hashlib
    import json
    from gpt_oss_vllm_amd import plugin
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "dequant_golden.json")))["cases"][name]
    c = ref.golden_dequant_case(name)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    am = plugin.resolve_absmax(t(c["absmax"]), t(c["nested_absmax"]), t(c["nested_map"])).reshape(-1)
    n = c["n_params"]
    per64 = am if am.numel() * 64 == n else plugin.absmax_per_block(am, n)
    assert per64.numel() * 64 == n and per64.dtype == torch.float32
    out = ref.nf4_dequant_numpy(c["q"], n, per64.numpy()).view(np.uint16)       # divisible branch only
    assert hashlib.sha256(out.tobytes()).hexdigest() == g["sha256"]


def test_per_block_absmax_rejects_unrepresentable_blocks():
    from gpt_oss_vllm_amd import plugin
    with pytest.raises(ValueError, match="blocks of 32"):
        plugin.absmax_per_block(torch.ones(8), 256)
    with pytest.raises(ValueError):
        plugin.absmax_per_block(torch.ones(3), 100)


def test_checkpoint_missing_expert_policy(tmp_path):
    """A matrix no shard holds: KeyError by default; with missing="zeros" an all-zero matrix for a missing gate_up, as the
    reference substitutes (patches/streaming_unsloth_bnb.py:261-264); a missing down raises either way (the reference
    attaches nothing for it, :285-288)."""
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint
    H, FF = 64, 64
    kg0, kd0, kd1 = (checkpoint.expert_weight_key(0, e, p) for e, p in ((0, "gate_up"), (0, "down"), (1, "down")))
    shard = str(tmp_path / "s.safetensors")
    save_file({kg0: torch.zeros((2 * FF * H // 2, 1), dtype=torch.uint8), kg0 + ".absmax": torch.ones(2 * FF * H // 64),
               kd1: torch.zeros((H * FF // 2, 1), dtype=torch.uint8), kd1 + ".absmax": torch.ones(H * FF // 64)}, shard)
    with pytest.raises(KeyError):
        list(checkpoint.iter_layer_tensors([shard], 0, 1))
    with pytest.raises(KeyError, match="down_projs.0"):       # expert 0 has no down matrix: zeros do not stand in for it
        list(checkpoint.iter_layer_tensors([shard], 0, 2, missing="zeros", dims=(H, FF)))
    save_file({kd0: torch.zeros((H * FF // 2, 1), dtype=torch.uint8), kd0 + ".absmax": torch.ones(H * FF // 64)},
              str(tmp_path / "t.safetensors"))
    got = dict(checkpoint.iter_layer_tensors([shard, str(tmp_path / "t.safetensors")], 0, 2, missing="zeros", dims=(H, FF)))
    kg1 = checkpoint.expert_weight_key(0, 1, "gate_up")       # expert 1 has no gate_up matrix
    assert got[kg1].numel() == FF * H and (got[kg1] == 0x77).all() and kg1 + ".absmax" not in got
    assert got[kd1].numel() == H * FF // 2
    with pytest.raises(ValueError):
        list(checkpoint.iter_layer_tensors([shard], 0, 1, missing="skip"))


def test_checkpoint_reader_schema(tmp_path, ref):
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint
    k = checkpoint.expert_weight_key(3, 5, "down")
    assert k == "model.layers.3.mlp.experts.down_projs.5.weight"
    a, b = str(tmp_path / "a.safetensors"), str(tmp_path / "b.safetensors")
    save_file({"unrelated": torch.zeros(1)}, a)
    save_file({k: torch.arange(16, dtype=torch.uint8).reshape(-1, 1), k + ".absmax": torch.ones(4, dtype=torch.uint8),
               k + ".nested_absmax": torch.ones(1), k + ".nested_quant_map": torch.linspace(0, 1, 256)}, b)
    t = checkpoint.read_expert([a, b], 3, 5, "down")           # first shard lacks the key -> second is used
    assert set(t) == {"weight", "absmax", "nested_absmax", "nested_quant_map"}
    assert checkpoint.read_expert([a], 3, 5, "down") is None    # nowhere -> None, as the reference returns
    assert checkpoint.read_expert([a, b], 3, 5, "gate_up") is None
    with pytest.raises(KeyError):
        list(checkpoint.iter_layer_tensors([a, b], 3, 1))


def test_router_cast_helper():
    from gpt_oss_vllm_amd import plugin

    class Block:
        pass
    blk = Block()
    blk.router = torch.nn.Linear(8, 4).to(torch.bfloat16)
    plugin.cast_router_to_fp16(blk)
    assert blk.router.weight.dtype == torch.float16 and blk.router.bias.dtype == torch.float16


def test_kernel_choice_for_expert_shards():
    """A shard sees every rank's token slots but only its share of the pairs: the choice follows the global count."""
    from gpt_oss_vllm_amd import lib
    assert lib.kernel_choice_flags(256, 4, 32) == lib.FLAG_FORCE_STREAM | lib.FLAG_STREAM_WIDE   # 8 ranks x 32 tokens: 32 pairs per expert, token tiles of 32
    assert lib.kernel_choice_flags(64, 4, 32) == lib.FLAG_FORCE_STREAM       # 2 ranks x 32 tokens: 8 pairs per expert, token tiles of 16
    assert lib.kernel_choice_flags(8192, 4, 32) == lib.FLAG_FORCE_TILE
    assert lib.kernel_choice_flags(8192, 4, 128) == lib.FLAG_FORCE_TILE
    assert lib.kernel_choice_flags(16, 32, 4) == lib.FLAG_FORCE_STREAM        # a token tile of <= 16 never takes the tile kernel
