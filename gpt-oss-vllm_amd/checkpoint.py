"""Reader for Unsloth / bitsandbytes NF4 expert tensors stored in safetensors shards.

Key schema (patches/streaming_unsloth_bnb.py:118-146, patches/unsloth_direct_loader.py:68-82 of the reference):
    model.layers.{L}.mlp.experts.{gate_up_projs|down_projs}.{E}.weight            packed uint8 [n/2, 1]
    ...weight.absmax | .nested_absmax | .nested_quant_map | .quant_map            quantisation state
The first shard that holds a key wins (streaming_unsloth_bnb.py:131-135); a key no shard holds yields None
(:230).  Tensors are handed to the plug-in's weight_loader under their checkpoint names, exactly as vLLM's
loader would, so a layer can be filled without vLLM.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import torch

PROJ_KEYS = {"gate_up": "gate_up_projs", "down": "down_projs"}
STATE_SUFFIXES = ("absmax", "nested_absmax", "nested_quant_map", "quant_map", "quant_state.bitsandbytes__nf4")


def discover_shards(cache_dir: Optional[str] = None) -> List[str]:
    """The safetensors shards of an Unsloth gpt-oss checkpoint in the Hugging Face cache, found the way
    patches/patch_11_consolidate_unsloth.py:36-43 finds them: first any `models--*unsloth*gpt-oss*` snapshot,
    then the 20b 4-bit repository by name; sorted, possibly empty."""
    import glob
    import os
    root = os.path.expanduser(cache_dir if cache_dir is not None else "~/.cache/huggingface/hub/")
    files = sorted(glob.glob(os.path.join(root, "models--*unsloth*gpt-oss*/snapshots/*/*.safetensors")))
    if not files:
        files = sorted(glob.glob(os.path.join(root, "models--unsloth--gpt-oss-20b-unsloth-bnb-4bit/snapshots/*/*.safetensors")))
    return files


def expert_weight_key(layer_idx: int, expert_idx: int, proj: str) -> str:
    return f"model.layers.{layer_idx}.mlp.experts.{PROJ_KEYS[proj]}.{expert_idx}.weight"


def read_expert(checkpoint_paths: Iterable[str], layer_idx: int, expert_idx: int, proj: str
                ) -> Optional[Dict[str, torch.Tensor]]:
    """{'weight': uint8, 'absmax': ..., ...} of one expert matrix, or None if no shard has it."""
    from safetensors import safe_open
    key = expert_weight_key(layer_idx, expert_idx, proj)
    for path in checkpoint_paths:
        with safe_open(path, framework="pt", device="cpu") as f:
            keys = set(f.keys())
            if key not in keys:
                continue
            out = {"weight": f.get_tensor(key)}
            for sfx in STATE_SUFFIXES:
                if f"{key}.{sfx}" in keys:
                    out[sfx] = f.get_tensor(f"{key}.{sfx}")
            return out
    return None


ZERO_CODE_BYTE = 0x77   # two NF4 codes 7 = 0.0 (patches/streaming_unsloth_bnb.py:101-106)


def iter_layer_tensors(checkpoint_paths: List[str], layer_idx: int, num_experts: int, missing: str = "raise",
                       dims: Optional[Tuple[int, int]] = None) -> Iterator[Tuple[str, torch.Tensor]]:
    """(checkpoint name, tensor) pairs of every expert tensor of one layer, in expert order.  An expert matrix no shard
    holds raises KeyError (missing="raise").  With missing="zeros" and dims=(H, FF) a missing gate_up is replaced by an
    all-zero matrix (every code the NF4 zero, no absmax), which is what the reference substitutes for it
    (patches/streaming_unsloth_bnb.py:261-264); a missing down still raises: the reference attaches nothing for it
    (:285-288) and has no forward that could say what such an expert computes.  Without biases an expert with a zero
    gate_up contributes nothing (h = silu(0) * 0 = 0); with a down bias b2 it still contributes w * b2."""
    if missing not in ("raise", "zeros"):
        raise ValueError(f"missing={missing!r}: expected 'raise' or 'zeros'")
    for e in range(num_experts):
        for proj in ("gate_up", "down"):
            t = read_expert(checkpoint_paths, layer_idx, e, proj)
            if t is None:
                if missing == "raise" or dims is None or proj != "gate_up":
                    raise KeyError(f"{expert_weight_key(layer_idx, e, proj)} is in none of the shards")
                H, FF = dims
                n = 2 * FF * H if proj == "gate_up" else H * FF
                t = {"weight": torch.full((n // 2, 1), ZERO_CODE_BYTE, dtype=torch.uint8)}
            key = expert_weight_key(layer_idx, e, proj)
            yield key, t["weight"]
            for sfx in STATE_SUFFIXES:
                if sfx in t:
                    yield f"{key}.{sfx}", t[sfx]


def load_layer(method, layer, checkpoint_paths: List[str], layer_idx: int, missing: str = "raise") -> None:
    """Fill one FusedMoE layer (already through method.create_weights) from the shards and convert it."""
    E, H, FF = layer._m355_dims
    for name, tensor in iter_layer_tensors(checkpoint_paths, layer_idx, E, missing=missing, dims=(H, FF)):
        method.weight_loader(layer, None, tensor, name, shard_id=None, expert_id=None)
    method.process_weights_after_loading(layer)
