"""vLLM quantization plug-in surface for the MI355X NF4 MoE path.

Mirrors the interface the reference occupies with patches/bnb_moe_4bit.py (a QuantizationConfig
subclass injected per layer as FusedMoE(quant_config=...), patches/patch_6_add_bnb_moe.py:78-105) and with
the FusedMoE.weight_loader hook of patches/patch_9_fusedmoe_unsloth.py:24-51 -- same method names,
argument meaning and return conventions -- but hands the work to libmoe355.so instead of bitsandbytes.

vLLM is optional at import time: when it is installed the classes derive from its base classes and
`register()` adds the method under the name "mi355_nf4_moe"; when it is not, duck-typed bases let the
same classes be driven directly (tests, bench).  Nothing here falls back to a CPU or eager path: the
forward raises if the HIP library or the GPU is missing.
"""
from __future__ import annotations

import json
from typing import Any, Dict, List, Optional

import torch

from . import lib

try:  # pragma: no cover - exercised only where vLLM is installed
    from vllm.model_executor.layers.quantization.base_config import QuantizationConfig as _QConfigBase
    from vllm.model_executor.layers.fused_moe.layer import FusedMoEMethodBase as _MoEMethodBase
    HAS_VLLM = True
except Exception:  # vLLM absent (or its import chain unusable): same surface, plain bases
    HAS_VLLM = False

    class _QConfigBase:  # type: ignore[no-redef]
        def __init__(self):
            self.packed_modules_mapping: Dict[str, List[str]] = {}

    class _MoEMethodBase:  # type: ignore[no-redef]
        def __init__(self, *args, **kwargs):
            pass

QUANT_NAME = "mi355_nf4_moe"
MAX_TOP_K_PRESIZE = 4   # top-k the configured max_tokens workspace is sized for (gpt-oss routes 4 experts per token)
BLOCKSIZE = 64   # bitsandbytes' default NF4 block; the reference never passes another (streaming_unsloth_bnb.py:36-44)


# ---------------------------------------------------------------------------------------------
# absmax un-double-quantisation, exactly the reference's torch expressions (host, load time)
# ---------------------------------------------------------------------------------------------
def resolve_absmax(absmax: Optional[torch.Tensor], nested_absmax: Optional[torch.Tensor],
                   nested_quant_map: Optional[torch.Tensor]) -> torch.Tensor:
    """fp32 absmax per block by the rules of patches/streaming_unsloth_bnb.py:148-173, computed with the
    same torch CPU operations (so the fp32 mean of nested_absmax is the reference's own value)."""
    if nested_absmax is not None:
        if nested_quant_map is None or absmax is None:
            raise ValueError("nested_absmax needs absmax (uint8) and nested_quant_map")
        na = nested_absmax.detach().cpu().float()
        nm = nested_quant_map.detach().cpu().float()
        a = nm[absmax.detach().cpu().to(torch.uint8).long()]
        if na.numel() == 1:
            return a * na.item()
        if na.shape == a.shape:
            return a * na
        try:
            return a * na
        except RuntimeError:
            return a * na.mean()     # the branch real checkpoints take (:165-171)
    if absmax is None:
        return torch.ones(1)
    return absmax.detach().cpu().float()


def resolve_absmax_table(absmax: Optional[torch.Tensor], nested_absmax: Optional[torch.Tensor],
                         nested_quant_map: Optional[torch.Tensor]):
    """(codes uint8 [nblk], table fp32 [256]) with table[codes] bit for bit resolve_absmax(...), or None when the
    absmax is not one scalar times a 256-entry map (element-wise or broadcast nested scales, plain fp32 absmax).
    Real checkpoints take the scalar branches (one nested scale per 256 blocks -> the mean fallback,
    patches/streaming_unsloth_bnb.py:165-171), so their absmax can stay one byte per block on the GPU."""
    if nested_absmax is None or nested_quant_map is None or absmax is None:
        return None
    na = nested_absmax.detach().cpu().float()
    nm = nested_quant_map.detach().cpu().float().reshape(-1)
    codes = absmax.detach().cpu().to(torch.uint8)
    if nm.numel() != 256:
        return None
    if na.numel() == 1:
        return codes.reshape(-1), nm * na.item()
    if na.shape == codes.shape:
        return None
    try:
        torch.broadcast_shapes(na.shape, codes.shape)
        return None
    except RuntimeError:
        return codes.reshape(-1), nm * na.mean()


def absmax_per_block(absmax: torch.Tensor, n_params: int) -> torch.Tensor:
    """One fp32 scale per 64-weight block that reproduces, bit for bit, what the reference's dequantiser multiplies
    each weight by for ANY count of absmax entries (patches/streaming_unsloth_bnb.py:197-218), so that the kernels'
    fixed 64-weight blocks cover every branch:
      * the count divides n_params: blocks of n_params / count weights (:197-202) -- each entry is repeated over its
        64-weight pieces (blocks of 64, 128, 192, ...);
      * a single entry (scalar absmax, or none -> ones(1), :173,216-218): that value for every block;
      * any other count (:203-215): the block size the reference guesses (64 if the entries cover n_params / 64
        blocks, else 128 if they cover n_params / 128, else 64), the entries it then uses, and 1.0 for the weights
        beyond them, which it leaves unscaled.
    Raises where a block of the reference is no whole number of 64-weight blocks (nothing a checkpoint with the
    kernels' shapes produces)."""
    a = absmax.detach().cpu().float().reshape(-1)
    if n_params % BLOCKSIZE:
        raise ValueError(f"{n_params} weights are no whole number of {BLOCKSIZE}-weight blocks")
    nblk = n_params // BLOCKSIZE
    cnt = a.numel()
    if cnt == 1:
        return a.expand(nblk).contiguous()
    if n_params % cnt == 0:
        bs = n_params // cnt
        if bs % BLOCKSIZE:
            raise ValueError(f"{cnt} absmax entries for {n_params} weights are blocks of {bs}: the kernels need "
                             f"whole multiples of {BLOCKSIZE} (the reference's branch :197-202 at another block size)")
        return a.repeat_interleave(bs // BLOCKSIZE)
    bs = 64
    for cand in (64, 128):                      # :205-209
        if (n_params + cand - 1) // cand <= cnt:
            bs = cand
            break
    used = min((n_params + bs - 1) // bs, cnt)  # :211-215: blocks beyond the entries stay unscaled
    out = torch.ones(nblk, dtype=torch.float32)
    piece = a[:used].repeat_interleave(bs // BLOCKSIZE)[:nblk]
    out[:piece.numel()] = piece
    return out


# ---------------------------------------------------------------------------------------------
# "true_bnb" layout: a tensor read the way bitsandbytes itself reads it, not the way the reference's loader does
# ---------------------------------------------------------------------------------------------
NF4_CODE = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
            -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
            0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
            0.7229568362236023, 1.0)     # the table of patches/streaming_unsloth_bnb.py:101-106 (= bitsandbytes' NF4)
BNB_NESTED_BLOCK = 256
QUANT_STATE_SUFFIX = "quant_state.bitsandbytes__nf4"


def parse_quant_state(blob: Optional[torch.Tensor]) -> Dict[str, Any]:
    """The JSON bitsandbytes packs into `<weight>.quant_state.bitsandbytes__nf4` (a uint8 tensor):
    quant_type, blocksize, dtype, shape, nested_blocksize, nested_offset, ..."""
    if blob is None:
        return {}
    raw = bytes(blob.detach().cpu().to(torch.uint8).reshape(-1).tolist())
    try:
        return json.loads(raw.decode("utf-8"))
    except (UnicodeDecodeError, ValueError) as exc:
        raise ValueError(f"unreadable {QUANT_STATE_SUFFIX}: {exc}") from None


def resolve_absmax_bnb(absmax: torch.Tensor, nested_absmax: Optional[torch.Tensor],
                       nested_quant_map: Optional[torch.Tensor], nested_offset: float,
                       nested_blocksize: int = BNB_NESTED_BLOCK) -> torch.Tensor:
    """bitsandbytes' own un-double-quantisation: absmax[b] = code[absmax_u8[b]] * nested_absmax[b // 256] + offset
    (the reference instead multiplies by nested_absmax.mean() and never adds the offset,
    patches/streaming_unsloth_bnb.py:148-171)."""
    if nested_absmax is None:
        return absmax.detach().cpu().float().reshape(-1)
    if nested_quant_map is None:
        raise ValueError("nested_absmax needs nested_quant_map")
    a = nested_quant_map.detach().cpu().float()[absmax.detach().cpu().to(torch.uint8).long().reshape(-1)]
    scale = nested_absmax.detach().cpu().float().reshape(-1).repeat_interleave(nested_blocksize)[:a.numel()]
    if scale.numel() != a.numel():
        raise ValueError(f"{nested_absmax.numel()} nested scales cannot cover {a.numel()} absmax codes")
    return a * scale + float(nested_offset)


# ---------------------------------------------------------------------------------------------
# QuantizationConfig
# ---------------------------------------------------------------------------------------------
class Mi355Nf4MoEConfig(_QConfigBase):
    """4-bit NF4 for the MoE experts only; router and attention stay fp16 (patches/bnb_moe_4bit.py:22-56)."""

    LAYOUTS = ("reference", "true_bnb")
    GATE_UP_ORDERS = ("concat", "interleaved")
    ABSMAX_STORAGE = ("auto", "fp32")
    PREFIX_FILTERS = ("experts", "reference")
    REFERENCE_PREFIX_PARTS = ("w13", "w2", "gate", "up", "down")   # patches/bnb_moe_4bit.py:46

    def __init__(self, layout: str = "reference", gate_up_order: str = "concat", absmax_storage: str = "auto",
                 max_tokens: int = 0, prefix_filter: str = "experts"):
        """layout "reference": tensors are read as the reference's loader reads them (low nibble first, mean
        fallback for the nested scales, no offset) -- the parity contract.  "true_bnb": as bitsandbytes wrote
        them (high nibble first, one nested scale per 256 blocks, plus nested_offset)."""
        super().__init__()
        if layout not in self.LAYOUTS:
            raise ValueError(f"layout {layout!r}: expected one of {self.LAYOUTS}")
        if gate_up_order not in self.GATE_UP_ORDERS:
            raise ValueError(f"gate_up_order {gate_up_order!r}: expected one of {self.GATE_UP_ORDERS}")
        if absmax_storage not in self.ABSMAX_STORAGE:
            raise ValueError(f"absmax_storage {absmax_storage!r}: expected one of {self.ABSMAX_STORAGE}")
        if prefix_filter not in self.PREFIX_FILTERS:
            raise ValueError(f"prefix_filter {prefix_filter!r}: expected one of {self.PREFIX_FILTERS}")
        self.layout = layout
        # "experts": every layer whose prefix names experts gets the method (a FusedMoE layer's prefix is
        # "...mlp.experts", without a projection name).  "reference": the reference's own rule, which also wants one
        # of w13/w2/gate/up/down in the prefix (patches/bnb_moe_4bit.py:46), for callers that ask per projection.
        self.prefix_filter = prefix_filter
        # tokens per forward the workspace is sized for when the weights are converted (vLLM: max_num_batched_tokens);
        # 0 = grow on demand outside graph capture
        self.max_tokens = int(max_tokens)
        # "auto": a layer whose experts all resolve to (code byte, 256-entry table) keeps the absmax that way on the
        # GPU (1 B instead of 4 B per 64 weights, same results bit for bit); "fp32": always one fp32 per block.
        self.absmax_storage = absmax_storage
        # "concat": gate = rows [0, FF), up = rows [FF, 2FF) of gate_up, as the reference splits it
        # (patches/streaming_unsloth_bnb.py:269-270).  "interleaved": gate = even rows, up = odd rows, as gpt-oss
        # itself stores it (transformers models/gpt_oss/modeling_gpt_oss.py:80); goes with activation="swigluoai".
        self.gate_up_order = gate_up_order
        self.weight_bits = 4
        self.group_size = BLOCKSIZE
        self.compute_dtype = torch.float16
        self.use_nf4 = True

    @classmethod
    def get_name(cls) -> str:
        return QUANT_NAME

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 70   # same figure as the reference (:53-56); gfx950 reports 9.5

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Mi355Nf4MoEConfig":
        config = config or {}
        return cls(layout=config.get("layout", "reference"), gate_up_order=config.get("gate_up_order", "concat"),
                   absmax_storage=config.get("absmax_storage", "auto"), max_tokens=int(config.get("max_tokens", 0)),
                   prefix_filter=config.get("prefix_filter", "experts"))

    def get_scaled_act_names(self) -> List[str]:
        return []

    def get_quant_method(self, layer, prefix: str):
        """The method object for expert layers, None for everything else (router, attention stay fp16).
        The reference answers with the string "bnb_4bit" for prefixes that name experts (:43-48); FusedMoE
        needs an object with create_weights/apply, which is what is returned here."""
        if "experts" not in prefix:
            return None
        if self.prefix_filter == "reference" and not any(part in prefix for part in self.REFERENCE_PREFIX_PARTS):
            return None
        return Mi355Nf4MoEMethod(self)


def is_moe_weight(name: str) -> bool:
    """patches/bnb_moe_4bit.py:99-109 plus the Unsloth per-expert names of patch_9."""
    pats = ("experts.w13_weight", "experts.w2_weight", "experts.w13_bias", "experts.w2_bias", "gate_up_proj",
            "down_proj")
    return any(p in name for p in pats)


def get_moe_weight_names(model) -> List[str]:
    return [n for n, _ in model.named_parameters() if is_moe_weight(n)]


# ---------------------------------------------------------------------------------------------
# FusedMoE method
# ---------------------------------------------------------------------------------------------
_SUFFIXES = (QUANT_STATE_SUFFIX, "nested_quant_map", "nested_absmax", "quant_map", "absmax")


class Mi355Nf4MoEMethod(_MoEMethodBase):
    """create_weights / weight_loader / process_weights_after_loading / apply for a FusedMoE layer."""

    def __init__(self, quant_config: Mi355Nf4MoEConfig, moe=None):
        try:
            super().__init__(moe)
        except TypeError:
            super().__init__()
        self.quant_config = quant_config

    # -- weights -------------------------------------------------------------------------------
    def create_weights(self, layer, num_experts: int, hidden_size: int, intermediate_size_per_partition: int,
                       params_dtype: torch.dtype = torch.float16, **extra_weight_attrs):
        E, H, FF = num_experts, hidden_size, intermediate_size_per_partition
        if H % 64 or FF % 64:
            raise ValueError(f"hidden_size {H} and intermediate_size {FF} must be multiples of 64")
        dev = extra_weight_attrs.pop("device", None)

        def param(shape, dtype):
            p = torch.nn.Parameter(torch.empty(shape, dtype=dtype, device=dev), requires_grad=False)
            # vLLM's model loaders call `param.weight_loader(param, loaded_weight, weight_name, shard_id=..,
            # expert_id=..[, return_success=..])`: FusedMoE.weight_loader bound to its layer, the signature
            # patches/patch_9_fusedmoe_unsloth.py:19 hooks
            p.weight_loader = lambda param, loaded_weight, weight_name=None, shard_id=None, expert_id=None, \
                return_success=False: self.weight_loader(layer, param, loaded_weight, weight_name, shard_id=shard_id,
                                                         expert_id=expert_id, return_success=return_success)
            for key, val in extra_weight_attrs.items():
                if key != "weight_loader":
                    setattr(p, key, val)
            return p

        specs = {"w13_qweight": ((E, FF * H), torch.uint8), "w13_absmax": ((E, 2 * FF * H // BLOCKSIZE), torch.float32),
                 "w2_qweight": ((E, H * FF // 2), torch.uint8), "w2_absmax": ((E, H * FF // BLOCKSIZE), torch.float32)}
        for name, (shape, dtype) in specs.items():
            p = param(shape, dtype)
            if isinstance(layer, torch.nn.Module):
                layer.register_parameter(name, p)
            else:
                setattr(layer, name, p)
        layer._m355_dims = (E, H, FF)
        layer._m355_bias = {}         # "gate_up_proj_bias" | "down_proj_bias" -> fp16 [E][...] as loaded
        layer._m355_raw = {}          # (expert, "gate_up_projs"|"down_projs") -> {"weight"| suffix: tensor}
        layer._m355_ready = False
        layer._m355_ws = None         # the forward's workspace: one buffer, grown to the largest batch seen
        layer._m355_shard = None      # (expert_map identity, e_first) of vLLM expert parallelism, resolved once

    def weight_loader(self, layer, param, loaded_weight: torch.Tensor, weight_name: str, shard_id=None,
                      expert_id=None, return_success: bool = False):
        """Same contract as the hook patches/patch_9_fusedmoe_unsloth.py:24-51 puts into FusedMoE.weight_loader:
        names holding 'gate_up_projs' / 'down_projs' are parsed as
        model.layers.L.mlp.experts.<proj>.E.weight[.suffix]; the packed tensor is kept as loaded and converted
        later; returns `True if return_success else None`."""
        if not isinstance(weight_name, str) or not isinstance(loaded_weight, torch.Tensor):
            raise TypeError("weight_loader(param, loaded_weight, weight_name, shard_id, expert_id): loaded_weight must be "
                            "a tensor and weight_name its checkpoint name")
        for bias_name in ("gate_up_proj_bias", "down_proj_bias"):     # fused [E][...] tensors of the gpt-oss checkpoint
            if weight_name.endswith(bias_name):
                layer._m355_bias[bias_name] = loaded_weight
                layer._m355_ready = False
                return True if return_success else None
        if "gate_up_projs" not in weight_name and "down_projs" not in weight_name:
            raise ValueError(f"not an Unsloth per-expert MoE tensor: {weight_name}")
        parts = weight_name.split(".")
        if len(parts) < 7:
            raise ValueError(f"cannot parse expert index from {weight_name}")
        expert_idx = int(parts[6])       # patch_9:35
        proj_type = parts[5]             # patch_9:36
        suffix = "weight"
        for sfx in _SUFFIXES:
            if weight_name.endswith("." + sfx):
                suffix = sfx
                break
        layer._m355_raw.setdefault((expert_idx, proj_type), {})[suffix] = loaded_weight
        layer._m355_ready = False
        return True if return_success else None

    def process_weights_after_loading(self, layer) -> None:
        """Resolve each expert's absmax (reference rules), move the packed blocks to the GPU unchanged and
        repack them into the kernels' tile order.  Nothing is re-quantised (the reference re-quantises through
        bitsandbytes when it moves an expert to the GPU, streaming_unsloth_bnb.py:48-65)."""
        E, H, FF = layer._m355_dims
        dev = layer.w13_qweight.device
        if dev.type != "cuda":
            raise lib.Mi355LibraryError("expert weights must live on the GPU before process_weights_after_loading")
        # byte absmax needs every expert of both matrices to resolve to (codes, table)
        def as_codes(t):
            ct = resolve_absmax_table(t.get("absmax"), t.get("nested_absmax"), t.get("nested_quant_map"))
            return ct if ct is not None and "weight" in t and ct[0].numel() * (BLOCKSIZE // 2) == t["weight"].numel() else None

        use_codes = (self.quant_config.absmax_storage == "auto" and self.quant_config.layout == "reference"
                     and len(layer._m355_raw) == 2 * E and all(as_codes(t) is not None for t in layer._m355_raw.values()))
        if use_codes:
            layer._m355_codes = {"w13": torch.empty((E, 2 * FF * H // BLOCKSIZE), dtype=torch.uint8, device=dev),
                                 "w2": torch.empty((E, H * FF // BLOCKSIZE), dtype=torch.uint8, device=dev)}
            layer._m355_tables = {"w13": torch.empty((E, 256), dtype=torch.float32, device=dev),
                                  "w2": torch.empty((E, 256), dtype=torch.float32, device=dev)}
        else:
            layer._m355_codes = layer._m355_tables = None
        for (e, proj), t in sorted(layer._m355_raw.items()):
            if not (0 <= e < E) or "weight" not in t:
                raise ValueError(f"incomplete or out-of-range expert tensor ({e}, {proj})")
            rows, cols = (2 * FF, H) if proj == "gate_up_projs" else (H, FF)
            q = t["weight"].reshape(-1).to(torch.uint8)
            if q.numel() * 2 != rows * cols:
                raise ValueError(f"expert {e} {proj}: {q.numel()} packed bytes, expected {rows * cols // 2}")
            if self.quant_config.layout == "true_bnb":
                state = parse_quant_state(t.get(QUANT_STATE_SUFFIX))
                if state.get("quant_type", "nf4") != "nf4" or int(state.get("blocksize", BLOCKSIZE)) != BLOCKSIZE:
                    raise ValueError(f"expert {e} {proj}: quant_state {state} is not NF4 with blocksize {BLOCKSIZE}")
                if "quant_map" in t and not torch.equal(t["quant_map"].detach().cpu().float().reshape(-1),
                                                        torch.tensor(NF4_CODE, dtype=torch.float32)):
                    raise ValueError(f"expert {e} {proj}: quant_map is not the NF4 table the kernels hold")
                if t.get("absmax") is None:
                    raise ValueError(f"expert {e} {proj}: no absmax")
                am = resolve_absmax_bnb(t["absmax"], t.get("nested_absmax"), t.get("nested_quant_map"),
                                        float(state.get("nested_offset", 0.0)),
                                        int(state.get("nested_blocksize", BNB_NESTED_BLOCK))).reshape(-1)
                q = (q >> 4) | (q << 4)      # the kernels take the low nibble first
            else:
                am = resolve_absmax(t.get("absmax"), t.get("nested_absmax"), t.get("nested_quant_map")).reshape(-1)
            if am.numel() * BLOCKSIZE != rows * cols:
                if self.quant_config.layout != "reference":
                    raise ValueError(f"expert {e} {proj}: {am.numel()} absmax entries, expected one per {BLOCKSIZE}")
                am = absmax_per_block(am, rows * cols)     # the reference's other branches (:197-218), same bits
            interleaved = proj == "gate_up_projs" and self.quant_config.gate_up_order == "interleaved"
            qt, at = lib.repack_nf4(q.to(dev).contiguous(), am.float().to(dev).contiguous(), rows, cols,
                                    row_order=1 if interleaved else 0)
            if proj == "gate_up_projs":
                layer.w13_qweight.data[e], layer.w13_absmax.data[e] = qt, at
            else:
                layer.w2_qweight.data[e], layer.w2_absmax.data[e] = qt, at
            if use_codes:
                codes, table = resolve_absmax_table(t.get("absmax"), t.get("nested_absmax"), t.get("nested_quant_map"))
                tag = "w13" if proj == "gate_up_projs" else "w2"
                layer._m355_codes[tag][e] = lib.repack_absmax_u8(codes.to(dev).contiguous(), rows, cols,
                                                                 row_order=1 if interleaved else 0)
                layer._m355_tables[tag][e] = table.to(dev)
        layer._m355_b13 = layer._m355_b2 = None
        b13 = layer._m355_bias.get("gate_up_proj_bias")
        if b13 is not None:
            if tuple(b13.shape) != (E, 2 * FF):
                raise ValueError(f"gate_up_proj_bias is {tuple(b13.shape)}, expected {(E, 2 * FF)}")
            if self.quant_config.gate_up_order == "interleaved":      # the kernels want gate entries, then up entries
                b13 = torch.cat([b13[:, 0::2], b13[:, 1::2]], dim=1)
            layer._m355_b13 = b13.to(device=dev, dtype=torch.float16).contiguous()
        b2 = layer._m355_bias.get("down_proj_bias")
        if b2 is not None:
            if tuple(b2.shape) != (E, H):
                raise ValueError(f"down_proj_bias is {tuple(b2.shape)}, expected {(E, H)}")
            layer._m355_b2 = b2.to(device=dev, dtype=torch.float16).contiguous()
        layer._m355_bias = {}
        layer._m355_raw = {}
        if self.quant_config.max_tokens > 0:
            layer._m355_ws = torch.empty(max(256, lib.moe_workspace_bytes(self.quant_config.max_tokens, MAX_TOP_K_PRESIZE, H, FF, E)),
                                         dtype=torch.uint8, device=dev)
        lib.init(dev.index if dev.index is not None else torch.cuda.current_device())   # kernel attributes, before any graph capture
        layer._m355_ready = True

    # -- forward -------------------------------------------------------------------------------
    def apply(self, layer, x: torch.Tensor, router_logits: torch.Tensor, top_k: int, renormalize: bool = True,
              use_grouped_topk: bool = False, topk_group=None, num_expert_group=None, global_num_experts: int = -1,
              expert_map: Optional[torch.Tensor] = None, custom_routing_function=None, scoring_func: str = "softmax",
              e_score_correction_bias=None, apply_router_weight_on_input: bool = False, activation: str = "silu",
              **kwargs) -> torch.Tensor:
        """FusedMoE forward: top-k + softmax over the selected logits, the routed experts, the weighted sum --
        the semantics of patches/unsloth_to_vllm_converter.py:282-325."""
        if not layer._m355_ready:
            raise RuntimeError("process_weights_after_loading has not run for this layer")
        if not renormalize or use_grouped_topk or custom_routing_function is not None or scoring_func != "softmax" \
                or e_score_correction_bias is not None or apply_router_weight_on_input:
            raise NotImplementedError("only top-k followed by softmax over the selected logits is implemented")
        if activation not in ("silu", "swigluoai"):
            raise NotImplementedError(f"activation {activation!r}: silu (the reference's) or swigluoai (gpt-oss's own)")
        act_mode = lib.ACT_SILU_CONCAT if activation == "silu" else lib.ACT_SWIGLU_OAI
        E, H, FF = layer._m355_dims
        shape = x.shape
        x2 = x.reshape(-1, H)
        in_dtype = x2.dtype
        if in_dtype != torch.float16:
            x2 = x2.to(torch.float16)
        lg = router_logits.reshape(-1, router_logits.shape[-1])
        if lg.dtype != torch.float16:
            lg = lg.to(torch.float16)
        e_first = self._shard_first_expert(layer, expert_map)
        T = x2.shape[0]
        flags = lib.FLAG_FP16_COMBINE
        if expert_map is not None and global_num_experts > 0:     # a shard: see lib.kernel_choice_flags
            flags |= lib.kernel_choice_flags(T, top_k, global_num_experts)
        ws = self._workspace(layer, T, top_k, x2.device)
        a13, a2 = layer.w13_absmax.data, layer.w2_absmax.data
        if getattr(layer, "_m355_codes", None) is not None:
            a13 = (layer._m355_codes["w13"], layer._m355_tables["w13"])
            a2 = (layer._m355_codes["w2"], layer._m355_tables["w2"])
        # top-k + softmax over the selected logits and the experts in one library call (a decode batch picks the top-k
        # inside the first GEMM launch: no launch of its own for it)
        out, _, _ = lib.moe_nf4_forward_logits(x2.contiguous(), lg.contiguous(), top_k, layer.w13_qweight.data, a13,
                                               layer.w2_qweight.data, a2, E_local=E, e_first=e_first, H=H, FF=FF,
                                               workspace=ws, act_mode=act_mode, flags=flags,
                                               b13=getattr(layer, "_m355_b13", None), b2=getattr(layer, "_m355_b2", None))
        if in_dtype != torch.float16:
            out = out.to(in_dtype)
        return out.reshape(shape)


    @staticmethod
    def _shard_first_expert(layer, expert_map: Optional[torch.Tensor]) -> int:
        """First global expert id of this rank's shard under vLLM expert parallelism (expert_map[global id] = local id
        or -1, a contiguous shard).  Read back from the GPU once per map (the first eager forward, or
        `set_expert_map`), never on the hot path and never under graph capture."""
        if expert_map is None:
            return 0
        key = (id(expert_map), expert_map.data_ptr(), expert_map.numel())
        cached = layer._m355_shard
        if cached is not None and cached[0] == key:
            return cached[1]
        if expert_map.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("expert_map seen for the first time during graph capture: run one eager forward first, "
                               "or call method.set_expert_map(layer, expert_map) after loading")
        owned = torch.nonzero(expert_map >= 0).flatten()
        e_first = int(owned[0].item()) if owned.numel() else 0
        E = layer._m355_dims[0]
        if owned.numel() and (owned.numel() != E or int(owned[-1].item()) != e_first + E - 1):
            raise ValueError(f"expert_map must map one contiguous range of {E} experts to this rank")
        layer._m355_shard = (key, e_first, expert_map)    # (the map is kept alive so its identity cannot be reused)
        return e_first

    def set_expert_map(self, layer, expert_map: Optional[torch.Tensor]) -> int:
        """Resolve the shard of `expert_map` ahead of the first forward (one device-to-host read, at load time)."""
        layer._m355_shard = None
        return self._shard_first_expert(layer, expert_map)

    @staticmethod
    def _workspace(layer, T: int, top_k: int, device) -> torch.Tensor:
        """One workspace per layer, sized for the largest batch seen so far (or for the configured max_tokens): a smaller
        batch carves its lists from the front of the same buffer, so continuous batching does not reallocate.  Under
        graph capture a too-small buffer is not replaced (the capture's pool would own the new one): the call takes a
        temporary from that pool instead."""
        E, H, FF = layer._m355_dims
        need = max(256, lib.moe_workspace_bytes(T, top_k, H, FF, E))
        ws = layer._m355_ws
        if ws is not None and ws.numel() >= need and ws.device == device:
            return ws
        fresh = torch.empty(need, dtype=torch.uint8, device=device)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            layer._m355_ws = fresh
        return fresh


def cast_router_to_fp16(mlp_block) -> None:
    """What patches/monkey_patch_router_fp16.py:6-17 does after MLPBlock.__init__: the router nn.Linear runs in
    fp16 so its logits match the fp16 expert path."""
    router = getattr(mlp_block, "router", None)
    if isinstance(router, torch.nn.Linear) and router.weight.dtype != torch.float16:
        with torch.no_grad():
            router.to(dtype=torch.float16)


def register() -> bool:
    """Register the config with vLLM under QUANT_NAME (a `vllm.general_plugins` entry point may call this).
    Returns False when vLLM is not installed; site-packages is never edited."""
    if not HAS_VLLM:
        return False
    from vllm.model_executor.layers.quantization import register_quantization_config  # pragma: no cover
    register_quantization_config(QUANT_NAME)(Mi355Nf4MoEConfig)  # pragma: no cover
    return True  # pragma: no cover
