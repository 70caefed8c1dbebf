"""Host side of the operators around the hot path in a gpt-oss decoder layer (RMSNorm, rotary tables, KV write).

The reference leaves these to upstream vLLM; their arithmetic is taken from transformers'
models/gpt_oss/modeling_gpt_oss.py (GptOssRMSNorm :47-62, GptOssRotaryEmbedding.forward :187-200,
_apply_rotary_emb :217-225) and modeling_rope_utils._compute_yarn_parameters.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

# gpt-oss-20b (transformers GptOssConfig defaults)
GPT_OSS_ROPE = dict(rope_theta=150000.0, factor=32.0, beta_fast=32.0, beta_slow=1.0,
                    original_max_position_embeddings=4096, truncate=False)
RMS_EPS = 1e-5


def yarn_inv_freq(head_dim: int, rope_theta: float, factor: float, beta_fast: float, beta_slow: float,
                  original_max_position_embeddings: int, truncate: bool = True) -> Tuple[torch.Tensor, float]:
    """(inv_freq fp32 [head_dim/2], attention scaling) of YaRN, computed as _compute_yarn_parameters does: the
    same torch fp32 expressions in the same order, so the values are that function's bit for bit."""
    dim = head_dim

    def correction_dim(num_rotations):
        return (dim * math.log(original_max_position_embeddings / (num_rotations * 2 * math.pi))) / (2 * math.log(rope_theta))

    low, high = correction_dim(beta_fast), correction_dim(beta_slow)
    if truncate:
        low, high = math.floor(low), math.ceil(high)
    low, high = max(low, 0), min(high, dim - 1)
    if low == high:
        high += 0.001
    pos_freqs = rope_theta ** (torch.arange(0, dim, 2).to(dtype=torch.float) / dim)
    extrapolation = 1.0 / pos_freqs
    interpolation = 1.0 / (factor * pos_freqs)
    ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (high - low), 0, 1)
    extrapolation_factor = 1 - ramp
    inv_freq = interpolation * (1 - extrapolation_factor) + extrapolation * extrapolation_factor
    scaling = 1.0 if factor <= 1 else 0.1 * math.log(factor) + 1.0
    return inv_freq, scaling


def rope_tables(max_pos: int, head_dim: int = 64, dtype: torch.dtype = torch.float16, **rope) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos, sin [max_pos][head_dim/2] in `dtype`, the values GptOssRotaryEmbedding.forward returns for positions
    0..max_pos-1: fp32 outer product, cos/sin, times the attention scaling, then the cast."""
    params = dict(GPT_OSS_ROPE)
    params.update(rope)
    inv_freq, scaling = yarn_inv_freq(head_dim, **params)
    freqs = (inv_freq[:, None].float() @ torch.arange(max_pos)[None, :].float()).transpose(0, 1)
    return (freqs.cos() * scaling).to(dtype).contiguous(), (freqs.sin() * scaling).to(dtype).contiguous()


def validate_paging(block_tables: torch.Tensor, slots: torch.Tensor, ctx_lens: torch.Tensor, kcache: torch.Tensor) -> None:
    """Host-side check (synchronises) that the paging metadata stays inside the cache: the kernels trust it, and an
    out-of-range block id is an out-of-bounds access on the GPU.  Meant to run once per configuration, before the
    step is launched or captured."""
    nblocks, block_size = kcache.shape[0], kcache.shape[1]
    bt = block_tables.detach().cpu()
    used = (ctx_lens.detach().cpu().to(torch.int64) + block_size - 1) // block_size
    if bt.shape[0] != used.numel():
        raise ValueError(f"block_tables has {bt.shape[0]} rows for {used.numel()} sequences")
    if int(used.max()) > bt.shape[1]:
        raise ValueError(f"a context of {int(ctx_lens.max())} tokens needs {int(used.max())} blocks, block_tables holds {bt.shape[1]}")
    for b in range(bt.shape[0]):
        row = bt[b, :int(used[b])]
        if row.numel() and (int(row.min()) < 0 or int(row.max()) >= nblocks):
            raise ValueError(f"sequence {b}: block id outside [0, {nblocks})")
    sl = slots.detach().cpu()
    if int(sl.max()) >= nblocks * block_size:
        raise ValueError(f"slot {int(sl.max())} outside the cache of {nblocks * block_size} rows")


# ---------------------------------------------------------------------------------------------
# One gpt-oss decoder layer at decode time, as a fixed sequence of library calls on preallocated buffers
# ---------------------------------------------------------------------------------------------
class PendingCombine:
    """The expert output of a layer whose forward skipped the combine step: the experts' rows and the routing still sit
    in that layer's workspace.  The next layer consumes it in its first kernel; materialize() runs the plain combine."""

    def __init__(self, layer):
        self.layer, self.workspace, self.w = layer, layer.moe_ws, layer.wts
        self.FF, self.E, self.flags = layer.FF, layer.E, 0

    def materialize(self):
        lay, lib, w = self.layer, self.layer.lib, self.layer.w
        skip = lib.FLAG_SKIP_ROUTE | lib.FLAG_SKIP_GATE_UP | lib.FLAG_SKIP_DOWN
        lib.moe_nf4_forward(lay.h2, lay.ids, lay.wts, w["w13_q"], w["w13_absmax"], w["w2_q"], w["w2_absmax"], E_local=lay.E,
                            e_first=0, H=lay.H, FF=lay.FF, workspace=lay.moe_ws, out=lay.moe_out, act_mode=lay.act_mode,
                            flags=self.flags | skip, b13=w.get("b13"), b2=w.get("b2"))
        return lay.moe_out


class GptOssDecodeLayer:
    """GptOssDecoderLayer.forward (transformers models/gpt_oss/modeling_gpt_oss.py:347-379) for one new token per
    sequence: RMSNorm -> fused q/k/v projection -> rotary + KV-cache write -> paged attention (sinks, optional sliding
    window) -> o projection + residual -> RMSNorm -> router top-k -> NF4 experts.  The second residual add is left to
    the next layer's first RMSNorm call (or `finish`), which fuses it.

    weights: dict with ln1, ln2 [H]; wqkv [(Hq + 2 Hkv) D][H], bqkv; wo [H][Hq D], bo; sinks [Hq] or None;
    router_w [E][H], router_b [E] or None; w13_q, w13_absmax, w2_q, w2_absmax (tile order), b13 / b2 or None -- fp16
    unless NF4.  Every intermediate lives in buffers sized for max_tokens, so a step allocates nothing and can be
    captured in a HIP graph.
    """

    def __init__(self, weights, *, hidden: int, heads: int, kv_heads: int, head_dim: int, ffn: int, experts: int,
                 top_k: int, max_tokens: int, eps: float = RMS_EPS, window: int = 0, act_mode: int = 1,
                 device=None):
        from . import lib
        self.lib, self.w = lib, weights
        self.H, self.Hq, self.Hkv, self.D, self.FF, self.E, self.k = hidden, heads, kv_heads, head_dim, ffn, experts, top_k
        self.eps, self.window, self.act_mode, self.T = eps, window, act_mode, max_tokens
        dev = device if device is not None else weights["wqkv"].device
        f16 = dict(dtype=torch.float16, device=dev)
        T = max_tokens
        self.h1 = torch.empty((T, hidden), **f16)
        self.resid = torch.empty((T, hidden), **f16)
        self.qkv = torch.empty((T, (heads + 2 * kv_heads) * head_dim), **f16)
        self.q = torch.empty((T, heads, head_dim), **f16)
        self.attn = torch.empty((T, heads, head_dim), **f16)
        self.resid2 = torch.empty((T, hidden), **f16)
        self.h2 = torch.empty((T, hidden), **f16)
        self.ids = torch.empty((T, top_k), dtype=torch.int32, device=dev)
        self.wts = torch.empty((T, top_k), dtype=torch.float32, device=dev)
        self.moe_out = torch.empty((T, hidden), **f16)
        self.moe_ws = torch.empty(max(256, lib.moe_workspace_bytes(T, top_k, hidden, ffn, experts)), dtype=torch.uint8, device=dev)
        self.attn_ws = None

    def write_kv_only(self, x, positions, slots, kcache, vcache, cos, sin):
        """RMSNorm + q/k/v projection + rotary + cache write for tokens whose attention output is not needed
        (filling the cache with a prompt through the decode kernels)."""
        lib, w = self.lib, self.w
        h = lib.add_rmsnorm(x, None, w["ln1"], self.eps)
        qkv = lib.linear_f16(h, w["wqkv"], w.get("bqkv"))
        lib.rope_kv_write(qkv, cos, sin, positions, slots, kcache, vcache, Hq=self.Hq, Hkv=self.Hkv, D=self.D)

    def prefill(self, resid_in, delta, positions, slots, seq_starts, max_seq: int, kcache, vcache, cos, sin):
        """The same layer over whole prompts: resid_in [tokens][H] holds the sequences one after another, seq_starts
        int32 [B + 1] their first rows, positions the rotary positions, slots the cache rows of the tokens (negative:
        not cached).  Attention is causal within each sequence (m355_attn_prefill) over the rotated keys of this
        call, so the sequences must start at position 0 of their context.  Buffers are allocated per call (prompt
        sizes vary).  Returns (residual stream after attention, expert output), as forward does."""
        lib, w = self.lib, self.w
        T, dev = resid_in.shape[0], resid_in.device
        resid = torch.empty_like(resid_in)
        h1 = lib.add_rmsnorm(resid_in, delta, w["ln1"], self.eps, resid_out=resid)
        qkv = lib.linear_f16(h1, w["wqkv"], w.get("bqkv"))
        # rotated keys and the values of the prompt, token-major, for the attention; then the same rows into the cache
        k_rows = torch.empty((T, 1, self.Hkv, self.D), dtype=torch.float16, device=dev)
        v_rows = torch.empty_like(k_rows)
        q = lib.rope_kv_write(qkv, cos, sin, positions, torch.arange(T, dtype=torch.int64, device=dev), k_rows, v_rows,
                              Hq=self.Hq, Hkv=self.Hkv, D=self.D)
        lib.rope_kv_write(qkv, cos, sin, positions, slots, kcache, vcache, Hq=self.Hq, Hkv=self.Hkv, D=self.D, q_out=q)
        attn = lib.attn_prefill(q, k_rows.view(T, self.Hkv, self.D), v_rows.view(T, self.Hkv, self.D), seq_starts, max_seq,
                                scale=self.D ** -0.5, sinks=w.get("sinks"), window=self.window)
        resid2 = lib.linear_f16(attn.view(T, self.Hq * self.D), w["wo"], w.get("bo"), resid=resid)
        h2 = lib.add_rmsnorm(resid2, None, w["ln2"], self.eps)
        moe_out, _, _ = lib.moe_nf4_forward_routed(h2, w["router_w"], w.get("router_b"), self.k, w["w13_q"], w["w13_absmax"],
                                                   w["w2_q"], w["w2_absmax"], E_local=self.E, e_first=0, H=self.H, FF=self.FF,
                                                   act_mode=self.act_mode, flags=0, b13=w.get("b13"), b2=w.get("b2"))
        return resid2, moe_out

    def forward(self, resid_in, delta, positions, slots, block_tables, ctx_lens, kcache, vcache, cos, sin, max_ctx: int,
                defer_combine: bool = False):
        """resid_in [T][H]: the residual stream entering the layer, delta: the previous layer's expert output still to
        be added (None for the first layer; a PendingCombine when that layer ran with defer_combine).  ctx_lens counts
        the new token.  Returns (residual stream after attention, expert output): their sum is the layer's output.
        defer_combine: the expert output is returned as a PendingCombine (the experts' rows still unsummed in this
        layer's workspace); the next layer's first kernel then forms it together with the residual add and its
        RMSNorm, one launch less per layer, same bits."""
        lib, w = self.lib, self.w
        T = resid_in.shape[0]
        if T != self.T:
            raise ValueError(f"built for {self.T} tokens per step, got {T}")
        # residual add (or the previous layer's combine), input norm, q/k/v projection, rotary embedding and cache write:
        # one launch up to 4 tokens, the norm's launch and the projection's one or two beyond (same bits)
        pending = (delta.workspace, delta.w, delta.FF, delta.E, delta.flags) if isinstance(delta, PendingCombine) else delta
        lib.norm_qkv_rope_kv_write(resid_in, pending, w["ln1"], self.eps, w["wqkv"], w.get("bqkv"), cos, sin, positions, slots,
                                   kcache, vcache, Hq=self.Hq, Hkv=self.Hkv, D=self.D, resid_out=self.resid, q_out=self.q,
                                   h_scratch=self.h1, qkv_scratch=self.qkv)
        need = lib.paged_attn_workspace_bytes(T, self.Hq, self.D, max_ctx)
        if need and (self.attn_ws is None or self.attn_ws.numel() < need):
            self.attn_ws = torch.empty(need, dtype=torch.uint8, device=self.q.device)
        lib.paged_attn_decode(self.q, kcache, vcache, block_tables, ctx_lens, max_ctx=max_ctx, scale=self.D ** -0.5,
                              workspace=self.attn_ws, out=self.attn, sinks=w.get("sinks"), window=self.window)
        lib.linear_f16(self.attn.view(T, self.Hq * self.D), w["wo"], w.get("bo"), resid=self.resid, out=self.resid2)
        # the router rides in the launch of the norm in front of the MoE block (one workgroup per token has the normalised
        # row at hand).  Measured against the norm alone followed by m355_moe_nf4_forward_routed (logits as a launch of
        # their own over every CU, top-k inside the gate_up launch): the whole stack runs 1.387 against 1.391 ms per step
        # at bs=1 (one launch more there), and 2.795 against 2.851 at bs=32 where the router is a launch of its own
        lib.add_rmsnorm_router(self.resid2, None, w["ln2"], self.eps, w["router_w"], w.get("router_b"), self.k, out=self.h2,
                               ids=self.ids, w=self.wts)
        lib.moe_nf4_forward(self.h2, self.ids, self.wts, w["w13_q"], w["w13_absmax"], w["w2_q"], w["w2_absmax"],
                            E_local=self.E, e_first=0, H=self.H, FF=self.FF, workspace=self.moe_ws, out=self.moe_out,
                            act_mode=self.act_mode, flags=lib.FLAG_SKIP_COMBINE if defer_combine else 0, b13=w.get("b13"),
                            b2=w.get("b2"))
        if defer_combine:
            return self.resid2, PendingCombine(self)
        return self.resid2, self.moe_out
