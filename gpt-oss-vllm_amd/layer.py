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
