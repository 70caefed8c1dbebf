"""CPU oracle for the NF4 MoE decode path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It
wraps oracle/libm355_ref.so (the C restatement, see m355_ref.c for the reference lines each function
follows), holds a numpy twin of the dequantizer for cross-checking the C code, and the synthetic
input generators shared by the golden fixtures and the GPU parity tests.

Pinned against the reference's own Python by tests/golden/ (see tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libm355_ref.so")

FLAG_FP16_COMBINE = 1

# streaming_unsloth_bnb.py:101-106
NF4_QUANT_MAP = np.array(
    [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
     -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
     0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
     0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0], dtype=np.float32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "m355_ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libm355_ref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.m355_ref_last_error.restype = ctypes.c_char_p
        L.m355_ref_threads.restype = ctypes.c_int
        L.m355_ref_f32_to_f16.restype = ctypes.c_uint16
        L.m355_ref_f32_to_f16.argtypes = [ctypes.c_float]
        L.m355_ref_f16_to_f32.restype = ctypes.c_float
        L.m355_ref_f16_to_f32.argtypes = [ctypes.c_uint16]
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.m355_ref_resolve_absmax.argtypes = [vp, vp, i64, vp, vp, i64, vp, vp]
        L.m355_ref_nf4_dequant.argtypes = [vp, i64, vp, i64, vp]
        L.m355_ref_router_topk.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        L.m355_ref_moe_nf4_forward.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp,
                                               i32, i32, i32, i32, i32, i32, vp]
        L.m355_ref_paged_attn_decode.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]
        L.m355_ref_fill_bytes.argtypes = [ctypes.c_uint64, vp, i64]
        L.m355_ref_fill_bytes.restype = None
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().m355_ref_last_error().decode()}")


def _c(a, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    assert a.dtype == dtype, (a.dtype, dtype)
    return a


def threads() -> int:
    return int(lib().m355_ref_threads())


# ---------------------------------------------------------------------------------------------
# C oracle wrappers
# ---------------------------------------------------------------------------------------------
def resolve_absmax(absmax, nested_map=None, nested_absmax=None, mean_override=None) -> np.ndarray:
    """fp32 per-block absmax by the rules of streaming_unsloth_bnb.py:148-173."""
    if absmax is None:
        n = 1
        a8 = a32 = None
    else:
        absmax = np.ascontiguousarray(absmax).reshape(-1)
        n = absmax.size
        a8 = absmax if absmax.dtype == np.uint8 else None
        a32 = absmax.astype(np.float32) if absmax.dtype != np.uint8 else None
    nm = _c(nested_map, np.float32)
    na = _c(nested_absmax, np.float32)
    mo = None if mean_override is None else np.array([mean_override], dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    _check(lib().m355_ref_resolve_absmax(_p(a8), _p(a32), n, _p(nm), _p(na), 0 if na is None else na.size,
                                         _p(mo), _p(out)), "resolve_absmax")
    return out


def nf4_dequant(q, n_params: int, absmax_f32) -> np.ndarray:
    """Packed NF4 bytes -> flat fp16 (as uint16 bit patterns viewed as float16)."""
    q = _c(np.asarray(q).reshape(-1), np.uint8)
    a = _c(np.asarray(absmax_f32).reshape(-1), np.float32)
    assert q.size * 2 == n_params
    out = np.empty(n_params, dtype=np.float16)
    _check(lib().m355_ref_nf4_dequant(_p(q), n_params, _p(a), a.size, _p(out)), "nf4_dequant")
    return out


def router_topk(x, wr, br, k: int):
    x = _c(x, np.float16)
    wr = _c(wr, np.float16)
    br = _c(br, np.float16)
    T, H = x.shape
    E = wr.shape[0]
    ids = np.empty((T, k), dtype=np.int32)
    w = np.empty((T, k), dtype=np.float32)
    logits = np.empty((T, E), dtype=np.float16)
    _check(lib().m355_ref_router_topk(_p(x), _p(wr), _p(br), T, H, E, k, _p(ids), _p(w), _p(logits)), "router_topk")
    return ids, w, logits


def moe_forward(x, ids, w, *, w13_q=None, w13_absmax=None, w2_q=None, w2_absmax=None, w13_f16=None,
                w2_f16=None, E_local: int, e_first: int = 0, H: int, FF: int, blocksize: int = 64,
                flags: int = FLAG_FP16_COMBINE) -> np.ndarray:
    x = _c(x, np.float16)
    ids = _c(ids, np.int32)
    w = _c(w, np.float32)
    T, k = ids.shape
    out = np.empty((T, H), dtype=np.float16)
    _check(lib().m355_ref_moe_nf4_forward(
        _p(x), T, _p(ids), _p(w), k, _p(_c(w13_q, np.uint8)), _p(_c(w13_absmax, np.float32)),
        _p(_c(w2_q, np.uint8)), _p(_c(w2_absmax, np.float32)), _p(_c(w13_f16, np.float16)),
        _p(_c(w2_f16, np.float16)), E_local, e_first, H, FF, blocksize, flags, _p(out)), "moe_forward")
    return out


def paged_attn_decode(q, kcache, vcache, block_tables, ctx_lens, scale: float) -> np.ndarray:
    q = _c(q, np.float16)
    kcache = _c(kcache, np.float16)
    vcache = _c(vcache, np.float16)
    block_tables = _c(block_tables, np.int32)
    ctx_lens = _c(ctx_lens, np.int32)
    B, Hq, D = q.shape
    _, block_size, Hkv, _ = kcache.shape
    out = np.empty_like(q)
    _check(lib().m355_ref_paged_attn_decode(_p(q), _p(kcache), _p(vcache), _p(block_tables), _p(ctx_lens), B,
                                            Hq, Hkv, D, block_size, block_tables.shape[1], scale, _p(out)),
           "paged_attn_decode")
    return out


def paged_attn_decode_ex(q, kcache, vcache, block_tables, ctx_lens, scale: float, sinks=None, window: int = 0) -> np.ndarray:
    """Decode attention with gpt-oss's extras, in fp32 numpy: a sink logit per query head joins the softmax
    denominator and is then dropped (eager_attention_forward, transformers models/gpt_oss/modeling_gpt_oss.py:243-251);
    window > 0 keeps the last `window` cached tokens.  fp16 in / fp16 out like the C function above."""
    q = np.asarray(q, dtype=np.float16)
    B, Hq, D = q.shape
    _, block_size, Hkv, _ = kcache.shape
    G = Hq // Hkv
    out = np.zeros_like(q)
    for b in range(B):
        L = int(ctx_lens[b])
        lo = max(L - window, 0) if window > 0 else 0
        if L <= lo:
            continue
        rows = [int(block_tables[b, t // block_size]) * block_size + t % block_size for t in range(lo, L)]
        K = np.asarray(kcache, dtype=np.float16).reshape(-1, Hkv, D)[rows].astype(np.float32)
        V = np.asarray(vcache, dtype=np.float16).reshape(-1, Hkv, D)[rows].astype(np.float32)
        for h in range(Hq):
            s = (K[:, h // G, :] @ q[b, h].astype(np.float32)) * np.float32(scale)
            logits = s if sinks is None else np.concatenate([s, [np.float32(sinks[h])]])
            p = np.exp(logits - logits.max())
            p = p / p.sum()
            out[b, h] = (p[:len(s)] @ V[:, h // G, :]).astype(np.float16)
    return out


# ---------------------------------------------------------------------------------------------
# numpy twin of the dequantizer (cross-checks the C code; same reference lines)
# ---------------------------------------------------------------------------------------------
def attn_prefill(q, k, v, seq_starts, scale: float, sinks=None, window: int = 0) -> np.ndarray:
    """Causal self-attention of packed prompts in fp32 numpy (what flash-attn's varlen call computes; the reference
    holds no fixture for it): q [tokens][Hq][D], k / v [tokens][Hkv][D]; query i of a sequence sees keys 0..i of it
    (the last `window` of them when window > 0); a sink adds exp(sink) to the denominator."""
    q32, k32, v32 = (np.asarray(a, dtype=np.float16).astype(np.float32) for a in (q, k, v))
    T, Hq, D = q32.shape
    G = Hq // k32.shape[1]
    out = np.zeros((T, Hq, D), np.float32)
    for b in range(len(seq_starts) - 1):
        s0, s1 = int(seq_starts[b]), int(seq_starts[b + 1])
        L = s1 - s0
        if L <= 0:
            continue
        i, j = np.arange(L)[:, None], np.arange(L)[None, :]
        ok = (j <= i) & ((j > i - window) if window > 0 else True)
        for h in range(Hq):
            sc = (q32[s0:s1, h] @ k32[s0:s1, h // G].T) * scale
            sc = np.where(ok, sc, -np.inf)
            m = sc.max(axis=1, keepdims=True)
            if sinks is not None:
                m = np.maximum(m, np.float32(sinks[h]))
            p = np.exp(sc - m)
            den = p.sum(axis=1, keepdims=True)
            if sinks is not None:
                den = den + np.exp(np.float32(sinks[h]) - m)
            out[s0:s1, h] = (p @ v32[s0:s1, h // G]) / den
    return out.astype(np.float16)


def nf4_dequant_numpy(q, n_params: int, absmax_f32) -> np.ndarray:
    q = np.asarray(q, dtype=np.uint8).reshape(-1)
    a = np.asarray(absmax_f32, dtype=np.float32).reshape(-1)
    idx = np.empty(n_params, dtype=np.uint8)
    idx[0::2] = q & 0x0F          # streaming_unsloth_bnb.py:181,186
    idx[1::2] = (q >> 4) & 0x0F   # :182,187
    v = NF4_QUANT_MAP[idx]        # :193
    if a.size > 1 and n_params % a.size == 0:   # :199-202
        bs = n_params // a.size
        v = (v.reshape(-1, bs) * a[:, None]).reshape(-1)
    elif a.size > 1:                             # :204-215
        bs = 64
        for cand in (64, 128):
            if (n_params + cand - 1) // cand <= a.size:
                bs = cand
                break
        nb = min((n_params + bs - 1) // bs, a.size)
        end = min(nb * bs, n_params)
        scale = np.repeat(a[:nb], bs)[:end]
        v = v.copy()
        v[:end] *= scale
    else:                                        # :216-218
        v = v * a[0]
    return v.astype(np.float32).astype(np.float16)   # :222


# ---------------------------------------------------------------------------------------------
# Operators around the hot path (transformers models/gpt_oss/modeling_gpt_oss.py): numpy restatements
# ---------------------------------------------------------------------------------------------
def add_rmsnorm(x, delta, weight, eps: float):
    """(r, y): r = fp16(x + delta) (delta None: x); y = fp16(weight * (float(r) * rsqrt(mean(float(r)^2) + eps)))
    -- GptOssRMSNorm.forward (:57-62) behind the decoder layer's residual add."""
    r = np.asarray(x, dtype=np.float16)
    if delta is not None:
        r = (r + np.asarray(delta, dtype=np.float16)).astype(np.float16)
    rf = r.astype(np.float32)
    var = np.mean(rf * rf, axis=-1, keepdims=True, dtype=np.float32)
    n = rf * (np.float32(1.0) / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return r, (np.asarray(weight, dtype=np.float16).astype(np.float32) * n).astype(np.float16)


def rope_rotate(x, cos, sin):
    """_apply_rotary_emb (:217-225) on fp16 tensors: x [..., D], cos/sin [..., D/2] broadcastable; every product and
    sum is rounded to fp16 as torch's half tensors round."""
    x = np.asarray(x, dtype=np.float16)
    h = x.shape[-1] // 2
    a, b = x[..., :h], x[..., h:]
    c, s = np.asarray(cos, dtype=np.float16), np.asarray(sin, dtype=np.float16)
    m = lambda u, v: (u.astype(np.float32) * v.astype(np.float32)).astype(np.float16)
    first = (m(a, c).astype(np.float32) - m(b, s).astype(np.float32)).astype(np.float16)
    second = (m(b, c).astype(np.float32) + m(a, s).astype(np.float32)).astype(np.float16)
    return np.concatenate([first, second], axis=-1)


def moe_forward_oai(x, ids, w, w13_f16, w2_f16, b13=None, b2=None, alpha: float = 1.702, limit: float = 7.0) -> np.ndarray:
    """gpt-oss's own expert block (GptOssExperts.forward / _apply_gate, transformers
    models/gpt_oss/modeling_gpt_oss.py:79-112) in fp32 numpy: w13_f16 [E][2FF][H] with gate rows at even and up
    rows at odd positions, b13 [E][2FF] interleaved the same way, w2_f16 [E][H][FF], b2 [E][H].  Gate and up are
    rounded to fp16 where the GEMM output would be; returns fp16 [T][H]."""
    x = np.asarray(x, dtype=np.float16).astype(np.float32)
    T, k = ids.shape
    out = np.zeros((T, w2_f16.shape[1]), np.float32)
    for t in range(T):
        for j in range(k):
            e = int(ids[t, j])
            gu = np.asarray(w13_f16[e], dtype=np.float32) @ x[t]
            if b13 is not None:
                gu = gu + np.asarray(b13[e], dtype=np.float32)
            gu = gu.astype(np.float16).astype(np.float32)
            gate, up = np.minimum(gu[0::2], limit), np.clip(gu[1::2], -limit, limit)
            h = ((up + 1.0) * gate / (1.0 + np.exp(-alpha * gate))).astype(np.float16).astype(np.float32)
            y = np.asarray(w2_f16[e], dtype=np.float32) @ h
            if b2 is not None:
                y = y + np.asarray(b2[e], dtype=np.float32)
            out[t] += np.float32(w[t, j]) * y.astype(np.float16).astype(np.float32)
    return out.astype(np.float16)


# ---------------------------------------------------------------------------------------------
# bitsandbytes' own NF4 layout (the "true_bnb" loader mode).  bitsandbytes is not installed here and the
# reference never reads a tensor this way (its loader takes the low nibble first and ignores nested_offset,
# patches/streaming_unsloth_bnb.py:148-190), so this restates the published algorithm of bitsandbytes 0.4x
# `quantize_4bit` / `dequantize_4bit` with compress_statistics=True -- parity unpinned: no fixture of the
# reference covers it; the tests check quantise -> store -> load -> dequantise round trips instead.
#   element 2i   = code[byte >> 4], element 2i+1 = code[byte & 15]       (high nibble first)
#   absmax[b]    = nested_code[absmax_u8[b]] * nested_absmax[b // 256] + nested_offset
#   w            = fp16(code * absmax[block])                              (fp32 product, one rounding)
# ---------------------------------------------------------------------------------------------
BNB_NESTED_BLOCK = 256


def bnb_dynamic_map() -> np.ndarray:
    """A monotone signed 256-entry fp32 codebook in [-1, 1] with a dense centre, shaped like bitsandbytes'
    dynamic map (the loader only ever looks values up in the table a checkpoint carries)."""
    i = np.arange(256, dtype=np.float64)
    v = np.sign(i - 127.5) * (np.abs(i - 127.5) / 127.5) ** 2.0
    v[255] = 1.0
    return np.sort(v).astype(np.float32)


def bnb_resolve_absmax(absmax_u8, nested_map, nested_absmax, nested_offset: float) -> np.ndarray:
    a = np.asarray(nested_map, dtype=np.float32)[np.asarray(absmax_u8, dtype=np.uint8).reshape(-1)]
    scale = np.repeat(np.asarray(nested_absmax, dtype=np.float32).reshape(-1), BNB_NESTED_BLOCK)[:a.size]
    return (a * scale + np.float32(nested_offset)).astype(np.float32)


def bnb_swap_nibbles(q) -> np.ndarray:
    q = np.asarray(q, dtype=np.uint8)
    return ((q >> 4) | (q << 4)).astype(np.uint8)


def bnb_nf4_dequant(q, n_params: int, absmax_f32, blocksize: int = 64) -> np.ndarray:
    """fp16 weights of a high-nibble-first NF4 tensor (bitsandbytes dequantize_4bit)."""
    return nf4_dequant_numpy(bnb_swap_nibbles(np.asarray(q).reshape(-1)), n_params, absmax_f32)


def bnb_quantize_nf4(w_f32, blocksize: int = 64, nested_map=None):
    """Quantise a flat fp32 array as bitsandbytes' quantize_4bit(quant_type='nf4', compress_statistics=True):
    returns (packed u8 [n/2], absmax_u8 [n/blocksize], nested_absmax f32 [ceil(nblk/256)], nested_map, offset)."""
    w = np.asarray(w_f32, dtype=np.float32).reshape(-1, blocksize)
    absmax = np.abs(w).max(axis=1).astype(np.float32)
    norm = w / np.maximum(absmax, np.float32(1e-30))[:, None]
    idx = np.abs(norm[..., None] - NF4_QUANT_MAP[None, None, :]).argmin(axis=-1).astype(np.uint8).reshape(-1)
    packed = ((idx[0::2] << 4) | idx[1::2]).astype(np.uint8)
    nested_map = bnb_dynamic_map() if nested_map is None else np.asarray(nested_map, dtype=np.float32)
    offset = np.float32(absmax.mean())
    centred = absmax - offset
    nblk = centred.size
    pad = (-nblk) % BNB_NESTED_BLOCK
    c2 = np.concatenate([centred, np.zeros(pad, np.float32)]).reshape(-1, BNB_NESTED_BLOCK)
    nested_absmax = np.abs(c2).max(axis=1).astype(np.float32)
    n2 = c2 / np.maximum(nested_absmax, np.float32(1e-30))[:, None]
    a8 = np.abs(n2[..., None] - nested_map[None, None, :]).argmin(axis=-1).astype(np.uint8).reshape(-1)[:nblk]
    return packed, a8, nested_absmax, nested_map, float(offset)


# ---------------------------------------------------------------------------------------------
# Synthetic inputs (self-contained PRNG so the GPU box rebuilds identical bytes)
# ---------------------------------------------------------------------------------------------
def stream_seed(*parts: int) -> int:
    """Mix integers into one 64-bit stream id (splitmix-style)."""
    s = 0x355355355355
    for p in parts:
        s = (s ^ (int(p) + 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        s = (s * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        s ^= s >> 29
    return s


def fill_bytes(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    lib().m355_ref_fill_bytes(ctypes.c_uint64(seed), _p(out), n)
    return out


def uniform01(seed: int, n: int) -> np.ndarray:
    """n doubles in [0,1) from 53 random bits each."""
    raw = fill_bytes(seed, 8 * n).view(np.uint64)
    return (raw >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal_f16(seed: int, shape, std: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(seed, 2 * n)
    r = np.sqrt(-2.0 * np.log(1.0 - u[:n]))
    z = r * np.cos(2.0 * np.pi * u[n:])
    return (z * std).astype(np.float32).astype(np.float16).reshape(shape)


def nested_quant_map(seed: int = 7) -> np.ndarray:
    """A monotone 256-entry fp32 table shaped like bitsandbytes' dynamic map (values in (0, 1])."""
    u = np.sort(uniform01(stream_seed(seed, 1), 256))
    return ((np.arange(256) + 0.25 + 0.5 * u) / 256.0).astype(np.float32)


def nested_absmax_exact(seed: int, n: int) -> np.ndarray:
    """fp32 values k * 2^-14, k in [1, 2048): any-order fp32 sums of <= 8192 of them are exact."""
    kk = fill_bytes(seed, 2 * n).view(np.uint16).astype(np.int64) % 2047 + 1
    return (kk.astype(np.float64) * 2.0 ** -14).astype(np.float32)


def synth_expert_tensor(seed: int, n_params: int, blocksize: int = 64):
    """One double-quantized NF4 tensor in checkpoint form: packed u8 [n/2], absmax u8 [n/bs],
    nested_absmax fp32 [ceil(nblk/256)], nested_quant_map fp32 [256]."""
    nblk = n_params // blocksize
    q = fill_bytes(stream_seed(seed, 11), n_params // 2)
    a8 = fill_bytes(stream_seed(seed, 12), nblk)
    na = nested_absmax_exact(stream_seed(seed, 13), (nblk + 255) // 256)
    nm = nested_quant_map()
    return q, a8, na, nm


def synth_layer(seed: int, E: int, H: int, FF: int, blocksize: int = 64):
    """Packed experts of one layer in checkpoint order plus their resolved fp32 absmax.
    Returns dict with w13_q [E, FF*H] u8, w13_absmax [E, 2*FF*H/bs] f32, w2_q [E, H*FF/2] u8, w2_absmax."""
    n13, n2 = 2 * FF * H, H * FF
    w13_q = np.empty((E, n13 // 2), np.uint8)
    w2_q = np.empty((E, n2 // 2), np.uint8)
    w13_a = np.empty((E, n13 // blocksize), np.float32)
    w2_a = np.empty((E, n2 // blocksize), np.float32)
    for e in range(E):
        q, a8, na, nm = synth_expert_tensor(stream_seed(seed, e, 13), n13, blocksize)
        w13_q[e] = q
        w13_a[e] = resolve_absmax(a8, nm, na)
        q, a8, na, nm = synth_expert_tensor(stream_seed(seed, e, 2), n2, blocksize)
        w2_q[e] = q
        w2_a[e] = resolve_absmax(a8, nm, na)
    return dict(w13_q=w13_q, w13_absmax=w13_a, w2_q=w2_q, w2_absmax=w2_a)


# ---------------------------------------------------------------------------------------------
# Golden-case definitions shared by tests/golden/make_golden.py (which feeds them to the reference's
# own Python) and the tests (which feed them to this oracle and to the HIP library).
# ---------------------------------------------------------------------------------------------
N_GATE_UP = 5760 * 2880   # streaming_unsloth_bnb.py:123-124
N_DOWN = 2880 * 2880      # :127-128


def golden_dequant_case(name: str) -> dict:
    """Inputs of one dequant golden case: dict(proj, q, absmax, nested_absmax, nested_map)."""
    sd = stream_seed(*[ord(c) for c in name])
    proj = "gate_up" if name.startswith("gate_up") else "down"
    n = N_GATE_UP if proj == "gate_up" else N_DOWN
    nblk = n // 64
    q, a8, na, nm = synth_expert_tensor(sd, n)
    case = dict(proj=proj, n_params=n, q=q, absmax=a8, nested_absmax=na, nested_map=nm)
    kind = name.split("_", 2 if proj == "gate_up" else 1)[-1]
    if kind == "nested_mean":
        pass
    elif kind == "nested_mean_u005":
        # nested scales uniform in (0, 0.05): their fp32 sum depends on the order of the additions, so the mean the
        # reference takes (torch's) is pinned by this case, not just any correctly rounded mean
        case["nested_absmax"] = (uniform01(stream_seed(sd, 4), na.size) * 0.05).astype(np.float32)
    elif kind == "nested_scalar":
        case["nested_absmax"] = na[:1].copy()
    elif kind == "nested_elementwise":
        case["nested_absmax"] = nested_absmax_exact(stream_seed(sd, 5), nblk)
    elif kind == "absmax_f32":
        case.update(absmax=(uniform01(stream_seed(sd, 6), nblk) * 0.05 + 1e-4).astype(np.float32),
                    nested_absmax=None, nested_map=None)
    elif kind == "absmax_u8_plain":
        case.update(nested_absmax=None, nested_map=None)
    elif kind == "no_absmax":
        case.update(absmax=None, nested_absmax=None, nested_map=None)
    elif kind == "absmax_single":
        case.update(absmax=np.array([0.03125 * 1.3], np.float32), nested_absmax=None, nested_map=None)
    elif kind == "absmax_bs128":
        case.update(absmax=(uniform01(stream_seed(sd, 7), n // 128) * 0.05 + 1e-4).astype(np.float32),
                    nested_absmax=None, nested_map=None)
    elif kind in ("ragged_plus1", "ragged_bs128", "ragged_short"):
        cnt = {"ragged_plus1": nblk + 1, "ragged_bs128": 100000, "ragged_short": 50001}[kind]
        case.update(absmax=(uniform01(stream_seed(sd, 8), cnt) * 0.05 + 1e-4).astype(np.float32),
                    nested_absmax=None, nested_map=None)
    else:
        raise KeyError(name)
    return case


GOLDEN_DEQUANT_CASES = [
    "gate_up_nested_mean", "down_nested_mean", "down_nested_scalar", "down_nested_elementwise",
    "down_absmax_f32", "down_absmax_u8_plain", "down_no_absmax", "gate_up_absmax_single",
    "down_absmax_bs128", "down_ragged_plus1", "down_ragged_bs128", "down_ragged_short",
    "down_nested_mean_u005", "gate_up_nested_mean_u005",
]


def golden_dequant_oracle(case: dict) -> np.ndarray:
    """Run one golden case through the C oracle; returns flat fp16."""
    a = resolve_absmax(case["absmax"], case["nested_map"], case["nested_absmax"])
    return nf4_dequant(case["q"], case["n_params"], a)


def sample_positions(seed: int, n: int, count: int = 4096) -> np.ndarray:
    return (fill_bytes(stream_seed(seed, 99), 8 * count).view(np.uint64) % np.uint64(n)).astype(np.int64)


GOLDEN_FORWARD_CASES = {
    # name: (H, FF, E, k, T)
    "fwd_full_T1": (2880, 2880, 8, 4, 1),
    "fwd_full_T3": (2880, 2880, 8, 4, 3),
    "fwd_small_T37": (256, 192, 6, 2, 37),
}


def golden_forward_case(name: str) -> dict:
    H, FF, E, k, T = GOLDEN_FORWARD_CASES[name]
    sd = stream_seed(*[ord(c) for c in name])
    layer = synth_layer(sd, E, H, FF)
    x = normal_f16(stream_seed(sd, 21), (T, H), 1.0)
    logits = normal_f16(stream_seed(sd, 22), (T, E), 1.5)
    return dict(H=H, FF=FF, E=E, k=k, T=T, x=x, logits=logits, **layer)


def prompt_attention_case(name):
    """Seeded q, k, v, sinks of the prompt-attention cases ("prompt_full", "prompt_window": 100 tokens, 8 query heads on
    1 KV head, window 0 / 24): what tests/golden/make_layer_golden.py fed to transformers' eager_attention_forward."""
    S, Hq_, Hkv_, D_ = 100, 8, 1, 64
    sd = stream_seed(740, 0 if name == "prompt_full" else 1)
    return dict(S=S, Hq=Hq_, Hkv=Hkv_, D=D_, window=0 if name == "prompt_full" else 24,
                q=normal_f16(stream_seed(sd, 1), (S, Hq_, D_)), k=normal_f16(stream_seed(sd, 2), (S, Hkv_, D_)),
                v=normal_f16(stream_seed(sd, 3), (S, Hkv_, D_)), sinks=normal_f16(stream_seed(sd, 4), (Hq_,), 2.0))


def decoder_layer_case(name):
    """Seeded parameters and inputs of the whole-decoder-layer cases ("sliding", "full"): what
    tests/golden/make_layer_golden.py fed to transformers' GptOssDecoderLayer and what the GPU tests feed to the kernels."""
    H_, FF_, E_, k_, Hq_, Hkv_, D_, L_, window = 256, 192, 4, 2, 8, 1, 64, 200, 128
    sd = stream_seed(730, 0 if name == "sliding" else 1)
    n = lambda i, shape, std: normal_f16(stream_seed(sd, i), shape, std)
    lay = synth_layer(stream_seed(sd, 50), E_, H_, FF_)
    return dict(
        H=H_, FF=FF_, E=E_, k=k_, Hq=Hq_, Hkv=Hkv_, D=D_, L=L_, window=window, lay=lay,
        ln1=(1.0 + 0.1 * n(1, (H_,), 1.0).astype(np.float32)).astype(np.float16),
        ln2=(1.0 + 0.1 * n(2, (H_,), 1.0).astype(np.float32)).astype(np.float16),
        wqkv=n(3, ((Hq_ + 2 * Hkv_) * D_, H_), 0.06), bqkv=n(4, ((Hq_ + 2 * Hkv_) * D_,), 0.1),
        wo=n(5, (H_, Hq_ * D_), 0.04), bo=n(6, (H_,), 0.1), sinks=n(7, (Hq_,), 1.0),
        router_w=n(8, (E_, H_), 0.1), router_b=n(9, (E_,), 0.1),
        b13=n(10, (E_, 2 * FF_), 0.3), b2=n(11, (E_, H_), 0.3), x=n(12, (L_ + 1, H_), 1.0),
        w13_f16=np.stack([nf4_dequant(lay["w13_q"][e], 2 * FF_ * H_, lay["w13_absmax"][e]).reshape(2 * FF_, H_) for e in range(E_)]),
        w2_f16=np.stack([nf4_dequant(lay["w2_q"][e], H_ * FF_, lay["w2_absmax"][e]).reshape(H_, FF_) for e in range(E_)]))
