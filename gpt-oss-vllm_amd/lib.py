"""ctypes binding of libmoe355.so (the C ABI declared in include/m355.h) for torch tensors.

torch is used only for device memory and the current HIP stream.  There is no CPU or eager
fallback: if the shared library is missing, or a tensor is not on a GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M355_LIB_PATH") or os.path.join(_HERE, "libmoe355.so")

ACT_SILU_CONCAT = 0
ACT_SWIGLU_OAI = 1
FLAG_FP16_COMBINE = 1
FLAG_SKIP_ROUTE, FLAG_SKIP_GATE_UP, FLAG_SKIP_DOWN, FLAG_SKIP_COMBINE = 0x100, 0x200, 0x400, 0x800
FLAG_FORCE_STREAM, FLAG_FORCE_TILE, FLAG_STREAM_WIDE, FLAG_SEPARATE_COMBINE = 0x1000, 0x2000, 0x4000, 0x8000
FLAG_FOLD_COMBINE = 0x10000
FLAG_ROUTER_LAUNCH, FLAG_LOGITS_READY = 0x20000, 0x40000

# every symbol include/m355.h declares
EXPORTS = (
    "m355_version", "m355_last_error", "m355_nf4_dequant", "m355_repack_nf4", "m355_router_topk",
    "m355_topk_softmax", "m355_moe_nf4_forward", "m355_moe_workspace_bytes", "m355_paged_attn_decode",
    "m355_paged_attn_workspace_bytes", "m355_ep_row_bytes", "m355_ep_pack", "m355_ep_unpack", "m355_ep_reduce",
    "m355_add_rmsnorm", "m355_add_rmsnorm_router", "m355_rope_kv_write", "m355_linear_f16", "m355_paged_attn_decode_ex",
    "m355_repack_nf4_ex", "m355_moe_nf4_forward_ex", "m355_moe_nf4_forward_q8", "m355_repack_absmax_u8",
    "m355_moe_nf4_forward_routed", "m355_attn_prefill", "m355_moe_combine_add_rmsnorm", "m355_init",
    "m355_moe_nf4_forward_static", "m355_router_topk_ep_pack", "m355_moe_nf4_forward_ep_rows", "m355_ep_reduce_add_rmsnorm",
    "m355_qkv_rope_kv_write", "m355_moe_nf4_forward_logits", "m355_norm_qkv_rope_kv_write",
    "m355_time_next_launch", "m355_timed_launch_us",
)

_lib: Optional[ctypes.CDLL] = None


class Mi355LibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libmoe355.so (built by `make -C gpt-oss-vllm_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Mi355LibraryError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}`; "
            "this package has no fallback path")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    L.m355_version.restype = i32
    L.m355_init.argtypes = [i32]
    L.m355_init.restype = i32
    L.m355_last_error.restype = ctypes.c_char_p
    L.m355_nf4_dequant.argtypes = [vp, vp, i64, i32, vp, vp]
    L.m355_repack_nf4.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.m355_repack_nf4_ex.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.m355_repack_nf4_ex.restype = i32
    L.m355_moe_nf4_forward_ex.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                          vp, vp, f32, f32, vp, sz, vp, vp]
    L.m355_moe_nf4_forward_ex.restype = i32
    L.m355_moe_nf4_forward_static.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                              vp, vp, f32, f32, vp, vp]
    L.m355_moe_nf4_forward_static.restype = i32
    L.m355_moe_nf4_forward_q8.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                          vp, vp, f32, f32, vp, sz, vp, vp]
    L.m355_moe_nf4_forward_q8.restype = i32
    L.m355_moe_nf4_forward_routed.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                              i32, i32, i32, i32, i32, i32, vp, vp, f32, f32, vp, sz, vp, vp]
    L.m355_moe_nf4_forward_routed.restype = i32
    L.m355_moe_nf4_forward_logits.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                              i32, i32, i32, i32, i32, i32, vp, vp, f32, f32, vp, sz, vp, vp]
    L.m355_moe_nf4_forward_logits.restype = i32
    L.m355_attn_prefill.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                    f32, vp, i32, vp, vp]
    L.m355_attn_prefill.restype = i32
    L.m355_repack_absmax_u8.argtypes = [vp, i32, i32, i32, vp, vp]
    L.m355_repack_absmax_u8.restype = i32
    L.m355_router_topk.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    L.m355_topk_softmax.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.m355_moe_nf4_forward.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                       vp, sz, vp, vp]
    L.m355_moe_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.m355_moe_workspace_bytes.restype = sz
    L.m355_paged_attn_decode.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, vp]
    L.m355_paged_attn_decode_ex.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, sz, vp, vp]
    L.m355_paged_attn_decode_ex.restype = i32
    L.m355_paged_attn_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.m355_paged_attn_workspace_bytes.restype = sz
    L.m355_ep_row_bytes.argtypes = [i32, i32]
    L.m355_ep_row_bytes.restype = sz
    L.m355_router_topk_ep_pack.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp]
    L.m355_router_topk_ep_pack.restype = i32
    L.m355_moe_nf4_forward_ep_rows.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
                                               vp, vp, f32, f32, vp, sz, vp, vp]
    L.m355_moe_nf4_forward_ep_rows.restype = i32
    L.m355_ep_reduce_add_rmsnorm.argtypes = [vp, i32, vp, vp, f32, i32, i32, vp, vp, vp]
    L.m355_ep_reduce_add_rmsnorm.restype = i32
    L.m355_ep_pack.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.m355_ep_unpack.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.m355_ep_reduce.argtypes = [vp, i32, i32, i32, vp, vp]
    L.m355_linear_f16.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, vp]
    L.m355_linear_f16.restype = i32
    L.m355_add_rmsnorm.argtypes = [vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, vp]
    L.m355_add_rmsnorm_router.argtypes = [vp, vp, vp, ctypes.c_float, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.m355_add_rmsnorm_router.restype = i32
    L.m355_moe_combine_add_rmsnorm.argtypes = [vp, sz, i32, i32, i32, i32, vp, i32, vp, vp, ctypes.c_float, i32, vp, vp, vp]
    L.m355_moe_combine_add_rmsnorm.restype = i32
    L.m355_rope_kv_write.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.m355_qkv_rope_kv_write.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.m355_qkv_rope_kv_write.restype = i32
    L.m355_norm_qkv_rope_kv_write.argtypes = [vp, vp, vp, sz, i32, i32, i32, vp, i32, vp, f32, i32, i32, vp, vp, vp, vp, vp, vp,
                                              i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.m355_norm_qkv_rope_kv_write.restype = i32
    L.m355_time_next_launch.argtypes = []
    L.m355_time_next_launch.restype = i32
    L.m355_timed_launch_us.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.m355_timed_launch_us.restype = i32
    for fn in ("m355_ep_pack", "m355_ep_unpack", "m355_ep_reduce", "m355_nf4_dequant", "m355_repack_nf4", "m355_router_topk", "m355_topk_softmax",
               "m355_moe_nf4_forward", "m355_paged_attn_decode", "m355_add_rmsnorm", "m355_rope_kv_write"):
        getattr(L, fn).restype = i32
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise Mi355LibraryError(f"{what} failed ({rc}): {load().m355_last_error().decode()}")


def _dev(t: torch.Tensor, dtype: torch.dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise Mi355LibraryError(f"{name} must be a GPU tensor; this path has no CPU implementation")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


class _StreamArg(int):
    """The current stream's handle; its type marks the stream argument of a library call for prepare()."""


def _stream() -> int:
    return _StreamArg(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


class _Recorder:
    """Stands in for the loaded library while prepare() runs: every call that ends in a stream argument is carried out
    and written down as (name, C function, arguments)."""

    def __init__(self, L):
        self._L, self.calls = L, []

    def __getattr__(self, name):
        fn = getattr(self._L, name)

        def call(*args):
            if args and isinstance(args[-1], _StreamArg):
                self.calls.append((name, fn, args[:-1]))
            return fn(*args)
        return call


class Prepared:
    """Library calls with their arguments checked and converted once.  Calling the object issues them again, in order,
    on the stream that is current then: for loops that repeat the same calls on the same buffers without a HIP graph
    (a step that holds collectives, or vLLM's --enforce-eager), where the per-call checks and conversions of the
    wrappers above cost more host time than the launch itself.  The tensors must stay alive and keep their addresses:
    the object holds on to `owner` (normally the closure that made the calls) for that."""

    def __init__(self, calls, device_index, owner):
        self._dev, self._owner = device_index, owner
        self._calls = []
        for name, fn, args in calls:
            conv = tuple(a if a is None else t(a) for t, a in zip(fn.argtypes, args))
            self._calls.append((name, fn, conv))

    def __call__(self):
        st = torch._C._cuda_getCurrentRawStream(self._dev)
        for name, fn, args in self._calls:
            rc = fn(*args, st)
            if rc != 0:
                _check(rc, name)


def prepare(fn) -> Prepared:
    """Run fn() once (its library calls are carried out as usual) and return a Prepared that repeats those calls."""
    global _lib
    real = load()
    rec = _Recorder(real)
    _lib = rec
    try:
        fn()
    finally:
        _lib = real
    return Prepared(rec.calls, torch.cuda.current_device(), fn)


def version() -> int:
    return int(load().m355_version())


def init(device: Optional[int] = None) -> None:
    """Per-device set-up (kernel attributes) ahead of the first call, so that the first call may already sit inside a
    HIP graph capture.  m355_init."""
    if device is None:
        device = torch.cuda.current_device()
    _check(load().m355_init(int(device)), "m355_init")


def nf4_dequant(q: torch.Tensor, absmax: torch.Tensor, blocksize: int = 64, out: Optional[torch.Tensor] = None):
    """Packed NF4 (uint8, checkpoint order) + fp32 absmax -> flat fp16.  m355_nf4_dequant."""
    n = q.numel() * 2
    if out is None:
        out = torch.empty(n, dtype=torch.float16, device=q.device)
    if absmax.numel() * blocksize != n:
        raise ValueError(f"absmax has {absmax.numel()} entries, expected {n // blocksize}")
    _check(load().m355_nf4_dequant(_dev(q, torch.uint8, "q"), _dev(absmax, torch.float32, "absmax"), n, blocksize,
                                   _dev(out, torch.float16, "out"), _stream()), "m355_nf4_dequant")
    return out


def repack_nf4(q: torch.Tensor, absmax: torch.Tensor, rows: int, cols: int, row_order: int = 0):
    """One expert matrix, checkpoint order -> tile order; row_order 1 = gate/up rows interleaved in the
    checkpoint.  m355_repack_nf4_ex."""
    if q.numel() * 2 != rows * cols or absmax.numel() * 64 != rows * cols:
        raise ValueError("repack_nf4: sizes do not match rows*cols")
    qt = torch.empty_like(q)
    at = torch.empty_like(absmax)
    _check(load().m355_repack_nf4_ex(_dev(q, torch.uint8, "q"), _dev(absmax, torch.float32, "absmax"), rows, cols,
                                     int(row_order), _dev(qt, torch.uint8, "q_tiled"),
                                     _dev(at, torch.float32, "absmax_tiled"), _stream()), "m355_repack_nf4")
    return qt, at


def router_topk(x: torch.Tensor, wr: torch.Tensor, br: Optional[torch.Tensor], k: int, want_logits: bool = False,
                ids: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None):
    """fp16 router linear + top-k + softmax.  Returns (ids int32 [T,k], w fp32 [T,k][, logits fp16 [T,E]])."""
    T, H = x.shape
    E = wr.shape[0]
    if ids is None:
        ids = torch.empty((T, k), dtype=torch.int32, device=x.device)
    if w is None:
        w = torch.empty((T, k), dtype=torch.float32, device=x.device)
    _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w")
    logits = torch.empty((T, E), dtype=torch.float16, device=x.device) if want_logits else None
    _check(load().m355_router_topk(_dev(x, torch.float16, "x"), _dev(wr, torch.float16, "wr"),
                                   None if br is None else _dev(br, torch.float16, "br"), T, H, E, k,
                                   ids.data_ptr(), w.data_ptr(), None if logits is None else logits.data_ptr(),
                                   _stream()), "m355_router_topk")
    return (ids, w, logits) if want_logits else (ids, w)


def topk_softmax(logits: torch.Tensor, k: int):
    T, E = logits.shape
    ids = torch.empty((T, k), dtype=torch.int32, device=logits.device)
    w = torch.empty((T, k), dtype=torch.float32, device=logits.device)
    _check(load().m355_topk_softmax(_dev(logits, torch.float16, "logits"), T, E, k, ids.data_ptr(), w.data_ptr(),
                                    _stream()), "m355_topk_softmax")
    return ids, w


def moe_workspace_bytes(T: int, k: int, H: int, FF: int, E_local: int) -> int:
    return int(load().m355_moe_workspace_bytes(T, k, H, FF, E_local))


TILE_MIN_MEAN_PAIRS = 80   # the library's own crossover between its two MoE GEMM kernels (csrc/moe.hip)
WIDE_MIN_MEAN_PAIRS = 16   # ... and, within the streaming kernel, between token tiles of 16 and of 32


def kernel_choice_flags(tokens: int, top_k: int, global_experts: int) -> int:
    """The library chooses between its two GEMM kernels from tokens * top_k / E_local, which is right when every
    expert is local.  A rank that holds a shard sees the tokens of all ranks but only its share of the pairs: there
    the caller, who knows the global expert count, fixes the choice with this flag."""
    many = tokens > 16 and tokens * top_k >= TILE_MIN_MEAN_PAIRS * global_experts
    if many:
        return FLAG_FORCE_TILE
    wide = tokens > 16 and tokens * top_k >= WIDE_MIN_MEAN_PAIRS * global_experts
    return FLAG_FORCE_STREAM | (FLAG_STREAM_WIDE if wide else 0)


def moe_nf4_forward(x, ids, w, w13_q, w13_absmax, w2_q, w2_absmax, *, E_local: int, e_first: int, H: int, FF: int,
                    workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                    act_mode: int = ACT_SILU_CONCAT, flags: int = FLAG_FP16_COMBINE,
                    b13: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None, alpha: float = 1.702,
                    limit: float = 7.0):
    """One MoE layer over this rank's experts (weights in tile order); optional per-expert biases b13 [E][2FF]
    (gate entries then up entries), b2 [E][H].  w13_absmax / w2_absmax are fp32 tensors, or (codes uint8, table fp32
    [E][256]) pairs for the byte form.  m355_moe_nf4_forward_ex / m355_moe_nf4_forward_q8."""
    T, k = ids.shape
    need = moe_workspace_bytes(T, k, H, FF, E_local)
    if workspace is None:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty((T, H), dtype=torch.float16, device=x.device)
    if b13 is not None and tuple(b13.shape) != (E_local, 2 * FF):
        raise ValueError(f"b13 is {tuple(b13.shape)}, expected {(E_local, 2 * FF)}")
    if b2 is not None and tuple(b2.shape) != (E_local, H):
        raise ValueError(f"b2 is {tuple(b2.shape)}, expected {(E_local, H)}")
    if isinstance(w13_absmax, tuple) != isinstance(w2_absmax, tuple):
        raise ValueError("w13_absmax and w2_absmax must both be fp32 tensors or both (codes, table) pairs")
    if isinstance(w13_absmax, tuple):
        (c13, t13), (c2, t2) = w13_absmax, w2_absmax
        if tuple(t13.shape) != (E_local, 256) or tuple(t2.shape) != (E_local, 256):
            raise ValueError(f"absmax tables must be {(E_local, 256)}")
        _check(load().m355_moe_nf4_forward_q8(
            _dev(x, torch.float16, "x"), T, _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"), k,
            _dev(w13_q, torch.uint8, "w13_q"), _dev(c13, torch.uint8, "w13_absmax codes"), _dev(t13, torch.float32, "w13 table"),
            _dev(w2_q, torch.uint8, "w2_q"), _dev(c2, torch.uint8, "w2_absmax codes"), _dev(t2, torch.float32, "w2 table"),
            E_local, e_first, H, FF, act_mode, flags, None if b13 is None else _dev(b13, torch.float16, "b13"),
            None if b2 is None else _dev(b2, torch.float16, "b2"), float(alpha), float(limit),
            _dev(workspace, torch.uint8, "workspace"), workspace.numel(), _dev(out, torch.float16, "out"), _stream()),
            "m355_moe_nf4_forward_q8")
        return out
    _check(load().m355_moe_nf4_forward_ex(
        _dev(x, torch.float16, "x"), T, _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"), k,
        _dev(w13_q, torch.uint8, "w13_q"), _dev(w13_absmax, torch.float32, "w13_absmax"),
        _dev(w2_q, torch.uint8, "w2_q"), _dev(w2_absmax, torch.float32, "w2_absmax"), E_local, e_first, H, FF,
        act_mode, flags, None if b13 is None else _dev(b13, torch.float16, "b13"),
        None if b2 is None else _dev(b2, torch.float16, "b2"), float(alpha), float(limit),
        _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
        _dev(out, torch.float16, "out"), _stream()), "m355_moe_nf4_forward")
    return out


STATIC_WORKSPACE_BYTES = 8 << 20   # M355_STATIC_WORKSPACE_BYTES of include/m355.h


def moe_nf4_forward_static(x, ids, w, w13_q, w13_absmax, w2_q, w2_absmax, *, E_local: int, e_first: int, H: int, FF: int,
                           out: torch.Tensor, act_mode: int = ACT_SILU_CONCAT, flags: int = FLAG_FP16_COMBINE,
                           b13: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None, alpha: float = 1.702,
                           limit: float = 7.0):
    """moe_nf4_forward (fp32 absmax) with the workspace inside the library: nothing is allocated, here or there, so `out`
    must be given.  Shapes beyond the library's static workspace raise.  m355_moe_nf4_forward_static."""
    T, k = ids.shape
    _check(load().m355_moe_nf4_forward_static(
        _dev(x, torch.float16, "x"), T, _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"), k,
        _dev(w13_q, torch.uint8, "w13_q"), _dev(w13_absmax, torch.float32, "w13_absmax"),
        _dev(w2_q, torch.uint8, "w2_q"), _dev(w2_absmax, torch.float32, "w2_absmax"), E_local, e_first, H, FF,
        act_mode, flags, None if b13 is None else _dev(b13, torch.float16, "b13"),
        None if b2 is None else _dev(b2, torch.float16, "b2"), float(alpha), float(limit),
        _dev(out, torch.float16, "out"), _stream()), "m355_moe_nf4_forward_static")
    return out


def paged_attn_workspace_bytes(B: int, Hq: int, D: int, max_ctx: int) -> int:
    return int(load().m355_paged_attn_workspace_bytes(B, Hq, D, max_ctx))


def moe_nf4_forward_routed(x, wr, br, k: int, w13_q, w13_absmax, w2_q, w2_absmax, *, E_local: int, e_first: int, H: int,
                           FF: int, ids: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None,
                           workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                           act_mode: int = ACT_SILU_CONCAT, flags: int = FLAG_FP16_COMBINE,
                           b13: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None, alpha: float = 1.702,
                           limit: float = 7.0):
    """Router (x @ wr.T + br -> top-k -> softmax) and the MoE layer in one call; ids [T][k] and w [T][k] receive the
    routing (the bits router_topk gives).  Up to 24 tokens the logits are a launch of their own over every CU and the
    gate_up launch picks the top-k (FLAG_ROUTER_LAUNCH: router_topk in front, as beyond 24 tokens).
    Returns (out, ids, w).  m355_moe_nf4_forward_routed."""
    T, E_global = x.shape[0], wr.shape[0]
    if ids is None:
        ids = torch.empty((T, k), dtype=torch.int32, device=x.device)
    if w is None:
        w = torch.empty((T, k), dtype=torch.float32, device=x.device)
    need = moe_workspace_bytes(T, k, H, FF, E_local)
    if workspace is None:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty((T, H), dtype=torch.float16, device=x.device)
    if isinstance(w13_absmax, tuple) != isinstance(w2_absmax, tuple):
        raise ValueError("w13_absmax and w2_absmax must both be fp32 tensors or both (codes, table) pairs")
    if isinstance(w13_absmax, tuple):
        (c13, t13), (c2, t2) = w13_absmax, w2_absmax
        a13 = (None, _dev(c13, torch.uint8, "w13_absmax codes"), _dev(t13, torch.float32, "w13 table"))
        a2 = (None, _dev(c2, torch.uint8, "w2_absmax codes"), _dev(t2, torch.float32, "w2 table"))
    else:
        a13 = (_dev(w13_absmax, torch.float32, "w13_absmax"), None, None)
        a2 = (_dev(w2_absmax, torch.float32, "w2_absmax"), None, None)
    _check(load().m355_moe_nf4_forward_routed(
        _dev(x, torch.float16, "x"), T, _dev(wr, torch.float16, "wr"), None if br is None else _dev(br, torch.float16, "br"),
        E_global, k, _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"), _dev(w13_q, torch.uint8, "w13_q"), *a13,
        _dev(w2_q, torch.uint8, "w2_q"), *a2, E_local, e_first, H, FF, act_mode, flags,
        None if b13 is None else _dev(b13, torch.float16, "b13"), None if b2 is None else _dev(b2, torch.float16, "b2"),
        float(alpha), float(limit), _dev(workspace, torch.uint8, "workspace"), workspace.numel(),
        _dev(out, torch.float16, "out"), _stream()), "m355_moe_nf4_forward_routed")
    return out, ids, w


def time_next_launch():
    """Arm this thread: the next streaming MoE GEMM launch is bracketed by events of its own.  m355_time_next_launch."""
    _check(load().m355_time_next_launch(), "m355_time_next_launch")


def timed_launch_us() -> float:
    """Duration in microseconds of the launch armed by time_next_launch() (waits for it).  m355_timed_launch_us."""
    us = ctypes.c_float(0.0)
    _check(load().m355_timed_launch_us(ctypes.byref(us)), "m355_timed_launch_us")
    return float(us.value)


def moe_nf4_forward_logits(x, logits, k: int, w13_q, w13_absmax, w2_q, w2_absmax, *, E_local: int, e_first: int, H: int,
                           FF: int, ids: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None,
                           workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                           act_mode: int = ACT_SILU_CONCAT, flags: int = FLAG_FP16_COMBINE,
                           b13: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None, alpha: float = 1.702,
                           limit: float = 7.0):
    """Top-k + softmax over finished router logits [T][E_global] fp16 and the MoE layer in one call; ids [T][k] and
    w [T][k] receive the routing (the bits topk_softmax gives).  Up to 24 tokens the gate_up launch picks the top-k.
    Returns (out, ids, w).  m355_moe_nf4_forward_logits."""
    T, E_global = x.shape[0], logits.shape[-1]
    if logits.shape[0] != T:
        raise ValueError(f"logits for {logits.shape[0]} tokens, x for {T}")
    if ids is None:
        ids = torch.empty((T, k), dtype=torch.int32, device=x.device)
    if w is None:
        w = torch.empty((T, k), dtype=torch.float32, device=x.device)
    need = moe_workspace_bytes(T, k, H, FF, E_local)
    if workspace is None:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.empty((T, H), dtype=torch.float16, device=x.device)
    if isinstance(w13_absmax, tuple) != isinstance(w2_absmax, tuple):
        raise ValueError("w13_absmax and w2_absmax must both be fp32 tensors or both (codes, table) pairs")
    if isinstance(w13_absmax, tuple):
        (c13, t13), (c2, t2) = w13_absmax, w2_absmax
        a13 = (None, _dev(c13, torch.uint8, "w13_absmax codes"), _dev(t13, torch.float32, "w13 table"))
        a2 = (None, _dev(c2, torch.uint8, "w2_absmax codes"), _dev(t2, torch.float32, "w2 table"))
    else:
        a13 = (_dev(w13_absmax, torch.float32, "w13_absmax"), None, None)
        a2 = (_dev(w2_absmax, torch.float32, "w2_absmax"), None, None)
    _check(load().m355_moe_nf4_forward_logits(
        _dev(x, torch.float16, "x"), T, _dev(logits, torch.float16, "logits"), E_global, k, _dev(ids, torch.int32, "ids"),
        _dev(w, torch.float32, "w"), _dev(w13_q, torch.uint8, "w13_q"), *a13, _dev(w2_q, torch.uint8, "w2_q"), *a2, E_local,
        e_first, H, FF, act_mode, flags, None if b13 is None else _dev(b13, torch.float16, "b13"),
        None if b2 is None else _dev(b2, torch.float16, "b2"), float(alpha), float(limit),
        _dev(workspace, torch.uint8, "workspace"), workspace.numel(), _dev(out, torch.float16, "out"), _stream()),
        "m355_moe_nf4_forward_logits")
    return out, ids, w


def paged_attn_decode(q, kcache, vcache, block_tables, ctx_lens, *, max_ctx: int, scale: float,
                      workspace: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                      sinks: Optional[torch.Tensor] = None, window: int = 0):
    """Decode attention over a paged KV cache; sinks [Hq] fp16 and a sliding window are optional.
    m355_paged_attn_decode_ex."""
    B, Hq, D = q.shape
    _, block_size, Hkv, _ = kcache.shape
    need = paged_attn_workspace_bytes(B, Hq, D, max_ctx)
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    if out is None:
        out = torch.empty_like(q)
    if sinks is not None and sinks.numel() != Hq:
        raise ValueError(f"sinks has {sinks.numel()} entries, expected one per query head ({Hq})")
    _check(load().m355_paged_attn_decode_ex(
        _dev(q, torch.float16, "q"), _dev(kcache, torch.float16, "kcache"), _dev(vcache, torch.float16, "vcache"),
        _dev(block_tables, torch.int32, "block_tables"), _dev(ctx_lens, torch.int32, "ctx_lens"), B, Hq, Hkv, D,
        block_size, block_tables.shape[1], max_ctx, scale,
        None if sinks is None else _dev(sinks, torch.float16, "sinks"), int(window),
        None if workspace is None else _dev(workspace, torch.uint8, "workspace"),
        0 if workspace is None else workspace.numel(), _dev(out, torch.float16, "out"), _stream()),
        "m355_paged_attn_decode")
    return out


def ep_row_bytes(H: int, k: int) -> int:
    return int(load().m355_ep_row_bytes(H, k))


def ep_pack(x, ids, w, n_ranks: int, experts_per_rank: int, send: torch.Tensor):
    """Fill send [n_ranks, T, row_bytes] (uint8) for the dispatch all-to-all.  m355_ep_pack."""
    T, k = ids.shape
    H = x.shape[1]
    if send.numel() != n_ranks * T * ep_row_bytes(H, k):
        raise ValueError("ep_pack: send buffer has the wrong size")
    _check(load().m355_ep_pack(_dev(x, torch.float16, "x"), _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"),
                               T, k, H, n_ranks, experts_per_rank, _dev(send, torch.uint8, "send"), _stream()),
           "m355_ep_pack")
    return send


def router_topk_ep_pack(x: torch.Tensor, wr: torch.Tensor, br: Optional[torch.Tensor], k: int, n_ranks: int,
                        experts_per_rank: int, send: torch.Tensor, ids: Optional[torch.Tensor] = None,
                        w: Optional[torch.Tensor] = None):
    """router_topk and ep_pack in one launch.  Returns (ids, w); `send` [n_ranks, T, row_bytes] is filled.
    m355_router_topk_ep_pack."""
    T, H = x.shape
    E = wr.shape[0]
    if ids is None:
        ids = torch.empty((T, k), dtype=torch.int32, device=x.device)
    if w is None:
        w = torch.empty((T, k), dtype=torch.float32, device=x.device)
    if send.numel() != n_ranks * T * ep_row_bytes(H, k):
        raise ValueError("router_topk_ep_pack: send buffer has the wrong size")
    _check(load().m355_router_topk_ep_pack(
        _dev(x, torch.float16, "x"), _dev(wr, torch.float16, "wr"), None if br is None else _dev(br, torch.float16, "br"), T, H, E, k,
        _dev(ids, torch.int32, "ids"), _dev(w, torch.float32, "w"), n_ranks, experts_per_rank, _dev(send, torch.uint8, "send"),
        _stream()), "m355_router_topk_ep_pack")
    return ids, w


def moe_nf4_forward_ep_rows(rows: torch.Tensor, NT: int, k: int, w13_q, w13_absmax, w2_q, w2_absmax, *, E_local: int,
                            e_first: int, H: int, FF: int, workspace: torch.Tensor, out: torch.Tensor,
                            act_mode: int = ACT_SILU_CONCAT, flags: int = FLAG_FP16_COMBINE, b13: Optional[torch.Tensor] = None,
                            b2: Optional[torch.Tensor] = None, alpha: float = 1.702, limit: float = 7.0):
    """The MoE layer on the NT token slots of an expert-parallel receive buffer, read in place (activations, ids and
    weights inside the rows).  m355_moe_nf4_forward_ep_rows."""
    if rows.numel() != NT * ep_row_bytes(H, k):
        raise ValueError("moe_nf4_forward_ep_rows: rows buffer has the wrong size")
    if isinstance(w13_absmax, tuple) != isinstance(w2_absmax, tuple):
        raise ValueError("w13_absmax and w2_absmax must both be fp32 tensors or both (codes, table) pairs")
    if isinstance(w13_absmax, tuple):
        (c13, t13), (c2, t2) = w13_absmax, w2_absmax
        a13 = (None, _dev(c13, torch.uint8, "w13_absmax codes"), _dev(t13, torch.float32, "w13 table"))
        a2 = (None, _dev(c2, torch.uint8, "w2_absmax codes"), _dev(t2, torch.float32, "w2 table"))
    else:
        a13 = (_dev(w13_absmax, torch.float32, "w13_absmax"), None, None)
        a2 = (_dev(w2_absmax, torch.float32, "w2_absmax"), None, None)
    _check(load().m355_moe_nf4_forward_ep_rows(
        _dev(rows, torch.uint8, "rows"), NT, k, _dev(w13_q, torch.uint8, "w13_q"), *a13, _dev(w2_q, torch.uint8, "w2_q"), *a2,
        E_local, e_first, H, FF, act_mode, flags, None if b13 is None else _dev(b13, torch.float16, "b13"),
        None if b2 is None else _dev(b2, torch.float16, "b2"), float(alpha), float(limit),
        _dev(workspace, torch.uint8, "workspace"), workspace.numel(), _dev(out, torch.float16, "out"), _stream()),
        "m355_moe_nf4_forward_ep_rows")
    return out


def ep_reduce_add_rmsnorm(recv: torch.Tensor, n_ranks: int, x: torch.Tensor, weight: torch.Tensor, eps: float,
                          resid_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """add_rmsnorm(x, delta) with delta = the sum over ranks of recv [n_ranks, T, H], as ep_reduce forms it.
    m355_ep_reduce_add_rmsnorm."""
    T, H = x.shape
    if recv.numel() != n_ranks * T * H:
        raise ValueError("ep_reduce_add_rmsnorm: recv buffer has the wrong size")
    if out is None:
        out = torch.empty_like(x)
    _check(load().m355_ep_reduce_add_rmsnorm(
        _dev(recv, torch.float16, "recv"), n_ranks, _dev(x, torch.float16, "x"), _dev(weight, torch.float16, "weight"), float(eps),
        T, H, None if resid_out is None else _dev(resid_out, torch.float16, "resid_out"), _dev(out, torch.float16, "out"),
        _stream()), "m355_ep_reduce_add_rmsnorm")
    return out


def ep_unpack(recv, NT: int, k: int, H: int, x_all, ids_all, w_all):
    """Split received rows into activations, expert ids and routing weights.  m355_ep_unpack."""
    if recv.numel() != NT * ep_row_bytes(H, k):
        raise ValueError("ep_unpack: recv buffer has the wrong size")
    _check(load().m355_ep_unpack(_dev(recv, torch.uint8, "recv"), NT, k, H, _dev(x_all, torch.float16, "x_all"),
                                 _dev(ids_all, torch.int32, "ids_all"), _dev(w_all, torch.float32, "w_all"), _stream()),
           "m355_ep_unpack")


def ep_reduce(recv, n_ranks: int, T: int, H: int, out):
    """out[t] = sum over ranks of recv[r, t].  m355_ep_reduce."""
    if recv.numel() != n_ranks * T * H:
        raise ValueError("ep_reduce: recv buffer has the wrong size")
    _check(load().m355_ep_reduce(_dev(recv, torch.float16, "recv"), n_ranks, T, H, _dev(out, torch.float16, "out"),
                                 _stream()), "m355_ep_reduce")
    return out


def add_rmsnorm(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, eps: float,
                resid_out: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """y = rmsnorm(x + delta) * weight on fp16 rows; resid_out (optional) receives x + delta.  m355_add_rmsnorm."""
    T, H = x.shape
    if out is None:
        out = torch.empty_like(x)
    _check(load().m355_add_rmsnorm(_dev(x, torch.float16, "x"), None if delta is None else _dev(delta, torch.float16, "delta"),
                                   _dev(weight, torch.float16, "weight"), float(eps), T, H,
                                   None if resid_out is None else _dev(resid_out, torch.float16, "resid_out"),
                                   _dev(out, torch.float16, "out"), _stream()), "m355_add_rmsnorm")
    return out


def add_rmsnorm_router(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, eps: float, wr: torch.Tensor,
                       br: Optional[torch.Tensor], k: int, resid_out: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None, ids: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None):
    """add_rmsnorm and router_topk on its output in one launch.  Returns (out, ids, w).  m355_add_rmsnorm_router."""
    T, H = x.shape
    E = wr.shape[0]
    if out is None:
        out = torch.empty_like(x)
    if ids is None:
        ids = torch.empty((T, k), dtype=torch.int32, device=x.device)
    if w is None:
        w = torch.empty((T, k), dtype=torch.float32, device=x.device)
    if wr.shape[1] != H or ids.shape != (T, k) or w.shape != (T, k):
        raise ValueError("add_rmsnorm_router: shapes do not match")
    _check(load().m355_add_rmsnorm_router(_dev(x, torch.float16, "x"), None if delta is None else _dev(delta, torch.float16, "delta"),
                                          _dev(weight, torch.float16, "weight"), float(eps), T, H,
                                          None if resid_out is None else _dev(resid_out, torch.float16, "resid_out"),
                                          _dev(out, torch.float16, "out"), _dev(wr, torch.float16, "wr"),
                                          None if br is None else _dev(br, torch.float16, "br"), E, k, _dev(ids, torch.int32, "ids"),
                                          _dev(w, torch.float32, "w"), _stream()), "m355_add_rmsnorm_router")
    return out, ids, w


def moe_combine_add_rmsnorm(moe_workspace: torch.Tensor, w: torch.Tensor, *, FF: int, E_local: int, flags: int,
                            x: torch.Tensor, weight: torch.Tensor, eps: float, resid_out: Optional[torch.Tensor] = None,
                            out: Optional[torch.Tensor] = None):
    """add_rmsnorm(x, delta) where delta is what the combine step of a MoE forward would write: the forward (same T, k, H,
    FF, E_local, `flags` its flags) ran with FLAG_SKIP_COMBINE on `moe_workspace`, `w` are its routing weights [T][k].
    m355_moe_combine_add_rmsnorm."""
    T, H = x.shape
    k = w.shape[1]
    if w.shape[0] != T:
        raise ValueError(f"w has {w.shape[0]} rows, x has {T}")
    if out is None:
        out = torch.empty_like(x)
    _check(load().m355_moe_combine_add_rmsnorm(
        _dev(moe_workspace, torch.uint8, "moe_workspace"), moe_workspace.numel(), T, k, FF, E_local, _dev(w, torch.float32, "w"),
        int(flags), _dev(x, torch.float16, "x"), _dev(weight, torch.float16, "weight"), float(eps), H,
        None if resid_out is None else _dev(resid_out, torch.float16, "resid_out"), _dev(out, torch.float16, "out"), _stream()),
        "m355_moe_combine_add_rmsnorm")
    return out


def rope_kv_write(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor, slots: torch.Tensor,
                  kcache: torch.Tensor, vcache: torch.Tensor, *, Hq: int, Hkv: int, D: int,
                  q_out: Optional[torch.Tensor] = None):
    """Rotary embedding of q and k, q to q_out [T][Hq][D], k and v into the paged cache.  m355_rope_kv_write."""
    T = qkv.shape[0]
    if qkv.shape[1] != (Hq + 2 * Hkv) * D:
        raise ValueError(f"qkv rows hold {qkv.shape[1]} values, expected {(Hq + 2 * Hkv) * D}")
    if cos.shape != sin.shape or cos.shape[1] != D // 2:
        raise ValueError("cos/sin must be [max_pos][D/2]")
    if kcache.shape != vcache.shape or kcache.shape[2] != Hkv or kcache.shape[3] != D:
        raise ValueError("caches must be [blocks][block_size][Hkv][D]")
    if q_out is None:
        q_out = torch.empty((T, Hq, D), dtype=torch.float16, device=qkv.device)
    _check(load().m355_rope_kv_write(_dev(qkv, torch.float16, "qkv"), _dev(cos, torch.float16, "cos"),
                                     _dev(sin, torch.float16, "sin"), _dev(positions, torch.int32, "positions"),
                                     _dev(slots, torch.int64, "slots"), T, Hq, Hkv, D, kcache.shape[1], cos.shape[0],
                                     _dev(q_out, torch.float16, "q_out"), _dev(kcache, torch.float16, "kcache"),
                                     _dev(vcache, torch.float16, "vcache"), _stream()), "m355_rope_kv_write")
    return q_out


def qkv_rope_kv_write(x: torch.Tensor, wqkv: torch.Tensor, bqkv: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor,
                      positions: torch.Tensor, slots: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, *, Hq: int,
                      Hkv: int, D: int, q_out: Optional[torch.Tensor] = None, qkv_scratch: Optional[torch.Tensor] = None):
    """The q/k/v projection of x [T][K] followed by rope_kv_write on its result, in one call (one launch up to 32 tokens;
    beyond, the two launches through qkv_scratch [T][(Hq + 2 Hkv) D], allocated here when missing).  The bits of
    linear_f16 then rope_kv_write.  m355_qkv_rope_kv_write."""
    T, K = x.shape
    N = (Hq + 2 * Hkv) * D
    if wqkv.shape != (N, K):
        raise ValueError(f"wqkv is {tuple(wqkv.shape)}, expected {(N, K)}")
    if cos.shape != sin.shape or cos.shape[1] != D // 2:
        raise ValueError("cos/sin must be [max_pos][D/2]")
    if kcache.shape != vcache.shape or kcache.shape[2] != Hkv or kcache.shape[3] != D:
        raise ValueError("caches must be [blocks][block_size][Hkv][D]")
    if q_out is None:
        q_out = torch.empty((T, Hq, D), dtype=torch.float16, device=x.device)
    if qkv_scratch is None and T > 4:
        qkv_scratch = torch.empty((T, N), dtype=torch.float16, device=x.device)
    _check(load().m355_qkv_rope_kv_write(
        _dev(x, torch.float16, "x"), T, _dev(wqkv, torch.float16, "wqkv"), None if bqkv is None else _dev(bqkv, torch.float16, "bqkv"),
        K, _dev(cos, torch.float16, "cos"), _dev(sin, torch.float16, "sin"), _dev(positions, torch.int32, "positions"),
        _dev(slots, torch.int64, "slots"), Hq, Hkv, D, kcache.shape[1], cos.shape[0],
        None if qkv_scratch is None else _dev(qkv_scratch, torch.float16, "qkv_scratch"), _dev(q_out, torch.float16, "q_out"),
        _dev(kcache, torch.float16, "kcache"), _dev(vcache, torch.float16, "vcache"), _stream()), "m355_qkv_rope_kv_write")
    return q_out


def norm_qkv_rope_kv_write(x: torch.Tensor, delta, ln_weight: torch.Tensor, eps: float, wqkv: torch.Tensor,
                           bqkv: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor,
                           slots: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, *, Hq: int, Hkv: int, D: int,
                           resid_out: Optional[torch.Tensor] = None, q_out: Optional[torch.Tensor] = None,
                           h_scratch: Optional[torch.Tensor] = None, qkv_scratch: Optional[torch.Tensor] = None):
    """add_rmsnorm(x, delta) followed by qkv_rope_kv_write on the normalised rows, in one call (one launch up to 4
    tokens).  delta: None, a tensor [T][K], or (moe_workspace, w, FF, E_local, flags) for the expert outputs a MoE forward
    left uncombined (FLAG_SKIP_COMBINE), as moe_combine_add_rmsnorm takes them.  resid_out receives x + delta.
    m355_norm_qkv_rope_kv_write."""
    T, K = x.shape
    N = (Hq + 2 * Hkv) * D
    if wqkv.shape != (N, K):
        raise ValueError(f"wqkv is {tuple(wqkv.shape)}, expected {(N, K)}")
    if cos.shape != sin.shape or cos.shape[1] != D // 2:
        raise ValueError("cos/sin must be [max_pos][D/2]")
    if kcache.shape != vcache.shape or kcache.shape[2] != Hkv or kcache.shape[3] != D:
        raise ValueError("caches must be [blocks][block_size][Hkv][D]")
    if q_out is None:
        q_out = torch.empty((T, Hq, D), dtype=torch.float16, device=x.device)
    if T > 4:
        if h_scratch is None:
            h_scratch = torch.empty((T, K), dtype=torch.float16, device=x.device)
        if qkv_scratch is None:
            qkv_scratch = torch.empty((T, N), dtype=torch.float16, device=x.device)
    ws_ptr, ws_bytes, k, FF, E_local, cw, flags, delta_ptr = None, 0, 0, 0, 0, None, 0, None
    if isinstance(delta, tuple):
        ws_t, w_t, FF, E_local, flags = delta
        if w_t.shape[0] != T:
            raise ValueError(f"w has {w_t.shape[0]} rows, x has {T}")
        ws_ptr, ws_bytes, k, cw = _dev(ws_t, torch.uint8, "moe_workspace"), ws_t.numel(), w_t.shape[1], _dev(w_t, torch.float32, "w")
    elif delta is not None:
        delta_ptr = _dev(delta, torch.float16, "delta")
    _check(load().m355_norm_qkv_rope_kv_write(
        _dev(x, torch.float16, "x"), delta_ptr, ws_ptr, ws_bytes, k, int(FF), int(E_local), cw, int(flags),
        _dev(ln_weight, torch.float16, "ln_weight"), float(eps), T, K, _dev(wqkv, torch.float16, "wqkv"),
        None if bqkv is None else _dev(bqkv, torch.float16, "bqkv"), _dev(cos, torch.float16, "cos"), _dev(sin, torch.float16, "sin"),
        _dev(positions, torch.int32, "positions"), _dev(slots, torch.int64, "slots"), Hq, Hkv, D, kcache.shape[1], cos.shape[0],
        None if resid_out is None else _dev(resid_out, torch.float16, "resid_out"),
        None if h_scratch is None else _dev(h_scratch, torch.float16, "h_scratch"),
        None if qkv_scratch is None else _dev(qkv_scratch, torch.float16, "qkv_scratch"), _dev(q_out, torch.float16, "q_out"),
        _dev(kcache, torch.float16, "kcache"), _dev(vcache, torch.float16, "vcache"), _stream()), "m355_norm_qkv_rope_kv_write")
    return q_out


def linear_f16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
               resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """out = x @ weight.T + bias (+ resid) for decode-sized x; weight in nn.Linear layout [N][K].  m355_linear_f16."""
    T, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight is {tuple(weight.shape)}, x has {K} columns")
    if out is None:
        out = torch.empty((T, N), dtype=torch.float16, device=x.device)
    _check(load().m355_linear_f16(_dev(x, torch.float16, "x"), T, _dev(weight, torch.float16, "weight"),
                                  None if bias is None else _dev(bias, torch.float16, "bias"), N, K,
                                  None if resid is None else _dev(resid, torch.float16, "resid"),
                                  _dev(out, torch.float16, "out"), _stream()), "m355_linear_f16")
    return out


def repack_absmax_u8(codes: torch.Tensor, rows: int, cols: int, row_order: int = 0):
    """absmax code bytes of one expert matrix, checkpoint order -> tile order.  m355_repack_absmax_u8."""
    if codes.numel() * 64 != rows * cols:
        raise ValueError("repack_absmax_u8: sizes do not match rows*cols")
    out = torch.empty_like(codes)
    _check(load().m355_repack_absmax_u8(_dev(codes, torch.uint8, "codes"), rows, cols, int(row_order),
                                        _dev(out, torch.uint8, "out"), _stream()), "m355_repack_absmax_u8")
    return out


def attn_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, seq_starts: torch.Tensor, max_seq: int, *, scale: float,
                 sinks: Optional[torch.Tensor] = None, window: int = 0, out: Optional[torch.Tensor] = None):
    """Causal self-attention of packed prompts: q [tokens][Hq][64], k / v [tokens][Hkv][64] fp16 (the token dimension may
    be strided, e.g. slices of a fused qkv tensor), seq_starts int32 [B + 1] on the GPU.  m355_attn_prefill."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not t.is_cuda or t.dtype != torch.float16 or t.stride(2) != 1 or t.stride(1) != D:
            raise Mi355LibraryError(f"{name} must be a GPU fp16 tensor [tokens][heads][{D}] with contiguous heads")
    if k.stride(0) != v.stride(0):
        raise ValueError("k and v must share their token stride")
    if out is None:
        out = torch.empty((T, Hq, D), dtype=torch.float16, device=q.device)
    B = seq_starts.numel() - 1
    _check(load().m355_attn_prefill(q.data_ptr(), k.data_ptr(), v.data_ptr(), _dev(seq_starts, torch.int32, "seq_starts"), B,
                                    int(max_seq), Hq, Hkv, D, q.stride(0), k.stride(0), out.stride(0), float(scale),
                                    None if sinks is None else _dev(sinks, torch.float16, "sinks"), int(window),
                                    _dev(out, torch.float16, "out"), _stream()), "m355_attn_prefill")
    return out
