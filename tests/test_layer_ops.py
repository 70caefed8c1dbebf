"""Operators around the hot path (residual add + RMSNorm, rotary embedding + KV-cache write).

Golden outputs come from transformers' gpt_oss modules (tests/golden/make_layer_golden.py); inputs are regenerated
here from the same seeds.  CPU tests pin the numpy restatements and the host-side YaRN tables to those outputs; GPU
tests pin the kernels to both.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref as oracle_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H, HQ, HKV, D = 2880, 64, 8, 64
ROPE_POSITIONS = list(range(64)) + [255, 1000, 4095, 4096, 50000, 131071]
TOKEN_POSITIONS = [5, 17, 4100]


def _g(name):
    return np.load(os.path.join(GOLD, name))


def _norm_inputs():
    x = oracle_ref.normal_f16(oracle_ref.stream_seed(700, 1), (5, H))
    delta = oracle_ref.normal_f16(oracle_ref.stream_seed(700, 2), (5, H), 0.5)
    weight = (1.0 + 0.1 * oracle_ref.normal_f16(oracle_ref.stream_seed(700, 3), (H,)).astype(np.float32)).astype(np.float16)
    return x, delta, weight


def _ulps(a, b):
    return np.abs(a.view(np.int16).astype(np.int32) - b.view(np.int16).astype(np.int32)).max()


def test_oracle_rmsnorm_vs_transformers():
    x, delta, weight = _norm_inputs()
    r, y = oracle_ref.add_rmsnorm(x, delta, weight, 1e-5)
    assert np.array_equal(r.view(np.uint16), _g("layer_rmsnorm.resid.npy").view(np.uint16))
    assert _ulps(y, _g("layer_rmsnorm.y.npy")) <= 1      # the fp32 mean is summed in a different order


def test_host_rope_tables_vs_transformers():
    from gpt_oss_vllm_amd import layer
    cos, sin = layer.rope_tables(131072)
    idx = torch.tensor(ROPE_POSITIONS)
    assert np.array_equal(cos[idx].numpy().view(np.uint16), _g("layer_rope.cos.npy").view(np.uint16))
    assert np.array_equal(sin[idx].numpy().view(np.uint16), _g("layer_rope.sin.npy").view(np.uint16))


def test_oracle_rope_vs_transformers():
    from gpt_oss_vllm_amd import layer
    cos, sin = layer.rope_tables(4200)
    q = oracle_ref.normal_f16(oracle_ref.stream_seed(701, 1), (3, HQ, D))
    k = oracle_ref.normal_f16(oracle_ref.stream_seed(701, 2), (3, HKV, D))
    c = cos[TOKEN_POSITIONS].numpy()[:, None, :]
    s = sin[TOKEN_POSITIONS].numpy()[:, None, :]
    assert np.array_equal(oracle_ref.rope_rotate(q, c, s).view(np.uint16), _g("layer_rope.q.npy").view(np.uint16))
    assert np.array_equal(oracle_ref.rope_rotate(k, c, s).view(np.uint16), _g("layer_rope.k.npy").view(np.uint16))


@pytest.mark.gpu
def test_gpu_add_rmsnorm():
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    x, delta, weight = _norm_inputs()
    xd, dd, wd = (torch.from_numpy(a).to(cuda) for a in (x, delta, weight))
    resid = torch.empty_like(xd)
    y = lib.add_rmsnorm(xd, dd, wd, 1e-5, resid_out=resid)
    torch.cuda.synchronize()
    assert np.array_equal(resid.cpu().numpy().view(np.uint16), _g("layer_rmsnorm.resid.npy").view(np.uint16))
    assert _ulps(y.cpu().numpy(), _g("layer_rmsnorm.y.npy")) <= 1
    # no residual: the plain norm; small and ragged row counts
    for T, Hs in ((1, 64), (3, 8), (2, 8192)):
        xs = oracle_ref.normal_f16(oracle_ref.stream_seed(702, T, Hs), (T, Hs))
        ws = oracle_ref.normal_f16(oracle_ref.stream_seed(703, Hs), (Hs,))
        got = lib.add_rmsnorm(torch.from_numpy(xs).to(cuda), None, torch.from_numpy(ws).to(cuda), 1e-6)
        _, want = oracle_ref.add_rmsnorm(xs, None, ws, 1e-6)
        assert _ulps(got.cpu().numpy(), want) <= 1
    assert lib.add_rmsnorm(torch.empty((0, 64), dtype=torch.float16, device=cuda), None, wd[:64].contiguous(), 1e-5).shape == (0, 64)
    with pytest.raises(lib.Mi355LibraryError, match="unsupported shape"):
        lib.add_rmsnorm(torch.zeros((1, 60), dtype=torch.float16, device=cuda), None, wd[:60].contiguous(), 1e-5)


@pytest.mark.gpu
def test_gpu_rope_kv_write():
    from gpt_oss_vllm_amd import layer, lib
    cuda = torch.device("cuda:0")
    cos, sin = layer.rope_tables(4200)
    T, block_size, nblocks = 3, 16, 6
    q = oracle_ref.normal_f16(oracle_ref.stream_seed(701, 1), (T, HQ, D))
    k = oracle_ref.normal_f16(oracle_ref.stream_seed(701, 2), (T, HKV, D))
    v = oracle_ref.normal_f16(oracle_ref.stream_seed(701, 3), (T, HKV, D))
    qkv = np.concatenate([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], axis=1)
    slots = np.array([37, -1, 80], dtype=np.int64)      # the middle token is padding: its K/V must not be written
    kc = torch.full((nblocks, block_size, HKV, D), 7.0, dtype=torch.float16, device=cuda)
    vc = torch.full((nblocks, block_size, HKV, D), 7.0, dtype=torch.float16, device=cuda)
    q_out = lib.rope_kv_write(torch.from_numpy(qkv).to(cuda), cos.to(cuda), sin.to(cuda),
                              torch.tensor(TOKEN_POSITIONS, dtype=torch.int32, device=cuda), torch.from_numpy(slots).to(cuda),
                              kc, vc, Hq=HQ, Hkv=HKV, D=D)
    torch.cuda.synchronize()
    assert np.array_equal(q_out.cpu().numpy().view(np.uint16), _g("layer_rope.q.npy").view(np.uint16))
    kg = _g("layer_rope.k.npy")
    kc_h, vc_h = kc.cpu().numpy().reshape(-1, HKV, D), vc.cpu().numpy().reshape(-1, HKV, D)
    for t in (0, 2):
        assert np.array_equal(kc_h[slots[t]].view(np.uint16), kg[t].view(np.uint16))
        assert np.array_equal(vc_h[slots[t]].view(np.uint16), v[t].view(np.uint16))
    untouched = np.ones(nblocks * block_size, dtype=bool)
    untouched[[37, 80]] = False
    assert (kc_h[untouched] == 7.0).all() and (vc_h[untouched] == 7.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("T,N,K,bias,resid", [(1, 5120, 2880, True, False), (32, 2880, 4096, True, True),
                                              (40, 64, 256, True, True), (17, 48, 64, False, False), (5, 16, 3072, False, True),
                                              # up to 4 tokens: the half-row kernel (rows not a multiple of the row batch,
                                              # fewer rows than workgroups, K with a ragged last piece, the largest K)
                                              (1, 2880, 4096, True, True), (2, 5120, 2880, False, True), (3, 48, 64, True, False),
                                              (4, 2880, 4096, True, True), (4, 7184, 192, False, False), (1, 16, 4096, True, True),
                                              # from 128 tokens on: the tiled kernel (whole and ragged token tiles, row groups)
                                              (128, 5120, 2880, True, False), (300, 2880, 4096, True, True),
                                              (1000, 48, 64, False, True), (2049, 272, 128, True, True), (257, 512, 192, False, False)])
def test_gpu_linear_f16(T, N, K, bias, resid):
    """q/k/v and o projection shapes of gpt-oss-20b plus ragged small ones, against an fp32 matmul of the same fp16
    inputs (tolerance: fp16 rounding of the result plus fp32 summation order)."""
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    sd = oracle_ref.stream_seed(710, T, N, K)
    x = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 1), (T, K))
    w = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 2), (N, K), 0.02)
    b = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 3), (N,), 0.1) if bias else None
    r = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 4), (T, N)) if resid else None
    want = x.astype(np.float32) @ w.astype(np.float32).T
    if bias:
        want = want + b.astype(np.float32)
    want = want.astype(np.float16)
    if resid:
        want = (want + r).astype(np.float16)
    dev = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    got = lib.linear_f16(dev(x), dev(w), dev(b), dev(r))
    torch.cuda.synchronize()
    err = np.abs(got.cpu().numpy().astype(np.float32) - want.astype(np.float32)).max()
    assert err <= 2e-3 * np.abs(want.astype(np.float32)).max() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T,Hq,Hkv,D,K,bias", [(1, 64, 8, 64, 2880, True), (4, 64, 8, 64, 2880, True), (3, 4, 2, 32, 192, False),
                                               (2, 6, 3, 16, 4096, True), (7, 8, 4, 64, 256, True),
                                               # 5 to 32 tokens: the MFMA form with the rotation in its epilogue
                                               (32, 64, 8, 64, 2880, True), (17, 64, 8, 64, 2880, False), (5, 64, 8, 64, 2880, True),
                                               (16, 4, 2, 32, 4096, True), (9, 3, 1, 16, 320, False), (32, 8, 8, 64, 3072, True),
                                               # beyond: the two launches
                                               (33, 8, 2, 64, 256, True), (20, 4, 2, 32, 4096, False)])
def test_gpu_qkv_rope_kv_write_equals_the_two_launches(T, Hq, Hkv, D, K, bias):
    """m355_qkv_rope_kv_write (one launch up to 32 tokens) against m355_linear_f16 followed by m355_rope_kv_write: q, and
    every word of both caches (a token with a negative slot writes nothing), bit for bit."""
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    sd = oracle_ref.stream_seed(720, T, Hq, K)
    N, block_size, nblocks, max_pos = (Hq + 2 * Hkv) * D, 16, 5, 40
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    x = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 1), (T, K)))
    w = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 2), (N, K), 0.03))
    b = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 3), (N,), 0.2)) if bias else None
    cos = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 4), (max_pos, D // 2)))
    sin = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 5), (max_pos, D // 2)))
    positions = torch.tensor([(7 * t + 3) % max_pos for t in range(T)], dtype=torch.int32, device=cuda)
    slots = torch.tensor([(11 * t + 5) % (nblocks * block_size) if t != 1 else -1 for t in range(T)], dtype=torch.int64, device=cuda)
    outs = []
    for fused in (False, True):
        kc = torch.full((nblocks, block_size, Hkv, D), 7.0, dtype=torch.float16, device=cuda)
        vc = torch.full((nblocks, block_size, Hkv, D), 7.0, dtype=torch.float16, device=cuda)
        if fused:
            q = lib.qkv_rope_kv_write(x, w, b, cos, sin, positions, slots, kc, vc, Hq=Hq, Hkv=Hkv, D=D)
        else:
            q = lib.rope_kv_write(lib.linear_f16(x, w, b), cos, sin, positions, slots, kc, vc, Hq=Hq, Hkv=Hkv, D=D)
        outs.append((q, kc, vc))
    torch.cuda.synchronize()
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16), c.view(torch.int16))
    assert not (outs[1][1] == 7.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("T,Hq,Hkv,D,K,bias,mode", [(1, 64, 8, 64, 2880, True, "delta"), (1, 64, 8, 64, 2880, False, "pending"),
                                                    (4, 64, 8, 64, 2880, True, "pending"), (3, 8, 2, 32, 256, False, "none"),
                                                    (2, 4, 4, 64, 4096, True, "delta"), (6, 8, 2, 32, 256, True, "pending"),
                                                    (5, 8, 2, 32, 320, False, "delta")])
def test_gpu_norm_qkv_rope_kv_write_equals_the_launches(T, Hq, Hkv, D, K, bias, mode):
    """m355_norm_qkv_rope_kv_write (one launch up to 4 tokens) against the norm's launch (m355_add_rmsnorm, or
    m355_moe_combine_add_rmsnorm on the expert outputs a forward left uncombined) followed by m355_qkv_rope_kv_write: the new
    residual stream, q and every word of both caches, bit for bit."""
    from gpt_oss_vllm_amd import lib
    from test_gpu_parity import tiled_layer
    cuda = torch.device("cuda:0")
    sd = oracle_ref.stream_seed(730, T, Hq, K)
    N, block_size, nblocks, max_pos = (Hq + 2 * Hkv) * D, 16, 5, 40
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    x = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 1), (T, K)))
    w = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 2), (N, K), 0.03))
    b = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 3), (N,), 0.2)) if bias else None
    cos = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 4), (max_pos, D // 2)))
    sin = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 5), (max_pos, D // 2)))
    ln = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 6), (K,)))
    positions = torch.tensor([(7 * t + 3) % max_pos for t in range(T)], dtype=torch.int32, device=cuda)
    slots = torch.tensor([(11 * t + 5) % (nblocks * block_size) if t != 1 else -1 for t in range(T)], dtype=torch.int64, device=cuda)
    delta = None
    if mode == "delta":
        delta = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 7), (T, K)))
    elif mode == "pending":      # a MoE forward of the same tokens that stopped before its combine step
        E, k, FF = 8, 2, 192
        layer = oracle_ref.synth_layer(oracle_ref.stream_seed(sd, 8), E, K, FF)
        tl = tiled_layer(lib, layer, E, K, FF, cuda)
        lg = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 9), (T, E), 1.5)
        ids, wts, _ = oracle_ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
        ws = torch.empty(max(256, lib.moe_workspace_bytes(T, k, K, FF, E)), dtype=torch.uint8, device=cuda)
        xe = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 10), (T, K)))
        wts_d = dev(wts)
        lib.moe_nf4_forward(xe, dev(ids), wts_d, *tl, E_local=E, e_first=0, H=K, FF=FF, workspace=ws,
                            flags=lib.FLAG_FP16_COMBINE | lib.FLAG_SKIP_COMBINE)
        delta = (ws, wts_d, FF, E, lib.FLAG_FP16_COMBINE)
    outs = []
    for fused in (False, True):
        kc = torch.full((nblocks, block_size, Hkv, D), 7.0, dtype=torch.float16, device=cuda)
        vc = torch.full((nblocks, block_size, Hkv, D), 7.0, dtype=torch.float16, device=cuda)
        resid = torch.full((T, K), float("nan"), dtype=torch.float16, device=cuda)
        if fused:
            q = lib.norm_qkv_rope_kv_write(x, delta, ln, 1e-5, w, b, cos, sin, positions, slots, kc, vc, Hq=Hq, Hkv=Hkv, D=D,
                                           resid_out=resid)
        else:
            if mode == "pending":
                h = lib.moe_combine_add_rmsnorm(delta[0], delta[1], FF=delta[2], E_local=delta[3], flags=delta[4], x=x, weight=ln,
                                                eps=1e-5, resid_out=resid)
            else:
                h = lib.add_rmsnorm(x, delta, ln, 1e-5, resid_out=resid)
            q = lib.qkv_rope_kv_write(h, w, b, cos, sin, positions, slots, kc, vc, Hq=Hq, Hkv=Hkv, D=D)
        outs.append((q, kc, vc, resid))
    torch.cuda.synchronize()
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16), c.view(torch.int16))
    assert not torch.isnan(outs[1][3]).any() and not (outs[1][1] == 7.0).all()


@pytest.mark.gpu
def test_gpu_linear_f16_prompt_size():
    """8192 tokens through the q/k/v projection shape (256-token tiles): sampled tokens against an fp32 matmul, and
    the result is the same whatever else shares the launch (a token's row alone through the small-batch kernel)."""
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    T, N, K = 8192, 5120, 2880
    g = torch.Generator(device=cuda)
    g.manual_seed(5)
    x = torch.randn((T, K), device=cuda, generator=g).half()
    w = (torch.randn((N, K), device=cuda, generator=g) * 0.05).half()
    b = torch.randn((N,), device=cuda, generator=g).half()
    r = torch.randn((T, N), device=cuda, generator=g).half()
    got = lib.linear_f16(x, w, b, r)
    sample = [0, 31, 32, 255, 256, 4097, 8000, 8191]
    small = lib.linear_f16(x[sample].contiguous(), w, b, r[sample].contiguous())
    torch.cuda.synchronize()
    xs, ws = x[sample].cpu().numpy().astype(np.float32), w.cpu().numpy().astype(np.float32)
    want = ((xs @ ws.T + b.cpu().numpy().astype(np.float32)).astype(np.float16) + r[sample].cpu().numpy()).astype(np.float32)
    tol = 2e-3 * np.abs(want).max() + 1e-5
    assert np.abs(got[sample].cpu().numpy().astype(np.float32) - want).max() <= tol
    assert np.abs(got[sample].float().cpu().numpy() - small.float().cpu().numpy()).max() <= tol


ATTN_CASES = (("sinks_full", 300, 0), ("sinks_window", 300, 128), ("sinks_short", 70, 128))


def _attn_case(ctx, window):
    sd = oracle_ref.stream_seed(705, ctx, window)
    q = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 1), (1, HQ, D))
    k = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 2), (ctx, HKV, D))
    v = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 3), (ctx, HKV, D))
    sinks = oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 4), (HQ,), 2.0)
    block_size = 16
    nblk = (ctx + block_size - 1) // block_size
    perm = np.random.default_rng(ctx).permutation(nblk).astype(np.int32)     # scattered pages
    kc = np.zeros((nblk, block_size, HKV, D), np.float16)
    vc = np.zeros((nblk, block_size, HKV, D), np.float16)
    for t in range(ctx):
        kc[perm[t // block_size], t % block_size] = k[t]
        vc[perm[t // block_size], t % block_size] = v[t]
    return q, kc, vc, perm[None, :].copy(), np.array([ctx], np.int32), sinks


@pytest.mark.parametrize("name,ctx,window", ATTN_CASES)
def test_oracle_attention_sinks_window_vs_transformers(name, ctx, window):
    q, kc, vc, bt, cl, sinks = _attn_case(ctx, window)
    got = oracle_ref.paged_attn_decode_ex(q, kc, vc, bt, cl, D ** -0.5, sinks=sinks, window=window)
    want = _g(f"layer_attn.{name}.npy")
    assert np.abs(got[0].astype(np.float32) - want).max() <= 2e-3 * np.abs(want).max() + 1e-5
    # without extras the numpy function is the C oracle's attention
    plain = oracle_ref.paged_attn_decode_ex(q, kc, vc, bt, cl, D ** -0.5)
    base = oracle_ref.paged_attn_decode(q, kc, vc, bt, cl, D ** -0.5)
    assert np.abs(plain.astype(np.float32) - base.astype(np.float32)).max() <= 2e-3 * np.abs(base.astype(np.float32)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name,ctx,window", ATTN_CASES)
def test_gpu_attention_sinks_window(name, ctx, window):
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    q, kc, vc, bt, cl, sinks = _attn_case(ctx, window)
    dev = lambda a: torch.from_numpy(a).to(cuda)
    out = lib.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(bt), dev(cl), max_ctx=ctx, scale=D ** -0.5,
                                sinks=dev(sinks), window=window)
    torch.cuda.synchronize()
    want = _g(f"layer_attn.{name}.npy")
    assert np.abs(out.cpu().numpy()[0].astype(np.float32) - want).max() <= 2e-3 * np.abs(want).max() + 1e-5
    # window without sinks, and a batch whose sequences sit on both sides of the window and of the partition size
    ctxs = [300, 5, 129, 600]
    B = len(ctxs)
    nblk_seq = (max(ctxs) + 15) // 16
    rng = np.random.default_rng(9)
    kcb = oracle_ref.normal_f16(oracle_ref.stream_seed(706, 1), (B * nblk_seq, 16, HKV, D))
    vcb = oracle_ref.normal_f16(oracle_ref.stream_seed(706, 2), (B * nblk_seq, 16, HKV, D))
    btb = rng.permutation(B * nblk_seq).astype(np.int32).reshape(B, nblk_seq)
    qb = oracle_ref.normal_f16(oracle_ref.stream_seed(706, 3), (B, HQ, D))
    clb = np.array(ctxs, np.int32)
    for sk in (None, sinks):
        got = lib.paged_attn_decode(dev(qb), dev(kcb), dev(vcb), dev(btb), dev(clb), max_ctx=max(ctxs), scale=D ** -0.5,
                                    sinks=None if sk is None else dev(sk), window=128)
        exp = oracle_ref.paged_attn_decode_ex(qb, kcb, vcb, btb, clb, D ** -0.5, sinks=sk, window=128)
        torch.cuda.synchronize()
        assert np.abs(got.cpu().numpy().astype(np.float32) - exp.astype(np.float32)).max() <= 2e-3 * np.abs(exp.astype(np.float32)).max() + 1e-5


@pytest.mark.parametrize("name", ["prompt_full", "prompt_window"])
def test_oracle_prompt_attention_vs_transformers(name):
    """Every query of a 100-token prompt at once (causal mask, sinks, optional window) against transformers'
    eager_attention_forward in fp32: pins the numpy prompt attention the GPU kernel is checked with."""
    p = oracle_ref.prompt_attention_case(name)
    got = oracle_ref.attn_prefill(p["q"], p["k"], p["v"], np.array([0, p["S"]], np.int32), p["D"] ** -0.5, p["sinks"], p["window"])
    want = _g(f"layer_attn.{name}.npy")
    assert np.abs(got.astype(np.float32) - want).max() <= 2e-3 * np.abs(want).max() + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["prompt_full", "prompt_window"])
def test_gpu_prompt_attention_vs_transformers(name):
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    p = oracle_ref.prompt_attention_case(name)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    out = lib.attn_prefill(dev(p["q"]), dev(p["k"]), dev(p["v"]), dev(np.array([0, p["S"]], np.int32)), p["S"],
                           scale=p["D"] ** -0.5, sinks=dev(p["sinks"]), window=p["window"])
    torch.cuda.synchronize()
    want = _g(f"layer_attn.{name}.npy")
    assert np.abs(out.cpu().numpy().astype(np.float32) - want).max() <= 2e-3 * np.abs(want).max() + 1e-5


def _oai_case():
    Hs, FFs, Es, ks, Ts = 256, 192, 4, 2, 9
    lay = oracle_ref.synth_layer(oracle_ref.stream_seed(720), Es, Hs, FFs)
    b13 = oracle_ref.normal_f16(oracle_ref.stream_seed(720, 1), (Es, 2 * FFs), 0.5)
    b2 = oracle_ref.normal_f16(oracle_ref.stream_seed(720, 2), (Es, Hs), 0.5)
    xs = oracle_ref.normal_f16(oracle_ref.stream_seed(720, 3), (Ts, Hs), 4.0)
    lg = oracle_ref.normal_f16(oracle_ref.stream_seed(720, 4), (Ts, Es), 1.5)
    ids, wts, _ = oracle_ref.router_topk(lg, np.eye(Es, dtype=np.float16), None, ks)
    return Hs, FFs, Es, lay, b13, b2, xs, ids, wts


def test_oracle_experts_oai_vs_transformers():
    Hs, FFs, Es, lay, b13, b2, xs, ids, wts = _oai_case()
    w13 = np.stack([oracle_ref.nf4_dequant(lay["w13_q"][e], 2 * FFs * Hs, lay["w13_absmax"][e]).reshape(2 * FFs, Hs) for e in range(Es)])
    w2 = np.stack([oracle_ref.nf4_dequant(lay["w2_q"][e], Hs * FFs, lay["w2_absmax"][e]).reshape(Hs, FFs) for e in range(Es)])
    got = oracle_ref.moe_forward_oai(xs, ids, wts, w13, w2, b13, b2).astype(np.float32)
    want = _g("layer_experts_oai.out.npy")
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max() + 1e-5


@pytest.mark.gpu
def test_gpu_experts_oai():
    """Interleaved gate/up rows (row_order 1 at repack), biases, clamp and alpha against transformers' GptOssExperts."""
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    Hs, FFs, Es, lay, b13, b2, xs, ids, wts = _oai_case()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    w13_q, w13_a, w2_q, w2_a = [], [], [], []
    for e in range(Es):
        qt, at = lib.repack_nf4(dev(lay["w13_q"][e]), dev(lay["w13_absmax"][e]), 2 * FFs, Hs, row_order=1)
        w13_q.append(qt), w13_a.append(at)
        qt, at = lib.repack_nf4(dev(lay["w2_q"][e]), dev(lay["w2_absmax"][e]), Hs, FFs)
        w2_q.append(qt), w2_a.append(at)
    b13_internal = np.concatenate([b13[:, 0::2], b13[:, 1::2]], axis=1)      # gate entries, then up entries
    out = lib.moe_nf4_forward(dev(xs), dev(ids), dev(wts), torch.stack(w13_q), torch.stack(w13_a), torch.stack(w2_q),
                              torch.stack(w2_a), E_local=Es, e_first=0, H=Hs, FF=FFs, act_mode=lib.ACT_SWIGLU_OAI, flags=0,
                              b13=dev(b13_internal), b2=dev(b2))
    torch.cuda.synchronize()
    want = _g("layer_experts_oai.out.npy")
    assert np.abs(out.cpu().numpy().astype(np.float32) - want).max() <= 2e-3 * np.abs(want).max() + 1e-5
    # biases alone, with the reference's activation: the silu path plus b13/b2 against the oracle's fp16 forward
    # on the same weights with the biases folded in by hand is covered by linearity: zero biases change nothing
    z13, z2 = torch.zeros((Es, 2 * FFs), dtype=torch.float16, device=cuda), torch.zeros((Es, Hs), dtype=torch.float16, device=cuda)
    args = (dev(xs), dev(ids), dev(wts), torch.stack(w13_q), torch.stack(w13_a), torch.stack(w2_q), torch.stack(w2_a))
    a = lib.moe_nf4_forward(*args, E_local=Es, e_first=0, H=Hs, FF=FFs)
    b = lib.moe_nf4_forward(*args, E_local=Es, e_first=0, H=Hs, FF=FFs, b13=z13, b2=z2)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_gpu_plugin_swigluoai():
    """The same block through the plug-in surface: gate_up_order="interleaved", fused bias tensors handed to
    weight_loader under their checkpoint names, apply(activation="swigluoai")."""
    from gpt_oss_vllm_amd import checkpoint, plugin
    cuda = torch.device("cuda:0")
    Hs, FFs, Es, lay, b13, b2, xs, ids, wts = _oai_case()
    method = plugin.Mi355Nf4MoEConfig(gate_up_order="interleaved").get_quant_method(None, "model.layers.0.mlp.experts")
    layer = torch.nn.Module()
    method.create_weights(layer, Es, Hs, FFs, torch.float16, device=cuda)
    for e in range(Es):
        for proj, tag in (("gate_up", "w13"), ("down", "w2")):
            key = checkpoint.expert_weight_key(0, e, proj)
            method.weight_loader(layer, None, torch.from_numpy(lay[tag + "_q"][e].copy()).reshape(-1, 1), key)
            method.weight_loader(layer, None, torch.from_numpy(lay[tag + "_absmax"][e].copy()), key + ".absmax")
    assert method.weight_loader(layer, None, torch.from_numpy(b13.copy()), "model.layers.0.mlp.experts.gate_up_proj_bias",
                                return_success=True) is True
    method.weight_loader(layer, None, torch.from_numpy(b2.copy()), "model.layers.0.mlp.experts.down_proj_bias")
    method.process_weights_after_loading(layer)
    lg = oracle_ref.normal_f16(oracle_ref.stream_seed(720, 4), (xs.shape[0], Es), 1.5)
    out = method.apply(layer, torch.from_numpy(xs).to(cuda), torch.from_numpy(lg).to(cuda), top_k=2, renormalize=True,
                       activation="swigluoai")
    want = _g("layer_experts_oai.out.npy")
    # the plug-in combines over k in fp16 as the reference does: one more fp16 rounding per term than the golden
    assert np.abs(out.cpu().numpy().astype(np.float32) - want).max() <= 4e-3 * np.abs(want).max() + 1e-5
    with pytest.raises(NotImplementedError):
        method.apply(layer, torch.from_numpy(xs).to(cuda), torch.from_numpy(lg).to(cuda), top_k=2, activation="gelu")


def _decoder_layer(name, max_tokens):
    """(layer object, case dict) of a whole-decoder-layer case on cuda:0."""
    from gpt_oss_vllm_amd import layer, lib
    cuda = torch.device("cuda:0")
    p = oracle_ref.decoder_layer_case(name)
    H_, FF_, E_ = p["H"], p["FF"], p["E"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    w13_q, w13_a, w2_q, w2_a = [], [], [], []
    for e in range(E_):
        qt, at = lib.repack_nf4(dev(p["lay"]["w13_q"][e]), dev(p["lay"]["w13_absmax"][e]), 2 * FF_, H_, row_order=1)
        w13_q.append(qt), w13_a.append(at)
        qt, at = lib.repack_nf4(dev(p["lay"]["w2_q"][e]), dev(p["lay"]["w2_absmax"][e]), H_, FF_)
        w2_q.append(qt), w2_a.append(at)
    weights = dict(ln1=dev(p["ln1"]), ln2=dev(p["ln2"]), wqkv=dev(p["wqkv"]), bqkv=dev(p["bqkv"]), wo=dev(p["wo"]), bo=dev(p["bo"]),
                   sinks=dev(p["sinks"]), router_w=dev(p["router_w"]), router_b=dev(p["router_b"]),
                   w13_q=torch.stack(w13_q), w13_absmax=torch.stack(w13_a), w2_q=torch.stack(w2_q), w2_absmax=torch.stack(w2_a),
                   b13=dev(np.concatenate([p["b13"][:, 0::2], p["b13"][:, 1::2]], axis=1)), b2=dev(p["b2"]))
    lyr = layer.GptOssDecodeLayer(weights, hidden=H_, heads=p["Hq"], kv_heads=p["Hkv"], head_dim=p["D"], ffn=FF_, experts=E_,
                                  top_k=p["k"], max_tokens=max_tokens, window=p["window"] if name == "sliding" else 0,
                                  act_mode=lib.ACT_SWIGLU_OAI)
    return lyr, p


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sliding", "full"])
def test_gpu_whole_decoder_layer_vs_transformers(name):
    """200 prompt tokens are written into the paged cache through the layer's own norm / projection / rotary
    kernels, then one new token goes through the whole layer; the result is the last row of transformers'
    GptOssDecoderLayer run over the 201-token sequence in fp32 (sliding-window layer and full-attention layer)."""
    from gpt_oss_vllm_amd import layer
    cuda = torch.device("cuda:0")
    lyr, p = _decoder_layer(name, 1)
    L = p["L"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    block_size = 16
    nblk = (L + 1 + block_size - 1) // block_size
    perm = np.random.default_rng(3).permutation(nblk).astype(np.int32)
    kc = torch.zeros((nblk, block_size, p["Hkv"], p["D"]), dtype=torch.float16, device=cuda)
    vc = torch.zeros_like(kc)
    cos, sin = layer.rope_tables(512)
    cos, sin = cos.to(cuda), sin.to(cuda)
    slot_of = lambda t: int(perm[t // block_size]) * block_size + t % block_size
    x = dev(p["x"])
    lyr.write_kv_only(x[:L].contiguous(), torch.arange(L, dtype=torch.int32, device=cuda),
                      torch.tensor([slot_of(t) for t in range(L)], dtype=torch.int64, device=cuda), kc, vc, cos, sin)
    resid2, moe_out = lyr.forward(x[L:].contiguous(), None, torch.tensor([L], dtype=torch.int32, device=cuda),
                                  torch.tensor([slot_of(L)], dtype=torch.int64, device=cuda), dev(perm[None, :].copy()),
                                  torch.tensor([L + 1], dtype=torch.int32, device=cuda), kc, vc, cos, sin, max_ctx=L + 1)
    torch.cuda.synchronize()
    got = (resid2.float() + moe_out.float()).cpu().numpy()[0]
    want = _g(f"layer_decoder.{name}.npy")
    assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("T,fp16", [(1, 0), (5, 1), (33, 0)])
def test_gpu_combine_fused_into_add_rmsnorm(T, fp16):
    """m355_moe_combine_add_rmsnorm (the combine step formed inside the next residual add + RMSNorm) against the two
    separate launches, bit for bit: forward with and without M355_FLAG_SKIP_COMBINE on the same routing."""
    from gpt_oss_vllm_amd import lib
    cuda = torch.device("cuda:0")
    E, k, H, FF = 8, 3, 256, 192
    sd = oracle_ref.stream_seed(T, fp16, 515)
    layer_np = oracle_ref.synth_layer(sd, E, H, FF)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    w13_q = torch.stack([lib.repack_nf4(dev(layer_np["w13_q"][e]), dev(layer_np["w13_absmax"][e]), 2 * FF, H)[0] for e in range(E)])
    w13_a = torch.stack([lib.repack_nf4(dev(layer_np["w13_q"][e]), dev(layer_np["w13_absmax"][e]), 2 * FF, H)[1] for e in range(E)])
    w2_q = torch.stack([lib.repack_nf4(dev(layer_np["w2_q"][e]), dev(layer_np["w2_absmax"][e]), H, FF)[0] for e in range(E)])
    w2_a = torch.stack([lib.repack_nf4(dev(layer_np["w2_q"][e]), dev(layer_np["w2_absmax"][e]), H, FF)[1] for e in range(E)])
    x = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 1), (T, H)))
    resid = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 3), (T, H)))
    lnw = dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 4), (H,), 0.1)) + 1
    ids, w = lib.topk_softmax(dev(oracle_ref.normal_f16(oracle_ref.stream_seed(sd, 2), (T, E), 1.5)), k)
    ws = torch.empty(max(256, lib.moe_workspace_bytes(T, k, H, FF, E)), dtype=torch.uint8, device=cuda)
    kw = dict(E_local=E, e_first=0, H=H, FF=FF, workspace=ws)
    delta = lib.moe_nf4_forward(x, ids, w, w13_q, w13_a, w2_q, w2_a, flags=fp16, **kw)
    r_want = torch.empty_like(resid)
    y_want = lib.add_rmsnorm(resid, delta, lnw, 1e-5, resid_out=r_want)
    lib.moe_nf4_forward(x, ids, w, w13_q, w13_a, w2_q, w2_a, flags=fp16 | lib.FLAG_SKIP_COMBINE, **kw)
    r_got = torch.empty_like(resid)
    y_got = lib.moe_combine_add_rmsnorm(ws, w, FF=FF, E_local=E, flags=fp16, x=resid, weight=lnw, eps=1e-5, resid_out=r_got)
    torch.cuda.synchronize()
    assert torch.isfinite(y_want.float()).all() and delta.float().abs().max() > 0
    assert torch.equal(r_got, r_want) and torch.equal(y_got, y_want)
    with pytest.raises(lib.Mi355LibraryError):   # a workspace too small for this forward
        lib.moe_combine_add_rmsnorm(ws[:256], w, FF=FF, E_local=E, flags=fp16, x=resid, weight=lnw, eps=1e-5)


@pytest.mark.gpu
def test_gpu_decoder_layers_deferred_combine_equals_plain():
    """Two decode steps through a layer chained to itself (the second consumes the first's expert output): with the
    combine deferred into the next add + RMSNorm the results equal the plain path bit for bit."""
    from gpt_oss_vllm_amd import layer
    cuda = torch.device("cuda:0")
    lyr, p = _decoder_layer("full", 1)
    L = p["L"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    block_size = 16
    nblk = (L + 2 + block_size - 1) // block_size
    perm = np.random.default_rng(5).permutation(nblk).astype(np.int32)
    cos, sin = layer.rope_tables(512)
    cos, sin = cos.to(cuda), sin.to(cuda)
    slot_of = lambda t: int(perm[t // block_size]) * block_size + t % block_size
    x = dev(p["x"])
    outs = []
    for defer in (False, True):
        kc = torch.zeros((nblk, block_size, p["Hkv"], p["D"]), dtype=torch.float16, device=cuda)
        vc = torch.zeros_like(kc)
        lyr.write_kv_only(x[:L].contiguous(), torch.arange(L, dtype=torch.int32, device=cuda),
                          torch.tensor([slot_of(t) for t in range(L)], dtype=torch.int64, device=cuda), kc, vc, cos, sin)
        args = lambda t: (torch.tensor([t], dtype=torch.int32, device=cuda), torch.tensor([slot_of(t)], dtype=torch.int64, device=cuda),
                          dev(perm[None, :].copy()), torch.tensor([t + 1], dtype=torch.int32, device=cuda), kc, vc, cos, sin)
        r1, d1 = lyr.forward(x[L:].contiguous(), None, *args(L), max_ctx=L + 2, defer_combine=defer)
        assert isinstance(d1, layer.PendingCombine) == defer
        r1 = r1.clone()
        r2, d2 = lyr.forward(r1, d1, *args(L + 1), max_ctx=L + 2, defer_combine=defer)
        d2 = d2.materialize() if defer else d2
        torch.cuda.synchronize()
        outs.append((r2.clone(), d2.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sliding", "full"])
def test_gpu_prompt_through_decoder_layer_vs_transformers(name):
    """The prompt pass of the layer: all 201 tokens at once (prompt attention kernel), every output row against
    transformers' GptOssDecoderLayer; then the cache the pass left serves a decode step: token 201 repeated as a new
    token at position 200 of a 200-token cache gives the prompt pass's own last row."""
    from gpt_oss_vllm_amd import layer
    cuda = torch.device("cuda:0")
    lyr, p = _decoder_layer(name, 1)
    S = p["L"] + 1
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    block_size = 16
    nblk = (S + block_size - 1) // block_size
    perm = np.random.default_rng(4).permutation(nblk).astype(np.int32)
    kc = torch.zeros((nblk, block_size, p["Hkv"], p["D"]), dtype=torch.float16, device=cuda)
    vc = torch.zeros_like(kc)
    cos, sin = layer.rope_tables(512)
    cos, sin = cos.to(cuda), sin.to(cuda)
    slots = torch.tensor([int(perm[t // block_size]) * block_size + t % block_size for t in range(S)], dtype=torch.int64, device=cuda)
    x = dev(p["x"])
    resid2, moe_out = lyr.prefill(x, None, torch.arange(S, dtype=torch.int32, device=cuda), slots,
                                  torch.tensor([0, S], dtype=torch.int32, device=cuda), S, kc, vc, cos, sin)
    torch.cuda.synchronize()
    got = (resid2.float() + moe_out.float()).cpu().numpy()
    want = _g(f"layer_decoder_rows.{name}.npy")
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max()
    # two prompts packed in one pass (the 201 tokens and their first 70) = each on its own
    both = torch.cat([x, x[:70]])
    pos2 = torch.cat([torch.arange(S), torch.arange(70)]).to(torch.int32).to(cuda)
    no_slot = torch.full((S + 70,), -1, dtype=torch.int64, device=cuda)           # negative slot: not cached
    r2, m2 = lyr.prefill(both, None, pos2, no_slot, torch.tensor([0, S, S + 70], dtype=torch.int32, device=cuda), S, kc, vc, cos, sin)
    torch.cuda.synchronize()
    got2 = (r2.float() + m2.float()).cpu().numpy()
    assert np.abs(got2[:S] - got).max() <= 2e-3 * np.abs(got).max()
    assert np.abs(got2[S:] - got[:70]).max() <= 2e-3 * np.abs(got).max()
    # decode on the cache the prompt pass filled
    resid_d, moe_d = lyr.forward(x[S - 1:].contiguous(), None, torch.tensor([S - 1], dtype=torch.int32, device=cuda),
                                 slots[S - 1:].contiguous(), dev(perm[None, :].copy()),
                                 torch.tensor([S], dtype=torch.int32, device=cuda), kc, vc, cos, sin, max_ctx=S)
    torch.cuda.synchronize()
    dec = (resid_d.float() + moe_d.float()).cpu().numpy()[0]
    assert np.abs(dec - got[-1]).max() <= 2e-3 * np.abs(got[-1]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("T,H,E,k,bias,delta", [(32, 2880, 32, 4, True, False), (1, 2880, 32, 4, False, True), (9, 256, 8, 3, True, True),
                                                (300, 512, 128, 4, True, False), (1100, 256, 32, 4, True, True),
                                                # either side of the token count at which m355_router_topk changes kernels, and a
                                                # shape its 16-token form does not take (more than 128 experts)
                                                (1023, 256, 32, 4, True, False), (1024, 256, 32, 4, False, True),
                                                (1030, 256, 160, 4, True, False)])
def test_add_rmsnorm_router_equals_the_two_launches(T, H, E, k, bias, delta, ref, m355, cuda):
    """The norm in front of the MoE block and the router on its output in one launch: the bits of m355_add_rmsnorm then
    m355_router_topk (from 1024 tokens the call is those two launches)."""
    sd = ref.stream_seed(9500, T, H, E)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    x = dev(ref.normal_f16(ref.stream_seed(sd, 1), (T, H)))
    d = dev(ref.normal_f16(ref.stream_seed(sd, 2), (T, H), 0.5)) if delta else None
    nw = dev(ref.normal_f16(ref.stream_seed(sd, 3), (H,)))
    wr = dev(ref.normal_f16(ref.stream_seed(sd, 4), (E, H), 0.05))
    br = dev(ref.normal_f16(ref.stream_seed(sd, 5), (E,), 0.3)) if bias else None
    r0 = torch.empty_like(x)
    y0 = m355.add_rmsnorm(x, d, nw, 1e-5, resid_out=r0)
    ids0, w0 = m355.router_topk(y0, wr, br, k)
    r1 = torch.empty_like(x)
    y1, ids1, w1 = m355.add_rmsnorm_router(x, d, nw, 1e-5, wr, br, k, resid_out=r1)
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(r1.view(torch.int16), r0.view(torch.int16))
    assert torch.equal(ids1, ids0) and torch.equal(w1.view(torch.int32), w0.view(torch.int32))
