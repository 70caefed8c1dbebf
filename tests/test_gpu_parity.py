"""GPU parity: every call goes through the C ABI (libmoe355.so) and is checked against the CPU oracle
and the committed golden outputs of the reference.  Run on the GPU box with -m gpu."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

# Tolerance for fp16 GEMM outputs against the reference / oracle (fp32 accumulation in a different order,
# codes rounded to fp16 before the matrix product): |d| <= 2e-3 * max|y| + 1e-5.
RTOL_MAX, ATOL = 2e-3, 1e-5


def _close(got, want, rtol=RTOL_MAX, atol=ATOL):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    lim = rtol * np.abs(want).max() + atol
    err = np.abs(got - want).max()
    assert err <= lim, f"max |d| {err} > {lim}"
    return err


def _to_dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def tiled_layer(m355, layer, E, H, FF, cuda):
    """Upload checkpoint-order experts and repack them on the GPU into tile order."""
    w13_q = torch.empty((E, FF * H), dtype=torch.uint8, device=cuda)
    w13_a = torch.empty((E, 2 * FF * H // 64), dtype=torch.float32, device=cuda)
    w2_q = torch.empty((E, H * FF // 2), dtype=torch.uint8, device=cuda)
    w2_a = torch.empty((E, H * FF // 64), dtype=torch.float32, device=cuda)
    for e in range(E):
        q, a = m355.repack_nf4(_to_dev(layer["w13_q"][e], cuda), _to_dev(layer["w13_absmax"][e], cuda), 2 * FF, H)
        w13_q[e], w13_a[e] = q, a
        q, a = m355.repack_nf4(_to_dev(layer["w2_q"][e], cuda), _to_dev(layer["w2_absmax"][e], cuda), H, FF)
        w2_q[e], w2_a[e] = q, a
    return w13_q, w13_a, w2_q, w2_a


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["gate_up_nested_mean", "down_nested_mean", "down_nested_elementwise",
                                  "down_absmax_f32", "down_absmax_bs128"])
def test_dequant_bit_exact_vs_reference_golden(name, ref, m355, cuda):
    gold = json.load(open(os.path.join(GOLD, "dequant_golden.json")))["cases"][name]
    case = ref.golden_dequant_case(name)
    a = ref.resolve_absmax(case["absmax"], case["nested_map"], case["nested_absmax"])
    n = case["n_params"]
    out = m355.nf4_dequant(_to_dev(case["q"], cuda), _to_dev(a, cuda), blocksize=n // a.size)
    torch.cuda.synchronize()
    bits = out.cpu().numpy().view(np.uint16)
    assert hashlib.sha256(bits.tobytes()).hexdigest() == gold["sha256"]
    want = ref.nf4_dequant(case["q"], n, a).view(np.uint16)
    assert np.array_equal(bits, want)


@pytest.mark.parametrize("name", ["gate_up_nested_mean", "down_nested_mean", "down_nested_scalar", "down_nested_elementwise",
                                  "down_absmax_f32", "down_absmax_u8_plain", "down_no_absmax", "gate_up_absmax_single",
                                  "down_absmax_bs128", "down_ragged_plus1", "down_ragged_bs128", "down_ragged_short",
                                  "down_nested_mean_u005", "gate_up_nested_mean_u005"])
def test_dequant_product_path_vs_reference_golden(name, ref, m355, cuda):
    """The scales come from the PRODUCT's host code (plugin.resolve_absmax: the reference's torch expressions, including
    its fp32 mean of inexactly summing nested scales; plugin.absmax_per_block: every other branch of
    patches/streaming_unsloth_bnb.py:197-218 expressed as one scale per 64 weights), the weights from the GPU kernel:
    sha256 of the fp16 bits equals the reference's for every branch of its dequantiser."""
    from gpt_oss_vllm_amd import plugin
    gold = json.load(open(os.path.join(GOLD, "dequant_golden.json")))["cases"][name]
    case = ref.golden_dequant_case(name)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    am = plugin.resolve_absmax(t(case["absmax"]), t(case["nested_absmax"]), t(case["nested_map"])).reshape(-1)
    n = case["n_params"]
    if am.numel() * 64 != n:
        am = plugin.absmax_per_block(am, n)
    out = m355.nf4_dequant(_to_dev(case["q"], cuda), am.float().contiguous().to(cuda), blocksize=64)
    torch.cuda.synchronize()
    bits = out.cpu().numpy().view(np.uint16)
    assert hashlib.sha256(bits.tobytes()).hexdigest() == gold["sha256"]


def test_dequant_rejects_bad_arguments(m355, cuda):
    q = torch.zeros(64, dtype=torch.uint8, device=cuda)
    a = torch.ones(2, dtype=torch.float32, device=cuda)
    with pytest.raises(m355.Mi355LibraryError):
        m355.nf4_dequant(q.cpu(), a)
    with pytest.raises(ValueError):
        m355.nf4_dequant(q, a, blocksize=32)
    out = torch.empty(128, dtype=torch.float16, device=cuda)
    rc = m355.load().m355_nf4_dequant(q.data_ptr(), a.data_ptr(), 128, 12, out.data_ptr(), None)
    assert rc < 0 and b"blocksize" in m355.load().m355_last_error()


def test_repack_layout(ref, m355, cuda):
    rows, cols = 96, 192
    q = ref.fill_bytes(5, rows * cols // 2)
    a = np.arange(rows * cols // 64, dtype=np.float32)
    qt, at = m355.repack_nf4(_to_dev(q, cuda), _to_dev(a, cuda), rows, cols)
    qt, at = qt.cpu().numpy(), at.cpu().numpy()
    NB, P = cols // 64, rows // 32
    qrows = q.reshape(rows, cols // 2)
    for p in range(P):
        for b in range(NB):
            for lane in range(64):
                r, g = lane & 15, lane >> 4
                base = ((p * NB + b) * 64 + lane) * 16
                c0 = (64 * b + 16 * g) // 2
                assert np.array_equal(qt[base:base + 8], qrows[16 * p + r, c0:c0 + 8])
                assert np.array_equal(qt[base + 8:base + 16], qrows[16 * (p + P) + r, c0:c0 + 8])
    am = a.reshape(rows, NB)
    for p in range(P):
        for half in range(2):
            for g in range(4):
                for b in range(NB):
                    for i in range(4):
                        o = (((p * 2 + half) * 4 + g) * NB + b) * 4 + i
                        assert at[o] == am[16 * (p + half * P) + 4 * g + i, b]


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,E,k,bias", [(1, 32, 4, False), (32, 32, 4, True), (7, 128, 4, False), (3, 8, 2, True),
                                        (1100, 32, 4, True), (1030, 128, 4, False), (1024, 40, 3, True)])   # T >= 1024: 16 tokens per workgroup
def test_router_vs_oracle(T, E, k, bias, ref, m355, cuda):
    H = 2880
    x = ref.normal_f16(ref.stream_seed(T, E, 1), (T, H))
    wr = ref.normal_f16(ref.stream_seed(T, E, 2), (E, H), 0.02)
    br = ref.normal_f16(ref.stream_seed(T, E, 3), (E,), 0.1) if bias else None
    ids_r, w_r, lg_r = ref.router_topk(x, wr, br, k)
    ids, w, lg = m355.router_topk(_to_dev(x, cuda), _to_dev(wr, cuda), None if br is None else _to_dev(br, cuda), k,
                                  want_logits=True)
    ids, w, lg = ids.cpu().numpy(), w.cpu().numpy(), lg.cpu().numpy()
    # logits are fp16 roundings of sums taken in different orders: allow one fp16 ulp
    assert np.abs(lg.astype(np.float32) - lg_r.astype(np.float32)).max() <= 2e-3 * max(1.0, np.abs(lg_r.astype(np.float32)).max())
    # selection must agree wherever the oracle's k-th and (k+1)-th logits are separated
    srt = np.sort(lg_r.astype(np.float32), axis=1)[:, ::-1]
    clear = (srt[:, k - 1] - srt[:, k]) > 4e-3 if E > k else np.ones(T, bool)
    gaps = np.diff(-np.sort(-lg_r.astype(np.float32), axis=1)[:, :k], axis=1) if k > 1 else np.ones((T, 1))
    clear &= (np.abs(gaps) > 4e-3).all(axis=1)
    assert clear.any()
    assert np.array_equal(ids[clear], ids_r[clear])
    assert np.abs(w[clear] - w_r[clear]).max() < 5e-3
    np.testing.assert_allclose(w.sum(axis=1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("name", ["fwd_full_T3", "fwd_small_T37"])
def test_topk_softmax_vs_reference_golden(name, ref, m355, cuda):
    c = ref.golden_forward_case(name)
    ids_g = np.load(os.path.join(GOLD, name + ".topk_ids.npy"))
    w_g = np.load(os.path.join(GOLD, name + ".topk_w.npy")).astype(np.float32)
    ids, w = m355.topk_softmax(_to_dev(c["logits"], cuda), c["k"])
    ids, w = ids.cpu().numpy(), w.cpu().numpy()
    lg = c["logits"].astype(np.float32)
    untied = np.array([len(np.unique(np.sort(r)[::-1][:c["k"] + 1])) == min(c["k"] + 1, c["E"]) for r in lg])
    assert np.array_equal(ids[untied], ids_g[untied])
    assert np.abs(w[untied] - w_g[untied]).max() <= 1e-3   # golden weights are fp16


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(("fwd_full_T1", "fwd_full_T3", "fwd_small_T37")))
def test_moe_forward_vs_reference_golden(name, ref, m355, cuda):
    c = ref.golden_forward_case(name)
    H, FF, E, k = c["H"], c["FF"], c["E"], c["k"]
    ids_g = np.load(os.path.join(GOLD, name + ".topk_ids.npy"))
    w_g = np.load(os.path.join(GOLD, name + ".topk_w.npy")).astype(np.float32)
    out_g = np.load(os.path.join(GOLD, name + ".out.npy"))
    tl = tiled_layer(m355, c, E, H, FF, cuda)
    out = m355.moe_nf4_forward(_to_dev(c["x"], cuda), _to_dev(ids_g, cuda), _to_dev(w_g, cuda), *tl,
                               E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), out_g)


@pytest.mark.parametrize("T,E,k", [(1, 32, 4), (32, 32, 4), (40, 4, 2)])
def test_moe_forward_full_size_vs_oracle(T, E, k, ref, m355, cuda):
    """gpt-oss-20b expert shapes; T=40 with 4 experts puts ~20 tokens on an expert (two token tiles)."""
    H = FF = 2880
    sd = ref.stream_seed(T, E, k, 77)
    layer = ref.synth_layer(sd, E, H, FF)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    wr = ref.normal_f16(ref.stream_seed(sd, 2), (E, H), 0.02)
    ids, w, _ = ref.router_topk(x, wr, None, k)
    want = ref.moe_forward(x, ids, w, **layer, E_local=E, H=H, FF=FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    out = m355.moe_nf4_forward(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda), *tl, E_local=E, e_first=0,
                               H=H, FF=FF)
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), want)
    # the layer is linear in the routing weights: doubling w doubles the output (fp32 combine)
    o1 = m355.moe_nf4_forward(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda), *tl, E_local=E, e_first=0,
                              H=H, FF=FF, flags=0)
    o2 = m355.moe_nf4_forward(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(2 * w, cuda), *tl, E_local=E, e_first=0,
                              H=H, FF=FF, flags=0)
    _close(o2.float().cpu().numpy(), 2 * o1.float().cpu().numpy(), rtol=1e-3)


def test_moe_expert_shards_sum_to_whole(ref, m355, cuda):
    """Expert-parallel contract: the outputs of disjoint expert shards add up to the unsharded layer."""
    H, FF, E, k, T = 256, 192, 8, 3, 19
    sd = ref.stream_seed(4242)
    layer = ref.synth_layer(sd, E, H, FF)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    lg = ref.normal_f16(ref.stream_seed(sd, 2), (T, E), 1.5)
    ids, w = m355.topk_softmax(_to_dev(lg, cuda), k)
    w13_q, w13_a, w2_q, w2_a = tiled_layer(m355, layer, E, H, FF, cuda)
    xd = _to_dev(x, cuda)
    whole = m355.moe_nf4_forward(xd, ids, w, w13_q, w13_a, w2_q, w2_a, E_local=E, e_first=0, H=H, FF=FF, flags=0)
    acc = torch.zeros((T, H), dtype=torch.float32, device=cuda)
    for r in range(4):
        sl = slice(2 * r, 2 * r + 2)
        part = m355.moe_nf4_forward(xd, ids, w, w13_q[sl].contiguous(), w13_a[sl].contiguous(), w2_q[sl].contiguous(),
                                    w2_a[sl].contiguous(), E_local=2, e_first=2 * r, H=H, FF=FF, flags=0)
        acc += part.float()
    _close(acc.cpu().numpy(), whole.float().cpu().numpy(), rtol=2e-3)
    want = ref.moe_forward(x, ids.cpu().numpy(), w.cpu().numpy(), **layer, E_local=E, H=H, FF=FF, flags=0)
    _close(whole.cpu().numpy(), want)


@pytest.mark.parametrize("T,E,k", [(150, 8, 2), (5, 160, 4), (64, 32, 4), (65, 32, 4), (3, 8, 2), (4, 160, 4),
                                   (200, 8, 2), (300, 6, 3)])   # the last two: 50 and 150 pairs per expert, the tile kernel
def test_moe_routing_paths_vs_oracle(T, E, k, ref, m355, cuda):
    """Pairs are bucketed inside the gate_up launch up to 256 pairs / 128 local experts and by the routing kernel
    beyond that: both sides of each limit against the oracle, and the stage-by-stage call (route, gate_up, down,
    combine as four calls with skip flags) against the one-call result, bit for bit."""
    H, FF = 256, 192
    sd = ref.stream_seed(T, E, k, 991)
    layer = ref.synth_layer(sd, E, H, FF)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    lg = ref.normal_f16(ref.stream_seed(sd, 2), (T, E), 1.5)
    ids, w = m355.topk_softmax(_to_dev(lg, cuda), k)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    xd = _to_dev(x, cuda)
    out = m355.moe_nf4_forward(xd, ids, w, *tl, E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    want = ref.moe_forward(x, ids.cpu().numpy(), w.cpu().numpy(), **layer, E_local=E, H=H, FF=FF)
    _close(out.cpu().numpy(), want)
    ws = torch.empty(m355.moe_workspace_bytes(T, k, H, FF, E), dtype=torch.uint8, device=cuda)
    staged = torch.empty_like(out)
    F = m355.FLAG_FP16_COMBINE
    every = m355.FLAG_SKIP_ROUTE | m355.FLAG_SKIP_GATE_UP | m355.FLAG_SKIP_DOWN | m355.FLAG_SKIP_COMBINE
    for keep in (m355.FLAG_SKIP_ROUTE, m355.FLAG_SKIP_GATE_UP, m355.FLAG_SKIP_DOWN, m355.FLAG_SKIP_COMBINE):
        m355.moe_nf4_forward(xd, ids, w, *tl, E_local=E, e_first=0, H=H, FF=FF, flags=F | (every & ~keep),
                             workspace=ws, out=staged)
    torch.cuda.synchronize()
    assert torch.equal(staged, out)


def test_moe_empty_batch_and_errors(m355, cuda):
    z = torch.zeros(256, dtype=torch.uint8, device=cuda)
    f = torch.zeros(64, dtype=torch.float32, device=cuda)
    x = torch.zeros((0, 64), dtype=torch.float16, device=cuda)
    ids = torch.zeros((0, 1), dtype=torch.int32, device=cuda)
    w = torch.zeros((0, 1), dtype=torch.float32, device=cuda)
    out = m355.moe_nf4_forward(x, ids, w, z, f, z, f, E_local=1, e_first=0, H=64, FF=64)
    assert out.shape == (0, 64)
    with pytest.raises(m355.Mi355LibraryError, match="unsupported shape"):
        m355.moe_nf4_forward(torch.zeros((1, 60), dtype=torch.float16, device=cuda),
                             torch.zeros((1, 1), dtype=torch.int32, device=cuda),
                             torch.ones((1, 1), dtype=torch.float32, device=cuda), z, f, z, f,
                             E_local=1, e_first=0, H=60, FF=64)


# ------------------------------------------------------------------------------------------------
# 32 sequences fill the chip by themselves, so contexts of three and four rounds of 256 tokens stay in one workgroup
_CTX32 = [1000, 1, 256, 257, 513, 768, 769, 1024, 5, 300, 900, 640] + [37 * i % 1024 + 1 for i in range(20)]


@pytest.mark.parametrize("ctx", [[1], [256, 17, 100], [700, 3, 257, 512], [0, 5], [300, 2, 512, 257], _CTX32, _CTX32[:31] + [0]])
def test_paged_attention_vs_oracle(ctx, ref, m355, cuda):
    """No reference output exists for attention (it lives in FlashAttention-2): parity unpinned, checked
    against the oracle's fp32 softmax attention.  Tolerance 2e-3 * max|out| + 1e-5."""
    B, Hq, Hkv, D, bs = len(ctx), 64, 8, 64, 16
    max_ctx = max(max(ctx), 1)
    max_blocks = (max_ctx + bs - 1) // bs
    nblk = B * max_blocks + 3
    sd = ref.stream_seed(*ctx, 9)
    q = ref.normal_f16(ref.stream_seed(sd, 1), (B, Hq, D))
    kc = ref.normal_f16(ref.stream_seed(sd, 2), (nblk, bs, Hkv, D))
    vc = ref.normal_f16(ref.stream_seed(sd, 3), (nblk, bs, Hkv, D))
    perm = np.argsort(ref.uniform01(ref.stream_seed(sd, 4), nblk)).astype(np.int32)   # scattered physical blocks
    bt = perm[:B * max_blocks].reshape(B, max_blocks).copy()
    cl = np.array(ctx, dtype=np.int32)
    scale = D ** -0.5
    want = ref.paged_attn_decode(q, kc, vc, bt, cl, scale)
    out = m355.paged_attn_decode(_to_dev(q, cuda), _to_dev(kc, cuda), _to_dev(vc, cuda), _to_dev(bt, cuda),
                                 _to_dev(cl, cuda), max_ctx=max_ctx, scale=scale)
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), want)


@pytest.mark.parametrize("B,max_len,window", [(3, 512, 128), (3, 512, 300), (32, 1024, 300), (32, 700, 0), (2, 2000, 700)])
def test_paged_attention_rounds_sinks_window_vs_oracle(B, max_len, window, ref, m355, cuda):
    """Contexts of several 256-token rounds (one workgroup per context up to two rounds, up to four when the batch fills
    the chip, partitions + merge beyond) with sinks and a sliding window that starts inside any of the rounds."""
    Hq, Hkv, D, bs = 64, 8, 64, 16
    sd = ref.stream_seed(B, max_len, window, 4711)
    lens = (ref.uniform01(ref.stream_seed(sd, 5), B) * max_len).astype(np.int32) + 1
    lens[0] = max_len
    max_blocks = (max_len + bs - 1) // bs
    nblk = B * max_blocks + 1
    q = ref.normal_f16(ref.stream_seed(sd, 1), (B, Hq, D))
    kc = ref.normal_f16(ref.stream_seed(sd, 2), (nblk, bs, Hkv, D))
    vc = ref.normal_f16(ref.stream_seed(sd, 3), (nblk, bs, Hkv, D))
    sinks = ref.normal_f16(ref.stream_seed(sd, 6), (Hq,))
    perm = np.argsort(ref.uniform01(ref.stream_seed(sd, 4), nblk)).astype(np.int32)
    bt = perm[:B * max_blocks].reshape(B, max_blocks).copy()
    scale = D ** -0.5
    want = ref.paged_attn_decode_ex(q, kc, vc, bt, lens, scale, sinks=sinks, window=window)
    out = m355.paged_attn_decode(_to_dev(q, cuda), _to_dev(kc, cuda), _to_dev(vc, cuda), _to_dev(bt, cuda),
                                 _to_dev(lens, cuda), max_ctx=max_len, scale=scale, sinks=_to_dev(sinks, cuda), window=window)
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), want)


# ------------------------------------------------------------------------------------------------
def test_ep_pack_unpack_reduce_kernels(ref, m355, cuda):
    H, k, T, N, epr = 256, 3, 7, 4, 2
    sd = ref.stream_seed(555)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    lg = ref.normal_f16(ref.stream_seed(sd, 2), (T, N * epr))
    ids, w, _ = ref.router_topk(lg, np.eye(N * epr, dtype=np.float16), None, k)
    rb = m355.ep_row_bytes(H, k)
    assert rb == (2 * H + 8 * k + 15) // 16 * 16
    send = torch.zeros(N * T * rb, dtype=torch.uint8, device=cuda)
    m355.ep_pack(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda), N, epr, send)
    buf = send.cpu().numpy().reshape(N, T, rb)
    for r in range(N):
        for t in range(T):
            got_ids = buf[r, t, 2 * H:2 * H + 4 * k].copy().view(np.int32)
            if ((ids[t] >= r * epr) & (ids[t] < (r + 1) * epr)).any():
                assert np.array_equal(buf[r, t, :2 * H].copy().view(np.float16).view(np.uint16), x[t].view(np.uint16))
                assert np.array_equal(got_ids, ids[t])
                assert np.array_equal(buf[r, t, 2 * H + 4 * k:2 * H + 8 * k].copy().view(np.float32), w[t])
            else:
                assert (got_ids == -1).all()
    # unpack what rank 1 would receive if every peer sent it this rank's chunk 1
    recv = send.reshape(N, T * rb)[1].repeat(N).contiguous()
    x_all = torch.zeros((N * T, H), dtype=torch.float16, device=cuda)
    ids_all = torch.zeros((N * T, k), dtype=torch.int32, device=cuda)
    w_all = torch.zeros((N * T, k), dtype=torch.float32, device=cuda)
    m355.ep_unpack(recv, N * T, k, H, x_all, ids_all, w_all)
    ids_all, x_all = ids_all.cpu().numpy(), x_all.cpu().numpy()
    for t in range(T):
        hit = ((ids[t] >= epr) & (ids[t] < 2 * epr)).any()
        for s in range(N):
            if hit:
                assert np.array_equal(ids_all[s * T + t], ids[t])
                assert np.array_equal(x_all[s * T + t].view(np.uint16), x[t].view(np.uint16))
            else:
                assert (ids_all[s * T + t] == -1).all()
    parts = ref.normal_f16(ref.stream_seed(sd, 3), (N, T, H))
    out = torch.empty((T, H), dtype=torch.float16, device=cuda)
    m355.ep_reduce(_to_dev(parts, cuda), N, T, H, out)
    want = parts.astype(np.float32).sum(0).astype(np.float16)
    assert np.abs(out.cpu().numpy().astype(np.float32) - want.astype(np.float32)).max() <= 2e-3 * np.abs(want).max()


def test_plugin_path_end_to_end(ref, m355, cuda, tmp_path):
    """config -> method -> create_weights -> weight_loader (names as in the checkpoint, double-quantised state)
    -> process_weights_after_loading -> apply, from a safetensors shard in the Unsloth key schema."""
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint, plugin
    H, FF, E, k, T = 256, 192, 4, 2, 9
    tensors, layer_ref = {}, dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
    for e in range(E):
        for proj, n, tag in (("gate_up", 2 * FF * H, "w13"), ("down", H * FF, "w2")):
            q, a8, na, nm = ref.synth_expert_tensor(ref.stream_seed(9, e, n), n)
            key = checkpoint.expert_weight_key(2, e, proj)
            tensors[key] = torch.from_numpy(q.copy()).reshape(-1, 1)
            tensors[key + ".absmax"] = torch.from_numpy(a8.copy())
            tensors[key + ".nested_absmax"] = torch.from_numpy(na.copy())
            tensors[key + ".nested_quant_map"] = torch.from_numpy(nm.copy())
            layer_ref[tag + "_q"].append(q)
            layer_ref[tag + "_absmax"].append(ref.resolve_absmax(a8, nm, na))
    layer_ref = {n: np.stack(v) for n, v in layer_ref.items()}
    shard = str(tmp_path / "model-00001.safetensors")
    save_file(tensors, shard)
    cfg = plugin.Mi355Nf4MoEConfig()
    method = cfg.get_quant_method(None, "model.layers.2.mlp.experts")
    layer = torch.nn.Module()
    method.create_weights(layer, E, H, FF, torch.float16, device=cuda)
    checkpoint.load_layer(method, layer, [shard], 2)
    # the same layer filled the way vLLM's model loader fills it: param.weight_loader(param, tensor, name, shard_id, expert_id)
    layer_v = torch.nn.Module()
    method.create_weights(layer_v, E, H, FF, torch.float16, device=cuda)
    for name, tensor in tensors.items():
        proj = "w13_qweight" if "gate_up_projs" in name else "w2_qweight"
        param = getattr(layer_v, proj)
        assert param.weight_loader(param, tensor, name, shard_id="w1", expert_id=int(name.split(".")[6]), return_success=True) is True
    method.process_weights_after_loading(layer_v)
    assert torch.equal(layer_v.w13_qweight, layer.w13_qweight) and torch.equal(layer_v.w2_qweight, layer.w2_qweight)
    assert torch.equal(layer_v._m355_codes["w13"], layer._m355_codes["w13"])
    x = ref.normal_f16(ref.stream_seed(9, 100), (T, H))
    lg = ref.normal_f16(ref.stream_seed(9, 101), (T, E), 1.5)
    out = method.apply(layer, _to_dev(x, cuda), _to_dev(lg, cuda), top_k=k, renormalize=True)
    ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    want = ref.moe_forward(x, ids, w, **layer_ref, E_local=E, H=H, FF=FF)
    _close(out.cpu().numpy(), want)
    # these tensors resolve to code byte + table, so the layer keeps the absmax as bytes; fp32 storage gives the same bits
    assert layer._m355_codes is not None
    method32 = plugin.Mi355Nf4MoEConfig(absmax_storage="fp32").get_quant_method(None, "model.layers.2.mlp.experts")
    layer32 = torch.nn.Module()
    method32.create_weights(layer32, E, H, FF, torch.float16, device=cuda)
    checkpoint.load_layer(method32, layer32, [shard], 2)
    assert layer32._m355_codes is None
    out32 = method32.apply(layer32, _to_dev(x, cuda), _to_dev(lg, cuda), top_k=k, renormalize=True)
    assert torch.equal(out32, out)
    # bf16 activations are accepted (converted at the boundary) and come back as bf16
    out_bf = method.apply(layer, _to_dev(x, cuda).to(torch.bfloat16), _to_dev(lg, cuda), top_k=k, renormalize=True)
    assert out_bf.dtype == torch.bfloat16
    with pytest.raises(NotImplementedError):
        method.apply(layer, _to_dev(x, cuda), _to_dev(lg, cuda), top_k=k, renormalize=False)


def test_expert_parallel_class_single_rank(ref, m355, cuda):
    from gpt_oss_vllm_amd import ep
    H, FF, E, k, T = 256, 192, 6, 2, 5
    layer = ref.synth_layer(321, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    x = ref.normal_f16(322, (T, H))
    lg = ref.normal_f16(323, (T, E))
    ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    moe = ep.ExpertParallelMoE(tl, num_experts=E, top_k=k, hidden=H, ffn=FF, max_tokens=T)
    out = moe.forward(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda))
    want = ref.moe_forward(x, ids, w, **layer, E_local=E, H=H, FF=FF)
    _close(out.cpu().numpy(), want)
    with pytest.raises(ValueError):
        moe.forward(_to_dev(x[:2], cuda), _to_dev(ids[:2], cuda), _to_dev(w[:2], cuda))


@pytest.mark.parametrize("T,E,k,H,FF,act", [(32, 32, 4, 2880, 2880, "silu"), (1, 32, 4, 2880, 2880, "silu"),
                                            (19, 8, 3, 256, 192, "oai"), (70, 4, 2, 256, 192, "silu")])
def test_moe_forward_byte_absmax_equals_fp32_absmax(T, E, k, H, FF, act, ref, m355, cuda):
    """absmax kept as the checkpoint's code byte + a 256-entry table per expert (m355_moe_nf4_forward_q8) gives the
    bits of the fp32-absmax entry point when table[code] is that fp32 absmax (the reference's
    nested_quant_map[code] * scalar, patches/streaming_unsloth_bnb.py:150-171)."""
    sd = ref.stream_seed(T, E, k, 4321)
    nm = ref.nested_quant_map()
    q13 = np.stack([ref.fill_bytes(ref.stream_seed(sd, e, 1), FF * H) for e in range(E)])
    q2 = np.stack([ref.fill_bytes(ref.stream_seed(sd, e, 2), H * FF // 2) for e in range(E)])
    c13 = np.stack([ref.fill_bytes(ref.stream_seed(sd, e, 3), 2 * FF * H // 64) for e in range(E)])
    c2 = np.stack([ref.fill_bytes(ref.stream_seed(sd, e, 4), H * FF // 64) for e in range(E)])
    scale = (0.01 + 0.02 * ref.uniform01(ref.stream_seed(sd, 5), 2 * E)).astype(np.float32)
    t13 = (nm[None, :] * scale[:E, None]).astype(np.float32)
    t2 = (nm[None, :] * scale[E:, None]).astype(np.float32)
    layer = dict(w13_q=q13, w13_absmax=np.stack([t13[e][c13[e]] for e in range(E)]),
                 w2_q=q2, w2_absmax=np.stack([t2[e][c2[e]] for e in range(E)]))
    w13_q, w13_a, w2_q, w2_a = tiled_layer(m355, layer, E, H, FF, cuda)
    c13_t = torch.stack([m355.repack_absmax_u8(_to_dev(c13[e], cuda), 2 * FF, H) for e in range(E)])
    c2_t = torch.stack([m355.repack_absmax_u8(_to_dev(c2[e], cuda), H, FF) for e in range(E)])
    x = _to_dev(ref.normal_f16(ref.stream_seed(sd, 6), (T, H)), cuda)
    lg = ref.normal_f16(ref.stream_seed(sd, 7), (T, E), 1.5)
    ids, w = m355.topk_softmax(_to_dev(lg, cuda), k)
    kw = dict(E_local=E, e_first=0, H=H, FF=FF)
    if act == "oai":
        kw.update(act_mode=m355.ACT_SWIGLU_OAI,
                  b13=_to_dev(ref.normal_f16(ref.stream_seed(sd, 8), (E, 2 * FF), 0.1), cuda),
                  b2=_to_dev(ref.normal_f16(ref.stream_seed(sd, 9), (E, H), 0.1), cuda))
    want = m355.moe_nf4_forward(x, ids, w, w13_q, w13_a, w2_q, w2_a, **kw)
    got = m355.moe_nf4_forward(x, ids, w, w13_q, (c13_t, _to_dev(t13, cuda)), w2_q, (c2_t, _to_dev(t2, cuda)), **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(want.float()).all() and want.float().abs().max() > 0
    assert torch.equal(got, want)
    if T <= 32 and H == 2880:
        ref_out = ref.moe_forward(x.cpu().numpy(), ids.cpu().numpy(), w.cpu().numpy(), **layer, E_local=E, H=H, FF=FF)
        _close(got.cpu().numpy(), ref_out)


# ------------------------------------------------------------------------------------------------
# The kernel for many tokens per expert (256-token tiles, M355_FLAG_FORCE_TILE) against the oracle and against the
# kernel for few tokens per expert on the same inputs.
@pytest.mark.parametrize("T,E,k,H,FF,act", [(40, 4, 2, 256, 192, "silu"), (300, 4, 2, 256, 192, "oai"),
                                            (700, 2, 2, 256, 192, "silu"), (33, 40, 4, 256, 192, "silu"),
                                            (96, 4, 2, 2880, 2880, "silu"), (70, 4, 2, 2880, 2880, "oai")])
def test_moe_tile_kernel_vs_oracle_and_stream_kernel(T, E, k, H, FF, act, ref, m355, cuda):
    sd = ref.stream_seed(T, E, k, 2025)
    layer = ref.synth_layer(sd, E, H, FF)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    lg = ref.normal_f16(ref.stream_seed(sd, 2), (T, E), 1.5)
    ids, w = m355.topk_softmax(_to_dev(lg, cuda), k)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    xd = _to_dev(x, cuda)
    kw = dict(E_local=E, e_first=0, H=H, FF=FF)
    b13 = b2 = None
    if act == "oai":
        b13 = ref.normal_f16(ref.stream_seed(sd, 8), (E, 2 * FF), 0.1)
        b2 = ref.normal_f16(ref.stream_seed(sd, 9), (E, H), 0.1)
        kw.update(act_mode=m355.ACT_SWIGLU_OAI, b13=_to_dev(b13, cuda), b2=_to_dev(b2, cuda))
    F = m355.FLAG_FP16_COMBINE
    tile = m355.moe_nf4_forward(xd, ids, w, *tl, flags=F | m355.FLAG_FORCE_TILE, **kw)
    stream = m355.moe_nf4_forward(xd, ids, w, *tl, flags=F | m355.FLAG_FORCE_STREAM, **kw)
    # the streaming kernel's form with token tiles of 32 (ragged last tiles, experts with 0 .. several tiles)
    wide = m355.moe_nf4_forward(xd, ids, w, *tl, flags=F | m355.FLAG_FORCE_STREAM | m355.FLAG_STREAM_WIDE, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(tile.float()).all() and torch.isfinite(wide.float()).all()
    _close(tile.cpu().numpy(), stream.cpu().numpy())
    _close(wide.cpu().numpy(), stream.cpu().numpy())
    ids_h, w_h = ids.cpu().numpy(), w.cpu().numpy()
    if act == "oai":
        w13 = np.stack([ref.nf4_dequant(layer["w13_q"][e], 2 * FF * H, layer["w13_absmax"][e]).reshape(2 * FF, H) for e in range(E)])
        w2 = np.stack([ref.nf4_dequant(layer["w2_q"][e], H * FF, layer["w2_absmax"][e]).reshape(H, FF) for e in range(E)])
        # the kernels' row order is gate rows then up rows; the gpt-oss oracle takes them interleaved
        w13_il = np.empty_like(w13)
        w13_il[:, 0::2], w13_il[:, 1::2] = w13[:, :FF], w13[:, FF:]
        b13_il = np.empty_like(b13)
        b13_il[:, 0::2], b13_il[:, 1::2] = b13[:, :FF], b13[:, FF:]
        want = ref.moe_forward_oai(x, ids_h, w_h, w13_il, w2, b13_il, b2)
        _close(tile.cpu().numpy(), want, rtol=5e-3)
        _close(wide.cpu().numpy(), want, rtol=5e-3)
    else:
        want = ref.moe_forward(x, ids_h, w_h, **layer, E_local=E, H=H, FF=FF)
        _close(tile.cpu().numpy(), want)
        _close(wide.cpu().numpy(), want)


def test_moe_tile_kernel_prefill_size(ref, m355, cuda):
    """4096 tokens through 32 full-size experts (about 512 per expert: two 256-token tiles each, the kernel is chosen
    from the shape, no flag): sampled tokens against the oracle (a token's output depends on no other token),
    byte absmax against fp32 absmax bit for bit, and linearity in the routing weights."""
    T, E, k, H, FF = 4096, 32, 4, 2880, 2880
    g = torch.Generator(device=cuda)
    g.manual_seed(99)
    nm = ref.nested_quant_map()
    q13 = torch.randint(0, 256, (E, FF * H), dtype=torch.uint8, device=cuda, generator=g)
    q2 = torch.randint(0, 256, (E, H * FF // 2), dtype=torch.uint8, device=cuda, generator=g)
    c13 = torch.randint(0, 256, (E, 2 * FF * H // 64), dtype=torch.uint8, device=cuda, generator=g)
    c2 = torch.randint(0, 256, (E, H * FF // 64), dtype=torch.uint8, device=cuda, generator=g)
    t13 = (torch.from_numpy(nm)[None, :] * (0.01 + 0.02 * torch.rand((E, 1)))).float().to(cuda)
    t2 = (torch.from_numpy(nm)[None, :] * (0.01 + 0.02 * torch.rand((E, 1)))).float().to(cuda)
    a13, a2 = torch.gather(t13, 1, c13.long()), torch.gather(t2, 1, c2.long())
    x = torch.randn((T, H), device=cuda, generator=g).half()
    lg = (torch.randn((T, E), device=cuda, generator=g) * 1.5).half()
    ids, w = m355.topk_softmax(lg, k)
    kw = dict(E_local=E, e_first=0, H=H, FF=FF)
    out = m355.moe_nf4_forward(x, ids, w, q13, a13, q2, a2, flags=0, **kw)
    out8 = m355.moe_nf4_forward(x, ids, w, q13, (c13, t13), q2, (c2, t2), flags=0, **kw)
    out2 = m355.moe_nf4_forward(x, ids, 2 * w, q13, a13, q2, a2, flags=0, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, out8)
    _close(out2.float().cpu().numpy(), 2 * out.float().cpu().numpy(), rtol=1e-3)
    # oracle on 6 sampled tokens; the tensors here are already in tile order, so undo it on the host for the experts used
    sample = [0, 255, 256, 1777, 4000, 4095]
    ids_h, w_h = ids.cpu().numpy(), w.cpu().numpy()
    used = sorted({int(e) for t in sample for e in ids_h[t]})
    remap = {e: i for i, e in enumerate(used)}

    def untile(qt, at, rows, cols):
        """inverse of m355_repack_nf4 for one expert (host)."""
        NB, P = cols // 64, rows // 32
        qt = qt.reshape(P, NB, 4, 16, 2, 8)                       # [p][b][g][r][half][8 bytes]
        q = np.empty((rows, cols // 2), np.uint8).reshape(2, P, 16, NB, 4, 8)   # [half][p][r][b][g][8]
        q[:] = qt.transpose(4, 0, 3, 1, 2, 5)
        at = at.reshape(P, 2, 4, NB, 4)                            # [p][half][g4][b][i]
        a = np.empty((2, P, 4, 4, NB), np.float32)                 # [half][p][g4][i][b]
        a[:] = at.transpose(1, 0, 2, 4, 3)
        return q.reshape(-1), a.reshape(-1)
    sub = dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
    for e in used:
        q, a = untile(q13[e].cpu().numpy(), a13[e].cpu().numpy(), 2 * FF, H)
        sub["w13_q"].append(q), sub["w13_absmax"].append(a)
        q, a = untile(q2[e].cpu().numpy(), a2[e].cpu().numpy(), H, FF)
        sub["w2_q"].append(q), sub["w2_absmax"].append(a)
    sub = {n: np.stack(v) for n, v in sub.items()}
    ids_s = np.vectorize(remap.get)(ids_h[sample]).astype(np.int32)
    want = ref.moe_forward(x[sample].cpu().numpy(), ids_s, w_h[sample], **sub, E_local=len(used), H=H, FF=FF, flags=0)
    _close(out[sample].cpu().numpy(), want)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,E,k,bias,H,FF", [(1, 32, 4, False, 2880, 2880), (3, 32, 4, True, 2880, 2880),
                                             (8, 128, 4, False, 256, 192), (9, 32, 4, True, 256, 192),
                                             (2, 256, 5, True, 256, 192), (40, 8, 2, False, 256, 192)]
                         # every batch size of the fused-router form on the gpt-oss-20b routing shape (32 experts, top-4):
                         # the router's picks and the per-expert pair masks share one LDS buffer in that launch
                         + [(T, 32, 4, T % 2 == 0, 256, 192) for T in range(2, 9)] + [(5, 32, 4, False, 2880, 2880)]
                         # logits as a launch of their own, top-k inside gate_up (up to 24 tokens, 256 experts), and around the limits
                         + [(24, 32, 4, False, 2880, 2880), (25, 32, 4, True, 256, 192), (24, 64, 4, False, 256, 192),
                            (20, 48, 3, True, 256, 192), (17, 5, 5, False, 256, 192), (24, 65, 4, False, 256, 192),
                            (13, 128, 4, True, 256, 192), (24, 256, 8, False, 256, 192), (3, 257, 4, False, 256, 192),
                            (32, 32, 4, False, 2880, 2880)])
def test_router_and_experts_in_one_call(T, E, k, bias, H, FF, ref, m355, cuda):
    """m355_moe_nf4_forward_routed against router_topk followed by the forward: identical ids, weights and output
    (up to 24 tokens and 256 experts the logits are a launch of their own and the gate_up launch picks the top-k with the
    router kernel's code, from a lane per expert up to 64 experts and from a row of LDS per wave beyond; past those limits
    the call launches the router kernel)."""
    sd = ref.stream_seed(T, E, k, 808)
    layer = ref.synth_layer(sd, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    x = _to_dev(ref.normal_f16(ref.stream_seed(sd, 1), (T, H)), cuda)
    wr = _to_dev(ref.normal_f16(ref.stream_seed(sd, 2), (E, H), 0.05), cuda)
    br = _to_dev(ref.normal_f16(ref.stream_seed(sd, 3), (E,), 0.1), cuda) if bias else None
    ids0, w0 = m355.router_topk(x, wr, br, k)
    want = m355.moe_nf4_forward(x, ids0, w0, *tl, E_local=E, e_first=0, H=H, FF=FF)
    out, ids, w = m355.moe_nf4_forward_routed(x, wr, br, k, *tl, E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    assert torch.equal(ids, ids0) and torch.equal(w, w0)
    assert torch.equal(out, want)
    # the router kept as a launch of its own: the same again
    out2, ids2, w2 = m355.moe_nf4_forward_routed(x, wr, br, k, *tl, E_local=E, e_first=0, H=H, FF=FF,
                                                 flags=m355.FLAG_FP16_COMBINE | m355.FLAG_ROUTER_LAUNCH)
    torch.cuda.synchronize()
    assert torch.equal(ids2, ids0) and torch.equal(w2, w0) and torch.equal(out2, want)
    # gate_up alone, then again on the logits the first call left in the workspace (what the bench's roofline leg times)
    ws = torch.empty(max(256, m355.moe_workspace_bytes(T, k, H, FF, E)), dtype=torch.uint8, device=cuda)
    part = m355.FLAG_FP16_COMBINE | m355.FLAG_SKIP_DOWN | m355.FLAG_SKIP_COMBINE
    _, ids3, w3 = m355.moe_nf4_forward_routed(x, wr, br, k, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, flags=part)
    _, ids4, w4 = m355.moe_nf4_forward_routed(x, wr, br, k, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws,
                                              flags=part | m355.FLAG_LOGITS_READY)
    torch.cuda.synchronize()
    assert torch.equal(ids3, ids0) and torch.equal(w3, w0) and torch.equal(ids4, ids0) and torch.equal(w4, w0)
    # a shard (local experts 2..) sees the same routing and adds up with the rest as the separate calls do
    if E >= 8:
        sl = slice(2, E)
        shard = tuple(t[sl].contiguous() for t in tl)
        o_sh, ids_sh, _ = m355.moe_nf4_forward_routed(x, wr, br, k, *shard, E_local=E - 2, e_first=2, H=H, FF=FF)
        want_sh = m355.moe_nf4_forward(x, ids0, w0, *shard, E_local=E - 2, e_first=2, H=H, FF=FF)
        torch.cuda.synchronize()
        assert torch.equal(ids_sh, ids0) and torch.equal(o_sh, want_sh)


def test_timed_launch_reports_the_kernels_own_duration(ref, m355, cuda):
    """m355_time_next_launch / m355_timed_launch_us: the next streaming-kernel launch carries its own start / stop events;
    the duration is positive, below what an event pair around the call measures, and asking without a launch is an error."""
    T, E, k, H, FF = 4, 8, 2, 256, 192
    layer = ref.synth_layer(4242, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    x = _to_dev(ref.normal_f16(4243, (T, H)), cuda)
    lg = ref.normal_f16(4244, (T, E), 1.5)
    ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    idd, wd = _to_dev(ids, cuda), _to_dev(w, cuda)
    ws = torch.empty(max(256, m355.moe_workspace_bytes(T, k, H, FF, E)), dtype=torch.uint8, device=cuda)
    only_gate_up = m355.FLAG_FP16_COMBINE | m355.FLAG_SKIP_DOWN | m355.FLAG_SKIP_COMBINE
    m355.moe_nf4_forward(x, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, flags=only_gate_up)
    torch.cuda.synchronize()
    m355.time_next_launch()
    with pytest.raises(m355.Mi355LibraryError):
        m355.timed_launch_us()                      # armed, but nothing launched yet
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    m355.moe_nf4_forward(x, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, flags=only_gate_up)
    b.record()
    own = m355.timed_launch_us()
    b.synchronize()
    assert 0.5 < own < a.elapsed_time(b) * 1e3 + 1.0
    # not armed any more: the next call launches as usual and the figure stays that of the timed launch
    m355.moe_nf4_forward(x, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, flags=only_gate_up)
    torch.cuda.synchronize()
    assert m355.timed_launch_us() == own


@pytest.mark.parametrize("T,E,k,H,FF", [(1, 32, 4, 2880, 2880), (5, 32, 4, 256, 192), (24, 32, 4, 256, 192), (25, 32, 4, 256, 192),
                                        (7, 128, 4, 256, 192), (40, 8, 2, 256, 192), (3, 300, 6, 256, 192), (300, 32, 4, 256, 192)])
def test_logits_and_experts_in_one_call(T, E, k, H, FF, ref, m355, cuda):
    """m355_moe_nf4_forward_logits (what the plug-in's apply() calls with FusedMoE's router_logits) against
    m355_topk_softmax followed by the forward: identical ids, weights and output, on a full layer and on a shard."""
    sd = ref.stream_seed(T, E, k, 909)
    layer = ref.synth_layer(sd, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    x = _to_dev(ref.normal_f16(ref.stream_seed(sd, 1), (T, H)), cuda)
    lg_h = ref.normal_f16(ref.stream_seed(sd, 2), (T, E), 1.5)
    lg_h[0, : min(E, 6)] = lg_h[0, 0]          # ties: the lower expert id wins
    lg = _to_dev(lg_h, cuda)
    ids0, w0 = m355.topk_softmax(lg, k)
    want = m355.moe_nf4_forward(x, ids0, w0, *tl, E_local=E, e_first=0, H=H, FF=FF)
    out, ids, w = m355.moe_nf4_forward_logits(x, lg, k, *tl, E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    assert torch.equal(ids, ids0) and torch.equal(w, w0) and torch.equal(out, want)
    if E >= 8:
        shard = tuple(t[2:E].contiguous() for t in tl)
        o_sh, ids_sh, _ = m355.moe_nf4_forward_logits(x, lg, k, *shard, E_local=E - 2, e_first=2, H=H, FF=FF)
        want_sh = m355.moe_nf4_forward(x, ids0, w0, *shard, E_local=E - 2, e_first=2, H=H, FF=FF)
        torch.cuda.synchronize()
        assert torch.equal(ids_sh, ids0) and torch.equal(o_sh, want_sh)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lens,window,use_sinks", [([1], 0, False), ([37], 0, False), ([200, 3, 65], 0, True),
                                                   ([130, 64], 48, True), ([300], 128, False), ([0, 5, 33], 0, False)])
def test_attention_prefill_vs_oracle(lens, window, use_sinks, ref, m355, cuda):
    """Prompt attention lives in flash-attn in the reference (no fixture): parity unpinned, checked against the
    oracle's fp32 causal attention.  q, k, v are slices of one fused [tokens][(Hq + 2 Hkv) D] tensor, as a model's qkv
    projection leaves them.  Tolerance 2e-3 * max|out| + 1e-5 (probabilities are rounded to fp16 for the second product)."""
    Hq, Hkv, D = 64, 8, 64
    T = sum(lens)
    sd = ref.stream_seed(*lens, window, 31)
    qkv = ref.normal_f16(ref.stream_seed(sd, 1), (max(T, 1), (Hq + 2 * Hkv) * D))[:T]
    sinks = ref.normal_f16(ref.stream_seed(sd, 2), (Hq,), 1.0) if use_sinks else None
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    qkv_d = _to_dev(qkv, cuda) if T else torch.empty((0, (Hq + 2 * Hkv) * D), dtype=torch.float16, device=cuda)
    q = qkv_d[:, :Hq * D].view(T, Hq, D)
    k = qkv_d[:, Hq * D:(Hq + Hkv) * D].view(T, Hkv, D)
    v = qkv_d[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    scale = D ** -0.5
    out = m355.attn_prefill(q, k, v, _to_dev(starts, cuda), max(lens), scale=scale,
                            sinks=None if sinks is None else _to_dev(sinks, cuda), window=window)
    torch.cuda.synchronize()
    qn = qkv[:, :Hq * D].reshape(T, Hq, D)
    kn = qkv[:, Hq * D:(Hq + Hkv) * D].reshape(T, Hkv, D)
    vn = qkv[:, (Hq + Hkv) * D:].reshape(T, Hkv, D)
    want = ref.attn_prefill(qn, kn, vn, starts, scale, sinks, window)
    _close(out.cpu().numpy(), want)


def test_attention_prefill_last_token_equals_decode(ref, m355, cuda):
    """The last query of a prompt through the prefill kernel = the decode kernel's answer for the same cache."""
    Hq, Hkv, D, bs, L = 64, 8, 64, 16, 150
    sd = ref.stream_seed(77, 150)
    q = ref.normal_f16(ref.stream_seed(sd, 1), (L, Hq, D))
    k = ref.normal_f16(ref.stream_seed(sd, 2), (L, Hkv, D))
    v = ref.normal_f16(ref.stream_seed(sd, 3), (L, Hkv, D))
    nblk = (L + bs - 1) // bs
    kc = np.zeros((nblk, bs, Hkv, D), np.float16)
    vc = np.zeros((nblk, bs, Hkv, D), np.float16)
    kc.reshape(-1, Hkv, D)[:L], vc.reshape(-1, Hkv, D)[:L] = k, v
    starts = _to_dev(np.array([0, L], np.int32), cuda)
    pre = m355.attn_prefill(_to_dev(q, cuda), _to_dev(k, cuda), _to_dev(v, cuda), starts, L, scale=D ** -0.5)
    dec = m355.paged_attn_decode(_to_dev(q[L - 1:L], cuda), _to_dev(kc, cuda), _to_dev(vc, cuda),
                                 _to_dev(np.arange(nblk, dtype=np.int32).reshape(1, nblk), cuda),
                                 _to_dev(np.array([L], np.int32), cuda), max_ctx=L, scale=D ** -0.5)
    torch.cuda.synchronize()
    _close(pre[L - 1].cpu().numpy(), dec[0].cpu().numpy())


# ------------------------------------------------------------------------------------------------
# Prepared calls (arguments checked and converted once) repeat the wrappers' library calls exactly.
@pytest.mark.gpu
def test_prepared_calls_repeat_the_wrappers(ref, m355, cuda):
    T, E, k, H = 8, 32, 4, 2880
    x = _to_dev(ref.normal_f16(ref.stream_seed(77, 1), (T, H)), cuda)
    wr = _to_dev(ref.normal_f16(ref.stream_seed(77, 2), (E, H), 0.02), cuda)
    ids = torch.empty((T, k), dtype=torch.int32, device=cuda)
    w = torch.empty((T, k), dtype=torch.float32, device=cuda)
    logits_a = m355.router_topk(x, wr, None, k, want_logits=True)[2].clone()
    want_ids, want_w = m355.router_topk(x, wr, None, k)
    prepared = m355.prepare(lambda: m355.router_topk(x, wr, None, k, ids=ids, w=w))
    ids.fill_(-1)
    w.zero_()
    prepared()
    torch.cuda.synchronize()
    assert torch.equal(ids, want_ids) and torch.equal(w, want_w)
    assert torch.isfinite(logits_a.float()).all()
    # a failing call inside prepare() raises as the wrapper does, and the library is back in place afterwards
    with pytest.raises(m355.Mi355LibraryError):
        m355.prepare(lambda: m355.router_topk(x, wr, None, 64, ids=torch.empty((T, 64), dtype=torch.int32, device=cuda),
                                              w=torch.empty((T, 64), dtype=torch.float32, device=cuda)))
    assert m355.router_topk(x, wr, None, k)[0].shape == (T, k)


# ------------------------------------------------------------------------------------------------
def _small_plugin_layer(ref, cuda, E, H, FF, max_tokens=0, drop=()):
    """A FusedMoE-like layer filled through the plug-in from synthetic checkpoint tensors; `drop` lists (expert, proj)
    matrices that are withheld.  Returns (method, layer, oracle tensors)."""
    from gpt_oss_vllm_amd import checkpoint, plugin
    method = plugin.Mi355Nf4MoEConfig(max_tokens=max_tokens).get_quant_method(None, "model.layers.0.mlp.experts")
    layer = torch.nn.Module()
    method.create_weights(layer, E, H, FF, torch.float16, device=cuda)
    lay = dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
    for e in range(E):
        for proj, n, tag in (("gate_up", 2 * FF * H, "w13"), ("down", H * FF, "w2")):
            q, a8, na, nm = ref.synth_expert_tensor(ref.stream_seed(19, e, n), n)
            am = ref.resolve_absmax(a8, nm, na)
            if (e, proj) in drop:
                q, am = np.full_like(q, 0x77), np.ones_like(am)
            else:
                key = checkpoint.expert_weight_key(0, e, proj)
                for sfx, val in (("", q.reshape(-1, 1)), (".absmax", a8), (".nested_absmax", na), (".nested_quant_map", nm)):
                    method.weight_loader(layer, None, torch.from_numpy(val.copy()), key + sfx)
            lay[tag + "_q"].append(q), lay[tag + "_absmax"].append(am)
    return method, layer, {n: np.stack(v) for n, v in lay.items()}


def test_plugin_apply_under_graph_capture_with_expert_map(ref, m355, cuda):
    """FusedMoE.apply with vLLM's expert_map (expert parallelism) inside a HIP graph: the shard's first expert is read
    back once (eagerly, or by set_expert_map), the workspace is the layer's one buffer whatever the batch size, and
    nothing in apply() synchronises or allocates per call, so the capture succeeds and replays give the eager bits."""
    E_loc, E_glob, H, FF, k = 4, 8, 256, 192, 2
    method, layer, lay = _small_plugin_layer(ref, cuda, E_loc, H, FF, max_tokens=64)
    method.process_weights_after_loading(layer)
    emap = torch.full((E_glob,), -1, dtype=torch.int32, device=cuda)
    emap[2:6] = torch.arange(4, dtype=torch.int32, device=cuda)
    with pytest.raises(ValueError):
        method.set_expert_map(layer, torch.tensor([0, -1, 1, 2, 3, -1, -1, -1], dtype=torch.int32, device=cuda))
    assert method.set_expert_map(layer, emap) == 2
    ws0 = layer._m355_ws
    outs = {}
    for T in (9, 33, 5):      # continuous batching: the batch size changes from step to step
        x = _to_dev(ref.normal_f16(ref.stream_seed(29, T, 1), (T, H)), cuda)
        lg = _to_dev(ref.normal_f16(ref.stream_seed(29, T, 2), (T, E_glob), 1.5), cuda)
        eager = method.apply(layer, x, lg, top_k=k, renormalize=True, expert_map=emap, global_num_experts=E_glob)
        assert layer._m355_ws is ws0                      # no reallocation
        ids, w, _ = ref.router_topk(lg.cpu().numpy(), np.eye(E_glob, dtype=np.float16), None, k)
        want = ref.moe_forward(x.cpu().numpy(), ids, w, **lay, E_local=E_loc, e_first=2, H=H, FF=FF)
        _close(eager.cpu().numpy(), want)
        outs[T] = (x, lg, eager)
    x, lg, eager = outs[33]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        method.apply(layer, x, lg, top_k=k, renormalize=True, expert_map=emap, global_num_experts=E_glob)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = method.apply(layer, x, lg, top_k=k, renormalize=True, expert_map=emap, global_num_experts=E_glob)
    captured.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)
    # a map never seen before, first met during capture: refused with a clear message instead of a device-to-host read
    other = emap.clone()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="set_expert_map"):
        with torch.cuda.graph(g2):
            method.apply(layer, x, lg, top_k=k, renormalize=True, expert_map=other, global_num_experts=E_glob)


def test_plugin_missing_expert_as_zeros(ref, m355, cuda, tmp_path):
    """checkpoint.load_layer(missing="zeros"): an expert whose gate_up matrix no shard holds contributes nothing, as with
    the zeros the reference substitutes (patches/streaming_unsloth_bnb.py:261-264)."""
    from safetensors.torch import save_file
    from gpt_oss_vllm_amd import checkpoint, plugin
    E, H, FF, k, T = 4, 256, 192, 2, 11
    tensors, lay = {}, dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
    for e in range(E):
        for proj, n, tag in (("gate_up", 2 * FF * H, "w13"), ("down", H * FF, "w2")):
            q, a8, na, nm = ref.synth_expert_tensor(ref.stream_seed(39, e, n), n)
            am = ref.resolve_absmax(a8, nm, na)
            if (e, proj) == (1, "gate_up"):
                q, am = np.full_like(q, 0x77), np.ones_like(am)
            else:
                key = checkpoint.expert_weight_key(5, e, proj)
                tensors.update({key: torch.from_numpy(q.copy()).reshape(-1, 1), key + ".absmax": torch.from_numpy(a8.copy()),
                                key + ".nested_absmax": torch.from_numpy(na.copy()),
                                key + ".nested_quant_map": torch.from_numpy(nm.copy())})
            lay[tag + "_q"].append(q), lay[tag + "_absmax"].append(am)
    lay = {n: np.stack(v) for n, v in lay.items()}
    shard = str(tmp_path / "m.safetensors")
    save_file(tensors, shard)
    method = plugin.Mi355Nf4MoEConfig().get_quant_method(None, "model.layers.5.mlp.experts")
    layer = torch.nn.Module()
    method.create_weights(layer, E, H, FF, torch.float16, device=cuda)
    with pytest.raises(KeyError):
        checkpoint.load_layer(method, layer, [shard], 5)
    layer = torch.nn.Module()
    method.create_weights(layer, E, H, FF, torch.float16, device=cuda)
    checkpoint.load_layer(method, layer, [shard], 5, missing="zeros")
    x = ref.normal_f16(ref.stream_seed(39, 100), (T, H))
    lg = ref.normal_f16(ref.stream_seed(39, 101), (T, E), 1.5)
    out = method.apply(layer, _to_dev(x, cuda), _to_dev(lg, cuda), top_k=k, renormalize=True)
    ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    assert (ids == 1).any()
    want = ref.moe_forward(x, ids, w, **lay, E_local=E, H=H, FF=FF)
    _close(out.cpu().numpy(), want)


def test_init_before_first_call_allows_immediate_capture(ref, m355, cuda):
    m355.init(0)
    m355.init()      # idempotent
    with pytest.raises(m355.Mi355LibraryError):
        m355.init(10 ** 6)
