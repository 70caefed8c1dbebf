"""GPU: the fused forms of the expert-parallel steps against the separate kernels they replace, through the C ABI.

router + pack in one launch          == m355_router_topk then m355_ep_pack            (every byte of ids, w and send)
experts on the received rows         == m355_ep_unpack then m355_moe_nf4_forward       (every bit of the output rows)
reduce inside add + RMSNorm          == m355_ep_reduce then m355_add_rmsnorm           (every bit of both outputs)
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import _close, _to_dev, tiled_layer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,E,k,H,N,bias", [(1, 32, 4, 2880, 8, True), (32, 32, 4, 2880, 8, True), (7, 8, 3, 256, 4, False),
                                            (300, 128, 4, 512, 8, True), (1500, 32, 4, 256, 2, True)])
def test_router_and_pack_in_one_launch(T, E, k, H, N, bias, ref, m355, cuda):
    sd = ref.stream_seed(7100, T, E)
    x = _to_dev(ref.normal_f16(ref.stream_seed(sd, 1), (T, H)), cuda)
    wr = _to_dev(ref.normal_f16(ref.stream_seed(sd, 2), (E, H), 0.05), cuda)
    br = _to_dev(ref.normal_f16(ref.stream_seed(sd, 3), (E,), 0.5), cuda) if bias else None
    epr = E // N
    rb = m355.ep_row_bytes(H, k)
    ids0, w0 = m355.router_topk(x, wr, br, k)
    send0 = torch.full((N * T * rb,), 0xA5, dtype=torch.uint8, device=cuda)   # bytes the kernels leave alone stay 0xA5
    m355.ep_pack(x, ids0, w0, N, epr, send0)
    send1 = torch.full((N * T * rb,), 0xA5, dtype=torch.uint8, device=cuda)
    ids1 = torch.empty_like(ids0)
    w1 = torch.empty_like(w0)
    m355.router_topk_ep_pack(x, wr, br, k, N, epr, send1, ids=ids1, w=w1)
    torch.cuda.synchronize()
    assert torch.equal(ids1, ids0)
    assert torch.equal(w1.view(torch.int32), w0.view(torch.int32))
    a, b = send0.view(N, T, rb), send1.view(N, T, rb)
    # the padding at the end of a row is never written by either
    assert torch.equal(a[:, :, :2 * H + 8 * k], b[:, :, :2 * H + 8 * k])
    assert torch.equal(a, b)


@pytest.mark.parametrize("T,N,E,k,H,FF,act,flags_extra", [
    (4, 8, 32, 4, 2880, 2880, "silu", 0),       # a rank of 8 at the decode batch: 32 slots, 4 local experts
    (1, 8, 32, 4, 2880, 2880, "silu", 0),
    (5, 2, 8, 3, 256, 192, "oai", 0),
    (40, 2, 8, 2, 256, 192, "silu", 0),
    (300, 2, 8, 2, 256, 192, "silu", 0),        # past the streaming kernel's range: the tile kernel gathers from the rows
])
def test_experts_read_the_received_rows(T, N, E, k, H, FF, act, flags_extra, ref, m355, cuda):
    sd = ref.stream_seed(7200, T, N, H)
    epr, rank = E // N, 1
    layer = ref.synth_layer(ref.stream_seed(sd, 9), E, H, FF)
    sl = slice(rank * epr, (rank + 1) * epr)
    local = {n: v[sl] for n, v in layer.items()}
    tl = tiled_layer(m355, local, epr, H, FF, cuda)
    rb = m355.ep_row_bytes(H, k)
    NT = N * T
    # what rank 1 receives: from every peer s its T tokens' rows of chunk 1
    recv = torch.full((NT * rb,), 0x5A, dtype=torch.uint8, device=cuda)
    xs, idss, wss = [], [], []
    for s in range(N):
        x = ref.normal_f16(ref.stream_seed(sd, 1, s), (T, H))
        lg = ref.normal_f16(ref.stream_seed(sd, 2, s), (T, E), 1.5)
        ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
        send = torch.full((N * T * rb,), 0x5A, dtype=torch.uint8, device=cuda)
        m355.ep_pack(_to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda), N, epr, send)
        recv.view(N, T * rb)[s] = send.view(N, T * rb)[rank]
        xs.append(x), idss.append(ids), wss.append(w)
    act_kw = {}
    if act == "oai":
        act_kw = dict(act_mode=m355.ACT_SWIGLU_OAI,
                      b13=_to_dev(ref.normal_f16(ref.stream_seed(sd, 5), (epr, 2 * FF), 0.1), cuda),
                      b2=_to_dev(ref.normal_f16(ref.stream_seed(sd, 6), (epr, H), 0.1), cuda))
    flags = m355.FLAG_FP16_COMBINE | m355.kernel_choice_flags(NT, k, E) | flags_extra
    ws = torch.empty(max(256, m355.moe_workspace_bytes(NT, k, H, FF, epr)), dtype=torch.uint8, device=cuda)
    # the two-step form
    x_all = torch.zeros((NT, H), dtype=torch.float16, device=cuda)
    ids_all = torch.empty((NT, k), dtype=torch.int32, device=cuda)
    w_all = torch.empty((NT, k), dtype=torch.float32, device=cuda)
    m355.ep_unpack(recv, NT, k, H, x_all, ids_all, w_all)
    want = torch.empty((NT, H), dtype=torch.float16, device=cuda)
    m355.moe_nf4_forward(x_all, ids_all, w_all, *tl, E_local=epr, e_first=rank * epr, H=H, FF=FF, workspace=ws, out=want,
                         flags=flags, **act_kw)
    got = torch.full((NT, H), float("nan"), dtype=torch.float16, device=cuda)
    m355.moe_nf4_forward_ep_rows(recv, NT, k, *tl, E_local=epr, e_first=rank * epr, H=H, FF=FF, workspace=ws, out=got,
                                 flags=flags, **act_kw)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # and against the oracle: the partial sum over this rank's experts of every slot
    if act == "silu":
        x_np, ids_np, w_np = np.concatenate(xs), np.concatenate(idss), np.concatenate(wss)
        ref_out = ref.moe_forward(x_np, ids_np, w_np, **local, E_local=epr, e_first=rank * epr, H=H, FF=FF)
        _close(got.cpu().numpy(), ref_out)


def test_experts_on_rows_rejects_bad_sizes(m355, cuda):
    H, k = 256, 2
    rows = torch.zeros(3 * m355.ep_row_bytes(H, k), dtype=torch.uint8, device=cuda)
    dummy = torch.zeros(16, dtype=torch.uint8, device=cuda)
    with pytest.raises(ValueError):
        m355.moe_nf4_forward_ep_rows(rows, 4, k, dummy, dummy.float(), dummy, dummy.float(), E_local=1, e_first=0, H=H, FF=192,
                                     workspace=dummy, out=torch.empty((4, H), dtype=torch.float16, device=cuda))


@pytest.mark.parametrize("T,N,H", [(32, 8, 2880), (1, 8, 2880), (4, 2, 2880), (9, 4, 256), (300, 8, 512)])
def test_reduce_inside_add_rmsnorm(T, N, H, ref, m355, cuda):
    sd = ref.stream_seed(7300, T, N, H)
    recv = _to_dev(ref.normal_f16(ref.stream_seed(sd, 1), (N, T, H)), cuda)
    x = _to_dev(ref.normal_f16(ref.stream_seed(sd, 2), (T, H)), cuda)
    wt = _to_dev(ref.normal_f16(ref.stream_seed(sd, 3), (H,)), cuda)
    delta = torch.empty((T, H), dtype=torch.float16, device=cuda)
    m355.ep_reduce(recv, N, T, H, delta)
    r0 = torch.empty_like(x)
    y0 = torch.empty_like(x)
    m355.add_rmsnorm(x, delta, wt, 1e-5, resid_out=r0, out=y0)
    r1 = torch.empty_like(x)
    y1 = torch.empty_like(x)
    m355.ep_reduce_add_rmsnorm(recv, N, x, wt, 1e-5, resid_out=r1, out=y1)
    torch.cuda.synchronize()
    assert torch.equal(r1.view(torch.int16), r0.view(torch.int16))
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
    # the sum itself: fp32 over ranks, one rounding
    want = recv.float().sum(0).half()
    assert (delta.float() - want.float()).abs().max().item() <= 2e-3 * want.float().abs().max().item()


def test_expert_parallel_class_one_rank_rehearsal(ref, m355, cuda, tmp_path):
    """ExpertParallelMoE with the exchange forced on one rank: route_and_pack -> dispatch -> experts (on the rows) ->
    combine -> finish_add_rmsnorm equals router_topk -> moe_nf4_forward -> add_rmsnorm up to the rounding of the partial
    rows."""
    import torch.distributed as dist
    from gpt_oss_vllm_amd import ep
    H, FF, E, k, T = 256, 192, 8, 3, 6
    layer = ref.synth_layer(7400, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    x = _to_dev(ref.normal_f16(7401, (T, H)), cuda)
    wr = _to_dev(ref.normal_f16(7402, (E, H), 0.1), cuda)
    resid = _to_dev(ref.normal_f16(7403, (T, H)), cuda)
    nw = _to_dev(ref.normal_f16(7404, (H,)), cuda)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1)
    try:
        moe = ep.ExpertParallelMoE(tl, num_experts=E, top_k=k, hidden=H, ffn=FF, max_tokens=T, force_exchange=True,
                                   exchange=lambda recv, send: recv.copy_(send))
        assert moe.in_place
        ids = torch.empty((T, k), dtype=torch.int32, device=cuda)
        w = torch.empty((T, k), dtype=torch.float32, device=cuda)
        moe.route_and_pack(x, wr, None, ids, w)
        moe.dispatch()
        moe.experts()
        moe.combine()
        r1 = torch.empty_like(x)
        y1 = torch.empty_like(x)
        moe.finish_add_rmsnorm(resid, nw, 1e-5, r1, y1)
        ids0, w0 = m355.router_topk(x, wr, None, k)
        out0 = m355.moe_nf4_forward(x, ids0, w0, *tl, E_local=E, e_first=0, H=H, FF=FF, flags=moe.flags | moe.shard_flags)
        r0 = torch.empty_like(x)
        y0 = torch.empty_like(x)
        m355.add_rmsnorm(resid, out0, nw, 1e-5, resid_out=r0, out=y0)
        torch.cuda.synchronize()
        assert torch.equal(ids, ids0)
        # one rank: the partial rows ARE the layer's output, so the sum over ranks adds nothing
        assert torch.equal(r1.view(torch.int16), r0.view(torch.int16))
        assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
    finally:
        if own:
            dist.destroy_process_group()
