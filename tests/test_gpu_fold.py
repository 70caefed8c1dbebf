"""GPU: the combine step folded into the down stage's launch (the streaming kernel's default for small batches, forced
here at every size with M355_FLAG_FOLD_COMBINE) against the combine as its own launch (M355_FLAG_SEPARATE_COMBINE): the
same bits, on repeated calls that share one workspace."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _close, _to_dev, tiled_layer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,E,k,H,FF,extra", [
    (32, 32, 4, 2880, 2880, 0), (1, 32, 4, 2880, 2880, 0), (7, 8, 3, 256, 192, 0), (16, 8, 2, 320, 192, 0),
    (70, 4, 2, 256, 192, "stream"), (70, 4, 2, 256, 192, "wide"), (200, 8, 2, 256, 192, "wide"),
])
def test_folded_combine_equals_separate_launch(T, E, k, H, FF, extra, ref, m355, cuda):
    sd = ref.stream_seed(8100, T, E, H)
    layer = ref.synth_layer(ref.stream_seed(sd, 1), E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    ws = torch.empty(max(256, m355.moe_workspace_bytes(max(T, 64), k, H, FF, E)), dtype=torch.uint8, device=cuda)
    flags = m355.FLAG_FP16_COMBINE | m355.FLAG_FOLD_COMBINE | {0: 0, "stream": m355.FLAG_FORCE_STREAM, "wide": m355.FLAG_FORCE_STREAM | m355.FLAG_STREAM_WIDE}[extra]
    for rep, Tn in enumerate((T, max(1, T // 2), T)):       # the same workspace, changing sizes
        x = ref.normal_f16(ref.stream_seed(sd, 2, rep), (Tn, H))
        lg = ref.normal_f16(ref.stream_seed(sd, 3, rep), (Tn, E), 1.5)
        ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
        xd, idd, wd = _to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda)
        got = torch.full((Tn, H), float("nan"), dtype=torch.float16, device=cuda)
        m355.moe_nf4_forward(xd, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, out=got, flags=flags)
        want = torch.full((Tn, H), float("nan"), dtype=torch.float16, device=cuda)
        m355.moe_nf4_forward(xd, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, out=want,
                             flags=(flags & ~m355.FLAG_FOLD_COMBINE) | m355.FLAG_SEPARATE_COMBINE)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (rep, Tn)
        if rep == 0 and H <= 320:
            _close(got.cpu().numpy(), ref.moe_forward(x, ids, w, **layer, E_local=E, H=H, FF=FF))


def test_folded_combine_on_an_expert_shard(ref, m355, cuda):
    """Tokens without a local expert come out as zero rows, a batch without any local pair as all zeros."""
    T, E, k, H, FF = 24, 8, 2, 256, 192
    layer = ref.synth_layer(8200, E, H, FF)
    local = {n: v[2:4] for n, v in layer.items()}
    tl = tiled_layer(m355, local, 2, H, FF, cuda)
    x = ref.normal_f16(8201, (T, H))
    lg = ref.normal_f16(8202, (T, E), 1.5)
    ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
    ws = torch.empty(max(256, m355.moe_workspace_bytes(T, k, H, FF, 2)), dtype=torch.uint8, device=cuda)
    for ids_case in (ids, np.where((ids >= 2) & (ids < 4), ids + 4, ids).astype(np.int32)):   # second: nobody routes to experts 2, 3
        outs = []
        for fl in (m355.FLAG_FP16_COMBINE | m355.FLAG_FOLD_COMBINE, m355.FLAG_FP16_COMBINE | m355.FLAG_SEPARATE_COMBINE):
            o = torch.full((T, H), float("nan"), dtype=torch.float16, device=cuda)
            m355.moe_nf4_forward(_to_dev(x, cuda), _to_dev(ids_case, cuda), _to_dev(w, cuda), *tl, E_local=2, e_first=2, H=H, FF=FF,
                                 workspace=ws, out=o, flags=fl)
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
        hit = ((ids_case >= 2) & (ids_case < 4)).any(1)
        assert (outs[0][~torch.from_numpy(hit).to(cuda)] == 0).all()
        assert not torch.isnan(outs[0]).any()


def test_folded_combine_under_load(ref, m355, cuda):
    """The hand-off of y between workgroups inside the down launch, where an idle chip would hide a missing wait: the
    folded form (the default up to 3 tokens at top-4, forced here) for decode batches of 1, 4 and 8 against the separate combine launch, bit for bit, 240 calls
    back to back on one workspace, while a second stream keeps the chip busy with streaming kernels of uneven sizes (some
    fill every CU, some only a part).  Consecutive calls carry different inputs, so a y row left over from the call
    before would belong to another token; every word of every output is compared."""
    E, k, H, FF, CALLS, N_IN = 32, 4, 2880, 2880, 240, 6
    layer = ref.synth_layer(8300, E, H, FF)
    tl = tiled_layer(m355, layer, E, H, FF, cuda)
    ws = torch.empty(m355.moe_workspace_bytes(8, k, H, FF, E), dtype=torch.uint8, device=cuda)
    cases = []
    for T in (1, 4, 8):
        for i in range(N_IN):
            x = ref.normal_f16(ref.stream_seed(8301, T, i), (T, H))
            lg = ref.normal_f16(ref.stream_seed(8302, T, i), (T, E), 1.5)
            ids, w, _ = ref.router_topk(lg, np.eye(E, dtype=np.float16), None, k)
            xd, idd, wd = _to_dev(x, cuda), _to_dev(ids, cuda), _to_dev(w, cuda)
            want = torch.empty((T, H), dtype=torch.float16, device=cuda)
            m355.moe_nf4_forward(xd, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, out=want,
                                 flags=m355.FLAG_FP16_COMBINE | m355.FLAG_SEPARATE_COMBINE)
            cases.append((xd, idd, wd, want))
    torch.cuda.synchronize()
    # the load: NF4 dequantisation (reads 0.52 B, writes 2 B per weight) of 1 M .. 64 M weights on a second stream
    nmax = 64 << 20
    q = torch.randint(0, 256, (nmax // 2,), dtype=torch.uint8, device=cuda)
    am = torch.rand(nmax // 64, dtype=torch.float32, device=cuda)
    dq = torch.empty(nmax, dtype=torch.float16, device=cuda)
    side = torch.cuda.Stream(device=cuda)
    outs = [None] * CALLS
    for i in range(CALLS):
        n = (1 << 20) << (i * 5 % 7)
        with torch.cuda.stream(side):
            m355.nf4_dequant(q[: n // 2], am[: n // 64], 64, out=dq[:n])
        xd, idd, wd, _ = cases[(i * 7) % len(cases)]      # 7 is coprime to 18: sizes and inputs change from call to call
        outs[i] = torch.full_like(cases[(i * 7) % len(cases)][3], float("nan"))
        m355.moe_nf4_forward(xd, idd, wd, *tl, E_local=E, e_first=0, H=H, FF=FF, workspace=ws, out=outs[i],
                             flags=m355.FLAG_FP16_COMBINE | m355.FLAG_FOLD_COMBINE)
    torch.cuda.synchronize()
    bad = [i for i in range(CALLS) if not torch.equal(outs[i].view(torch.int16), cases[(i * 7) % len(cases)][3].view(torch.int16))]
    assert not bad, f"{len(bad)} of {CALLS} folded calls differ from the separate combine, first at call {bad[0]}"
