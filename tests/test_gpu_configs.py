"""GPU parity at the full sizes of the multi-GPU and 120b-shape workloads: a rank of an 8-way expert shard of
gpt-oss-20b (256 token slots, 4 local experts out of 32, first local expert != 0, rows of other ranks marked -1) and the
128-expert layer (decode batches of 32 and 1, and an 8192-token prompt), each against the CPU oracle through the C ABI.
Weights are drawn on the GPU directly in the kernels' tile order; the oracle gets the experts it needs in checkpoint
order through the inverse of m355_repack_nf4 (host)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL_MAX, ATOL = 2e-3, 1e-5     # |d| <= 2e-3 * max|y| + 1e-5, as in test_gpu_parity.py


def _close(got, want, rtol=RTOL_MAX, atol=ATOL):
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    lim = rtol * np.abs(want).max() + atol
    err = np.abs(got - want).max()
    assert err <= lim, f"max |d| {err} > {lim}"


def untile(qt, at, rows, cols):
    """Inverse of m355_repack_nf4 for one expert matrix (host): tile order -> checkpoint order."""
    NB, P = cols // 64, rows // 32
    qt = qt.reshape(P, NB, 4, 16, 2, 8)                                     # [p][b][g][r][half][8 bytes]
    q = np.empty((rows, cols // 2), np.uint8).reshape(2, P, 16, NB, 4, 8)   # [half][p][r][b][g][8]
    q[:] = qt.transpose(4, 0, 3, 1, 2, 5)
    at = at.reshape(P, 2, 4, NB, 4)                                         # [p][half][g4][b][i]
    a = np.empty((2, P, 4, 4, NB), at.dtype)                                # [half][p][g4][i][b]
    a[:] = at.transpose(1, 0, 2, 4, 3)
    return q.reshape(-1), a.reshape(-1)


class TiledLayer:
    """E experts of [2FF][H] and [H][FF] NF4 weights drawn on the GPU in tile order, absmax as code byte + table."""

    def __init__(self, ref, E, H, FF, cuda, seed):
        g = torch.Generator(device=cuda)
        g.manual_seed(seed)
        self.E, self.H, self.FF = E, H, FF
        nm = torch.from_numpy(ref.nested_quant_map())
        self.q13 = torch.randint(0, 256, (E, FF * H), dtype=torch.uint8, device=cuda, generator=g)
        self.q2 = torch.randint(0, 256, (E, H * FF // 2), dtype=torch.uint8, device=cuda, generator=g)
        self.c13 = torch.randint(0, 256, (E, 2 * FF * H // 64), dtype=torch.uint8, device=cuda, generator=g)
        self.c2 = torch.randint(0, 256, (E, H * FF // 64), dtype=torch.uint8, device=cuda, generator=g)
        cg = torch.Generator()
        cg.manual_seed(seed)
        self.t13 = (nm[None, :] * (0.01 + 0.02 * torch.rand((E, 1), generator=cg))).float().to(cuda)
        self.t2 = (nm[None, :] * (0.01 + 0.02 * torch.rand((E, 1), generator=cg))).float().to(cuda)

    def q8(self):
        return self.q13, (self.c13, self.t13), self.q2, (self.c2, self.t2)

    def fp32(self):
        return self.q13, torch.gather(self.t13, 1, self.c13.long()), self.q2, torch.gather(self.t2, 1, self.c2.long())

    def oracle_experts(self, experts):
        """Checkpoint-order tensors of the listed local experts for ref.moe_forward."""
        sub = dict(w13_q=[], w13_absmax=[], w2_q=[], w2_absmax=[])
        for e in experts:
            q, c = untile(self.q13[e].cpu().numpy(), self.c13[e].cpu().numpy(), 2 * self.FF, self.H)
            sub["w13_q"].append(q), sub["w13_absmax"].append(self.t13[e].cpu().numpy()[c])
            q, c = untile(self.q2[e].cpu().numpy(), self.c2[e].cpu().numpy(), self.H, self.FF)
            sub["w2_q"].append(q), sub["w2_absmax"].append(self.t2[e].cpu().numpy()[c])
        return {n: np.stack(v) for n, v in sub.items()}


def oracle_rows(ref, layer, x, ids, w, rows, e_first=0, flags=1):
    """ref.moe_forward for the token rows `rows`, restricted to the local experts those rows use (renumbered densely)."""
    E = layer.E
    ids_s = ids[rows]
    local = sorted({int(e) - e_first for e in ids_s.reshape(-1) if e_first <= e < e_first + E})
    if not local:
        return np.zeros((len(rows), layer.H), np.float16)
    remap = {e + e_first: i for i, e in enumerate(local)}
    ids_r = np.vectorize(lambda e: remap.get(int(e), -1))(ids_s).astype(np.int32)
    sub = layer.oracle_experts(local)
    return ref.moe_forward(x[rows], ids_r, w[rows], **sub, E_local=len(local), e_first=0, H=layer.H, FF=layer.FF, flags=flags)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("e_first", [0, 12, 28])
def test_rank_of_eight_way_shard_full_size(e_first, ref, m355, cuda):
    """What one of 8 ranks computes per layer at a global batch of 256 (32 sequences per rank): 256 received token slots,
    its 4 of the 32 experts, top-4 ids over all 32 experts, slots that route nothing here marked ids = -1 / w = 0 as the
    exchange's unpack step leaves them, kernel form chosen as the expert-parallel layer chooses it."""
    T, E_glob, E_loc, k, H, FF = 256, 32, 4, 4, 2880, 2880
    layer = TiledLayer(ref, E_loc, H, FF, cuda, seed=1000 + e_first)
    g = torch.Generator(device=cuda)
    g.manual_seed(77 + e_first)
    x = torch.randn((T, H), device=cuda, generator=g).half()
    lg = (torch.randn((T, E_glob), device=cuda, generator=g) * 1.5).half()
    ids, w = m355.topk_softmax(lg, k)
    here = ((ids >= e_first) & (ids < e_first + E_loc)).any(dim=1)
    ids = torch.where(here[:, None], ids, torch.full_like(ids, -1)).contiguous()
    w = torch.where(here[:, None], w, torch.zeros_like(w)).contiguous()
    assert 0 < int(here.sum()) < T
    flags = m355.FLAG_FP16_COMBINE | m355.kernel_choice_flags(T, k, E_glob)
    assert flags & m355.FLAG_FORCE_STREAM and flags & m355.FLAG_STREAM_WIDE
    out = m355.moe_nf4_forward(x, ids, w, *layer.q8(), E_local=E_loc, e_first=e_first, H=H, FF=FF, flags=flags)
    out32 = m355.moe_nf4_forward(x, ids, w, *layer.fp32(), E_local=E_loc, e_first=e_first, H=H, FF=FF, flags=flags)
    narrow = m355.moe_nf4_forward(x, ids, w, *layer.q8(), E_local=E_loc, e_first=e_first, H=H, FF=FF,
                                  flags=m355.FLAG_FP16_COMBINE | m355.FLAG_FORCE_STREAM)
    torch.cuda.synchronize()
    assert torch.equal(out, out32)
    rows = list(range(T))
    want = oracle_rows(ref, layer, x.cpu().numpy(), ids.cpu().numpy(), w.cpu().numpy(), rows, e_first=e_first)
    _close(out.cpu().numpy(), want)
    _close(narrow.cpu().numpy(), want)
    assert not out[~here].any()            # slots of other ranks come back as zero rows


@pytest.fixture(scope="module")
def layer128(ref, cuda):
    return TiledLayer(ref, 128, 2880, 2880, cuda, seed=120)


@pytest.mark.parametrize("T", [32, 1, 7])
def test_120b_shape_decode(T, layer128, ref, m355, cuda):
    """128 experts of gpt-oss size: a decode batch of 32 (streaming kernel, routing kernel since 128 pairs over 128 experts
    is inside the inline limits: both paths below), 1 and 7 tokens (router computed inside the gate_up launch)."""
    E, k, H, FF = 128, 4, 2880, 2880
    g = torch.Generator(device=cuda)
    g.manual_seed(500 + T)
    x = torch.randn((T, H), device=cuda, generator=g).half()
    wr = (torch.randn((E, H), device=cuda, generator=g) * 0.02).half()
    out, ids, w = m355.moe_nf4_forward_routed(x, wr, None, k, *layer128.q8(), E_local=E, e_first=0, H=H, FF=FF)
    ids0, w0 = m355.router_topk(x, wr, None, k)
    sep = m355.moe_nf4_forward(x, ids0, w0, *layer128.q8(), E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    assert torch.equal(ids, ids0) and torch.equal(w, w0) and torch.equal(out, sep)
    ids_r, _, _ = ref.router_topk(x.cpu().numpy(), wr.cpu().numpy(), None, k)
    assert (np.sort(ids_r, axis=1) == np.sort(ids.cpu().numpy(), axis=1)).mean() > 0.9     # near-ties may differ
    rows = list(range(T)) if T <= 8 else [0, 5, 13, 21, 30, 31]
    want = oracle_rows(ref, layer128, x.cpu().numpy(), ids.cpu().numpy(), w.cpu().numpy(), rows)
    _close(out[rows].cpu().numpy(), want)


def test_120b_shape_prompt(layer128, ref, m355, cuda):
    """8192 prompt tokens over 128 experts: about 256 (token, expert) pairs per expert, one 256-token tile and a ragged
    second one for most experts; the library picks the tile kernel from the shape.  Sampled rows against the oracle, byte
    absmax against fp32 absmax bit for bit."""
    T, E, k, H, FF = 8192, 128, 4, 2880, 2880
    g = torch.Generator(device=cuda)
    g.manual_seed(8192)
    x = torch.randn((T, H), device=cuda, generator=g).half()
    lg = (torch.randn((T, E), device=cuda, generator=g) * 1.5).half()
    ids, w = m355.topk_softmax(lg, k)
    out = m355.moe_nf4_forward(x, ids, w, *layer128.q8(), E_local=E, e_first=0, H=H, FF=FF)
    out32 = m355.moe_nf4_forward(x, ids, w, *layer128.fp32(), E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    assert torch.equal(out, out32)
    rows = [0, 255, 256, 4097, 8000, 8191]
    want = oracle_rows(ref, layer128, x[rows].cpu().numpy(), ids[rows].cpu().numpy(), w[rows].cpu().numpy(), list(range(len(rows))))
    _close(out[rows].cpu().numpy(), want)


@pytest.mark.parametrize("T,E,k,H,FF", [(5, 4, 2, 3136, 3136), (37, 3, 2, 3136, 2880), (20, 2, 1, 2880, 4096)])
def test_general_kernel_beyond_48_blocks(T, E, k, H, FF, ref, m355, cuda):
    """Hidden or intermediate sizes above 3072 (more than 48 blocks of 64 along K) leave the register-resident streaming
    form for the general kernel (activations staged in LDS, one launch row per active expert); several token tiles per
    expert at T = 37."""
    sd = ref.stream_seed(T, E, k, H, FF)
    lay = ref.synth_layer(sd, E, H, FF)
    x = ref.normal_f16(ref.stream_seed(sd, 1), (T, H))
    lg = ref.normal_f16(ref.stream_seed(sd, 2), (T, E), 1.5)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)  # noqa: E731
    ids, w = m355.topk_softmax(dev(lg), k)
    w13_q = torch.empty((E, FF * H), dtype=torch.uint8, device=cuda)
    w13_a = torch.empty((E, 2 * FF * H // 64), dtype=torch.float32, device=cuda)
    w2_q = torch.empty((E, H * FF // 2), dtype=torch.uint8, device=cuda)
    w2_a = torch.empty((E, H * FF // 64), dtype=torch.float32, device=cuda)
    for e in range(E):
        w13_q[e], w13_a[e] = m355.repack_nf4(dev(lay["w13_q"][e]), dev(lay["w13_absmax"][e]), 2 * FF, H)
        w2_q[e], w2_a[e] = m355.repack_nf4(dev(lay["w2_q"][e]), dev(lay["w2_absmax"][e]), H, FF)
    out = m355.moe_nf4_forward(dev(x), ids, w, w13_q, w13_a, w2_q, w2_a, E_local=E, e_first=0, H=H, FF=FF)
    torch.cuda.synchronize()
    want = ref.moe_forward(x, ids.cpu().numpy(), w.cpu().numpy(), **lay, E_local=E, H=H, FF=FF)
    _close(out.cpu().numpy(), want)
