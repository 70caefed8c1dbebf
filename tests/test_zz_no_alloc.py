"""GPU: the entry point that allocates nothing (m355_moe_nf4_forward_static, workspace in a device array inside the
library) against m355_moe_nf4_forward_ex with a caller-owned workspace: same bits, on two calls in a row, the second
with a smaller input (the array keeps the first call's contents), through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _layer(ref, m355, cuda, E, H, FF, seed):
    lay = ref.synth_layer(seed, E, H, FF)
    w13_q = torch.empty((E, FF * H), dtype=torch.uint8, device=cuda)
    w13_a = torch.empty((E, 2 * FF * H // 64), dtype=torch.float32, device=cuda)
    w2_q = torch.empty((E, H * FF // 2), dtype=torch.uint8, device=cuda)
    w2_a = torch.empty((E, H * FF // 64), dtype=torch.float32, device=cuda)
    for e in range(E):
        w13_q[e], w13_a[e] = m355.repack_nf4(_dev(lay["w13_q"][e], cuda), _dev(lay["w13_absmax"][e], cuda), 2 * FF, H)
        w2_q[e], w2_a[e] = m355.repack_nf4(_dev(lay["w2_q"][e], cuda), _dev(lay["w2_absmax"][e], cuda), H, FF)
    return lay, (w13_q, w13_a, w2_q, w2_a)


@pytest.mark.parametrize("H,FF,E,k,sizes", [(256, 192, 6, 2, (70, 9)), (2880, 2880, 4, 2, (24, 3))])
def test_static_workspace_entry_equals_workspace_entry(H, FF, E, k, sizes, ref, m355, cuda):
    lay, tl = _layer(ref, m355, cuda, E, H, FF, ref.stream_seed(H, FF, E, 606))
    L = m355.load()
    # set-up on the device the tensors live on: kernel attributes and the address of the library's device array
    m355.init(cuda.index if cuda.index is not None else torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    for T in sizes:       # the second call is smaller: it must not read what the first left in the shared array
        x = _dev(ref.normal_f16(ref.stream_seed(606, T, 1), (T, H)), cuda)
        lg = _dev(ref.normal_f16(ref.stream_seed(606, T, 2), (T, E), 1.5), cuda)
        b13 = _dev(ref.normal_f16(ref.stream_seed(606, T, 3), (E, 2 * FF), 0.1), cuda)
        b2 = _dev(ref.normal_f16(ref.stream_seed(606, T, 4), (E, H), 0.1), cuda)
        ids, w = m355.topk_softmax(lg, k)
        for act, bias in ((m355.ACT_SILU_CONCAT, False), (m355.ACT_SWIGLU_OAI, True)):
            want = m355.moe_nf4_forward(x, ids, w, *tl, E_local=E, e_first=0, H=H, FF=FF, act_mode=act,
                                        b13=b13 if bias else None, b2=b2 if bias else None)
            got = torch.full((T, H), float("nan"), dtype=torch.float16, device=cuda)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]       # device memory free (hipMemGetInfo)
            rc = L.m355_moe_nf4_forward_static(      # straight through the C ABI
                x.data_ptr(), T, ids.data_ptr(), w.data_ptr(), k, tl[0].data_ptr(), tl[1].data_ptr(), tl[2].data_ptr(),
                tl[3].data_ptr(), E, 0, H, FF, act, m355.FLAG_FP16_COMBINE, b13.data_ptr() if bias else None,
                b2.data_ptr() if bias else None, ctypes.c_float(1.702), ctypes.c_float(7.0), got.data_ptr(), stream)
            assert rc == 0, L.m355_last_error()
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] >= free_before   # the call took no device memory
            assert torch.isfinite(got.float()).all()
            assert torch.equal(got, want)
            # the stronger check: under stream capture a hipMalloc / hipFree is an error that invalidates the capture, so a
            # call that is captured, instantiated and replayed with the right result has made none
            got2 = torch.full((T, H), float("nan"), dtype=torch.float16, device=cuda)
            side = torch.cuda.Stream(device=cuda)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    rc = L.m355_moe_nf4_forward_static(
                        x.data_ptr(), T, ids.data_ptr(), w.data_ptr(), k, tl[0].data_ptr(), tl[1].data_ptr(), tl[2].data_ptr(),
                        tl[3].data_ptr(), E, 0, H, FF, act, m355.FLAG_FP16_COMBINE, b13.data_ptr() if bias else None,
                        b2.data_ptr() if bias else None, ctypes.c_float(1.702), ctypes.c_float(7.0), got2.data_ptr(),
                        side.cuda_stream)
            assert rc == 0, L.m355_last_error()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(got2, want)
            if act == m355.ACT_SILU_CONCAT and H == 256:
                ref_out = ref.moe_forward(x.cpu().numpy(), ids.cpu().numpy(), w.cpu().numpy(), **lay, E_local=E, H=H, FF=FF)
                lim = 2e-3 * np.abs(ref_out.astype(np.float32)).max() + 1e-5
                assert np.abs(got.cpu().numpy().astype(np.float32) - ref_out.astype(np.float32)).max() <= lim


def test_static_workspace_entry_rejects_oversize_and_bad_shapes(m355, cuda):
    L = m355.load()
    H = FF = 2880
    T, k, E = 400, 4, 4          # 1600 pairs x 11.5 KB of h and y rows: beyond the 8 MiB array
    assert m355.moe_workspace_bytes(T, k, H, FF, E) > m355.STATIC_WORKSPACE_BYTES
    x = torch.zeros((T, H), dtype=torch.float16, device=cuda)
    ids = torch.zeros((T, k), dtype=torch.int32, device=cuda)
    w = torch.zeros((T, k), dtype=torch.float32, device=cuda)
    q = torch.zeros(16, dtype=torch.uint8, device=cuda)
    a = torch.zeros(16, dtype=torch.float32, device=cuda)
    out = torch.zeros((T, H), dtype=torch.float16, device=cuda)
    with pytest.raises(m355.Mi355LibraryError, match="static one holds"):
        m355.moe_nf4_forward_static(x, ids, w, q, a, q, a, E_local=E, e_first=0, H=H, FF=FF, out=out)
    rc = L.m355_moe_nf4_forward_static(x.data_ptr(), 4, ids.data_ptr(), w.data_ptr(), 0, q.data_ptr(), a.data_ptr(),
                                       q.data_ptr(), a.data_ptr(), E, 0, H, FF, 0, 0, None, None, ctypes.c_float(0), ctypes.c_float(0),
                                       out.data_ptr(), None)
    assert rc == -2
    # an empty batch is a no-op, as for the other entry points
    assert L.m355_moe_nf4_forward_static(None, 0, None, None, k, None, None, None, None, E, 0, H, FF, 0, 0, None, None,
                                         ctypes.c_float(0), ctypes.c_float(0), None, None) == 0
