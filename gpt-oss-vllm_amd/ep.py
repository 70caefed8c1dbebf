"""Expert parallelism for one MoE layer: experts sharded over the ranks of a node, tokens exchanged by
two all-to-alls (dispatch, combine) through torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-GPU (tensor_parallel_size=1, patches/test_gptoss_bnb.py:30); this module is the
multi-GPU layer the path gains here.  Expert e lives on rank e // (E / world_size).

Per layer and rank, with T local tokens and N ranks:
  pack      send[r, t] = (x[t], ids[t], w[t]) if token t routes an expert to rank r, else ids = -1
            (route_and_pack: written by the router's own launch)
  dispatch  all_to_all_single of fixed-size chunks (capacity = T rows per peer, so no size exchange)
  experts   partial[n*T + t] = sum over this rank's experts of w * expert(x): one fused MoE forward that reads
            activations, ids and weights straight from the received rows
  combine   all_to_all_single of the partial rows back to the owners
  reduce    out[t] = sum over ranks of their partial rows (fp32 sum in rank order, rounded to fp16 once; the k
            contributions of one rank are combined there in the forward's own way, so the result differs from the
            single-GPU sum over k by the rounding of the per-rank partial rows), or finish_add_rmsnorm: the same sum
            formed inside the residual add + RMSNorm that follows the MoE block

`ops` supplies the device work.  The product uses HipOps (the C ABI); tests may pass a CPU stand-in to
exercise the exchange logic over gloo.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class HipOps:
    """Device work through libmoe355.so.  Raises if the library or the GPU is missing."""

    def __init__(self):
        from . import lib
        lib.load()
        self.lib = lib

    def row_bytes(self, H, k):
        return self.lib.ep_row_bytes(H, k)

    def pack(self, x, ids, w, n_ranks, epr, send):
        self.lib.ep_pack(x, ids, w, n_ranks, epr, send)

    def unpack(self, recv, NT, k, H, x_all, ids_all, w_all):
        self.lib.ep_unpack(recv, NT, k, H, x_all, ids_all, w_all)

    def reduce(self, recv, n_ranks, T, H, out):
        self.lib.ep_reduce(recv, n_ranks, T, H, out)

    def reduce_add_rmsnorm(self, recv, n_ranks, x, weight, eps, resid_out, out):
        self.lib.ep_reduce_add_rmsnorm(recv, n_ranks, x, weight, eps, resid_out=resid_out, out=out)

    def route_and_pack(self, x, wr, br, k, n_ranks, epr, send, ids, w):
        self.lib.router_topk_ep_pack(x, wr, br, k, n_ranks, epr, send, ids=ids, w=w)

    def moe(self, x, ids, w, weights, E_local, e_first, H, FF, workspace, out, flags, **act):
        self.lib.moe_nf4_forward(x, ids, w, *weights, E_local=E_local, e_first=e_first, H=H, FF=FF,
                                 workspace=workspace, out=out, flags=flags, **act)

    def moe_rows(self, rows, NT, k, weights, E_local, e_first, H, FF, workspace, out, flags, **act):
        self.lib.moe_nf4_forward_ep_rows(rows, NT, k, *weights, E_local=E_local, e_first=e_first, H=H, FF=FF,
                                         workspace=workspace, out=out, flags=flags, **act)

    def workspace_bytes(self, T, k, H, FF, E_local):
        return self.lib.moe_workspace_bytes(T, k, H, FF, E_local)

    def kernel_choice_flags(self, tokens, top_k, global_experts):
        return self.lib.kernel_choice_flags(tokens, top_k, global_experts)


class ExpertParallelMoE:
    """One MoE layer's expert shard plus the buffers of its token exchange (allocated once, reused)."""

    def __init__(self, weights, *, num_experts: int, top_k: int, hidden: int, ffn: int, max_tokens: int,
                 group: Optional[dist.ProcessGroup] = None, ops=None, device=None, flags: int = 1,
                 force_exchange: bool = False, exchange=None, act_mode: int = 0, b13=None, b2=None,
                 alpha: float = 1.702, limit: float = 7.0):
        self.group = group
        # exchange(recv, send): the all-to-all of equal chunks.  Default: torch.distributed (RCCL on device tensors).
        self.exchange = exchange if exchange is not None else (
            lambda recv, send: dist.all_to_all_single(recv, send, group=self.group))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if num_experts % self.world:
            raise ValueError(f"{num_experts} experts do not split over {self.world} ranks")
        self.E, self.k, self.H, self.FF = num_experts, top_k, hidden, ffn
        self.epr = num_experts // self.world
        self.e_first = self.rank * self.epr
        # (w13_q, w13_absmax, w2_q, w2_absmax) of the local experts, tile order; an absmax is an fp32 tensor or a
        # (codes uint8, table fp32 [E_local][256]) pair
        self.weights = weights
        # activation and biases of the local experts (b13 [E_local][2FF], b2 [E_local][H]), as lib.moe_nf4_forward takes them
        self.act = dict(act_mode=act_mode, b13=b13, b2=b2, alpha=alpha, limit=limit) if (act_mode or b13 is not None or b2 is not None) else {}
        self.ops = ops if ops is not None else HipOps()
        self.flags = flags
        # a single rank normally skips the exchange; force_exchange runs pack / all-to-all / unpack / reduce anyway
        # (a one-rank rehearsal of the multi-GPU path, needs an initialised process group)
        self.force_exchange = force_exchange and dist.is_initialized()
        self.T = max_tokens
        self.shard_flags = self.ops.kernel_choice_flags(self.world * max_tokens, top_k, num_experts)
        dev = device if device is not None else weights[0].device
        N, T, H, k = self.world, max_tokens, hidden, top_k
        rb = self.ops.row_bytes(H, k)
        self.row_bytes = rb
        self.send = torch.empty(N * T * rb, dtype=torch.uint8, device=dev)
        self.recv = torch.empty(N * T * rb, dtype=torch.uint8, device=dev)
        self.in_place = hasattr(self.ops, "moe_rows")   # the experts read the received rows as they lie
        if not self.in_place:
            self.x_all = torch.zeros((N * T, H), dtype=torch.float16, device=dev)
            self.ids_all = torch.empty((N * T, k), dtype=torch.int32, device=dev)
            self.w_all = torch.empty((N * T, k), dtype=torch.float32, device=dev)
        self.partial = torch.empty((N * T, H), dtype=torch.float16, device=dev)
        self.back = torch.empty((N * T, H), dtype=torch.float16, device=dev)
        self.workspace = torch.empty(max(256, self.ops.workspace_bytes(N * T, k, H, ffn, self.epr)),
                                     dtype=torch.uint8, device=dev)

    def forward(self, x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None):
        """x [T, H] fp16, ids [T, k] int32 (global expert ids), w [T, k] fp32 -> out [T, H] fp16.
        T must equal the capacity the layer was built with (decode batches are fixed-size)."""
        T = x.shape[0]
        if T != self.T:
            raise ValueError(f"this layer exchanges exactly {self.T} tokens per rank, got {T}")
        N, H, k = self.world, self.H, self.k
        if out is None:
            out = torch.empty((T, H), dtype=torch.float16, device=x.device)
        if N == 1 and not self.force_exchange:
            self.ops.moe(x, ids, w, self.weights, self.epr, 0, H, self.FF, self.workspace, out, self.flags, **self.act)
            return out
        self.pack(x, ids, w)
        self.dispatch()
        self.experts()
        self.combine()
        return self.finish(out)

    # The five pieces of forward(), callable one by one: pack / experts / finish launch kernels only (a caller may capture
    # them in HIP graphs), dispatch / combine are the two collectives.

    def pack(self, x, ids, w):
        self.ops.pack(x, ids, w, self.world, self.epr, self.send)

    def route_and_pack(self, x, wr, br, ids, w):
        """Router (x @ wr.T + br -> top-k -> softmax into ids / w) and pack in one launch."""
        self.ops.route_and_pack(x, wr, br, self.k, self.world, self.epr, self.send, ids, w)

    def dispatch(self):
        self.exchange(self.recv, self.send)

    def experts(self):
        # this rank sees the N * T token slots of all ranks but only the pairs routed to its experts
        if self.in_place:
            self.ops.moe_rows(self.recv, self.world * self.T, self.k, self.weights, self.epr, self.e_first, self.H, self.FF,
                              self.workspace, self.partial, self.flags | self.shard_flags, **self.act)
            return
        self.ops.unpack(self.recv, self.world * self.T, self.k, self.H, self.x_all, self.ids_all, self.w_all)
        self.ops.moe(self.x_all, self.ids_all, self.w_all, self.weights, self.epr, self.e_first, self.H, self.FF,
                     self.workspace, self.partial, self.flags | self.shard_flags, **self.act)

    def combine(self):
        self.exchange(self.back.view(-1), self.partial.view(-1))

    def finish(self, out):
        self.ops.reduce(self.back, self.world, self.T, self.H, out)
        return out

    def finish_add_rmsnorm(self, x, weight, eps, resid_out, out):
        """The reduce step inside the residual add + RMSNorm that follows: resid_out = x + sum over ranks, out = its norm."""
        self.ops.reduce_add_rmsnorm(self.back, self.world, x, weight, eps, resid_out, out)
        return out
