"""Row-sharded embedding tables over the GPUs of one node.

The reference has no model-parallel embedding (single ``torch.device``; DataParallel / DDP only in
vendored trainers); this is the MI355X-native scaling path for tables that should not
be replicated: ``owner(id) = id % W``, local row ``id // W``.  One exchange each way:

  forward   bucket lookups by owner -> all-to-all(row numbers) -> owner gathers rows (rbx_embed_fwd)
            -> all-to-all(rows) back -> un-permute
  backward  permute dY -> all-to-all(dY) -> owner scatter-adds into its shard's dense grad
            (rbx_embed_sort + rbx_embed_bwd: sorted, segmented, deterministic)

xGMI is point-to-point (7 links per GPU): an all-to-all keeps every link busy at once, so the
exchange time is max_peer_bytes / link_bw rather than a ring's 2(N-1)/N * S / link_bw.
Small tables stay replicated (their dense grads are all-reduced): cheaper than exchanging.

Two exchange modes:
  exact   variable-size all-to-all-v; needs the per-peer counts on the host (one sync per call).
  padded  every rank sends exactly ``capacity`` slots to every peer (empty slots carry row -1):
          no host sync, fixed shapes: the routing is one HIP call (rbx_route) and the step can be
          captured as hipGraph pieces between the RCCL exchanges (recbox_amd.graph.ShardedFMStep).  A lookup that does not fit sets the
          ``overflow`` flag (checked by the caller after the step) instead of being dropped silently.
"""
import math

import torch
from torch import nn

from . import comm


class HipLocalOps(object):
    """Local gather / scatter-add of one shard through the C ABI (plans are cached per shard shape)."""

    def __init__(self):
        self._plans = {}

    def _plan(self, weight):
        from . import ops
        from ._lib import FIELD_CATEGORICAL, POOL_NONE
        key = tuple(weight.shape)
        plan = self._plans.get(key)
        if plan is None:
            spec = ops.FieldSpec("shard", FIELD_CATEGORICAL, weight.shape[1], 0, param=0, pool=POOL_NONE,
                                 vocab=weight.shape[0])
            plan = ops.EmbedPlan([spec], weight.shape[1])
            self._plans[key] = plan
        return plan

    def gather(self, weight, rows):
        """rows [n] (row -1 = empty slot -> zero vector) -> [n, width]."""
        from . import ops
        from ._lib import check, lib
        n = rows.numel()
        out = torch.empty((n, weight.shape[1]), dtype=torch.float32, device=weight.device)
        if n == 0:
            return out
        plan = self._plan(weight)
        plan.bind_inputs([rows])
        plan.bind_params([weight.detach()])
        check(lib.rbx_embed_fwd(plan.arr, 1, n, ops._ptr(out), weight.shape[1], None, None, ops._stream()))
        return out

    def route(self, ids, world, capacity, base, overflow):
        """Wire slots of the padded exchange: (slot [B, T] int32, send [world * capacity] int64) -- rbx_route."""
        from . import ops
        send, slot = ops.route(ids, world, capacity, base, overflow)
        return slot, send

    def presort(self, weight, rows):
        """The id sort of ``scatter_add`` depends on the row numbers only: run it as soon as they are known (the
        result is passed to ``scatter_add`` as ``sorted_ws``)."""
        from . import ops
        from ._lib import check, lib
        n = rows.numel()
        if n == 0:
            return None
        plan = self._plan(weight)
        plan.bind_inputs([rows])
        plan.bind_params([weight.detach()], [weight.detach()])          # placeholder grad pointer: "trainable"
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
        check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(ws), ws_bytes, None, ops._stream()))
        return ws

    def scatter_add(self, weight, rows, dy, sorted_ws=None):
        """dense [n_local, width] gradient of ``gather`` w.r.t. weight (rows < 0 are skipped)."""
        from . import ops
        from ._lib import check, lib
        n = rows.numel()
        grad = torch.zeros_like(weight)
        if n == 0:
            return grad
        plan = self._plan(weight)
        plan.bind_inputs([rows])
        plan.bind_params([weight.detach()], [grad])
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        st = ops._stream()
        ws = sorted_ws
        if ws is None:
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
            check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(ws), ws_bytes, None, st))
        dy = dy.contiguous()
        check(lib.rbx_embed_bwd(plan.arr, 1, n, ops._ptr(dy), dy.stride(0), None, 0, ops._ptr(ws), ws_bytes, st))
        return grad


class _ExactLookup(torch.autograd.Function):
    """(owner, row) lookups, variable-size exchange (one host sync for the counts)."""

    @staticmethod
    def forward(ctx, owner, row, group, local_ops, out_shape, weight):
        rank, W = comm.world(group)
        owner = owner.reshape(-1)
        row = row.reshape(-1)
        perm = torch.argsort(owner, stable=True)                 # lookups grouped by destination rank
        send_rows = row[perm]
        send_counts_t = torch.bincount(owner, minlength=W)
        send_counts = send_counts_t.tolist()                     # sizes of the variable all-to-all (host sync)
        recv_counts = comm.exchange_counts(send_counts_t, group)
        recv_rows = comm.all_to_all_rows(send_rows, send_counts, recv_counts, group)      # row numbers out
        vecs = local_ops.gather(weight, recv_rows)                                        # owner-side gather
        back = comm.all_to_all_rows(vecs, recv_counts, send_counts, group)                # rows back
        out = torch.empty_like(back)
        out[perm] = back
        ctx.save_for_backward(perm, recv_rows, weight)
        ctx.meta = (send_counts, recv_counts, group, local_ops)
        return out.view(*out_shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dout):
        perm, recv_rows, weight = ctx.saved_tensors
        send_counts, recv_counts, group, local_ops = ctx.meta
        d_sorted = dout.reshape(-1, weight.shape[1])[perm].contiguous()
        d_recv = comm.all_to_all_rows(d_sorted, send_counts, recv_counts, group)          # dY to the owners
        return None, None, None, None, None, local_ops.scatter_add(weight, recv_rows, d_recv)


class _PaddedLookup(torch.autograd.Function):
    """Lookups ids[B, T], fixed-capacity exchange: static shapes, no host sync, graph-capturable."""

    @staticmethod
    def forward(ctx, ids, tables, capacity, weight):
        W, group, local_ops = tables.world_size, tables.group, tables.local_ops
        slot, send = tables.route(ids, capacity)
        slot = slot.reshape(-1).long()
        recv = comm.all_to_all_equal(send, group)                                          # [W * capacity] rows, -1 = empty
        vecs = local_ops.gather(weight, recv)
        back = comm.all_to_all_equal(vecs, group)                                          # same slot numbering
        width = weight.shape[1]
        out = torch.cat([back, back.new_zeros((1, width))], dim=0)[slot]                   # dump slot reads zeros
        ctx.save_for_backward(slot, recv, weight)
        ctx.meta = (capacity, group, local_ops, W)
        return out.view(*ids.shape, width)

    @staticmethod
    def backward(ctx, dout):
        slot, recv, weight = ctx.saved_tensors
        capacity, group, local_ops, W = ctx.meta
        width = weight.shape[1]
        dsend = dout.new_zeros((W * capacity + 1, width))
        dsend[slot] = dout.reshape(-1, width)                     # slots are unique except the dump slot (discarded)
        d_recv = comm.all_to_all_equal(dsend[:W * capacity].contiguous(), group)           # aligned with `recv`
        return None, None, None, local_ops.scatter_add(weight, recv, d_recv)


class ShardedEmbedding(nn.Module):
    """``nn.Embedding(num_embeddings, embedding_dim)`` whose rows live on ``id % world_size``.

    ``self.local`` is the shard's parameter holder (a real ``nn.Embedding``); checkpoints store
    one shard per rank.  ``padding_idx`` keeps its meaning on the owning rank."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, process_group=None, local_ops=None):
        super().__init__()
        rank, W = comm.world(process_group)
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.group, self.rank, self.world_size = process_group, rank, W
        n_local = (num_embeddings - rank + W - 1) // W
        local_pad = padding_idx // W if (padding_idx is not None and padding_idx % W == rank) else None
        self.local = nn.Embedding(max(n_local, 1), embedding_dim, padding_idx=local_pad)
        self.padding_idx = padding_idx
        self.local_ops = local_ops if local_ops is not None else HipLocalOps()

    @torch.no_grad()
    def load_full_table(self, full):
        """Copy this rank's rows out of a full [V, D] table (tests / checkpoint import)."""
        rows = torch.arange(self.rank, self.num_embeddings, self.world_size)
        self.local.weight[:rows.numel()].copy_(full[rows].to(self.local.weight.device))

    def forward(self, ids):
        flat = ids.long()
        return _ExactLookup.apply(flat % self.world_size, flat // self.world_size, self.group, self.local_ops,
                                  tuple(ids.shape), self.local.weight)

    def zero_pad_grad(self):
        """nn.Embedding(padding_idx) semantics for the shard that owns the pad row."""
        if self.local.padding_idx is not None and self.local.weight.grad is not None:
            self.local.weight.grad[self.local.padding_idx].zero_()


class ShardedTables(nn.Module):
    """T tables row-sharded together: one routing and one exchange per call for all of them.

    Rank r stores, back to back, its rows of every table (``base[r][t]`` = first local row of table t
    on rank r) in ONE packed weight ``self.weight`` of shape [rows_r, row_width]: floats [0, D) are
    the embedding row, float D is the table's dim-1 LR weight (``with_lr``), the rest pads the row to
    a multiple of 4 floats so that gathers, the wire format and the fused FM kernel stay float4-aligned.
    A lookup (t, id) goes to rank ``id % W``, local row ``base[id % W][t] + id // W`` -- computed by
    the requester, so only row numbers travel.

    ``capacity_factor``: None -> exact all-to-all-v (host sync); a float >= 1 -> padded, sync-free
    exchange with ``capacity = ceil(lookups / W * factor)`` slots per peer (see module docstring)."""

    def __init__(self, vocabs, embedding_dim, with_lr=True, capacity_factor=None, process_group=None, local_ops=None):
        super().__init__()
        rank, W = comm.world(process_group)
        self.vocabs, self.embedding_dim, self.with_lr = list(vocabs), embedding_dim, with_lr
        self.group, self.rank, self.world_size = process_group, rank, W
        self.lr_off = embedding_dim if with_lr else -1
        self.row_width = (embedding_dim + (1 if with_lr else 0) + 3) // 4 * 4
        counts = torch.tensor([[(v - r + W - 1) // W for v in self.vocabs] for r in range(W)], dtype=torch.long)
        base = torch.zeros_like(counts)
        base[:, 1:] = counts.cumsum(dim=1)[:, :-1]
        self.register_buffer("base", base, persistent=False)                       # [W, T]
        self.register_buffer("overflow", torch.zeros((), dtype=torch.bool), persistent=False)
        self.rows_local = int(counts[rank].sum())
        self.weight = nn.Parameter(torch.zeros(max(self.rows_local, 1), self.row_width))
        nn.init.normal_(self.weight[:, :embedding_dim + (1 if with_lr else 0)], std=1e-4)
        self.capacity_factor = capacity_factor
        self.local_ops = local_ops if local_ops is not None else HipLocalOps()

    @torch.no_grad()
    def load_full_tables(self, emb_tables, lr_tables=None):
        D = self.embedding_dim
        for t, v in enumerate(self.vocabs):
            rows = torch.arange(self.rank, v, self.world_size)
            lo = int(self.base[self.rank, t])
            self.weight[lo:lo + rows.numel(), :D].copy_(emb_tables[t][rows].to(self.weight.device))
            if self.with_lr and lr_tables is not None:
                self.weight[lo:lo + rows.numel(), D].copy_(lr_tables[t][rows].reshape(-1).to(self.weight.device))

    def local_rows_of(self, t):
        """(slice of local rows, global ids they hold) for table t on this rank."""
        lo = int(self.base[self.rank, t])
        ids = torch.arange(self.rank, self.vocabs[t], self.world_size)
        return slice(lo, lo + ids.numel()), ids

    def capacity_for(self, n_lookups):
        c = int(math.ceil(n_lookups / self.world_size * self.capacity_factor))
        return (c + 63) // 64 * 64

    def locate(self, ids):
        """ids [B, T] -> (owner rank, row number inside the owner's packed weight), both [B, T]."""
        ids = ids.long()
        W = self.world_size
        owner = ids % W
        t_index = torch.arange(ids.shape[1], device=ids.device).unsqueeze(0).expand_as(ids)
        return owner, self.base[owner, t_index] + ids // W

    def route(self, ids, capacity):
        """Wire slots of the padded exchange for ids ([B, T] tensor or a list of T id columns read in place):
        (slot [B, T], send [W * capacity] row numbers).  The work is the local backend's: rbx_route for
        ``HipLocalOps`` (three launches, no host sync)."""
        return self.local_ops.route(ids, self.world_size, capacity, self.base, self.overflow)

    def forward(self, ids):
        """ids [B, T] (one id per table per sample) -> packed rows [B, T, row_width]."""
        if self.capacity_factor is None:
            owner, row = self.locate(ids)
            return _ExactLookup.apply(owner, row, self.group, self.local_ops, tuple(ids.shape), self.weight)
        return _PaddedLookup.apply(ids, self, self.capacity_for(ids.numel()), self.weight)

    def split(self, packed):
        """packed [B, T, row_width] -> (E [B, T, D], L [B, T] or None) views."""
        E = packed[..., :self.embedding_dim]
        L = packed[..., self.embedding_dim] if self.with_lr else None
        return E, L
