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
        # persistent(): the dense gradient of the shard lives in ONE buffer that is cleared by row -- exactly the rows the
        # previous scatter_add stored, named by the sorted ids its presort left in the workspace -- instead of being
        # zero-filled in full at every step (the local shard of the Criteo-shaped tables is 379 MB / world of zeros per
        # step: 0.38 ms of the 0.66 ms step in a world of one).  Only for callers that sort and reduce a FIXED number of row
        # slots per step, one presort and one scatter_add each, in that order (recbox_amd.graph.ShardedFMStep).
        self._keep = None
        # row numbers travel as int32 (rbx_route32): a shard has fewer than 2^31 rows or rbx_embed_fwd could not address it
        self.wire_dtype = torch.int32

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
        """Wire slots of the padded exchange: (slot [B, T] int32, send [world * capacity] int32 row numbers) -- rbx_route32."""
        from . import ops
        send, slot = ops.route(ids, world, capacity, base, overflow, wire=self.wire_dtype)
        return slot, send

    def persistent(self, weight):
        """Switch to one persistent gradient buffer for this shard (see __init__)."""
        self._keep = {"grad": torch.zeros_like(weight), "ws": None, "ws_bytes": 0, "dirty": 0}

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
        keep = self._keep
        if keep is not None and keep["grad"].shape == weight.shape:
            plan.bind_params([weight.detach()], [keep["grad"]])
            ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
            if keep["ws"] is None or keep["ws_bytes"] != ws_bytes:
                if keep["dirty"]:
                    raise RuntimeError("persistent shard gradient: the number of row slots changed between steps")
                keep["ws"] = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
                keep["ws_bytes"] = ws_bytes
            if keep["dirty"]:                     # the rows the previous step stored: still named by the sorted ids in ws
                check(lib.rbx_embed_rezero(plan.arr, 1, keep["dirty"], ops._ptr(keep["ws"]), ws_bytes, ops._stream()))
                keep["dirty"] = 0
            check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(keep["ws"]), ws_bytes, None, ops._stream()))
            return keep["ws"]
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
        keep = self._keep
        kept = (keep is not None and keep["grad"].shape == weight.shape and sorted_ws is not None
                and sorted_ws is keep["ws"])
        grad = keep["grad"] if kept else torch.zeros_like(weight)
        if n == 0:
            return grad
        plan = self._plan(weight)
        plan.bind_inputs([rows])
        plan.bind_params([weight.detach()], [grad])
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        st = ops._stream()
        if kept:
            keep["dirty"] = n                     # (the reduce below STORES the sums of the rows it touches)
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
    def forward(ctx, owner, row, group, local_ops, out_shape, weight, tables=None):
        ctx.tables = tables
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
        grad = local_ops.scatter_add(weight, recv_rows, d_recv)
        if ctx.tables is not None:
            ctx.tables.clear_pad_grad(grad)
        return None, None, None, None, None, grad, None


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
        ctx.tables = tables
        return out.view(*ids.shape, width)

    @staticmethod
    def backward(ctx, dout):
        slot, recv, weight = ctx.saved_tensors
        capacity, group, local_ops, W = ctx.meta
        width = weight.shape[1]
        dsend = dout.new_zeros((W * capacity + 1, width))
        dsend[slot] = dout.reshape(-1, width)                     # slots are unique except the dump slot (discarded)
        d_recv = comm.all_to_all_equal(dsend[:W * capacity].contiguous(), group)           # aligned with `recv`
        return None, None, None, ctx.tables.clear_pad_grad(local_ops.scatter_add(weight, recv, d_recv))


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


def raise_if_overflowed(module, group, what):
    """Collective: has ANY rank's padded exchange run out of slots since the last call?  The flag is a device word the
    routing kernels raise instead of dropping lookups silently (a dropped lookup reads a zero row and trains nothing); this
    reads it (one host sync), agrees on it across the ranks, clears it and raises on EVERY rank -- so that no rank is left
    waiting in the next collective.  Call it once per step, or every N steps, from every rank."""
    flag = module.overflow.detach().float().reshape(1).clone()
    if comm.world(group)[1] > 1:
        import torch.distributed as dist
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    module.overflow.zero_()
    if float(flag.item()) > 0:
        raise RuntimeError("%s: an owner received more lookups from one rank than the exchange has slots for "
                           "(capacity_factor = %g); the affected lookups read zero rows.  Rebuild with a larger "
                           "capacity_factor (skewed ids need more than the default 1.25), or capacity_factor=None for "
                           "the exact all-to-all-v." % (what, module.capacity_factor))


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

    def __init__(self, vocabs, embedding_dim, with_lr=True, capacity_factor=None, process_group=None, local_ops=None,
                 padding_idx=None):
        """padding_idx: None, or one entry per table (None / the row that is nn.Embedding's ``padding_idx`` in the reference's
        table: it receives no gradient -- on the rank that owns it the row's gradient is cleared after the scatter-add -- and,
        loaded as the zero row it is there, reads as zero)."""
        super().__init__()
        rank, W = comm.world(process_group)
        self.vocabs, self.embedding_dim, self.with_lr = list(vocabs), embedding_dim, with_lr
        pads = list(padding_idx) if padding_idx is not None else [None] * len(self.vocabs)
        if len(pads) != len(self.vocabs):
            raise ValueError("ShardedTables: one padding_idx entry per table")
        self.padding_idx = pads
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
        mine = [int(base[rank, t]) + p // W for t, p in enumerate(pads) if p is not None and 0 <= p < self.vocabs[t] and p % W == rank]
        self.register_buffer("pad_rows", torch.tensor(mine, dtype=torch.long), persistent=False)
        if mine:
            with torch.no_grad():
                self.weight[self.pad_rows] = 0.0

    def clear_pad_grad(self, grad):
        """nn.Embedding(padding_idx) on the rank that owns a padding row: that row's gradient is zero (in place)."""
        if grad is not None and self.pad_rows.numel() > 0:
            grad.index_fill_(0, self.pad_rows, 0.0)
        return grad

    def raise_if_overflowed(self):
        """Collective; see ``raise_if_overflowed`` above."""
        if self.capacity_factor is not None:
            raise_if_overflowed(self, self.group, "ShardedTables")

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
            return _ExactLookup.apply(owner, row, self.group, self.local_ops, tuple(ids.shape), self.weight, self)
        return _PaddedLookup.apply(ids, self, self.capacity_for(ids.numel()), self.weight)

    def split(self, packed):
        """packed [B, T, row_width] -> (E [B, T, D], L [B, T] or None) views."""
        E = packed[..., :self.embedding_dim]
        L = packed[..., self.embedding_dim] if self.with_lr else None
        return E, L


# ------------------------------------------------------------------------------------------------------------
# Second generation: ONE exchange each way, pooled lookups reduced at the owner (csrc/rbx_shard.hip).
# BASELINE.json cfg 3 (YoutubeDNN over a row-sharded 10 M x 128 item table) and cfg 4 (DeepFM, large tables
# sharded, the rest data-parallel) run on this.
# ------------------------------------------------------------------------------------------------------------
class ShardCall(object):
    """Static description of what one layer call asks of a ``ShardedStore`` per sample:
    ``rows``  list of (table index, column offset in the output row) -- one id per sample each;
    ``pool``  None or (table index, column offset, seq_len, "mean" | "sum", mask_id, eps) -- a padded id sequence
              pooled with an id mask (rechub InputMask + AveragePooling / SumPooling, layers.py:135-148,187-210)."""

    def __init__(self, rows, pool=None):
        self.rows = [(int(t), int(o)) for t, o in rows]
        self.pool = pool
        self.T, self.P = len(self.rows), (1 if pool is not None else 0)
        self.key = (tuple(self.rows), None if pool is None else tuple(pool))

    def geometry(self, W, D, B, factor):
        L = self.pool[2] if self.pool is not None else 0
        cap_rows = -(-int(math.ceil(B * self.T / W * factor)) // 64) * 64 if self.T else 0
        cap_pool = -(-int(math.ceil(B * L / W * factor)) // 64) * 64 if self.P else 0
        return ShardGeom(W, D, self.T, self.P, B, max(cap_rows, 64) if self.T else 0, max(cap_pool, 64) if self.P else 0)


class ShardGeom(object):
    """Sizes of the wire format (the C side derives the same numbers: rbx_shard_int_chunk / rbx_shard_float_rows)."""

    def __init__(self, W, D, T, P, B, cap_rows, cap_pool):
        self.W, self.D, self.T, self.P, self.B, self.cap_rows, self.cap_pool = W, D, T, P, B, cap_rows, cap_pool
        self.off_offs = cap_pool
        self.off_rows = cap_pool + P * (B + 1)
        self.ichunk = (self.off_rows + cap_rows + 3) // 4 * 4
        self.frows = P * B + cap_rows
        self.n_keys = W * (cap_pool + cap_rows)

    def c_struct(self):
        from ._lib import rbx_shard_geom_t
        return rbx_shard_geom_t(self.W, self.D, self.T, self.P, self.B, self.cap_rows, self.cap_pool)


class HipShardOps(object):
    """The local work of the exchange through the C ABI (rbx_shard_* + the sorted scatter-add)."""

    def __init__(self):
        self._plans = {}
        # persistent(): ONE dense gradient buffer per shard, cleared by row (the rows the previous scatter stored, named by
        # the sorted keys its presort left in that step's workspace: rbx_embed_rezero) instead of a fresh zero-filled
        # [rows, D] tensor per step -- 5 GB at cfg 3 in a world of one, 640 MB at W = 8.  The gradient handed to autograd
        # then ALIASES that buffer: valid until the next backward, every step must start from ``weight.grad is None``
        # (zero_grad(set_to_none=True)), and nothing may write other rows into it in place.
        self._keep = None

    def persistent(self, weight):
        """Switch this shard to one persistent gradient buffer (see __init__).  Returns self."""
        self._keep = {"grad": None, "shape": tuple(weight.shape), "ws": None, "ws_bytes": 0, "dirty": 0}
        return self

    def end_step(self):
        """Step boundary (``sync_grads()`` calls it): a sort left pending by a forward whose backward never ran -- a step
        skipped after a NaN loss, an exception, a grad-enabled evaluation -- is forgotten instead of refusing every later
        lookup.  The rows that forward's sort named were not stored into: nothing to clear."""
        if self._keep is not None:
            self._keep["pending"] = False

    def route(self, geom, call, row_ids, pool_ids, vocabs, base, overflow, status):
        """-> (send int32 [W * ichunk], slot int32 [B, T], inv fp32 [B])."""
        from . import ops
        from ._lib import POOL_MEAN_ID, POOL_SUM_ID, RBX_NO_ID, check, lib, rbx_field_t
        dev = base.device
        g = geom.c_struct()
        # the wire sizes are derived twice (here and in C): they must agree before a buffer is sized by them
        if lib.rbx_shard_int_chunk(g) != geom.ichunk or lib.rbx_shard_float_rows(g) != geom.frows:
            raise RuntimeError("recbox_amd.sharded: wire geometry disagrees with librecbox_hip (%d/%d vs %d/%d)"
                               % (geom.ichunk, geom.frows, lib.rbx_shard_int_chunk(g), lib.rbx_shard_float_rows(g)))
        send = torch.empty(geom.W * geom.ichunk, dtype=torch.int32, device=dev)
        slot = torch.empty((geom.B, geom.T), dtype=torch.int32, device=dev) if geom.T else None
        inv = torch.empty(geom.B, dtype=torch.float32, device=dev) if geom.P else None
        rows = (rbx_field_t * max(geom.T, 1))()
        keep = []
        for f, t, (tbl, _) in zip(rows, row_ids, call.rows):
            t = ops._id_column(t)
            keep.append(t)
            f.ids, f.ids_stride_b, f.ids_stride_l = t.data_ptr(), t.stride(0), 0
            f.ids_dtype, f.vocab, f.seq_len = ops._DTYPE_CODE[t.dtype], vocabs[tbl], 1
        pool = None
        L = 0
        if geom.P:
            tbl, _, L, mode, mask_id, eps = call.pool
            t = ops._id_column(pool_ids)
            keep.append(t)
            pool = rbx_field_t()
            pool.ids, pool.ids_stride_b, pool.ids_stride_l = t.data_ptr(), t.stride(0), t.stride(1)
            pool.ids_dtype, pool.vocab, pool.seq_len = ops._DTYPE_CODE[t.dtype], vocabs[tbl], L
            pool.pool = POOL_MEAN_ID if mode == "mean" else POOL_SUM_ID
            pool.mask_id = RBX_NO_ID if mask_id is None else int(mask_id)
            pool.eps = eps
        ws_bytes = lib.rbx_shard_route_workspace_size(g, L)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        check(lib.rbx_shard_route(g, rows if geom.T else None, pool, ops._ptr(base), ops._ptr(send), ops._ptr(slot),
                                  ops._ptr(inv), ops._ptr(overflow), ops._ptr(status), ops._ptr(ws), ws_bytes,
                                  ops._stream()))
        return send, slot, inv

    def serve(self, geom, recv, weight, status):
        """-> (back fp32 [W * frows, D], keys int32 [n_keys], src int32 [n_keys])."""
        from . import ops
        from ._lib import check, lib
        dev = weight.device
        back = torch.empty((geom.W * geom.frows, geom.D), dtype=torch.float32, device=dev)
        keys = torch.empty(geom.n_keys, dtype=torch.int32, device=dev)
        src = torch.empty(geom.n_keys, dtype=torch.int32, device=dev)
        check(ops._timed(("shard_serve", geom.W, geom.B, geom.D),
                         lambda: lib.rbx_shard_serve(geom.c_struct(), ops._ptr(recv), ops._ptr(weight), weight.shape[0],
                                                     ops._ptr(back), ops._ptr(keys), ops._ptr(src), ops._ptr(status),
                                                     ops._stream())))
        return back, keys, src

    @staticmethod
    def _offs(geom, call):
        import ctypes
        offs = [o for _, o in call.rows] + ([call.pool[1]] if call.pool is not None else [])
        return (ctypes.c_int64 * len(offs))(*offs)

    def combine_fwd(self, geom, call, back, slot, inv, out):
        from . import ops
        from ._lib import check, lib
        check(lib.rbx_shard_combine_fwd(geom.c_struct(), ops._ptr(back), ops._ptr(slot), ops._ptr(inv), ops._ptr(out),
                                        out.stride(0) if geom.B > 1 else out.shape[1], self._offs(geom, call),
                                        ops._stream()))

    def combine_bwd(self, geom, call, dout, slot, inv):
        from . import ops
        from ._lib import check, lib
        gsend = torch.empty((geom.W * geom.frows, geom.D), dtype=torch.float32, device=dout.device)
        check(lib.rbx_shard_combine_bwd(geom.c_struct(), ops._ptr(dout), dout.stride(0) if geom.B > 1 else dout.shape[1],
                                        self._offs(geom, call), ops._ptr(slot), ops._ptr(inv), ops._ptr(gsend),
                                        ops._stream()))
        return gsend

    def _plan(self, weight):
        from . import ops
        from ._lib import FIELD_CATEGORICAL, POOL_NONE
        key = tuple(weight.shape)
        plan = self._plans.get(key)
        if plan is None:
            spec = ops.FieldSpec("shard", FIELD_CATEGORICAL, weight.shape[1], 0, param=0, pool=POOL_NONE,
                                 vocab=weight.shape[0])
            plan = self._plans[key] = ops.EmbedPlan([spec], weight.shape[1])
        return plan

    def presort(self, weight, keys):
        """The id sort of ``scatter`` needs the received row numbers only: it may run right after ``serve``.  With a
        persistent gradient (``persistent``) the sorted keys live in ONE workspace from step to step: the rows the previous
        backward stored -- still named by it -- are cleared first, then this step's keys are sorted over them (a captured
        step replays exactly that: a workspace of the step's own would freeze the keys of the capture)."""
        from . import ops
        from ._lib import check, lib
        n = keys.numel()
        if n == 0:
            return None
        plan = self._plan(weight)
        plan.bind_inputs([keys])
        keep = self._keep
        if keep is not None and keep["shape"] == tuple(weight.shape):
            if keep.get("pending"):
                raise RuntimeError("recbox_amd.sharded: a persistent shard gradient serves ONE lookup of its store per step "
                                   "(a second exchange started before the first one's backward); use fresh gradients.  "
                                   "If the earlier forward was abandoned without its backward (a skipped step, an "
                                   "exception, a grad-enabled evaluation), call model.sync_grads() or "
                                   "store.local_ops.end_step() at the step boundary")
            if keep["grad"] is None:
                keep["grad"] = torch.zeros_like(weight)                 # the only full fill
            plan.bind_params([weight.detach()], [keep["grad"]])
            ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
            if keep["dirty"]:
                check(lib.rbx_embed_rezero(plan.arr, 1, keep["dirty"], ops._ptr(keep["ws"]), keep["ws_bytes"], ops._stream()))
                keep["dirty"] = 0
            if keep.get("own") is None or keep["own_bytes"] < ws_bytes:
                keep["own"] = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
                keep["own_bytes"] = ws_bytes
            keep["ws"], keep["ws_bytes"] = keep["own"], keep["own_bytes"]
            check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(keep["own"]), keep["own_bytes"], None, ops._stream()))
            keep["pending"] = True
            return keep["own"]
        plan.bind_params([weight.detach()], [weight.detach()])            # placeholder grad pointer: "trainable"
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
        check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(ws), ws_bytes, None, ops._stream()))
        return ws

    def scatter(self, weight, keys, src, grecv, sorted_ws=None):
        """Dense gradient of the shard: grad[keys[i]] += grecv[src[i]] (keys < 0 skipped), sorted + segmented."""
        from . import ops
        from ._lib import check, lib
        n = keys.numel()
        keep = self._keep
        kept = keep is not None and keep["shape"] == tuple(weight.shape) and n > 0
        plan = self._plan(weight)
        if kept:
            if weight.grad is not None:
                raise RuntimeError("recbox_amd.sharded: a persistent shard gradient needs weight.grad to be None at every "
                                   "backward (zero_grad(set_to_none=True)): the gradient aliases one buffer")
            if keep["grad"] is None:
                keep["grad"] = torch.zeros_like(weight)                 # the only full fill
            grad = keep["grad"]
            if keep["dirty"]:
                # (a backward whose sort was not made ahead by ``presort``:) the rows the previous backward stored are named
                # by the sorted keys in ITS workspace (kept alive here)
                plan.bind_inputs([keys])
                plan.bind_params([weight.detach()], [grad])
                check(lib.rbx_embed_rezero(plan.arr, 1, keep["dirty"], ops._ptr(keep["ws"]), keep["ws_bytes"], ops._stream()))
                keep["dirty"] = 0
        else:
            grad = torch.zeros_like(weight)
        if n == 0:
            return grad
        plan.bind_inputs([keys])
        plan.bind_params([weight.detach()], [grad])
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        ws = sorted_ws
        if ws is None:
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
            check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(ws), ws_bytes, None, ops._stream()))
        elif kept and ws is keep.get("own"):
            ws_bytes = keep["own_bytes"]
        check(lib.rbx_embed_bwd_indexed(plan.arr, 1, n, ops._ptr(grecv), grecv.stride(0), ops._ptr(src), None, 0,
                                        ops._ptr(ws), ws_bytes, ops._stream()))
        if kept:
            keep["ws"], keep["ws_bytes"], keep["dirty"] = ws, ws_bytes, n
            keep["pending"] = False
            # autograd takes a gradient over as the parameter's .grad only when nobody else holds the TENSOR OBJECT -- handed
            # the buffer itself it would clone all of it (5 GB at cfg 3: the step went from 3.4 to 5.2 ms); a fresh view is
            # a new object over the same memory
            return grad.view(grad.shape)
        return grad


class _ShardLookup(torch.autograd.Function):
    """The exchange as ONE autograd node: fills the sharded lookups' slots of ``block`` (the [B, width] output rows of
    a layer call; allocated here when the call has no replicated features), backward returns the shard's dense
    gradient and hands ``dblock`` on to the replicated lookups untouched."""

    @staticmethod
    def forward(ctx, store, call, width, block, weight, *ids):
        W, group, lo = store.world_size, store.group, store.local_ops
        row_ids = ids[:call.T]
        pool_ids = ids[call.T] if call.P else None
        B = (row_ids[0] if call.T else pool_ids).shape[0]
        store.check_batch(B)                   # the wire sizes are derived from B on every rank: it must be the same one
        geom = call.geometry(W, store.embedding_dim, B, store.capacity_factor)
        dev = weight.device
        had_block = block is not None
        if block is None:
            if dev.type == "cuda":               # 16-byte aligned rows for the GEMM / float4 kernels that consume it
                from . import ops
                block = ops._padded_rows(B, width, dev)
            else:
                block = torch.empty((B, width), dtype=torch.float32, device=dev)
        else:
            ctx.mark_dirty(block)
        status = None
        if store.check_ids():
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        base = store.base_for(call)
        send, slot, inv = lo.route(geom, call, row_ids, pool_ids, store.vocabs, base, store.overflow, status)
        recv = torch.empty_like(send)
        comm.all_to_all_equal_into(recv, send, group).wait()                          # requests to the owners
        back, keys, src = lo.serve(geom, recv, weight.detach(), status)
        # the owner's id sort needs only the received row numbers: it runs beside the rows' way back
        sort = None
        if ctx.needs_input_grad[4] and getattr(store, "_grad_mode", True):     # (a forward under torch.no_grad() sorts nothing)
            sort = store.early_sort(lo, weight, keys)
        got = torch.empty_like(back)
        comm.all_to_all_equal_into(got, back, group).wait()                           # rows / partial sums back
        lo.combine_fwd(geom, call, got, slot, inv, block)
        if status is not None:
            # bit 0 is raised on the requester, bit 1 on the owner: the ranks agree on it before any of them raises, or the
            # others would walk into the next collective alone
            if W > 1:
                comm.all_reduce_max_(status, group)
            if int(status.item()) != 0:
                raise IndexError("index out of range in self")
        ctx.store, ctx.call, ctx.geom = store, call, geom
        ctx.had_block = had_block
        ctx.sort = sort
        ctx.save_for_backward(slot, inv, keys, src, weight)
        return block

    @staticmethod
    def backward(ctx, dblock):
        store, call, geom = ctx.store, ctx.call, ctx.geom
        slot, inv, keys, src, weight = ctx.saved_tensors
        lo = store.local_ops
        if store.on_backward_start is not None:
            store.on_backward_start()           # every consumer of this block has finished its backward by now
        if dblock.stride(1) != 1 or dblock.dtype != torch.float32:
            dblock = dblock.contiguous().float()
        gsend = lo.combine_bwd(geom, call, dblock, slot, inv)
        grecv = torch.empty_like(gsend)
        comm.all_to_all_equal_into(grecv, gsend, store.group).wait()                  # gradients to the owners
        ws = None
        if ctx.sort is not None:
            ctx.sort.join()
            ws = ctx.sort.ws
        grad = lo.scatter(weight, keys, src, grecv, sorted_ws=ws)
        return (None, None, None, dblock if ctx.had_block else None, grad) + (None,) * (call.T + call.P)


class ShardedStore(nn.Module):
    """Tables of one embedding dimension, row-sharded together over the ranks of ``process_group``:
    ``owner(id) = id % W``; rank r keeps, back to back in ONE weight ``[rows_r, D]``, its rows ``id // W`` of every
    table (``base[r][t]`` = first of them).  Checkpoints hold one shard per rank.  A table has no ``padding_idx``
    here (rechub's tables have none, initializers.py:17; the id mask of a pooled lookup is the feature's).
    Every rank must call ``lookup`` with the SAME batch size (the wire format has static sizes derived from it): use
    ``drop_last=True`` or pad the last batch; ``check_batch`` verifies it collectively the first time a size is seen."""

    def __init__(self, vocabs, embedding_dim, capacity_factor=1.25, process_group=None, local_ops=None):
        super().__init__()
        rank, W = comm.world(process_group)
        if embedding_dim % 4:
            raise NotImplementedError("ShardedStore: embedding_dim must be a multiple of 4 (got %d)" % embedding_dim)
        self.vocabs, self.embedding_dim = [int(v) for v in vocabs], int(embedding_dim)
        self.group, self.rank, self.world_size = process_group, rank, W
        counts = torch.tensor([[(v - r + W - 1) // W for v in self.vocabs] for r in range(W)], dtype=torch.long)
        base = torch.zeros_like(counts)
        base[:, 1:] = counts.cumsum(dim=1)[:, :-1]
        self.register_buffer("base", base, persistent=False)                         # [W, n_tables]
        self.register_buffer("overflow", torch.zeros((), dtype=torch.uint8), persistent=False)
        self.rows_local = int(counts[rank].sum())
        self.weight = nn.Parameter(torch.empty(max(self.rows_local, 1), self.embedding_dim))
        nn.init.normal_(self.weight, std=1e-4)
        self.capacity_factor = float(capacity_factor)
        self.local_ops = local_ops if local_ops is not None else HipShardOps()
        self.on_backward_start = None          # optional hook called when the exchange's backward starts
        self._bases = {}
        self.check_batch_size = "auto"
        self._seen_batch = set()

    def check_ids(self):
        from . import ops
        return ops.config.check_ids

    def check_batch(self, B):
        """Every rank derives the static wire sizes of the exchange from ITS batch size: they only agree when all ranks
        hold the same number of samples (``drop_last=True`` in the loader, or pad the last batch).  ``check_batch_size``:
        "auto" (default): "always" for eager calls, nothing inside a hipGraph capture (see below).
        (That is one 16-byte all-reduce and a host synchronisation per store per eagerly launched step -- the price of not
        hanging in the exchange; a loop whose loader guarantees equal batches -- ``drop_last=True`` -- should set
        ``store.check_batch_size = "once"``; captured steps pay nothing either way.)
        "once" checks a size collectively the first time THIS rank sees it -- a set-up error raises ValueError on
        every rank at the first step instead of hanging in the all-to-all; it cannot see a size that changes on one rank only
        (the other ranks do not enter the check).  "always" checks every call (one 16-byte all-reduce + host sync); False
        never."""
        mode = self.check_batch_size
        if mode == "auto":
            # a captured step has ONE batch size by construction (checked in its warm-up); launched eagerly, a loader's short
            # last batch on one rank only would walk into the all-to-all with other wire sizes than its peers: check every
            # call there -- a check that only the rank seeing a NEW size enters is itself a mismatched collective
            mode = "once" if (self.weight.is_cuda and torch.cuda.is_current_stream_capturing()) else "always"
        if not mode or self.world_size == 1 or (mode == "once" and B in self._seen_batch):
            return
        t = torch.tensor([B, -B], dtype=torch.int64, device=self.weight.device)
        comm.all_reduce_max_(t, self.group)
        hi, lo = int(t[0].item()), -int(t[1].item())
        if hi != lo:
            raise ValueError("recbox_amd.sharded: the ranks hold different batch sizes (%d..%d, this rank %d); the exchange "
                             "needs the same number of samples on every rank (drop_last=True)" % (lo, hi, B))
        self._seen_batch.add(B)

    def early_sort(self, lo, weight, keys):
        """The owner's id sort needs the received row numbers only: side stream, joined before the scatter-add."""
        from . import ops
        if not weight.is_cuda or not hasattr(lo, "presort"):
            return None
        work = ops.SideWork(weight.device, lambda: lo.presort(weight, keys), uses=(keys,))
        work.ws = work.result
        return work if work.ws is not None else None

    def base_for(self, call):
        """[W, T + P] int64: first local row, on every owner, of the table each lookup of ``call`` addresses."""
        b = self._bases.get(call.key)
        if b is None or b.device != self.base.device:
            tbl = [t for t, _ in call.rows] + ([call.pool[0]] if call.pool is not None else [])
            b = self._bases[call.key] = self.base[:, tbl].contiguous()
        return b

    def raise_if_overflowed(self):
        """Collective; see ``raise_if_overflowed`` above."""
        raise_if_overflowed(self, self.group, "ShardedStore")

    @torch.no_grad()
    def load_full_tables(self, tables):
        for t, v in enumerate(self.vocabs):
            rows = torch.arange(self.rank, v, self.world_size)
            lo = int(self.base[self.rank, t])
            self.weight[lo:lo + rows.numel()].copy_(tables[t][rows].to(self.weight.device))

    def local_rows_of(self, t):
        """(slice of local rows, global ids they hold) for table t on this rank."""
        lo = int(self.base[self.rank, t])
        ids = torch.arange(self.rank, self.vocabs[t], self.world_size)
        return slice(lo, lo + ids.numel()), ids

    def lookup(self, call, width, row_ids, pool_ids=None, block=None):
        """Fill the slots of ``call`` in ``block`` [B, width] (None: a new block) and return it."""
        ids = list(row_ids) + ([pool_ids] if call.P else [])
        self._grad_mode = torch.is_grad_enabled()        # (inside Function.forward grad mode is always off: told from here)
        return _ShardLookup.apply(self, call, width, block, self.weight, *ids)
