"""Row-sharded embedding table over the GPUs of one node.

The reference has no model-parallel embedding (single ``torch.device``; DataParallel / DDP only in
vendored trainers); this is the MI355X-native scaling path for tables that should not
be replicated: ``owner(id) = id % W``, local row ``id // W``.  One exchange each way:

  forward   bucket ids by owner -> all-to-all-v(ids) -> owner gathers rows (rbx_embed_fwd) ->
            all-to-all-v(rows) back -> un-permute
  backward  permute dY -> all-to-all-v(dY) -> owner scatter-adds into its shard's dense grad
            (rbx_embed_sort + rbx_embed_bwd: sorted, segmented, deterministic)

xGMI is point-to-point (7 links per GPU): an all-to-all keeps every link busy at once, so the
exchange time is max_peer_bytes / link_bw rather than a ring's 2(N-1)/N * S / link_bw.
Small tables stay replicated (``comm.all_reduce_grads``): cheaper than exchanging.
"""
import torch
from torch import nn

from . import comm


class HipLocalOps(object):
    """Local gather / scatter-add of one shard through the C ABI."""

    def gather(self, weight, rows):
        from . import _embed_host as host
        from ._lib import FIELD_CATEGORICAL
        holder = _Holder(weight)
        plan = host.Plan([host.Lookup("shard", FIELD_CATEGORICAL, holder, weight.shape[1])])
        return plan.run([rows]).detach()

    def scatter_add(self, weight, rows, dy):
        """dense [n_local, D] gradient of ``gather`` w.r.t. weight."""
        from . import ops
        from ._lib import FIELD_CATEGORICAL, POOL_NONE, check, lib
        n = rows.numel()
        grad = torch.zeros_like(weight)
        if n == 0:
            return grad
        spec = ops.FieldSpec("shard", FIELD_CATEGORICAL, weight.shape[1], 0, param=0, pool=POOL_NONE,
                             vocab=weight.shape[0])
        plan = ops.EmbedPlan([spec], weight.shape[1])
        plan.bind_inputs([rows])
        plan.bind_params([weight], [grad])
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, 1, n)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=weight.device)
        st = ops._stream()
        check(lib.rbx_embed_sort(plan.arr, 1, n, ops._ptr(ws), ws_bytes, None, st))
        dy = dy.contiguous()
        check(lib.rbx_embed_bwd(plan.arr, 1, n, ops._ptr(dy), dy.stride(0), None, 0, ops._ptr(ws), ws_bytes, st))
        return grad


class _Holder(nn.Embedding):
    """nn.Embedding view over an existing weight tensor (no copy), for the planner."""

    def __init__(self, weight):
        nn.Module.__init__(self)
        self.num_embeddings, self.embedding_dim = weight.shape
        self.padding_idx = None
        self.weight = weight


class _ShardedLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, group, local_ops):
        rank, W = comm.world(group)
        flat = ids.reshape(-1).long()
        owner = flat % W
        perm = torch.argsort(owner, stable=True)                 # rows grouped by destination rank
        send_rows = (flat // W)[perm]
        send_counts_t = torch.bincount(owner, minlength=W)
        send_counts = send_counts_t.tolist()                     # sizes of the variable all-to-all (host sync)
        recv_counts = comm.exchange_counts(send_counts_t, group)
        recv_rows = comm.all_to_all_rows(send_rows, send_counts, recv_counts, group)      # ids out
        vecs = local_ops.gather(weight, recv_rows)                                        # owner-side gather
        back = comm.all_to_all_rows(vecs, recv_counts, send_counts, group)                # rows back
        out = torch.empty_like(back)
        out[perm] = back
        ctx.save_for_backward(perm, recv_rows, weight)
        ctx.meta = (send_counts, recv_counts, group, local_ops, ids.shape)
        return out.view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dout):
        perm, recv_rows, weight = ctx.saved_tensors
        send_counts, recv_counts, group, local_ops, shape = ctx.meta
        d_sorted = dout.reshape(-1, weight.shape[1])[perm].contiguous()
        d_recv = comm.all_to_all_rows(d_sorted, send_counts, recv_counts, group)          # dY to the owners
        grad = local_ops.scatter_add(weight, recv_rows, d_recv)
        return None, grad, None, None


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
        out = _ShardedLookup.apply(ids, self.local.weight, self.group, self.local_ops)
        return out

    def zero_pad_grad(self):
        """nn.Embedding(padding_idx) semantics for the shard that owns the pad row."""
        if self.local.padding_idx is not None and self.local.weight.grad is not None:
            self.local.weight.grad[self.local.padding_idx].zero_()
