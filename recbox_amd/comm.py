"""Collectives used by the sharded embedding path, on top of ``torch.distributed``.

On the GPU box the backend is "nccl", which on ROCm builds IS RCCL over xGMI: variable-size
all-to-all maps to ``all_to_all_single`` (grouped send/recv on the current HIP stream, one
direct xGMI link per peer on a fully connected 8-GPU node).  gloo (CPU tests) has no
all-to-all, so the same exchange is expressed with batched point-to-point there.
"""
import torch
import torch.distributed as dist


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def exchange_counts(send_counts, group=None):
    """send_counts: int64 tensor [W] (rows this rank sends to each peer) -> recv_counts list[W]."""
    rank, W = world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return [int(send_counts[0])]
    if dist.get_backend(group) == "nccl":
        recv = torch.empty_like(send_counts)
        dist.all_to_all_single(recv, send_counts, group=group)
        return recv.tolist()
    send_counts = send_counts.cpu()                       # gloo moves host memory only
    table = [torch.empty_like(send_counts) for _ in range(W)]
    dist.all_gather(table, send_counts, group=group)
    return [int(t[rank]) for t in table]


def all_to_all_rows(x, send_counts, recv_counts, group=None):
    """x: [sum(send_counts), ...] rows grouped by destination rank -> [sum(recv_counts), ...]
    rows grouped by source rank."""
    rank, W = world(group)
    out = x.new_empty((sum(recv_counts),) + tuple(x.shape[1:]))
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(x)
        return out
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, x.contiguous(), list(recv_counts), list(send_counts), group=group)
        return out
    # gloo: batched point-to-point with the same semantics.  gloo has no device transport: device rows are
    # staged through the host (2-rank-on-one-GPU tests; the production backend is RCCL above)
    if x.is_cuda:
        return all_to_all_rows(x.cpu(), send_counts, recv_counts, group).to(x.device)
    ops, so, ro = [], 0, 0
    x = x.contiguous()
    for peer in range(W):
        s, r = send_counts[peer], recv_counts[peer]
        if peer == rank:
            out[ro:ro + r].copy_(x[so:so + s])
        else:
            if s > 0:
                ops.append(dist.P2POp(dist.isend, x[so:so + s].contiguous(), peer, group=group))
            if r > 0:
                ops.append(dist.P2POp(dist.irecv, out[ro:ro + r], peer, group=group))
        so += s
        ro += r
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def all_to_all_equal(x, group=None):
    """x: [W * c, ...] (block p goes to rank p) -> [W * c, ...] (block p came from rank p).  Equal
    splits need no size exchange, so on RCCL this is a single capturable collective."""
    rank, W = world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return x.clone()
    x = x.contiguous()
    if dist.get_backend(group) == "nccl":
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x, group=group)
        return out
    c = x.shape[0] // W
    return all_to_all_rows(x, [c] * W, [c] * W, group)


class _Done(object):
    """Handle of a collective that has already been enqueued in stream order."""

    def wait(self):
        return True


def all_to_all_equal_into(out, x, group=None, async_op=False):
    """all_to_all_equal into a caller-owned buffer (static buffers between hipGraph pieces).  Returns a handle
    whose ``wait()`` orders the current stream after the exchange; with ``async_op`` the exchange runs on RCCL's
    own stream and work enqueued before ``wait()`` overlaps with it."""
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(x)
    elif dist.get_backend(group) == "nccl":
        work = dist.all_to_all_single(out, x, group=group, async_op=async_op)
        if async_op:
            return work
    else:
        out.copy_(all_to_all_equal(x, group))
    return _Done()


def all_reduce_sum_(x, group=None, async_op=False):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(x, group=group, async_op=async_op)
        if async_op:
            return work
    return _Done()


def all_reduce_grads(params, group=None):
    """Data-parallel towers: one all-reduce (sum) per dense gradient."""
    _, W = world(group)
    if W == 1:
        return
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad, group=group)
