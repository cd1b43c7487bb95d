"""Collectives used by the sharded embedding path, on top of ``torch.distributed``.

On the GPU box the backend is "nccl", which on ROCm builds IS RCCL over xGMI: variable-size
all-to-all maps to ``all_to_all_single`` (grouped send/recv on the current HIP stream, one
direct xGMI link per peer on a fully connected 8-GPU node).  gloo (CPU tests) has no
all-to-all, so the same exchange is expressed with batched point-to-point there.
"""
import os

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


class _Joined(object):
    def __init__(self, cur, side):
        self.cur, self.side = cur, side

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)
        return True


class _Direct(object):
    """RCCL's grouped send/recv issued on the caller's stream through the C ABI (rbx_all_to_all), with the
    communicator torch.distributed built.  Off unless ``RECBOX_AMD_DIRECT_RCCL=1`` / ``comm.direct.enable()``:
    torch.distributed's own call stays the default and the fallback."""

    def __init__(self):
        self.on = os.environ.get("RECBOX_AMD_DIRECT_RCCL", "0") != "0"
        self.bound = False
        self.comms = {}
        self.stream = None              # for exchanges that overlap with compute (async_op)
        self.keep = None

    def enable(self, on=True):
        self.on = bool(on)

    def _bind(self):
        import ctypes
        import glob
        from . import _lib
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so.1", "librccl.so"]
        err = None
        for path in cands:
            try:
                rccl = ctypes.CDLL(path)          # already loaded by torch: this only takes a handle
                fns = [ctypes.cast(getattr(rccl, n), ctypes.c_void_p)
                       for n in ("ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString")]
                _lib.check(_lib.lib.rbx_comm_bind(*fns))
                self.keep, self.bound = rccl, True
                return
            except (OSError, AttributeError) as exc:
                err = exc
        raise RuntimeError("recbox_amd.comm: cannot bind RCCL (%s)" % (err,))

    def comm_ptr(self, group, device):
        key = (id(group), device.index)
        ptr = self.comms.get(key)
        if ptr is None:
            pg = group if group is not None else dist.distributed_c10d._get_default_group()
            ptr = int(pg._get_backend(torch.device("cuda", device.index))._comm_ptr())
            if ptr == 0:
                raise RuntimeError("recbox_amd.comm: the process group has no RCCL communicator yet")
            self.comms[key] = ptr
        return ptr

    def all_to_all(self, out, x, group, async_op):
        import ctypes
        from . import _lib
        if not self.bound:
            self._bind()
        W = dist.get_world_size(group)
        if not (x.is_contiguous() and out.is_contiguous()) or x.numel() * x.element_size() != out.numel() * out.element_size():
            raise ValueError("all_to_all_equal_into: contiguous buffers of equal size")
        nbytes = x.numel() * x.element_size()
        if nbytes % W:
            raise ValueError("all_to_all_equal_into: %d bytes do not split over %d ranks" % (nbytes, W))
        comm = self.comm_ptr(group, x.device)
        cur = torch.cuda.current_stream(x.device)
        st = cur
        if async_op:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=x.device)
            st = self.stream
            st.wait_stream(cur)
        _lib.check(_lib.lib.rbx_all_to_all(ctypes.c_void_p(comm), ctypes.c_void_p(x.data_ptr()),
                                           ctypes.c_void_p(out.data_ptr()), nbytes // W, W,
                                           ctypes.c_void_p(st.cuda_stream)))
        return _Joined(cur, st) if async_op else _Done()


    def self_check(self, group=None, device=None):
        """One small exchange both ways (this path and torch.distributed's) on a known pattern; switches the direct
        path off, with a warning, if it raises or differs.  Every rank must call it."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"):
            return self.on
        rank, W = dist.get_rank(group), dist.get_world_size(group)
        device = device or torch.device("cuda", torch.cuda.current_device())
        x = (torch.arange(W * 6, dtype=torch.float32, device=device) + 1000.0 * rank).reshape(W * 3, 2)
        want = torch.empty_like(x)
        dist.all_to_all_single(want, x, group=group)
        ok = torch.zeros(1, device=device)
        try:
            got = torch.empty_like(x)
            self.all_to_all(got, x, group, False)
            again = torch.empty_like(x)
            self.all_to_all(again, x, group, True).wait()
            ok.fill_(1.0 if (torch.equal(got, want) and torch.equal(again, want)) else 0.0)
        except Exception as exc:                      # noqa: BLE001 -- any failure means "use torch.distributed"
            import warnings
            warnings.warn("recbox_amd.comm: direct RCCL exchange unavailable (%s: %s)" % (type(exc).__name__, exc))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks take the same path
        self.on = bool(ok.item() > 0)
        return self.on


direct = _Direct()


def all_to_all_equal_into(out, x, group=None, async_op=False):
    """all_to_all_equal into a caller-owned buffer (static buffers between hipGraph pieces).  Returns a handle
    whose ``wait()`` orders the current stream after the exchange; with ``async_op`` the exchange runs on RCCL's
    own stream and work enqueued before ``wait()`` overlaps with it."""
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(x)
    elif dist.get_backend(group) == "nccl":
        if direct.on:
            return direct.all_to_all(out, x, group, async_op)
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


def all_gather_rows(x, group=None):
    """x [r, c] on every rank -> [W * r, c], rank order (the statistics of a synchronised BatchNorm).  gloo gathers host
    memory only: device rows are staged through the host there (tests on one GPU; the production backend is RCCL)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x.clone()
    W = dist.get_world_size(group)
    x = x.contiguous()
    if dist.get_backend(group) == "nccl":
        out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    host = x.cpu()
    parts = [torch.empty_like(host) for _ in range(W)]
    dist.all_gather(parts, host, group=group)
    return torch.cat(parts, dim=0).to(x.device)


def all_reduce_max_(x, group=None):
    """In-place MAX over the ranks (status words, flags, batch-size checks)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=group)
    return x


def all_reduce_grads(params, group=None):
    """Data-parallel towers: one all-reduce (sum) per dense gradient."""
    _, W = world(group)
    if W == 1:
        return
    for p in params:                    # the same sequence of collectives on every rank: a missing gradient counts as zeros
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        dist.all_reduce(p.grad, group=group)
