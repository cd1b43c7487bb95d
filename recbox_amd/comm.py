"""Collectives used by the sharded embedding path, on top of ``torch.distributed``.

On the GPU box the backend is "nccl", which on ROCm builds IS RCCL over xGMI: variable-size
all-to-all maps to ``all_to_all_single`` (grouped send/recv on the current HIP stream, one
direct xGMI link per peer on a fully connected 8-GPU node).  gloo (CPU tests) has no
all-to-all, so the same exchange is expressed with batched point-to-point there.
"""
import os

import torch
import torch.distributed as dist


# bench.py --force-sharded / tests: issue the all-reduces of the N > 1 step in a world of one too (RCCL then runs them as
# a local copy), so that a one-GPU box prices -- and exercises -- the step exactly as N > 1 ranks launch it
force_world_of_one = os.environ.get("RECBOX_AMD_FORCE_COLLECTIVES", "0") != "0"


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def multi(group=None):
    """Does this process group need its reductions issued (more than one rank, or ``force_world_of_one``)?"""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or force_world_of_one


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
        if direct.usable(x, group):
            direct.all_to_all(out, x, group, False)
        else:
            dist.all_to_all_single(out, x, group=group)
        return out
    c = x.shape[0] // W
    return all_to_all_rows(x, [c] * W, [c] * W, group)


class _Done(object):
    """Handle of a collective that has already been enqueued in stream order."""

    def wait(self):
        return True


class _Joined(object):
    """Handle of a collective issued on the side stream: ``wait()`` orders the ISSUING stream behind the collective itself
    (an event recorded right after it), not behind whatever was enqueued on the side stream later."""

    def __init__(self, cur, side):
        self.cur = cur
        self.done = torch.cuda.Event()
        self.done.record(side)

    def wait(self):
        # torch's Work.wait() contract: the stream current at wait() is ordered behind the collective -- and the issuing
        # stream too when that is another one (DenseGradSync issues from an autograd hook, waits in finish(): ADVICE r5)
        now = torch.cuda.current_stream(self.cur.device)
        now.wait_event(self.done)
        if now != self.cur:
            self.cur.wait_event(self.done)
        return True


_RBX_DTYPE = {torch.int32: 0, torch.int64: 1, torch.float32: 2, torch.float64: 3}
_RBX_OP = {"sum": 0, "max": 2, "min": 3}


class _Direct(object):
    """RCCL's collectives issued on the CALLER's stream through the C ABI (rbx_all_to_all = grouped ncclSend / ncclRecv,
    rbx_all_reduce, rbx_all_gather), with the communicator torch.distributed built.  torch.distributed runs every
    collective on RCCL's own stream behind two event joins (15-50 us of idle GPU apiece between short dependent pieces) and
    its wrapper refuses hipGraph capture; issued here the collectives are ordinary nodes of the step's stream, so a whole
    sharded step -- exchanges and all-reduces included -- captures into ONE hipGraph (recbox_amd.graph.GraphedStep).

    ``RECBOX_AMD_DIRECT_RCCL``: "auto" (default) -- the first collective of a process group on a GPU tensor runs
    ``self_check`` (every rank reaches it at the same call: collectives are issued in the same order everywhere), which
    compares this path with torch.distributed's on a known pattern, eagerly AND replayed from a captured hipGraph, and
    keeps it only if every rank agrees; "1" the same check, but a failure raises instead of falling back; "0" off."""

    def __init__(self):
        self.mode = os.environ.get("RECBOX_AMD_DIRECT_RCCL", "auto").lower()
        self.on = False
        self.capturable = False
        self.checked = set()
        self.bound = False
        self.stream = None              # for exchanges that overlap with compute (async_op)
        self.keep = None
        self.why = "not checked yet"

    def enable(self, on=True):
        """Force the policy: True = "1" (check, raise on failure), False = off, "auto" = check, fall back on failure."""
        self.mode = "auto" if on == "auto" else ("1" if on else "0")
        self.checked.clear()
        if not on:
            self.on = self.capturable = False

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
                _lib.check(_lib.lib.rbx_comm_bind_collectives(ctypes.cast(rccl.ncclAllReduce, ctypes.c_void_p),
                                                              ctypes.cast(rccl.ncclAllGather, ctypes.c_void_p),
                                                              ctypes.cast(rccl.ncclCommUserRank, ctypes.c_void_p)))
                self.keep, self.bound = rccl, True
                return
            except (OSError, AttributeError) as exc:
                err = exc
        raise RuntimeError("recbox_amd.comm: cannot bind RCCL (%s)" % (err,))

    @staticmethod
    def _live_ptr(group, device):
        """The RCCL communicator torch.distributed holds for ``group`` on ``device`` RIGHT NOW (0 = none yet).  Read on
        every use -- two pybind calls -- instead of cached: a communicator pointer kept across
        ``destroy_process_group()`` / ``init_process_group()`` would dangle, and ``id(group)`` (``id(None)`` for the
        default group) says nothing about which group is meant."""
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        try:
            return int(pg._get_backend(torch.device("cuda", device.index))._comm_ptr())
        except (RuntimeError, AttributeError):
            return 0

    def comm_ptr(self, group, device):
        ptr = self._live_ptr(group, device)
        if ptr == 0:
            raise RuntimeError("recbox_amd.comm: the process group has no RCCL communicator yet")
        return ptr

    def usable(self, x, group):
        """Should a collective on ``x`` take this path?  Runs the one-time check of the group when due.  A group is known
        by (device, its live communicator): a re-created default group has a new communicator and gets a new check."""
        if self.mode == "0" or not x.is_cuda:
            return False
        if not (dist.is_available() and dist.is_initialized()):
            self.checked.clear()                # whatever was checked belonged to a group that is gone
            self.on = self.capturable = False
            return False
        if dist.get_backend(group) != "nccl":
            return False
        ptr = self._live_ptr(group, x.device)
        if (x.device.index, ptr) not in self.checked:
            # (ptr == 0 -- no communicator yet, or a torch build without _comm_ptr -- is a state like any other: checked
            #  once, not on every collective; a communicator that appears later is a new pointer and gets its check)
            if torch.cuda.is_current_stream_capturing():
                return self.on and ptr != 0     # (the check syncs the host: a capture must come after a warm-up step)
            self.self_check(group, x.device)
            self.checked.add((x.device.index, ptr))
            if self._live_ptr(group, x.device) == 0:
                self.on = self.capturable = False       # nothing to issue on: torch.distributed's own path
        return self.on and self._live_ptr(group, x.device) != 0

    def _streams(self, x, async_op):
        cur = torch.cuda.current_stream(x.device)
        if not async_op:
            return cur, cur
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=x.device)
        self.stream.wait_stream(cur)
        return cur, self.stream

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
        cur, st = self._streams(x, async_op)
        _lib.check(_lib.lib.rbx_all_to_all(ctypes.c_void_p(comm), ctypes.c_void_p(x.data_ptr()),
                                           ctypes.c_void_p(out.data_ptr()), nbytes // W, W,
                                           ctypes.c_void_p(st.cuda_stream)))
        return _Joined(cur, st) if async_op else _Done()

    def all_reduce(self, x, group, op="sum", async_op=False):
        """In place over the ranks of ``group``; x: contiguous int32 / int64 / float32 / float64."""
        import ctypes
        from . import _lib
        if not self.bound:
            self._bind()
        if not x.is_contiguous() or x.dtype not in _RBX_DTYPE:
            raise ValueError("all_reduce: a contiguous int32 / int64 / float32 / float64 tensor")
        comm = self.comm_ptr(group, x.device)
        cur, st = self._streams(x, async_op)
        _lib.check(_lib.lib.rbx_all_reduce(ctypes.c_void_p(comm), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                           x.numel(), _RBX_DTYPE[x.dtype], _RBX_OP[op], ctypes.c_void_p(st.cuda_stream)))
        return _Joined(cur, st) if async_op else _Done()

    def all_gather(self, out, x, group):
        import ctypes
        from . import _lib
        if not self.bound:
            self._bind()
        if not (x.is_contiguous() and out.is_contiguous()):
            raise ValueError("all_gather: contiguous buffers")
        comm = self.comm_ptr(group, x.device)
        cur, st = self._streams(x, False)
        _lib.check(_lib.lib.rbx_all_gather(ctypes.c_void_p(comm), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           x.numel() * x.element_size(), ctypes.c_void_p(st.cuda_stream)))
        return _Done()

    def self_check(self, group=None, device=None):
        """One small exchange + all-reduce + all-gather both ways (this path and torch.distributed's) on a known pattern,
        eagerly and replayed from a captured hipGraph; keeps the direct path (``on``) only if every rank got identical
        results, and marks it ``capturable`` only if the replays did too.  Collective: every rank must call it at the same
        point (``usable`` does, at the first collective of a group)."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"):
            return False
        rank, W = dist.get_rank(group), dist.get_world_size(group)
        device = device or torch.device("cuda", torch.cuda.current_device())
        if self.mode == "0":
            self.on = self.capturable = False
            return False
        x = (torch.arange(W * 6, dtype=torch.float32, device=device) + 1000.0 * rank).reshape(W * 3, 2)
        want = torch.empty_like(x)
        dist.all_to_all_single(want, x, group=group)
        r_want = x.clone()
        dist.all_reduce(r_want, group=group)
        m_want = torch.tensor([rank, -rank], dtype=torch.int64, device=device)
        dist.all_reduce(m_want, op=dist.ReduceOp.MAX, group=group)
        g_want = x.new_empty((W,) + tuple(x.shape))
        dist.all_gather_into_tensor(g_want.view(-1), x.view(-1), group=group)
        self.checked.add((device.index, self._live_ptr(group, device)))     # (the collectives above created it)
        ok = torch.zeros(2, device=device)
        err = None
        try:
            got = torch.empty_like(x)
            self.all_to_all(got, x, group, False)
            again = torch.empty_like(x)
            self.all_to_all(again, x, group, True).wait()
            r_got = x.clone()
            self.all_reduce(r_got, group)
            r_async = x.clone()
            self.all_reduce(r_async, group, async_op=True).wait()
            m_got = torch.tensor([rank, -rank], dtype=torch.int64, device=device)
            self.all_reduce(m_got, group, op="max")
            g_got = torch.empty_like(g_want)
            self.all_gather(g_got, x, group)
            same = (torch.equal(got, want) and torch.equal(again, want) and torch.equal(r_got, r_want)
                    and torch.equal(r_async, r_want) and torch.equal(m_got, m_want) and torch.equal(g_got, g_want))
            ok[0] = 1.0 if same else 0.0
        except Exception as exc:                      # noqa: BLE001 -- any failure means "use torch.distributed"
            err = exc
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks take the same path
        eager_ok = bool(ok[0].item() > 0)
        cap_ok = False
        if eager_ok:
            # the same three collectives captured into a hipGraph and replayed twice on fresh inputs
            try:
                xs = x.clone()
                a_out, r_buf = torch.empty_like(x), torch.empty_like(x)
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                        self.all_to_all(a_out, xs, group, False)
                        r_buf.copy_(xs)
                        self.all_reduce(r_buf, group, async_op=True).wait()
                torch.cuda.current_stream(device).wait_stream(side)
                good = True
                for k in range(2):
                    xs.copy_(x + 7.0 * k)
                    graph.replay()
                    a_want, rr_want = torch.empty_like(x), (x + 7.0 * k)
                    dist.all_to_all_single(a_want, x + 7.0 * k, group=group)
                    dist.all_reduce(rr_want, group=group)
                    good = good and torch.equal(a_out, a_want) and torch.equal(r_buf, rr_want)
                ok[1] = 1.0 if good else 0.0
                # a hipGraph that holds RCCL kernel nodes must be gone before the communicator is destroyed:
                # destroy_process_group() hangs behind a live one (profiles/r04/teardown.txt: "keep" hangs, "release" leaves
                # in 0.4 s) -- so this one goes right away, and GraphedStep / ShardedFMStep have ``release()``
                torch.cuda.synchronize(device)
                del graph
            except Exception as exc:                  # noqa: BLE001
                err = err or exc
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            cap_ok = bool(ok[1].item() > 0)
        self.on, self.capturable = eager_ok, cap_ok
        self.why = ("ok" if eager_ok else "self-check failed%s" % ((": %s: %s" % (type(err).__name__, err)) if err else ""))
        if not eager_ok:
            if self.mode == "1":
                raise RuntimeError("recbox_amd.comm: RECBOX_AMD_DIRECT_RCCL=1 but the direct RCCL path is unusable (%s)" % self.why)
            import warnings
            warnings.warn("recbox_amd.comm: direct RCCL collectives unavailable (%s); using torch.distributed" % self.why)
        return self.on

    def shutdown(self):
        """Before ``destroy_process_group()``: forget the communicators (the next group gets its own check).  Graphs that
        captured collectives are the caller's to release first (``GraphedStep.release`` / ``ShardedFMStep.release``)."""
        self.checked.clear()
        self.on = self.capturable = False


direct = _Direct()


def all_to_all_equal_into(out, x, group=None, async_op=False):
    """all_to_all_equal into a caller-owned buffer (static buffers between hipGraph pieces).  Returns a handle
    whose ``wait()`` orders the current stream after the exchange; with ``async_op`` the exchange runs on RCCL's
    own stream and work enqueued before ``wait()`` overlaps with it."""
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(x)
    elif dist.get_backend(group) == "nccl":
        if direct.usable(x, group):
            return direct.all_to_all(out, x, group, async_op)
        work = dist.all_to_all_single(out, x, group=group, async_op=async_op)
        if async_op:
            return work
    else:
        out.copy_(all_to_all_equal(x, group))
    return _Done()


def all_reduce_sum_(x, group=None, async_op=False, force=False):
    """In-place SUM over the ranks (``force`` / ``force_world_of_one``: issued in a world of one too)."""
    if multi(group) or (force and dist.is_available() and dist.is_initialized()):
        if direct.usable(x, group) and x.is_contiguous() and x.dtype in _RBX_DTYPE:
            return direct.all_reduce(x, group, "sum", async_op)
        work = dist.all_reduce(x, group=group, async_op=async_op)
        if async_op:
            return work
    return _Done()


def _storage_spans(tensors, max_gap=16, min_cover=0.9):
    """Group gradient tensors by the storage they live in: [(span, members)] where ``span`` is ONE contiguous 1-D view of
    the storage range the members occupy -- when they are contiguous, tile the range up to alignment gaps of a few
    elements and share dtype -- else (None, [tensor]).  The dense gradients of one lookup are views of one flat buffer
    (ops._flat_zero_grads): reducing the span in place needs no flatten / un-flatten copies (65 copy kernels per FM step)."""
    by = {}
    for t in tensors:
        by.setdefault((t.untyped_storage().data_ptr(), t.dtype), []).append(t)
    out = []
    for (_, dtype), ts in by.items():
        ok = all(t.is_contiguous() for t in ts)
        if ok and len(ts) > 1:
            ts = sorted(ts, key=lambda t: t.storage_offset())
            lo, hi = ts[0].storage_offset(), max(t.storage_offset() + t.numel() for t in ts)
            end = lo
            for t in ts:
                if t.storage_offset() < end or t.storage_offset() - end > max_gap:
                    ok = False
                    break
                end = t.storage_offset() + t.numel()
            if ok and sum(t.numel() for t in ts) >= min_cover * (hi - lo):
                span = torch.empty(0, dtype=dtype, device=ts[0].device).set_(ts[0].untyped_storage(), lo, (hi - lo,))
                out.append((span, ts))
                continue
        for t in ts:
            out.append((None, [t]))
    return out


def all_reduce_coalesced_(tensors, group=None, async_op=False):
    """In-place SUM of every tensor over the ranks, with as few collectives as their memory layout allows: tensors that are
    views of one flat buffer are reduced as ONE span of it, in place (no flatten, no copy back); loose tensors are packed
    into one flat buffer and copied back on ``wait()``.  Every rank must pass the same layout.  Returns a handle."""
    if not tensors or not multi(group):
        return _Done()
    spans = _storage_spans([t for t in tensors])
    loose = [ts[0] for span, ts in spans if span is None]
    handles = []
    for span, ts in spans:
        if span is not None:
            handles.append(all_reduce_sum_(span, group, async_op))
    back = None
    if len(loose) == 1 and loose[0].is_contiguous():
        handles.append(all_reduce_sum_(loose[0], group, async_op))
    elif loose:
        flat = torch._utils._flatten_dense_tensors(loose)
        handles.append(all_reduce_sum_(flat, group, async_op))
        back = (flat, loose)
    return _Coalesced(handles, back)


class _Coalesced(object):
    def __init__(self, handles, back):
        self.handles, self.back = handles, back

    def wait(self):
        for h in self.handles:
            h.wait()
        if self.back is not None:
            flat, loose = self.back
            for t, r in zip(loose, torch._utils._unflatten_dense_tensors(flat, loose)):
                t.copy_(r)
            self.back = None
        return True


def all_gather_rows(x, group=None):
    """x [r, c] on every rank -> [W * r, c], rank order (the statistics of a synchronised BatchNorm).  gloo gathers host
    memory only: device rows are staged through the host there (tests on one GPU; the production backend is RCCL)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x.clone()
    W = dist.get_world_size(group)
    x = x.contiguous()
    if dist.get_backend(group) == "nccl":
        out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
        if direct.usable(x, group):
            direct.all_gather(out, x, group)
        else:
            dist.all_gather_into_tensor(out, x, group=group)
        return out
    host = x.cpu()
    parts = [torch.empty_like(host) for _ in range(W)]
    dist.all_gather(parts, host, group=group)
    return torch.cat(parts, dim=0).to(x.device)


def all_reduce_max_(x, group=None):
    """In-place MAX over the ranks (status words, flags, batch-size checks)."""
    if multi(group):
        if direct.usable(x, group) and x.is_contiguous() and x.dtype in _RBX_DTYPE:
            direct.all_reduce(x, group, "max")
        else:
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
        all_reduce_sum_(p.grad, group)
