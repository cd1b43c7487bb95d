"""hipGraph capture of a whole training step (forward + loss + backward).

The FM step is ~15 short kernels; launched eagerly from Python it is host-bound
(1.2 ms of interpreter time for 0.3-0.5 ms of GPU work on MI355X).  Capturing the
step once and replaying it removes the per-step host cost: this is the
"HIP graphs instead of a tracing compiler" half of the design.  The captured
region runs the SAME kernels through the same C ABI; inputs live in static
buffers that the caller refills (``copy_`` or in-place loader writes) before
each replay.
"""
import torch

from . import ops


class GraphedStep(object):
    """Capture ``fn()`` (which must read its inputs from pre-allocated tensors and leave
    its results in tensors reachable from the returned object) and replay it.

    Usage::
        step = GraphedStep(lambda: train_step(static_batch), warmup=3)
        for batch in loader:
            static_batch.copy_(batch, non_blocking=True)
            loss = step()           # replays fwd+bwd; parameter .grad tensors are static: consume them
                                    # (optimizer step) before the next replay

    ``fn`` may run SEVERAL steps in a row, each over its own static batch (a loader that keeps S batches resident): a replay
    then pays the graph-to-graph latency of the runtime -- ~12 us more than the gap between two steps inside one graph on
    MI355X, 5 % of the FM step -- once per S steps (``bench.py --steps-per-graph``, profiles/r06/fm_steps_per_graph.txt).
    What consumes the gradients (an optimiser step) then belongs inside ``fn`` too: after a replay ``p.grad`` holds the
    LAST step's gradients.
    """

    def __init__(self, fn, warmup=3, capture_error_mode="global", reuse_grads=True, check_every=0, params=None,
                 pool=None):
        # params: parameters whose ``.grad`` this step produces.  Several GraphedSteps over ONE model (one graph per
        # resident input buffer, replayed in rotation) each own the gradient tensors of their capture: with ``params``
        # a call re-points every ``p.grad`` at the tensors THIS graph wrote (the table gradients of the persistent
        # buffer are the same memory in every graph; the small dense ones are not).
        # pool: a ``torch.cuda.graph_pool_handle()`` / another graph's ``.pool()`` to share intermediate memory with
        # (graphs that are never replayed concurrently).
        # check_every: every that many replays, read the deferred id-range status word (one host sync) and raise
        # IndexError if a replay met an id outside its table (0 = never: call ops.check_deferred_ids() yourself)
        self.check_every, self.replays = int(check_every), 0
        # reuse_grads: False / True (the fused FM body) / "all" (also the generic lookup: ops.config.reuse_grad_buffers)
        if ops.config.check_ids:
            raise RuntimeError("GraphedStep: set recbox_amd.ops.config.check_ids = False first "
                               "(the id range check syncs the host, which cannot be captured)")
        self.fn = fn
        # Parameter gradients of a replayed step are static tensors that the next replay overwrites.  Under exactly
        # that contract the fused FM backward may keep ONE dense gradient buffer and clear only the rows the previous
        # replay wrote instead of re-filling 379 MB of zeros per step (ops.config.reuse_grad_buffers): ``fn`` must
        # start from ``p.grad = None`` for the embedding parameters, as a captured step does anyway.
        old = ops.config.reuse_grad_buffers
        ops.config.reuse_grad_buffers = "all" if "all" in (reuse_grads, old) else (bool(reuse_grads) or old)
        # Several GraphedSteps over the same parameters (one per rotating batch) warm up on a side stream each: the
        # parameters' AccumulateGrad nodes remember the first one, and autograd says so once per process ("stream does not
        # match ..."; it adds the cross-stream wait itself).  Intended here, so the notice is off for warm-up and capture.
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # warm up allocator + autograd on a side stream
                for _ in range(max(warmup, 2)):     # (two steps: the second one records the re-zero path)
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, pool=pool, capture_error_mode=capture_error_mode):
                self.out = fn()
            # ``stream``: the stream of the capture -- the one the parameters' AccumulateGrad nodes remember as long as the
            # captured graph's outputs keep them alive.  Eager steps between replays that run there
            # (``with torch.cuda.stream(step.stream)``) meet no cross-stream accumulate.  (ONE stream for warm-up and capture
            # of every GraphedStep was measured: the replayed FM step went from 0.234 to 0.234-0.273 ms, run to run.)
            self.stream = getattr(torch.cuda.graph, "default_capture_stream", None) or side
        finally:
            ops.config.reuse_grad_buffers = old
            if quiet is not None:
                quiet(True)
        self.grads = [(p, p.grad) for p in params] if params is not None else None

    def pool(self):
        return self.graph.pool()

    def release(self):
        """Drop the captured graph (a capture that holds RCCL kernel nodes must go before ``destroy_process_group()``)."""
        self.graph = None
        self.out = None

    def __call__(self):
        if ops._dropout_ticks:
            ops.bump_dropout_tick()          # a replay re-runs the captured seeds: the device tick makes the masks new
        self.graph.replay()
        if self.grads is not None:
            for p, g in self.grads:
                p.grad = g
        self.replays += 1
        if self.check_every and self.replays % self.check_every == 0:
            ops.check_deferred_ids()
        return self.out


class ShardedFMStep(object):
    """One training step of ``ShardedFM`` (row-sharded tables, padded sync-free exchange) as hipGraph
    pieces with the RCCL collectives launched between them:

        route      ids -> (owner, row) -> wire slots (rbx_route)          [graph]
        all-to-all row numbers to the owners                              RCCL
        serve      owners gather the packed rows (rbx_embed_fwd)          [graph]
        presort    owners sort the received row numbers (rbx_embed_sort)  [graph, side stream, joined before settle]
        localsort  id sort of the replicated tables' backward (rbx_fm_sort; needs X only)
                                                                          [graph, own stream from the start of the step,
                                                                           beside route / exchange / serve; joined before tail]
        all-to-all rows back                                              RCCL
        head       fused FM forward (remote rows read at their wire slots),
                   loss, dL/dlogit, dL/d(remote rows) written to the slots [graph]
        all-to-all dL/d(rows) to the owners                               RCCL      } side stream, behind presort,
        settle     owners scatter-add into their shard's dense grad                 } beside the tail (round 5)
                   (rbx_embed_sort + rbx_embed_bwd)                       [graph]
        tail       fused backward of the replicated tables, flat grads    [graph]
        all-reduce of the flat gradient of the replicated parameters      RCCL, asynchronous
        finish     replicated grads un-flattened                          [graph]

    Capturing a collective inside a hipGraph is not dependable on this stack (round 1: the capture of a
    torch.distributed all_to_all_single hung), so the graphs stop at the collectives; the host cost per step is
    5 graph launches + 4 collectives instead of ~150 eager kernel launches (``graphs="whole"``: ONE launch).  ``graphs=False`` runs the same
    pieces eagerly (tests, debugging).  Inputs ``X`` (dict of static tensors) and ``y`` are read in place:
    refill them before every call.  After a call, ``.grad`` of every parameter is what
    ``(bce(model(X), y) / world).backward(); model.sync_grads()`` leaves.  ``persistent_shard_grad``: the dense gradient of
    this rank's table shard aliases ONE buffer from step to step, cleared by row (HipLocalOps.persistent) instead of
    zero-filled in full -- consume it before the next call."""

    def __init__(self, model, X, y, graphs=True, warmup=2, loss_fn=None, persistent_shard_grad=True):
        """graphs: False (eager pieces) / True (hipGraph pieces with the collectives between them) / "whole" (ONE hipGraph
        holding the pieces AND the collectives: needs the collectives on the step's own stream, ``comm.direct`` usable and
        capturable -- checked here, collectively; otherwise the pieces) / "auto" = "whole" when possible, else True."""
        from . import comm
        if model.tables is None or model.tables.capacity_factor is None:
            raise ValueError("ShardedFMStep needs row-sharded tables with the padded exchange (capacity_factor)")
        if graphs and ops.config.check_ids:
            raise RuntimeError("ShardedFMStep: set recbox_amd.ops.config.check_ids = False first")
        self.model, self.X, self.y = model, X, y
        self.tables = tables = model.tables
        self.group = tables.group
        self.W = W = tables.world_size
        self.loss_fn = loss_fn or ops.binary_cross_entropy      # the ranking harness's mean BCE on sigmoid outputs
        self.fused_loss = loss_fn is None                       # ... which this step then evaluates on the logits in one pass
        ids = model.sharded_ids(X)
        self.B, self.T = ids.shape
        self.cap = cap = tables.capacity_for(ids.numel())
        dev, width = ids.device, tables.row_width
        # wire buffers: written by one piece / collective, read by the next
        self.recv = torch.full((W * cap,), -1, dtype=getattr(tables.local_ops, "wire_dtype", torch.long), device=dev)
        self.back = torch.zeros((W * cap, width), dtype=torch.float32, device=dev)
        self.d_recv = torch.zeros((W * cap, width), dtype=torch.float32, device=dev)
        self.replicated = [p for p in model.replicated_parameters() if p.requires_grad]
        self.sizes = [p.numel() for p in self.replicated]
        self.comm = comm
        self.pieces = [self._route, self._serve, self._head, self._tail, self._settle, self._finish]
        self.early = torch.cuda.Stream(device=dev)       # id sort of the replicated tables: beside route / exchange / serve
        # the owner-side id sort runs beside the rows' way back and the local forward, on a stream of its own
        self.side = torch.cuda.Stream(device=dev)
        self.sorted_ws = None
        self.local_sorted = None
        self.graphs = None
        self.reduced = None
        persistent = persistent_shard_grad and hasattr(tables.local_ops, "persistent")
        if persistent:
            # the shard's dense gradient: one buffer cleared by row instead of a full zero fill per step
            if graphs and warmup < 1:
                # the row re-zero is part of the captured owner-side sort only if a step has run before the capture (the
                # decision "are there rows to clear" is taken on the host, at capture time): without one, replays would
                # never clear what the replays before them stored
                raise ValueError("ShardedFMStep(graphs=True, persistent_shard_grad=True) needs warmup >= 1")
            tables.local_ops.persistent(tables.weight)
        self.multi = comm.multi(self.group)          # more than one rank (or a world of one told to issue its reductions)
        if not self.multi:
            self.pieces[-1] = lambda: None          # nothing to un-flatten without an all-reduce (and no empty graph to replay)
        self.whole = None
        if graphs in ("whole", "auto"):
            # every rank takes the same decision: usable() runs the collective self-check of the group when it is due
            ok = bool(comm.direct.usable(self.back, self.group) and comm.direct.capturable)
            if graphs == "whole" and not ok:
                import warnings
                warnings.warn("ShardedFMStep: the collectives cannot be captured on this stack (%s); hipGraph pieces instead"
                              % comm.direct.why)
            graphs = "whole" if ok else True
        if graphs == "whole":
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 1)):
                    self._run(self.pieces)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            keep = getattr(tables.local_ops, "_keep", None)
            if persistent and keep is not None and self.W * cap > 0 and not keep.get("dirty"):
                raise RuntimeError("ShardedFMStep: the warm-up left no rows to clear; the captured sort would never re-zero the "
                                   "persistent shard gradient")
            self.whole = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.whole, capture_error_mode="thread_local"):
                self._run(self.pieces)
        elif graphs:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):           # allocator, plans, RCCL communicator: warm before capturing
                for _ in range(warmup):
                    self._run(self.pieces)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graphs = []
            keep = getattr(tables.local_ops, "_keep", None)
            if persistent and keep is not None and self.W * cap > 0 and not keep.get("dirty"):
                raise RuntimeError("ShardedFMStep: the warm-up left no rows to clear; the captured sort would never re-zero the "
                                   "persistent shard gradient")
            for piece in self.pieces:
                if piece is self.pieces[-1]:
                    self.graphs.append(piece)           # (the join of the asynchronous all-reduce: a stream wait, not a graph)
                    continue
                if piece == self._head:                 # the two id sorts are captured on their own streams first
                    gs = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gs, stream=self.side, capture_error_mode="thread_local"):
                        self._presort()
                    self.presort_replay = gs.replay
                    gl = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gl, stream=self.early, capture_error_mode="thread_local"):
                        self._localsort()
                    self.localsort_replay = gl.replay
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    piece()
                self.graphs.append(g.replay)

    # ---- pieces (each one a fixed kernel sequence over static buffers) ----------------------------------
    def _route(self):
        # rbx_route reads the id columns of the batch in place (float64 / int columns, strided views)
        slot, self.send = self.tables.route([self.X[n] for n in self.model.sharded_names], self.cap)
        self.slot = slot.contiguous()

    def _serve(self):
        self.vecs = self.tables.local_ops.gather(self.tables.weight, self.recv)

    def _head(self):
        for p in self.replicated:
            p.grad = None
        # the fused FM kernel reads row (b, t) at wire slot slot[b, t] of the exchange buffer (no un-permute pass).
        # Fresh gradients here: _finish writes the ALL-REDUCED gradients into p.grad, i.e. rows this rank's batch never
        # touched -- a persistent buffer that is re-zeroed by this rank's sorted ids (ops.config.reuse_grad_buffers)
        # would keep them.
        reuse = ops.config.reuse_grad_buffers
        ops.config.reuse_grad_buffers = reuse and not self.multi
        try:
            self.logit = self.model.logits(self.X, packed=self.back, packed_index=self.slot,
                                           presorted=self.local_sorted)
        finally:
            ops.config.reuse_grad_buffers = reuse
        if self.fused_loss:                       # sigmoid + mean BCE + both backward steps: two kernels instead of nine
            self.loss, self.dlogit = ops.sigmoid_bce(self.logit, self.y, grad_scale=1.0 / self.W)
        else:
            leaf = self.logit.detach().requires_grad_()
            loss = self.loss_fn(torch.sigmoid(leaf), self.y)
            (loss / self.W).backward()            # global-mean loss: owners sum the contributions of every rank
            self.loss, self.dlogit = loss.detach(), leaf.grad
        # dL/d(remote rows) goes to the same wire slots and leaves for the owners while the local backward runs
        self.dsend = ops.fm_extra_grad(self.logit, self.dlogit, self.back, self.slot, self.tables.lr_off)
        ops.join_early_sort(self.logit)           # (a sort started by the forward itself must end inside this piece)

    def _tail(self):
        self.logit.backward(self.dlogit)          # fused backward of the replicated tables / numeric weights / bias
        # every gradient of the fused backward is a view of ONE flat buffer (ops._flat_zero_grads): the all-reduce takes
        # that buffer as it lies (comm.all_reduce_coalesced_) -- no torch.cat before it, no per-parameter copy after it
        # (round 3's _finish: 65 copy kernels per step, invisible in a world of one that skipped the reduction)
        for p in self.replicated:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self.flat = [p.grad for p in self.replicated]

    def _localsort(self):
        # ids of the replicated tables -> sorted (row, sample) pairs for the fused backward; needs X only
        self.local_sorted = self.model.presort_local(self.X)

    def _presort(self):
        self.sorted_ws = self.tables.local_ops.presort(self.tables.weight, self.recv)

    def _settle(self):
        self.tables.weight.grad = self.tables.clear_pad_grad(
            self.tables.local_ops.scatter_add(self.tables.weight, self.recv, self.d_recv, sorted_ws=self.sorted_ws))

    def _finish(self):
        if self.reduced is not None:
            self.reduced.wait()                   # (loose gradients, if any, are copied back here)
            self.reduced = None

    # ---- the step ------------------------------------------------------------------------------------------
    def _run(self, pieces):
        route, serve, head, tail, settle, finish = pieces
        comm, group = self.comm, self.group
        cur = torch.cuda.current_stream()
        # the id sort of the replicated tables' backward depends on X only: it fills the GPU while route -> exchange ->
        # serve -> exchange (short, dependent, partly on the wire) are under way, and is joined before the tail
        self.early.wait_stream(cur)
        with torch.cuda.stream(self.early):
            (self.localsort_replay if pieces is self.graphs else self._localsort)()
        route()
        comm.all_to_all_equal_into(self.recv, self.send, group)
        serve()
        # the owners' id sort needs only the row numbers: it runs on a side stream beside the rows' way back and
        # the local forward, and is joined right before the owner-side scatter-add
        # (enqueued BEHIND the rows' way back: in a replayed graph a kernel that waits for another hardware queue starts late --
        #  with the sort captured first, the exchange started 76 us after the gather it depends on, profiles/r05)
        served = cur.record_event()
        comm.all_to_all_equal_into(self.back, self.vecs, group)
        self.side.wait_event(served)
        with torch.cuda.stream(self.side):
            (self.presort_replay if pieces is self.graphs else self._presort)()
        head()
        # Round 5: the owners' half of the backward -- dL/d(rows) to the owners, scatter-add into the shard's gradient -- goes
        # to the side stream, behind the owners' id sort that is already there, and runs BESIDE the fused backward of the
        # replicated tables on this stream (before: exchange -> tail -> settle in a row on this one; world of one through
        # RCCL: 0.446 -> 0.402 ms, profiles/r05/sharded_owner_beside.txt)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            self.dsend.record_stream(self.side)
            comm.all_to_all_equal_into(self.d_recv, self.dsend, group)
            settle()
            g = self.tables.weight.grad
            if g is not None:
                g.record_stream(cur)
        cur.wait_stream(self.early)
        tail()
        self.reduced = comm.all_reduce_coalesced_(self.flat, group, async_op=True) if (self.flat and self.multi) else None
        cur.wait_stream(self.side)
        finish()
        return self.loss

    def __call__(self):
        if self.whole is not None:
            self.whole.replay()
            return self.loss
        return self._run(self.graphs if self.graphs is not None else self.pieces)

    def release(self):
        """Drop the captured graphs (they hold RCCL kernel nodes when the collectives were captured): call it -- and
        synchronise -- before ``destroy_process_group()``."""
        self.whole = None
        self.graphs = None
        self.presort_replay = self.localsort_replay = None
