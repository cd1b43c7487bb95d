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
            loss = step()           # replays fwd+bwd; parameter .grad tensors are static
    """

    def __init__(self, fn, warmup=3, capture_error_mode="global"):
        if ops.config.check_ids:
            raise RuntimeError("GraphedStep: set recbox_amd.ops.config.check_ids = False first "
                               "(the id range check syncs the host, which cannot be captured)")
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm up allocator + autograd on a side stream
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
