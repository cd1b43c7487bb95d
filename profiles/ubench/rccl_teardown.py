"""EXPERIMENT (world of one through RCCL): what makes destroy_process_group() hang after RCCL kernels were captured into a
hipGraph (rounds 2-3: it hung, the whole-step capture stayed out of bench.py), and what leaves cleanly.

    python -u profiles/ubench/rccl_teardown.py <variant>      (run every variant under `timeout 90`)

variants: nocapture | keep | release | release_sync | abort | shutdown
  nocapture     direct collectives issued eagerly only, then destroy_process_group()
  keep          capture + replay, the graph object still alive at destroy_process_group()
  release       capture + replay, graph dropped (del + gc) before destroy
  release_sync  the same + torch.cuda.synchronize() between the two
  abort         capture + replay, graph alive, the backend's abort() instead of destroy
  shutdown      capture + replay, graph dropped, backend.shutdown() then destroy
Prints "<variant>: left cleanly in X s" or is killed by the timeout."""
import gc
import os
import sys
import time
sys.path.insert(0, ".")
import torch
import torch.distributed as dist

variant = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29561")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from recbox_amd import comm
os.environ["RECBOX_AMD_DIRECT_RCCL_CAPTURE"] = "0"          # the self-check's own capture is what is under test here
comm.direct.enable(True)
assert comm.direct.self_check(device=dev)
x = torch.randn(1 << 20, device=dev)
out = torch.empty_like(x)
r = x.clone()
comm.direct.all_to_all(out, x, None, False)
comm.direct.all_reduce(r, None)
torch.cuda.synchronize()
assert torch.equal(out, x) and torch.equal(r, x)
g = None
if variant != "nocapture":
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            comm.direct.all_to_all(out, x, None, False)
            r.copy_(x)
            comm.direct.all_reduce(r, None, async_op=True).wait()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, x) and torch.equal(r, x)
print("%s: collectives ok, leaving" % variant, flush=True)
t0 = time.perf_counter()
if variant in ("release", "release_sync", "shutdown"):
    g = None
    gc.collect()
if variant in ("release_sync", "shutdown"):
    torch.cuda.synchronize()
backend = dist.distributed_c10d._get_default_group()._get_backend(dev)
if variant == "abort":
    backend.abort()
elif variant == "shutdown":
    backend.shutdown()
    dist.destroy_process_group()
else:
    dist.destroy_process_group()
print("%s: left cleanly in %.2f s" % (variant, time.perf_counter() - t0), flush=True)
