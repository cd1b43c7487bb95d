"""Host time per call of one ShardedFMStep (world of one through RCCL): where the Python thread spends the step.
    python profiles/ubench/sharded_host_time.py        (on the GPU box; prints microseconds per step and per call site)"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from recbox_amd import comm, ops
    from recbox_amd.graph import ShardedFMStep
    from recbox_amd.ranking.pytorch.models import ShardedFM
    ops.config.check_ids = False
    fmw = bench.CriteoFeatureMap(16)
    batch = bench.synthetic_batch(65536, 1, "uniform", dev)
    X, y = bench.slice_inputs(fmw.fm, batch)
    model = ShardedFM(fmw.fm, 16, shard_min_vocab=100000, capacity_factor=1.25).to(dev)
    bench.init_weights(model)
    step = ShardedFMStep(model, X, y, graphs=True)
    spent = defaultdict(float)

    def timed(name, fn):
        def run(*a, **k):
            t = time.perf_counter()
            out = fn(*a, **k)
            spent[name] += time.perf_counter() - t
            return out
        return run

    step.graphs = [timed("replay:" + p.__name__, g) for p, g in zip(step.pieces, step.graphs)]
    step.presort_replay = timed("replay:_presort", step.presort_replay)
    step.localsort_replay = timed("replay:_localsort", step.localsort_replay)

    class Comm(object):
        all_to_all_equal_into = staticmethod(timed("all_to_all_equal_into", comm.all_to_all_equal_into))
        all_reduce_sum_ = staticmethod(timed("all_reduce_sum_", comm.all_reduce_sum_))
    step.comm = Comm
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    spent.clear()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("host %.1f us/step to enqueue, %.1f us/step until the GPU is done" % (host / n * 1e6, total / n * 1e6))
    for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
        print("  %-28s %8.1f us/step" % (k, v / n * 1e6))
    print("  %-28s %8.1f us/step" % ("(rest: stream waits, python)", (host - sum(spent.values())) / n * 1e6))


if __name__ == "__main__":
    main()
