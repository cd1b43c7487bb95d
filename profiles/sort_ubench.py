"""The id sort of the fused FM backward (rbx_fm_sort) on its own, at the bench shape: us per call (eager launches, so the
host's launch cost is in it), nothing beside it.  Run it under rocprofv3 --kernel-trace --stats for the kernels' own times.
    python profiles/sort_ubench.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recbox_amd.ranking.pytorch.models import FM


def main():
    dev = torch.device("cuda:0")
    fmw = bench.CriteoFeatureMap(16)
    model = FM(fmw.fm, 16, fused=True).to(dev)
    batch = bench.synthetic_batch(65536, 1, "uniform", dev)
    X, _ = bench.slice_inputs(fmw.fm, batch)
    res = model.presort(X)
    for _ in range(5):
        res = model.presort(X, into=res)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    a.record()
    for _ in range(n):
        res = model.presort(X, into=res)
    b.record()
    torch.cuda.synchronize()
    print("rbx_fm_sort: %.1f us per call" % (a.elapsed_time(b) / n * 1e3))


if __name__ == "__main__":
    main()
