"""Owner-side gather of the packed [D | LR | pad] rows (ShardedTables) in isolation: row width, id order, empty slots.
    python profiles/ubench/serve_gather.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from recbox_amd.sharded import HipLocalOps  # noqa: E402


def timeit(fn, n=20):
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(12):
        big.fill_(0.0)                       # pre-load the queue: the events then bracket GPU time only
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


def main():
    ops = HipLocalOps()
    V, n = 5521898, 655360
    g = torch.Generator().manual_seed(0)
    for width in (16, 20, 24, 32):
        w = torch.randn(V, width, device="cuda")
        rows = torch.randint(0, V, (n,), generator=g).cuda()
        holes = rows.clone()
        holes[torch.rand(n, generator=g).cuda() < 0.2] = -1
        srt = torch.sort(rows).values
        for name, r in (("random", rows), ("20% empty", holes), ("sorted", srt)):
            us = timeit(lambda: ops.gather(w, r))
            print("width %2d floats (%3d B)  ids %-9s  %6.1f us   %5.2f G rows/s" % (width, width * 4, name, us, n / us / 1e3))
        del w


if __name__ == "__main__":
    main()
