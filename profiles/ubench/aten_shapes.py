"""Which ATen kernels (and on what shapes) are left in one eager step of a model mirror (configs 3-5)?
    python profiles/ubench/aten_shapes.py [deepfm|youtube|sasrec]
torch.profiler over 3 steps, grouped by op and input shape, device time per step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import models_bench as mb  # noqa: E402


def main():
    captured = {}

    def grab(name, step, *a, **k):
        captured["step"] = step

    mb.measure = grab
    getattr(mb, sys.argv[1] if len(sys.argv) > 1 else "deepfm")()
    step = captured["step"]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", 0.0)
        if dt > 20.0 * 3:
            rows.append((dt / 3.0, e.key, e.count // 3, str(e.input_shapes)[:150]))
    for dt, key, n, shp in sorted(rows, reverse=True)[:60]:
        print("%9.1f us/step  x%-3d %-42s %s" % (dt, n, key[:42], shp))


if __name__ == "__main__":
    main()
