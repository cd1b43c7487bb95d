"""cProfile of the eager (no hipGraph) single-GPU step of bench.py: where the Python thread spends its ~1.3 ms.
    python profiles/ubench/eager_host_profile.py [--fresh-grads]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def main():
    from recbox_amd import ops
    from recbox_amd.ranking.pytorch.models import FM
    from recbox_amd.ranking.pytorch.torch_utils import get_loss
    dev = torch.device("cuda", 0)
    ops.config.check_ids = False
    ops.config.reuse_grad_buffers = "--fresh-grads" not in sys.argv
    fmw = bench.CriteoFeatureMap(16)
    X, y = bench.slice_inputs(fmw.fm, bench.synthetic_batch(65536, 1, "uniform", dev))
    model = FM(fmw.fm, 16, fused=True).to(dev)
    bench.init_weights(model)
    loss_fn = get_loss("binary_crossentropy")

    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(X)["y_pred"], y, reduction="mean")
        loss.backward()

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("host %.0f us/step to enqueue, %.0f us/step until the GPU is done" % (host / 200 * 1e6, (time.perf_counter() - t0) / 200 * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
