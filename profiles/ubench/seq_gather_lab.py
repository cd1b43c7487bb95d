"""embed_seq_kernel at BASELINE cfg 3's shape (10 M x 128 table, B = 65 536, history <= 50 mean-pooled): the forward alone, for
A/B builds of the kernel's parameters (RECBOX_HIP_LIB=...).  Run on the GPU box:  python profiles/ubench/seq_gather_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd.rechub.basic.features import SequenceFeature, SparseFeature  # noqa: E402
from recbox_amd.rechub.basic.layers import EmbeddingLayer  # noqa: E402
from recbox_amd import ops  # noqa: E402
from profiles.ubench.kernels_bench import timeit  # noqa: E402

if __name__ == "__main__":
    ops.config.check_ids = False
    V, D, B, L = 10_000_000, 128, 65536, 50
    item = SparseFeature("item", V, D)
    hist = SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)
    with torch.device("cuda"):
        layer = EmbeddingLayer([item, hist])
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    h = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    x = {"hist": h.cuda()}
    nnz = int(lens.sum())
    with torch.no_grad():
        ts = [timeit(lambda: layer(x, [hist], squeeze_dim=True), iters=20) for _ in range(3)]
    t = min(ts)
    byt = nnz * (D * 4 + 8) + B * D * 4
    print("%-40s %7.1f us  %6.0f GB/s (%.3f of 8 TB/s)  [three runs: %s]" % (os.path.basename(os.environ.get("RECBOX_HIP_LIB", "default")), t * 1e6,
          byt / t / 1e9, byt / t / 8e12, " ".join("%.1f" % (q * 1e6) for q in ts)))
