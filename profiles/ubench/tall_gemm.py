"""Kernel breakdown of the tall-skinny Linear of SASRec ([B*L, 64] x [64, 64], fwd + bwd): run under rocprofv3 --kernel-trace --stats.
    python profiles/ubench/tall_gemm.py [M] [K] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from recbox_amd import ops  # noqa: E402

M, K, N = (int(a) for a in (sys.argv[1:4] + ["819200", "64", "64"][len(sys.argv) - 1:]))
x = torch.randn(M, K, device="cuda", requires_grad=True)
w = torch.randn(N, K, device="cuda", requires_grad=True)
b = torch.randn(N, device="cuda", requires_grad=True)
g = torch.randn(M, N, device="cuda")
for _ in range(12):
    x.grad = w.grad = b.grad = None
    y = ops.linear(x, w, b)
    y.backward(g)
torch.cuda.synchronize()
