"""The split-operand tower GEMMs at cfg 4's shapes (forward, dx, dW) for A/B builds (RECBOX_HIP_LIB=...).
Run on the GPU box:  python profiles/ubench/gemm_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from profiles.ubench.kernels_bench import timeit  # noqa: E402


def case(M, K, N):
    pitch = (K + 3) // 4 * 4
    x = torch.randn(M, pitch, device="cuda")[:, :K].requires_grad_(True)
    w = (torch.randn(N, K, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    with torch.no_grad():
        t = timeit(lambda: ops.linear(x, w, b, None), iters=20)
    y = ops.linear(x, w, b, None)
    g = torch.randn_like(y)
    tb = timeit(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True), iters=10)
    fl = 2.0 * M * K * N
    print("[%d,%d]x[%d,%d]^T  fwd %7.1f us (%.3f of the bf16 pipes)   dx + dW + db %7.1f us (%.3f)   checksum %.6e"
          % (M, K, N, K, t * 1e6, 6 * fl / t / 2.5e15, tb * 1e6, 12 * fl / tb / 2.5e15, float(y.double().sum())))


if __name__ == "__main__":
    torch.manual_seed(0)
    print(os.path.basename(os.environ.get("RECBOX_HIP_LIB", "default")))
    for (M, K, N) in [(65536, 1677, 400), (65536, 400, 400), (16384, 4096, 4096)]:
        case(M, K, N)
