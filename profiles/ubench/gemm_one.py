"""One shape of the split-operand tower GEMM, a few launches (for rocprofv3 --pmc passes):
    python profiles/ubench/gemm_one.py M K N [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402

if __name__ == "__main__":
    M, K, N = (int(a) for a in sys.argv[1:4])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    torch.manual_seed(0)
    pitch = (K + 3) // 4 * 4
    x = torch.randn(M, pitch, device="cuda")[:, :K]
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    with torch.no_grad():
        for _ in range(n):
            ops.linear(x, w, b, "relu")
    torch.cuda.synchronize()
