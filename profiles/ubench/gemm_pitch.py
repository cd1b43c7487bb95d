"""Tower layer 1 of BASELINE cfg 4 ([65536, 1677] x [400, 1677]^T, gemm_bxp_kernel) by the ROW PITCH of the activation block.
Run on the GPU box:  python profiles/ubench/gemm_pitch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from profiles.ubench.kernels_bench import timeit  # noqa: E402

if __name__ == "__main__":
    torch.manual_seed(0)
    M, K, N = 65536, 1677, 400
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    for pitch in (1677, 1680, 1684, 1688, 1696, 1712, 1728, 1760, 1792):
        buf = torch.randn(M, pitch, device="cuda")
        x = buf[:, :K]
        with torch.no_grad():
            t = timeit(lambda: ops.linear(x, w, b, "relu"), iters=20)
        print("pitch %5d floats (%6d B)  %7.1f us  %6.1f TF f32-equivalent" % (pitch, pitch * 4, t * 1e6, 2.0 * M * K * N / t / 1e12))
