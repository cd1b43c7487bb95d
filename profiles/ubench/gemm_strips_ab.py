"""gemm_bxw_kernel (column strips per wavefront, 256 x 224 tiles) against gemm_bxp_kernel (256 x 128 tiles) at the tower shapes
of BASELINE cfg 4 (B = 65 536), outputs bit-compared.  Run on the GPU box:  python profiles/ubench/gemm_strips_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from recbox_amd._lib import lib  # noqa: E402
from profiles.ubench.kernels_bench import timeit  # noqa: E402


def case(M, K, N, pitch=None):
    buf = torch.randn(M, pitch or K, device="cuda")
    x = buf[:, :K]
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = {}
    with torch.no_grad():
        for mode in (0, 2):
            ops.gemm_strips(mode)
            y = ops.linear(x, w, b, "relu")
            t = timeit(lambda: ops.linear(x, w, b, "relu"), iters=20)
            out[mode] = (y, t)
    ops.gemm_strips(1)
    fl = 2.0 * M * K * N
    (y0, t0), (y1, t1) = out[0], out[2]
    print("[%d,%d]x[%d,%d]^T  128-column tiles %7.1f us (%5.1f TF f32-equivalent, %.3f of the bf16 pipes)   strips %7.1f us (%5.1f TF, %.3f)"
          "   bit-identical %s  max|diff| %.3g" % (M, K, N, K, t0 * 1e6, fl / t0 / 1e12, 6 * fl / t0 / 2.5e15, t1 * 1e6, fl / t1 / 1e12,
                                                   6 * fl / t1 / 2.5e15, bool(torch.equal(y0, y1)), float((y0 - y1).abs().max())))


if __name__ == "__main__":
    torch.manual_seed(0)
    case(65536, 1677, 400, 1680)
    case(65536, 400, 400)
    case(65536, 400, 1677)
    case(65536, 1024, 1024)
    case(16384, 4096, 4096)
    case(5000, 333, 450)
