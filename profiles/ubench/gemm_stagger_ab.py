"""gemm_bxs_kernel (the two halves of a workgroup a phase apart, three LDS stages) against gemm_bxp_kernel (all eight wavefronts in step)
at the tower shapes of BASELINE cfg 4 (B = 65 536), outputs bit-compared.  Run on the GPU box:  python profiles/ubench/gemm_stagger_ab.py"""
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
        for mode in (0, 1, 2, 3):
            ops.gemm_stagger(mode)
            y = ops.linear(x, w, b, "relu")
            t = timeit(lambda: ops.linear(x, w, b, "relu"), iters=20)
            out[mode] = (y, t)
    ops.gemm_stagger(3)
    fl = 2.0 * M * K * N
    (y0, t0), (y1, t1) = out[0], out[3]
    print("   late = bit 0 / 1 / 2 of the wavefront number: %.1f / %.1f / %.1f us" % (out[1][1] * 1e6, out[2][1] * 1e6, out[3][1] * 1e6))
    print("[%d,%d]x[%d,%d]^T  in step (gemm_bxp) %7.1f us (%5.1f TF f32-equivalent, %.3f of the bf16 pipes)   a phase apart (gemm_bxs) %7.1f us (%5.1f TF, %.3f)"
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
