"""VERDICT r4 #4, first question: what does the tower GEMM gain when the ACTIVATION operand arrives as bf16 planes (written
once by its producer) instead of being split on the way to LDS by every workgroup that reads it?  gemm_bxp_kernel<true>
(planes of A registered with rbx_split_register, the same layout as the weights') against gemm_bxp_kernel<false> at the
shapes of BASELINE cfg 4's tower (B = 65 536; 1677 -> 400 -> 400 -> 400 and the dx products), results bit-compared.
Run on the GPU box:  python profiles/ubench/gemm_presplit_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from recbox_amd._lib import lib, check  # noqa: E402
from profiles.ubench.kernels_bench import timeit  # noqa: E402


def planes_of(x):
    M, K = x.shape
    nbytes = lib.rbx_split_bf16_size(M, K, 0)
    p = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib.rbx_split_bf16(x.data_ptr(), K, M, K, 0, p.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return p


def case(M, K, N):
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    with torch.no_grad():
        y0 = ops.linear(x, w, b, "relu")
        t0 = timeit(lambda: ops.linear(x, w, b, "relu"), iters=20)
        p = planes_of(x)
        ts = timeit(lambda: planes_of(x), iters=10)
        check(lib.rbx_split_register(x.data_ptr(), p.data_ptr(), M, K, 0))
        try:
            y1 = ops.linear(x, w, b, "relu")
            t1 = timeit(lambda: ops.linear(x, w, b, "relu"), iters=20)
        finally:
            lib.rbx_split_unregister(x.data_ptr())
    same = bool(torch.equal(y0, y1))
    fl = 2.0 * M * K * N
    print("[%d,%d]x[%d,%d]^T  split in the loop %7.1f us (%5.1f TF f32-equivalent, %.3f of the bf16 pipes)   planes of A %7.1f us "
          "(%5.1f TF, %.3f)   bit-identical %s   (stand-alone split of A: %.1f us, incl. the weights' split in both timings)"
          % (M, K, N, K, t0 * 1e6, fl / t0 / 1e12, 6 * fl / t0 / 2.5e15, t1 * 1e6, fl / t1 / 1e12, 6 * fl / t1 / 2.5e15, same,
             ts * 1e6))


if __name__ == "__main__":
    torch.manual_seed(0)
    for (M, K, N) in [(65536, 1677, 400), (65536, 400, 400), (65536, 400, 1677), (65536, 1024, 1024), (16384, 4096, 4096)]:
        case(M, K, N)
