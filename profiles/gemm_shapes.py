"""Time the three tower contractions of one nn.Linear through the C ABI at given shapes and row pitches.

usage: python profiles/gemm_shapes.py            (prints one line per shape: us and TFLOP/s for fwd, dx, dW)
A row pitch larger than K models an input block that is allocated with padded rows (x = block[:, :K]).
"""
import sys
import torch
from recbox_amd import ops

SHAPES = [  # M, N, K, pitch of x
    (65536, 400, 1677, 1677), (65536, 400, 1677, 1680), (65536, 400, 1677, 1696), (65536, 400, 1664, 1664),
    (65536, 400, 400, 400), (65536, 384, 384, 384), (65536, 512, 512, 512),
    (8192, 8192, 8192, 8192), (4096, 4096, 4096, 4096),
]
EDGES = [(65536, 384, 1792, 1792), (65536, 400, 1792, 1792), (65536, 384, 1677, 1677), (65536, 384, 1680, 1680),
         (65536, 400, 1680, 1680)]


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    print(f"{'M':>6} {'N':>5} {'K':>5} {'pitch':>5} | {'fwd us':>8} {'TF':>6} | {'dx us':>8} {'TF':>6} | {'dW us':>8} {'TF':>6}")
    for M, N, K, pitch in (EDGES if "--edges" in sys.argv else SHAPES):
        block = torch.randn(M, pitch, device=dev)
        x = block[:, :K]
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        dw = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        gf = 2.0 * M * N * K / 1e6
        t_f = timed(lambda: ops._lin_fwd(x, w, b, 1))
        t_x = timed(lambda: ops._lin_dx(dy, w))
        t_w = timed(lambda: ops._lin_dwdb(x, w, dy, dw, db))
        print(f"{M:>6} {N:>5} {K:>5} {pitch:>5} | {t_f:8.1f} {gf / t_f:6.1f} | {t_x:8.1f} {gf / t_x:6.1f} | {t_w:8.1f} {gf / t_w:6.1f}")
        del block, x, w, dy, dw


if __name__ == "__main__":
    sys.exit(main())
