"""Per-kernel roofline numbers at the shapes of BASELINE.json configs 3-5 (the bench line covers config 2).
Run on the GPU box:  python profiles/ubench/kernels_bench.py   -> prints one line per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from recbox_amd.rechub.basic.features import SequenceFeature, SparseFeature  # noqa: E402
from recbox_amd.rechub.basic.layers import EmbeddingLayer  # noqa: E402

ops.config.check_ids = False


_DELAY = None


def timeit(fn, iters=10, warm=3):
    """Seconds per call of fn.  Eager enqueueing costs 50-300 us of Python per call, more than the
    shorter kernels take, so the stream is first loaded with ~5 ms of fills: by the time the GPU reaches
    the start event every call of fn is already queued and the events bracket device time only."""
    global _DELAY
    if _DELAY is None:
        _DELAY = torch.empty(1 << 28, device="cuda")          # 1 GiB
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(24):
        _DELAY.zero_()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm(M, K, N, act):
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    w = torch.randn(N, K, device="cuda", requires_grad=True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    t = timeit(lambda: ops.linear(x, w, b, act))
    y = ops.linear(x, w, b, act)
    g = torch.randn_like(y)
    tb = timeit(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    fl = 2.0 * M * K * N
    print("linear fwd  [%d,%d]x[%d,%d]^T %-5s %8.1f us  %6.1f TFLOP/s  (fp32 MFMA peak 157)" % (M, K, N, K, act, t * 1e6, fl / t / 1e12))
    print("linear bwd  dx+dW+db                 %8.1f us  %6.1f TFLOP/s" % (tb * 1e6, 2 * fl / tb / 1e12))


def attention(B, H, L, D):
    q, k, v = (torch.randn(B, H, L, D, device="cuda", requires_grad=True) for _ in range(3))
    t = timeit(lambda: ops.attention(q, k, v, scale=D ** -0.5, causal=True, fill=float("-inf")))
    o, _ = ops.attention(q, k, v, scale=D ** -0.5, causal=True, fill=float("-inf"))
    g = torch.randn_like(o)
    tb = timeit(lambda: torch.autograd.grad(o, (q, k, v), g, retain_graph=True))
    fl = 4.0 * B * H * L * L * D / 2          # causal: half of QK^T and PV
    print("attention fwd B=%d H=%d L=%d d=%d causal %8.1f us  %6.1f TFLOP/s (fp32 VALU peak 157)" % (B, H, L, D, t * 1e6, fl / t / 1e12))
    print("attention bwd                           %8.1f us  %6.1f TFLOP/s" % (tb * 1e6, 2.5 * fl / tb / 1e12))


def gather_pool(V, D, B, L):
    item = SparseFeature("item", V, D)
    hist = SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)
    layer = EmbeddingLayer([item, hist]).cuda()
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    h = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    x = {"item": torch.randint(1, V, (B,), generator=g).cuda(), "hist": h.cuda()}
    nnz = int(lens.sum())
    t = timeit(lambda: layer(x, [hist], squeeze_dim=True))
    out = layer(x, [hist], squeeze_dim=True)
    gr = torch.randn_like(out)
    tb = timeit(lambda: torch.autograd.grad(out, layer.embed_dict["item"].weight, gr, retain_graph=True), iters=4)
    byt = nnz * D * 4 + B * L * 8 + B * D * 4
    print("gather+mean-pool fwd V=%d D=%d B=%d L<=%d (nnz %d)  %8.1f us  %7.1f GB/s of rows+ids+out" % (V, D, B, L, nnz, t * 1e6, byt / t / 1e9))
    print("  its backward (sort + segmented scatter-add + %d MB dense-grad zero fill)  %8.1f us" % (V * D * 4 >> 20, tb * 1e6))


def gather_dot(V, D, R, n_sets):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(V, D, generator=g) * 0.1).cuda().requires_grad_(True)
    x = torch.randn(R, D, generator=g).cuda().requires_grad_(True)
    sets = [torch.randint(0, V, (R,), generator=g).cuda() for _ in range(n_sets)]
    t = timeit(lambda: ops.gather_dot(x, sets, w))
    out = ops.gather_dot(x, sets, w)
    gr = torch.randn_like(out)
    tb = timeit(lambda: torch.autograd.grad(out, (x, w), gr, retain_graph=True), iters=4)
    byt = R * n_sets * (D * 4 + 8 + 4) + R * D * 4
    print("gather-dot fwd V=%d D=%d rows=%d sets=%d  %8.1f us  %7.1f GB/s of rows+ids+logits+x" % (V, D, R, n_sets, t * 1e6, byt / t / 1e9))
    print("  its backward (dx gather + sort + segmented scatter-add + %d MB dense-grad zero fill)  %8.1f us" % (V * D * 4 >> 20, tb * 1e6))


def retrieval(U, I, D, k=500):
    g = torch.Generator().manual_seed(0)
    u = torch.randn(U, D, generator=g).cuda()
    v = torch.randn(I, D, generator=g).cuda()
    scores = ops.linear(u, v)
    tg = timeit(lambda: ops.linear(u, v), iters=5)
    tk = timeit(lambda: ops.topk(scores, k), iters=5)
    print("retrieval  users=%d items=%d D=%d: scores GEMM %8.1f us (%5.1f TFLOP/s)   top-%d select %8.1f us (%6.1f GB/s of one "
          "sweep over the scores)" % (U, I, D, tg * 1e6, 2.0 * U * I * D / tg / 1e12, k, tk * 1e6, U * I * 4 / tk / 1e9))


def sampler(I, rows, negs):
    t = timeit(lambda: ops.negsample(I, rows, negs, seed=1, device="cuda"))
    corpus = [torch.randint(0, 1000, (I,)).cuda(), torch.randint(0, 1000, (I, 8), dtype=torch.int32).cuda()]
    idx = ops.negsample(I, rows, negs, seed=1, device="cuda").reshape(-1)
    tg = timeit(lambda: ops.gather_rows(corpus, idx))
    print("negative sampling items=%d rows=%d negs=%d  %8.1f us (%6.1f G draws/s)   corpus gather (8 B + 32 B rows) %8.1f us "
          "(%6.1f GB/s of rows+ids+out)" % (I, rows, negs, t * 1e6, rows * negs / t / 1e9, tg * 1e6,
                                          idx.numel() * (8 + 40 * 2) / tg / 1e9))


if __name__ == "__main__":
    if "eval" in sys.argv[1:]:
        retrieval(6040, 3706, 64)
        retrieval(1024, 1_000_000, 128)
        sampler(10_000_000, 65536, 4)
        sys.exit(0)
    if "dot" in sys.argv[1:]:
        gather_dot(1_000_000, 64, 4096 * 200, 2)
        sys.exit(0)
    if "gather" in sys.argv[1:]:
        gather_pool(10_000_000, 128, 65536, 50)
        gather_pool(10_000_000, 64, 65536, 50)
        gather_pool(10_000_000, 32, 65536, 50)
        gather_pool(1_000_000, 16, 65536, 50)
        sys.exit(0)
    print("# cfg 4 DeepFM tower (rechub style input 26*64+13 = 1677), batch 65 536")
    gemm(65536, 1677, 400, "relu")
    gemm(65536, 400, 400, "relu")
    gemm(65536, 400, 1, None)
    print("# cfg 5 SASRec attention core")
    attention(4096, 1, 200, 64)
    gemm(4096 * 200, 64, 64, None)
    gather_dot(1_000_000, 64, 4096 * 200, 2)
    print("# SURVEY 8f: loader and evaluation")
    retrieval(6040, 3706, 64)
    retrieval(1024, 1_000_000, 128)
    sampler(10_000_000, 65536, 4)
    print("# cfg 3 YoutubeDNN history pooling, one GPU holding the whole 10M x 128 table (5.1 GB)")
    gather_pool(10_000_000, 128, 65536, 50)
    print("# cfg 2 layer path pieces")
    gather_pool(1_000_000, 16, 65536, 50)
