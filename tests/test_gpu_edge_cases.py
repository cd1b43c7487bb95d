"""Empty / degenerate inputs through the ops added for K7, the norms, the loader and the retrieval path: a zero-size
batch must come back as correctly shaped empty tensors (and zero gradients), never a crash or a launch error."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_batches_everywhere():
    from recbox_amd import ops
    dev = "cuda"
    w = torch.randn(10, 8, device=dev, requires_grad=True)
    x = torch.zeros(0, 8, device=dev, requires_grad=True)
    out = ops.gather_dot(x, [torch.zeros(0, dtype=torch.long, device=dev)], w)
    assert out.shape == (0, 1)
    out.sum().backward()
    assert w.grad is not None and float(w.grad.abs().sum()) == 0.0 and x.grad.shape == (0, 8)
    ln = torch.nn.LayerNorm(8).to(dev)
    assert ops.layer_norm(torch.zeros(0, 3, 8, device=dev), ln).shape == (0, 3, 8)
    v, i = ops.topk(torch.zeros(0, 5, device=dev), 3)
    assert v.shape == (0, 3) and i.shape == (0, 3)
    v, i = ops.topk(torch.randn(2, 0, device=dev), 3)                       # rows without candidates: all padding
    assert bool((i == -1).all())
    assert ops.negsample(10, 0, 4, seed=1, device=dev).shape == (0, 4)
    assert ops.negsample(10, 5, 0, seed=1, pos=torch.arange(5, device=dev)).tolist() == [[0], [1], [2], [3], [4]]
    assert ops.gather_rows([torch.arange(6, device=dev).view(3, 2)], torch.zeros(0, dtype=torch.long, device=dev))[0].shape == (0, 2)
    of = torch.zeros((), dtype=torch.bool, device=dev)
    send, slot = ops.route(torch.zeros(0, 2, dtype=torch.long, device=dev), 2, 64, torch.zeros(2, 2, dtype=torch.long, device=dev), of)
    assert slot.shape == (0, 2) and bool((send == -1).all()) and not bool(of)
    y = ops.cross(torch.zeros(0, 4, device=dev), torch.zeros(0, 4, device=dev), torch.zeros(0, 4, device=dev))
    assert y.shape == (0, 4)
    assert ops.linear(torch.zeros(0, 8, device=dev), torch.randn(3, 8, device=dev)).shape == (0, 3)
    torch.cuda.synchronize()


def test_batch_norm_single_row_raises_like_torch():
    from recbox_amd import ops
    bn = torch.nn.BatchNorm1d(4).cuda()
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        ops.batch_norm(torch.randn(1, 4, device="cuda"), bn)
    bn.eval()
    assert ops.batch_norm(torch.randn(1, 4, device="cuda"), bn).shape == (1, 4)          # eval mode accepts it


def test_linear_strided_rows_match_contiguous():
    """A column block of a wider activation is read in place (row stride > K) and gives the same result and grads."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(0)
    wide = torch.randn(300, 52, generator=g).cuda()
    w = torch.randn(7, 37, generator=g).cuda().requires_grad_(True)
    b = torch.randn(7, generator=g).cuda().requires_grad_(True)
    xs = wide[:, :37].detach().requires_grad_(True)                                     # stride 52, 37 columns
    xc = wide[:, :37].contiguous().detach().requires_grad_(True)
    ys, yc = ops.linear(xs, w, b, "relu"), ops.linear(xc, w, b, "relu")
    assert torch.equal(ys, yc)
    r = torch.randn_like(ys)
    gs = torch.autograd.grad((ys * r).sum(), (xs, w, b))
    gc = torch.autograd.grad((yc * r).sum(), (xc, w, b))
    for a, c in zip(gs, gc):
        assert torch.equal(a.contiguous(), c.contiguous())


@pytest.mark.parametrize("D,R", [(64, 1_000_003), (16, 300_007)])
def test_hot_rows_with_nonzero_gradients_are_summed_by_several_workgroups(D, R):
    """One item that collects 70 % of a million lookups -- all with non-zero gradients -- and a second one with 20 %: the
    long fix-up of the segmented reduce shares each of these chains between up to 16 workgroups (every KW-th window of
    512 chunk partials, the last workgroup to arrive adds the partials in slot order).  Against float64, and bit-identical
    from run to run."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(D + R)
    V = 1000
    u = torch.rand(R, generator=g)
    ids = torch.randint(0, V, (R,), generator=g)
    ids[u < 0.7] = 7
    ids[(u >= 0.7) & (u < 0.9)] = 3
    x = torch.randn(R, D, generator=g)
    w = torch.randn(V, D, generator=g) * 0.1
    gout = torch.randn(R, 1, generator=g)
    want = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids, (gout.double() * x.double()) * 0.5)

    def run():
        xc = x.cuda()
        wc = w.cuda().requires_grad_(True)
        out = ops.gather_dot(xc, [ids.cuda()], wc, scale=0.5)
        out.backward(gout.cuda())
        return wc.grad.clone()

    a, b = run(), run()
    assert torch.equal(a, b)
    # f32 sums of up to 700 000 terms of size ~0.5: a few 1e-7 of the sum of magnitudes; one lost window of 512 chunks
    # (>= 8192 terms) would be off by tens
    assert float((a.double().cpu() - want).abs().max()) <= 5e-7 * R


@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(100, 1), (2048, 3), (65536, 2), (4096, 31), (9000, 7)])
def test_chained_sort_leaves_the_same_gradients_as_the_launch_per_pass_sort(B, L):
    """The id sort of the backward in 1 + passes launches (rbx_sort_chained, BwdPlan::chained: tiles of a table group chained
    by published counts) against the histogram / scan / scatter launches per pass: the same sorted pairs, hence BIT-identical
    gradients -- tables that sort in 1, 2, 3 and 4 passes in one call (200 / 30 000 / 3 000 000 / 20 000 000 rows: the
    shorter ones join the sort in a later pass), one id and pooled sequences per sample, 1 .. 64 tiles per table group
    (65 536 x 2 lookups = exactly 64 tiles, the largest chained group; 4096 x 31 = 62), ids drawn so that runs of equal ids
    cross tile borders -- and both against float64 index_add."""
    from recbox_amd import _embed_host as host, ops
    from recbox_amd._lib import FIELD_CATEGORICAL, POOL_NONE, POOL_SUM_ID
    g = torch.Generator().manual_seed(B * 131 + L)
    vocabs, D = [200, 30_000, 3_000_000, 20_000_000], 4
    tables = [torch.nn.Embedding(v, D).cuda() for v in vocabs]
    ids = []
    for v in vocabs:
        t = torch.randint(0, v, (B, L), generator=g)
        hot = torch.rand(B, L, generator=g) < 0.3               # 30 % of the lookups on 5 rows: long runs of equal keys
        t[hot] = torch.randint(0, 5, (int(hot.sum()),), generator=g) * (v // 7)
        ids.append(t if L > 1 else t[:, 0])
    lookups = [host.Lookup("f%d" % i, FIELD_CATEGORICAL, tables[i], D, pool=POOL_SUM_ID if L > 1 else POOL_NONE, seq_len=L,
                           mask_id=-1 if L > 1 else None)
               for i in range(len(vocabs))]
    plan = host.Plan(lookups)
    gout = torch.randn(B, plan.width, generator=g)
    was = ops.sort_chained()
    grads = {}
    try:
        for mode in (True, False):
            ops.sort_chained(mode)
            for t in tables:
                t.weight.grad = None
            out = plan.run([i.cuda() for i in ids])
            out.backward(gout.cuda())
            grads[mode] = [t.weight.grad.clone() for t in tables]
    finally:
        ops.sort_chained(was)
    for k, (a, b) in enumerate(zip(grads[True], grads[False])):
        assert torch.equal(a, b), "table %d" % k
        idk = ids[k].reshape(B, L)
        want = torch.zeros(vocabs[k], D, dtype=torch.float64)
        want.index_add_(0, idk.reshape(-1), gout[:, k * D:(k + 1) * D].double().repeat_interleave(L, dim=0))
        err = float((a.double().cpu() - want).abs().max())
        assert err <= 1e-5 * max(1.0, float(want.abs().max())), "table %d: %g" % (k, err)
