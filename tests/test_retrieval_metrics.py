"""retrieval evaluation.  CPU: the metric formulas (host form and vectorised form) against the
per-user values the live reference's metric classes produced (fixture).  GPU: rbx_topk against a full sort,
rbx_membership / rbx_penalize_members against numpy, evaluate_metrics against the reference's averages."""
import numpy as np
import pytest
import torch

from conftest import Fixture


def _lists(off, items, n):
    return dict((q, items[off[q]:off[q + 1]].tolist()) for q in range(n))


def _fixture():
    fx = Fixture("retrieval_metrics")
    X, O = fx["in"], fx["out"]
    n_q = len(X["train_off"]) - 1
    return X, O, _lists(X["train_off"], X["train_items"], n_q), _lists(X["valid_off"], X["valid_items"], n_q), \
        [str(m) for m in O["metrics"]]


def test_metric_formulas_match_reference_per_user():
    import recbox_amd.core.metrics as M
    X, O, train, valid, metrics = _fixture()
    funcs = [eval(m, vars(M)) for m in metrics]
    top, query = O["top50"], X["query"]
    for u in range(top.shape[0]):
        true_items = valid[int(query[u])]
        got = [f(top[u], true_items) for f in funcs]
        np.testing.assert_allclose(got, O["per_user"][u], rtol=1e-12, atol=1e-15)
    # vectorised form on the same hit flags
    hits = torch.tensor([[int(i) in set(valid[int(q)]) for i in row] for row, q in zip(top, query)])
    n_true = torch.tensor([float(len(valid[int(q)])) for q in query], dtype=torch.float64)
    for j, f in enumerate(funcs):
        np.testing.assert_allclose(f.batch(hits, n_true).numpy(), O["per_user"][:, j], rtol=1e-12, atol=1e-15)
    with pytest.raises(NotImplementedError):
        M.evaluate_metrics(np.zeros((1, 2)), np.zeros((3, 2)), {}, {}, [0], ["Bogus(k=1)"], device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("rows,n,k", [(5, 3706, 500), (3, 100, 500), (2, 70000, 500), (7, 1000, 50), (1, 1, 1),
                                      (4, 200000, 10), (2, 33000, 1024)])
def test_topk_equals_full_sort(rows, n, k):
    from recbox_amd import ops
    g = torch.Generator().manual_seed(n + k)
    s = torch.randn(rows, n, generator=g)
    s[:, ::7] = s[:, 1::7][:, :s[:, ::7].shape[1]] if n > 14 else s[:, ::7]     # exact ties, broken by the lower index
    if n > 10:
        s[0, :5] = float("-inf")
        s[0, 5] = float("inf")
    vals, idx = ops.topk(s.cuda(), k)
    order = torch.sort(s, dim=1, descending=True, stable=True)       # stable: equal scores keep ascending index
    kk = min(k, n)
    assert torch.equal(idx[:, :kk].cpu(), order.indices[:, :kk])
    assert torch.equal(vals[:, :kk].cpu(), order.values[:, :kk])
    if kk < k:
        assert bool((idx[:, kk:] == -1).all()) and bool((vals[:, kk:] < -3e38).all())
    remap = torch.randint(0, 2 ** 40, (rows, n), generator=g)          # caller-side ids are full 64-bit values
    _, idx2 = ops.topk(s.cuda(), min(k, 64), index=remap.cuda())
    if len(torch.unique(s)) == s.numel():                            # without ties the remapped ids follow the same order
        assert torch.equal(idx2.cpu()[:, :min(kk, 64)], torch.gather(remap, 1, order.indices[:, :min(kk, 64)]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(70000, 500), (300000, 64), (20000, 1000)])
def test_topk_long_rows_fast_and_exact_paths_side_by_side(n, k):
    """Long rows take the sampled-threshold path (one filtering sweep, exact selection among the candidates); rows it
    cannot serve -- constant rows, rows whose top values all sit in one 8 192-score segment, rows with fewer distinct
    large values than the sample suggests -- fall back to the exact multi-level path on the device.  Both kinds in one
    call, each row equal to a stable full sort."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(n)
    rows = [torch.randn(n, generator=g), torch.zeros(n), torch.randn(n, generator=g) * 1e-3]
    spike = torch.zeros(n)
    spike[5000:5000 + 2 * k] = torch.rand(2 * k, generator=g) + 1.0          # every winner inside one segment
    rows.append(spike)
    heavy = torch.randn(n, generator=g)
    heavy[::2] = heavy[0]                                                    # half the row is one value
    rows.append(heavy)
    rows.append(-torch.rand(n, generator=g))
    s = torch.stack(rows)
    vals, idx = ops.topk(s.cuda(), k)
    order = torch.sort(s, dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu(), order.indices[:, :k])
    assert torch.equal(vals.cpu(), order.values[:, :k])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(70000, 500), (8192, 1024), (100, 7), (20000, 1)])
def test_topk_with_massive_ties(n, k):
    """Degenerate score rows -- all equal, three distinct values, one value with a few larger ones -- over one and
    several selection levels: the winners are the lowest columns among the equal scores, as a stable sort gives them."""
    from recbox_amd import ops
    rows = torch.stack([torch.zeros(n), (torch.arange(n) % 3).float(), torch.ones(n)])
    rows[2, n // 2] = 5.0
    rows[2, n - 1] = 4.0
    vals, idx = ops.topk(rows.cuda(), k)
    order = torch.sort(rows, dim=1, descending=True, stable=True)
    kk = min(k, n)
    assert torch.equal(idx[:, :kk].cpu(), order.indices[:, :kk])
    assert torch.equal(vals[:, :kk].cpu(), order.values[:, :kk])


@pytest.mark.gpu
def test_membership_and_penalty():
    from recbox_amd import ops
    g = np.random.RandomState(2)
    n_q, n_items, rows, k = 9, 300, 50, 40
    lists = [np.sort(g.choice(n_items, g.randint(0, 80), replace=False)) for _ in range(n_q)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    items = np.concatenate(lists).astype(np.int64)
    cand = g.randint(-1, n_items, (rows, k)).astype(np.int64)
    query = g.randint(0, n_q, rows).astype(np.int64)
    want = np.array([[c in set(lists[q].tolist()) for c in row] for row, q in zip(cand, query)])
    args = [torch.from_numpy(a).cuda() for a in (cand, query, off, items)]
    assert (ops.membership(*args).cpu().numpy() == want).all()
    scores = g.randn(rows, k).astype(np.float32)
    ref = scores.copy()
    ref += -1e9 * want.astype(np.float64)                             # the reference's in-place add (metrics.py:62)
    got = ops.penalize_members_(torch.from_numpy(scores).cuda(), *args, penalty=-1e9).cpu().numpy()
    assert (got == ref).all()


@pytest.mark.gpu
def test_evaluate_metrics_matches_reference_fixture():
    import recbox_amd.core.metrics as M
    X, O, train, valid, metrics = _fixture()
    out = M.evaluate_metrics(X["user"], X["item"], train, valid, X["query"], metrics)
    assert list(out) == metrics
    np.testing.assert_allclose([out[m] for m in metrics], O["average"], rtol=1e-9, atol=1e-12)
    # index level: the 50 items per user that were scored
    dev = torch.device("cuda")
    n_q = len(X["train_off"]) - 1
    funcs = [eval(m, vars(M)) for m in metrics]
    res, top = M.evaluate_block(torch.tensor(X["user"], dtype=torch.float32, device=dev),
                                torch.tensor(X["item"], dtype=torch.float32, device=dev),
                                torch.tensor(X["query"], device=dev), M.build_csr(train, n_q, dev),
                                M.build_csr(valid, n_q, dev), funcs, 50)
    assert (top.cpu().numpy() == O["top50"]).all()
    np.testing.assert_allclose(res.cpu().numpy(), O["per_user"], rtol=1e-12, atol=1e-15)
