"""SURVEY 8f-1 on the GPU: rbx_negsample / rbx_gather_rows against the C oracle (bit-exact) and the loader
mirror against the fixture generated from the live reference's TrainDataset + collate_fn."""
import numpy as np
import pytest
import torch

from conftest import Fixture
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


def test_negsample_bit_exact_vs_oracle_and_properties():
    from recbox_amd import ops
    n_items, rows, negs = 3706, 20000, 7
    pos = (torch.arange(rows) * 13 % n_items).long()
    got = ops.negsample(n_items, rows, negs, seed=2019, offset=5, pos=pos.cuda()).cpu().numpy()
    want = C.negsample(n_items, rows, negs, seed=2019, offset=5, pos=pos.numpy())
    assert (got == want).all()
    assert (ops.negsample(n_items, rows, negs, seed=2019, offset=5).cpu().numpy() == want[:, 1:]).all()   # no pos column
    counts = np.bincount(got[:, 1:].reshape(-1), minlength=n_items)
    expected = rows * negs / n_items
    assert abs(((counts - expected) ** 2 / expected).sum() / (n_items - 1) - 1.0) < 0.1      # reduced chi2 ~ 1
    # exclusion (ignore_pos_items): CSR of 11 queries, 300 sorted items each
    g = np.random.RandomState(1)
    lists = [np.sort(g.choice(n_items, 300, replace=False)) for _ in range(11)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    items = np.concatenate(lists).astype(np.int64)
    query = (torch.arange(rows) % 11).long()
    got_e = ops.negsample(n_items, rows, negs, seed=3, pos=pos.cuda(), query=query.cuda(),
                          excl_offsets=torch.from_numpy(off).cuda(), excl_items=torch.from_numpy(items).cuda()).cpu().numpy()
    want_e = C.negsample(n_items, rows, negs, seed=3, pos=pos.numpy(), query=query.numpy(), excl_offsets=off, excl_items=items)
    assert (got_e == want_e).all()
    for q in range(11):
        assert not np.isin(got_e[query.numpy() == q][:, 1:], lists[q]).any()


def test_negsample_full_size_is_uniform_and_deterministic():
    """BASELINE cfg 3 scale: 10 M items, 65 536 x 4 negatives; same seed -> same block, different epochs differ."""
    from recbox_amd import ops
    a = ops.negsample(10_000_000, 65536, 4, seed=7, device="cuda")
    b = ops.negsample(10_000_000, 65536, 4, seed=7, device="cuda")
    c = ops.negsample(10_000_000, 65536, 4, seed=7, offset=65536 * 4, device="cuda")
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert int(a.min()) >= 0 and int(a.max()) < 10_000_000
    assert abs(float(a.double().mean()) / 5e6 - 1.0) < 0.01
    sample = C.negsample(10_000_000, 64, 4, seed=7)
    assert (a[:64].cpu().numpy() == sample).all()


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.float32, torch.float64, torch.uint8, torch.int16])
def test_gather_rows_bit_exact(dtype):
    from recbox_amd import ops
    g = torch.Generator().manual_seed(3)
    n, m = 1000, 4097
    cols = [torch.randint(0, 100, (n,), generator=g).to(dtype), torch.randint(0, 100, (n, 3), generator=g).to(dtype),
            torch.randint(0, 100, (n, 50), generator=g).to(dtype), torch.randint(0, 100, (n, 2, 4), generator=g).to(dtype)]
    idx = torch.randint(0, n, (m,), generator=g)
    outs = ops.gather_rows([c.cuda() for c in cols], idx.cuda())
    for c, o in zip(cols, outs):
        assert o.dtype == c.dtype and torch.equal(o.cpu(), c[idx])
        want, _ = C.gather_rows(c.numpy(), idx.numpy())
        assert (o.cpu().numpy() == want).all()
    assert ops.gather_rows([cols[0].cuda()], idx[:0].cuda())[0].shape == (0,)
    with pytest.raises(IndexError):
        ops.gather_rows([cols[0].cuda()], torch.tensor([0, n]).cuda())


def test_loader_batches_equal_reference_collate_fixture():
    from recbox_amd.matching.pytorch.dataloaders import collate, collate_unique
    fx = Fixture("matching_loader")
    dev = "cuda"
    data = {k: torch.from_numpy(v).to(dev) for k, v in fx["data"].items()}
    corpus = {k: torch.from_numpy(v).to(dev) for k, v in fx["corpus"].items()}
    labels = torch.from_numpy(fx["in"]["labels"]).to(dev)
    all_idx = torch.from_numpy(fx["in"]["all_item_indexes"]).to(dev)
    batch = torch.from_numpy(fx["in"]["batch_index"]).to(dev)
    u, it, lab, inv, _ = collate(data, corpus, labels, all_idx, batch)
    assert inv is None
    for k in fx["user"]:
        assert u[k].dtype == torch.from_numpy(fx["user"][k]).dtype and (u[k].cpu().numpy() == fx["user"][k]).all(), k
    for k in fx["item"]:
        assert tuple(it[k].shape) == fx["item"][k].shape and (it[k].cpu().numpy() == fx["item"][k]).all(), k
    assert (lab.cpu().numpy() == fx["out"]["labels"]).all() and lab.dtype == torch.float32
    u2, it2, lab2, inv2, _ = collate_unique(data, corpus, labels, all_idx, batch)
    for k in fx["item_u"]:
        assert (it2[k].cpu().numpy() == fx["item_u"][k]).all(), k
    assert (inv2.cpu().numpy() == fx["out_u"]["inverse_indexes"]).all()
    assert (lab2.cpu().numpy() == fx["out_u"]["labels"]).all()


def test_train_generator_epoch():
    """Mirror of the reference's TrainGenerator surface: len, re-sampling per epoch, batch structure, negatives
    never among the user's own items with ignore_pos_items."""
    from recbox_amd.matching.features import FeatureMap
    from recbox_amd.matching.pytorch.dataloaders import TrainGenerator
    fm = FeatureMap("toy", "/tmp", query_index="query_index", corpus_index="corpus_index", label_name="label")
    g = np.random.RandomState(0)
    N, I = 1000, 200
    data = {"user_id": g.randint(1, 50, N), "hist": g.randint(0, I, (N, 6)).astype(np.int32),
            "query_index": g.randint(0, 40, N), "corpus_index": g.randint(0, I, N), "label": np.ones(N)}
    corpus = {"item_id": np.arange(I) + 1, "cate": g.randint(1, 9, I)}
    gen = TrainGenerator(fm, data, corpus, batch_size=128, shuffle=True, num_negs=4, ignore_pos_items=True)
    assert len(gen) == 8
    seen, first = 0, None
    for user, item, labels, inv in gen:
        b = labels.shape[0]
        assert labels.shape == (b, 5) and float(labels[:, 1:].abs().sum()) == 0.0 and inv is None
        assert item["item_id"].shape == (b * 5,) and user["hist"].shape == (b, 6)
        assert "query_index" not in user and "label" not in user
        seen += b
    assert seen == N
    e1 = gen.all_item_indexes.clone()
    for _ in gen:
        pass
    assert torch.equal(e1[:, 0], gen.all_item_indexes[:, 0]) and not torch.equal(e1, gen.all_item_indexes)
    q = torch.as_tensor(data["query_index"])
    c = torch.as_tensor(data["corpus_index"])
    negs = gen.all_item_indexes[:, 1:].cpu()
    for user_q in range(5):
        own = set(c[q == user_q].tolist())
        assert not (set(negs[q == user_q].reshape(-1).tolist()) & own)


def test_negsample_exact_fallback_and_query_bounds():
    """a query that interacted with almost the whole corpus never gets one of its own items (exact draw over the
    complement after 64 rejections; HIP == C oracle bit for bit), and a query index outside the CSR raises IndexError."""
    from recbox_amd import ops
    n_items, rows, negs = 200, 512, 6
    allowed = {0: [7, 123], 1: [199], 2: [0, 1, 2, 3]}
    off, items = [0], []
    for q in range(3):
        items += [i for i in range(n_items) if i not in allowed[q]]
        off.append(len(items))
    query = np.arange(rows) % 3
    off, items = np.array(off), np.array(items)
    got = ops.negsample(n_items, rows, negs, seed=5, query=torch.from_numpy(query).cuda(),
                        excl_offsets=torch.from_numpy(off).cuda(), excl_items=torch.from_numpy(items).cuda()).cpu().numpy()
    want = C.negsample(n_items, rows, negs, seed=5, query=query, excl_offsets=off, excl_items=items)
    assert (got == want).all()
    for r in range(rows):
        assert set(got[r].tolist()) <= set(allowed[int(query[r])])
    bad = torch.from_numpy(query).cuda()
    bad[3] = 3                                                            # the CSR has 3 queries: 0, 1, 2
    with pytest.raises(IndexError):
        ops.negsample(n_items, rows, negs, seed=5, query=bad, excl_offsets=torch.from_numpy(off).cuda(),
                      excl_items=torch.from_numpy(items).cuda())
