"""Pin the plain-C oracle (oracle/recbox_oracle.c) against the fixtures generated from the live
reference.  CPU only."""
import ctypes

import numpy as np

from conftest import Fixture, assert_close
from oracle import c_oracle as C

TOL = 1e-5


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def test_embed_fwd_bwd_matches_ranking_fixture():
    lib = C.load()
    fx = Fixture("ranking_feature_embedding")
    X, P, G = fx["in"], fx["p"], fx["g"]
    pre = "embedding_layer.embedding_layers.%s.weight"
    W = {k: _f32(P[pre % k]) for k in ("n1", "c1", "n2", "hist", "c3")}
    dW = {k: np.zeros_like(v) for k, v in W.items()}
    ids = {k: np.ascontiguousarray(X[k]) for k in ("n1", "c1", "n2", "hist", "c2", "c3")}
    D, B = 8, 7
    fields = [C.field(ids["n1"], W["n1"].reshape(-1), dW["n1"].reshape(-1), kind=C.NUMERIC, dim=D, out_off=0),
              C.field(ids["c1"], W["c1"], dW["c1"], dim=D, out_off=D, padding_idx=0),
              C.field(ids["n2"], W["n2"].reshape(-1), dW["n2"].reshape(-1), kind=C.NUMERIC, dim=D, out_off=2 * D),
              C.field(ids["hist"], W["hist"], dW["hist"], dim=D, out_off=3 * D, pool=C.POOL_MEAN_VALUE,
                      padding_idx=0, eps=1e-12),
              C.field(ids["c2"], W["hist"], dW["hist"], dim=D, out_off=4 * D, padding_idx=0),   # share_embedding
              C.field(ids["c3"], W["c3"], dW["c3"], dim=D, out_off=5 * D)]
    arr = C.array_of(fields)
    out = np.zeros((B, 6 * D), np.float32)
    scale = np.zeros((6, B), np.float32)
    assert lib.orc_embed_fwd(arr, 6, B, C.ptr(out), 6 * D, C.ptr(scale)) == 0
    assert_close(out.reshape(B, 6, D), fx["out"]["emb"], TOL, "emb")
    R = _f32(X["R"]).reshape(B, 6 * D)
    lib.orc_embed_bwd(arr, 6, ctypes.c_int64(B), C.ptr(R), ctypes.c_int64(6 * D), C.ptr(scale))
    for k in W:
        assert_close(dW[k].reshape(G[pre % k].shape), G[pre % k], TOL, "grad " + k)


def test_fm_fused_matches_ranking_fm_fixture():
    lib = C.load()
    fx = Fixture("ranking_fm")
    X, P, G = fx["in"], fx["p"], fx["g"]
    names = ["I%d" % i for i in range(1, 14)] + ["C%d" % i for i in range(1, 27)]
    e_pre, l_pre = "embedding_layer.embedding_layer.embedding_layers.%s.weight", \
        "fm.lr_layer.embedding_layer.embedding_layer.embedding_layers.%s.weight"
    B, D = 64, 16
    keep, emb, lr = [], [], []
    for n in names:
        ids = np.ascontiguousarray(X[n])
        we, wl = _f32(P[e_pre % n]), _f32(P[l_pre % n])
        ge, gl = np.zeros_like(we), np.zeros_like(wl)
        num = n.startswith("I")
        keep.append((ids, we, wl, ge, gl))
        emb.append(C.field(ids, we.reshape(-1) if num else we, ge.reshape(-1) if num else ge,
                           kind=C.NUMERIC if num else C.CATEGORICAL, dim=D, padding_idx=None if num else 0))
        lr.append(C.field(ids, wl.reshape(-1) if num else wl, gl.reshape(-1) if num else gl,
                          kind=C.NUMERIC if num else C.CATEGORICAL, dim=1, padding_idx=None if num else 0))
    ea, la = C.array_of(emb), C.array_of(lr)
    bias = _f32(P["fm.lr_layer.bias"])
    logit, ssum = np.zeros(B, np.float32), np.zeros((B, D), np.float32)
    lib.orc_fm_fwd(ea, la, 39, ctypes.c_int64(B), C.ptr(bias), C.ptr(logit), C.ptr(ssum))
    assert_close(logit.reshape(B, 1), fx["out"]["logit"], TOL, "logit")
    y = _f32(X["label"]).reshape(-1)
    g = ((1.0 / (1.0 + np.exp(-logit.astype(np.float64))) - y) / B).astype(np.float32)   # d mean-BCE / d logit
    dbias = np.zeros(1, np.float32)
    lib.orc_fm_bwd(ea, la, 39, ctypes.c_int64(B), C.ptr(g), C.ptr(ssum), C.ptr(dbias))
    assert_close(dbias, G["fm.lr_layer.bias"], TOL)
    for n, (_, _, _, ge, gl) in zip(names, keep):
        assert_close(ge.reshape(G[e_pre % n].shape), G[e_pre % n], TOL, "emb grad " + n)
        assert_close(gl.reshape(G[l_pre % n].shape), G[l_pre % n], TOL, "lr grad " + n)


def test_interaction_modes_match_fixture():
    lib = C.load()
    fx = Fixture("inner_product")
    E = _f32(fx["in"]["E"])
    B, F, D = E.shape
    P = F * (F - 1) // 2
    for mode, key, shape in ((0, "product_sum", (B, 1)), (1, "bi_interaction", (B, D)),
                             (2, "inner_product", (B, P)), (3, "elementwise_product", (B, P, D))):
        out = np.zeros(shape, np.float32)
        lib.orc_interaction_fwd(C.ptr(E), ctypes.c_int64(B), F, D, mode, C.ptr(out))
        assert_close(out, fx["out"][key], TOL, key)
    x = np.arange(12, dtype=np.float32).reshape(2, 3, 2)
    out = np.zeros((2, 1), np.float32)
    lib.orc_interaction_fwd(C.ptr(x), ctypes.c_int64(2), 3, 2, 0, C.ptr(out))
    assert out.tolist() == [[31.0], [427.0]]


def test_rechub_id_mask_pools_match_fixture():
    lib = C.load()
    fx = Fixture("rechub_embedding")
    X, P = fx["in"], fx["p"]
    D, B = 8, 7
    W = {k: _f32(P["embed_dict.%s.weight" % k]) for k in ("uid", "iid", "tags")}
    ids = {k: np.ascontiguousarray(X[k]) for k in ("uid", "hist_mean", "hist_sum", "iid", "price", "age", "tags")}
    # squeeze layout: sparse slots in feature order, then the dense values
    fields = [C.field(ids["uid"], W["uid"], dim=D, out_off=0),
              C.field(ids["hist_mean"], W["iid"], dim=D, out_off=D, pool=C.POOL_MEAN_ID, mask_id=0, eps=1e-16),
              C.field(ids["hist_sum"], W["iid"], dim=D, out_off=2 * D, pool=C.POOL_SUM_ID, mask_id=0),
              C.field(ids["iid"], W["iid"], dim=D, out_off=3 * D),
              C.field(ids["tags"], W["tags"], dim=D, out_off=4 * D, pool=C.POOL_MEAN_ID, mask_id=-1, eps=1e-16),
              C.field(ids["price"], kind=C.DENSE, dim=1, out_off=5 * D),
              C.field(ids["age"], kind=C.DENSE, dim=1, out_off=5 * D + 1)]
    out = np.zeros((B, 5 * D + 2), np.float32)
    scale = np.zeros((7, B), np.float32)
    assert lib.orc_embed_fwd(C.array_of(fields), 7, ctypes.c_int64(B), C.ptr(out), ctypes.c_int64(5 * D + 2),
                             C.ptr(scale)) == 0
    assert_close(out, fx["out"]["squeezed"], TOL)


# ---- sampler + corpus gather ---------------------------------------------------------------
def test_philox_known_answers():
    """Random123 v1.14 kat_vectors for philox4x32-10: the generator the sampler is defined on."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert tuple(int(x) for x in C.philox4x32_10(ctr, key)) == want


def test_negsample_oracle_distribution_and_exclusion():
    """The reference's contract (h5_generator.py:61-84): column 0 = positives, negatives uniform over
    [0, num_items) with replacement; ignore_pos_items never returns an item of the query."""
    n_items, rows, negs = 50, 4000, 5
    pos = np.arange(rows, dtype=np.int64) % n_items
    s = C.negsample(n_items, rows, negs, seed=11, pos=pos)
    assert s.shape == (rows, 1 + negs) and (s[:, 0] == pos).all()
    neg = s[:, 1:]
    assert neg.min() >= 0 and neg.max() < n_items
    counts = np.bincount(neg.reshape(-1), minlength=n_items)
    expected = rows * negs / n_items
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 100.0                                     # 49 dof: P(chi2 > 100) < 3e-5
    assert (C.negsample(n_items, rows, negs, seed=11, pos=pos) == s).all()          # reproducible
    assert (C.negsample(n_items, rows, negs, seed=12, pos=pos) != s)[:, 1:].mean() > 0.9
    # exclusion: query q interacted with items {q, q+1, ..., q+9} mod 50
    query = np.arange(rows, dtype=np.int64) % 7
    off = np.arange(8, dtype=np.int64) * 10
    items = np.concatenate([np.sort((q + np.arange(10)) % n_items) for q in range(7)]).astype(np.int64)
    e = C.negsample(n_items, rows, negs, seed=11, pos=pos, query=query, excl_offsets=off, excl_items=items)
    for q in range(7):
        bad = set(items[off[q]:off[q + 1]].tolist())
        assert not (set(e[query == q][:, 1:].reshape(-1).tolist()) & bad)
    keep = ~np.isin(neg, items[:10]) | (query != 0)[:, None]
    assert (e[:, 1:][keep & (query == 0)[:, None]] == neg[keep & (query == 0)[:, None]]).all()   # attempt 0 kept when legal


def test_gather_rows_oracle_matches_reference_loader_fixture():
    """orc_gather_rows == TrainDataset.__getitem__ + collate_fn of the live reference (fixture)."""
    fx = Fixture("matching_loader")
    idx = fx["in"]["all_item_indexes"][fx["in"]["batch_index"]]
    for k, v in fx["corpus"].items():
        got, bad = C.gather_rows(v, idx.reshape(-1))
        assert not bad and got.dtype == fx["item"][k].dtype
        assert (got == fx["item"][k]).all(), k
    for k, v in fx["data"].items():
        got, _ = C.gather_rows(v, fx["in"]["batch_index"])
        assert (got == fx["user"][k]).all(), k
    _, bad = C.gather_rows(fx["corpus"]["item_id"], np.array([0, 99], dtype=np.int64))
    assert bad


def test_negsample_oracle_never_returns_an_excluded_item_even_for_a_query_that_saw_almost_everything():
    """after 64 rejected draws the sampler used to keep its last draw, i.e. could emit one of the query's own
    items as a negative.  Now: one exact draw over the complement (uniform over the items the query did not interact with,
    the reference's ignore_pos_items distribution)."""
    n_items, rows, negs = 200, 64, 8
    allowed = {0: [7, 123], 1: [199], 2: [0, 1, 2, 3]}
    off, items = [0], []
    for q in range(3):
        ex = [i for i in range(n_items) if i not in allowed[q]]
        items += ex
        off.append(len(items))
    query = np.arange(rows) % 3
    out = C.negsample(n_items, rows, negs, seed=5, query=query, excl_offsets=np.array(off), excl_items=np.array(items))
    for r in range(rows):
        assert set(out[r].tolist()) <= set(allowed[int(query[r])])
    assert {7, 123} == set(out[query == 0].reshape(-1).tolist())            # both survivors are drawn
    assert {0, 1, 2, 3} == set(out[query == 2].reshape(-1).tolist())
