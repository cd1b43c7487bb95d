"""Pin oracle/torch_ref.py against the fixtures produced by the LIVE reference.

CPU only.  Tolerance 1e-5 here (same ATen kernels on both sides; the 1e-4 bar
of BASELINE.json applies to the HIP path).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from conftest import Fixture, assert_close, assert_grads_close, load_params
from oracle import torch_ref as R

TOL = 1e-5


class _FM(object):
    def __init__(self, features, num_fields=None):
        self.features = OrderedDict(features)
        self.num_fields = num_fields if num_fields is not None else sum(
            1 for s in features.values() if s["type"] != "meta")


def ranking_embedding_features():
    f = OrderedDict()
    f["n1"] = {"source": "user", "type": "numeric"}
    f["c1"] = {"source": "user", "type": "categorical", "vocab_size": 11, "padding_idx": 0}
    f["n2"] = {"source": "item", "type": "numeric"}
    f["hist"] = {"source": "user", "type": "sequence", "vocab_size": 23, "padding_idx": 0, "max_len": 5,
                 "feature_encoder": "layers.MaskedAveragePooling()"}
    f["c2"] = {"source": "item", "type": "categorical", "vocab_size": 23, "padding_idx": 0,
               "share_embedding": "hist"}
    f["c3"] = {"source": "context", "type": "categorical", "vocab_size": 5}
    f["meta_id"] = {"source": "user", "type": "meta"}
    return f


CRITEO_SMALL_VOCABS = [37, 21, 101, 97, 13, 7, 53, 29, 3, 61, 43, 89, 31, 11, 47, 83, 5, 41, 19, 4, 71,
                       9, 8, 67, 17, 59]


def criteo_small_features():
    f = OrderedDict()
    for i in range(13):
        f["I%d" % (i + 1)] = {"source": "", "type": "numeric"}
    for i, v in enumerate(CRITEO_SMALL_VOCABS):
        f["C%d" % (i + 1)] = {"source": "", "type": "categorical", "vocab_size": v + 1, "padding_idx": 0}
    return f


def test_ranking_feature_embedding():
    fx = Fixture("ranking_feature_embedding")
    fm = _FM(ranking_embedding_features())
    layer = load_params(R.RefFeatureEmbedding(fm, 8), fx["p"])
    X = fx.tensors("in")
    out = layer(X)
    assert_close(out, fx["out"]["emb"], TOL, "emb")
    (out * X["R"]).sum().backward()
    assert_grads_close(layer, fx["g"], TOL)
    assert_close(layer(X, feature_source=["user"]), fx["out"]["emb_user"], TOL)
    assert_close(layer(X, feature_type="categorical"), fx["out"]["emb_cat"], TOL)
    assert_close(layer(X, dynamic_emb_dim=True), fx["out"]["emb_dyn"], TOL)
    # share_embedding aliases the same module object
    assert layer.embedding_layer.embedding_layers["c2"] is layer.embedding_layer.embedding_layers["hist"]


def test_ranking_fm():
    fx = Fixture("ranking_fm")
    fm = _FM(criteo_small_features())
    model = load_params(R.RefFMModel(fm, 16), fx["p"])
    X = fx.tensors("in")
    emb = model.embedding_layer(X)
    assert_close(emb, fx["out"]["feature_emb"], TOL)
    logit = model.fm(X, emb)
    assert_close(logit, fx["out"]["logit"], TOL)
    loss = R.ranking_bce_loss(torch.sigmoid(logit), X["label"])
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_inner_product():
    fx = Fixture("inner_product")
    t = fx.tensors("in")
    for mode in ("product_sum", "bi_interaction", "inner_product", "elementwise_product"):
        e = t["E"].clone().requires_grad_(True)
        o = R.inner_product_interaction(e, mode)
        assert_close(o, fx["out"][mode], TOL, mode)
        (o * t["R_" + mode]).sum().backward()
        assert_close(e.grad, fx["g"][mode], TOL, "grad " + mode)
    for rs in (1, 0):
        e = t["E"].clone().requires_grad_(True)
        o = R.rechub_fm(e, bool(rs))
        key = "rechub_fm_%d" % rs
        assert_close(o, fx["out"][key], TOL)
        (o * t["R_" + key]).sum().backward()
        assert_close(e.grad, fx["g"][key], TOL)


def test_inner_product_kats():
    """Known answers taken from the live reference."""
    x = torch.arange(12.0).reshape(2, 3, 2)
    assert R.inner_product_interaction(x, "product_sum").tolist() == [[31.0], [427.0]]
    assert R.inner_product_interaction(x, "bi_interaction").tolist() == [[8.0, 23.0], [188.0, 239.0]]
    assert R.inner_product_interaction(x, "inner_product").tolist() == [[3.0, 5.0, 23.0], [111.0, 137.0, 179.0]]
    assert R.rechub_fm(x, True).tolist() == [[31.0], [427.0]]


def test_pooling():
    fx = Fixture("pooling")
    t = fx.tensors("in")
    keep = t["keep"]
    cases = {"core_avg": lambda e: R.value_masked_mean(e),
             "core_sum": R.plain_sum,
             "rank_avg": lambda e: R.value_masked_mean(e),
             "rank_avg_mask": lambda e: R.value_masked_mean(e, mask=keep),
             "rank_sum": R.plain_sum,
             "rechub_avg": lambda e: R.id_masked_pool(e, keep.unsqueeze(1).float(), "mean"),
             "rechub_sum": lambda e: R.id_masked_pool(e, keep.unsqueeze(1).float(), "sum"),
             "rechub_avg_nomask": lambda e: R.id_masked_pool(e, None, "mean"),
             "rechub_sum_nomask": lambda e: R.id_masked_pool(e, None, "sum")}
    for key, fn in cases.items():
        e = t["E"].clone().requires_grad_(True)
        o = fn(e)
        assert_close(o, fx["out"][key], TOL, key)
        (o * t["R"]).sum().backward()
        assert_close(e.grad, fx["g"][key], TOL, "grad " + key)


def test_pooling_kats():
    s = torch.tensor([[[1., 2.], [3., 4.], [0., 0.]], [[0., 0.], [0., 0.], [0., 0.]]])
    assert R.value_masked_mean(s).tolist() == [[2.0, 3.0], [0.0, 0.0]]
    m = torch.tensor([[[1., 1., 0.]], [[0., 0., 0.]]])
    assert R.id_masked_pool(s, m, "mean").tolist() == [[2.0, 3.0], [0.0, 0.0]]
    assert R.id_masked_pool(s, m, "sum").tolist() == [[4.0, 6.0], [0.0, 0.0]]
    # value-mask corner: a non-zero row summing to 0 is not counted
    assert R.value_masked_mean(torch.tensor([[[1., -1.], [3., 4.]]])).tolist() == [[4.0, 3.0]]


def test_backward_kats():
    """Backward KATs of SURVEY.md 8c (shared table, padding row, all-pad sample)."""
    f = OrderedDict()
    f["h"] = {"source": "", "type": "sequence", "vocab_size": 6, "padding_idx": 0, "max_len": 4,
              "feature_encoder": "layers.MaskedAveragePooling()"}
    f["c"] = {"source": "", "type": "categorical", "vocab_size": 6, "padding_idx": 0, "share_embedding": "h"}
    layer = R.RefFeatureEmbedding(_FM(f), 2)
    tab = layer.embedding_layer.embedding_layers["h"]
    assert tab is layer.embedding_layer.embedding_layers["c"]
    with torch.no_grad():
        tab.weight.copy_(torch.arange(12.0).reshape(6, 2))
        tab.weight[0].zero_()
    X = {"h": torch.tensor([[1, 2, 2, 0], [0, 0, 0, 0]]), "c": torch.tensor([2, 0])}
    out = layer(X)
    assert_close(out, torch.tensor([[[10 / 3, 13 / 3], [4., 5.]], [[0., 0.], [0., 0.]]]), 1e-6)
    out.sum().backward()
    want = torch.zeros(6, 2)
    want[1] = 1 / 3
    want[2] = 5 / 3
    assert_close(tab.weight.grad, want, 1e-6)
    # rechub: id mask, no padding_idx on the table; concat gives the pad row gradient
    fe = [R.RefSequenceFeature("s", 6, 2, pooling="mean", padding_idx=0)]
    lay = R.RefRechubEmbeddingLayer(fe)
    with torch.no_grad():
        lay.embed_dict["s"].weight.copy_(torch.arange(12.0).reshape(6, 2))
    o = lay({"s": torch.tensor([[1, 2, 2, 0]])}, fe)
    assert_close(o, torch.tensor([[[10 / 3, 13 / 3]]]), 1e-6)
    o.sum().backward()
    want = torch.zeros(6, 2)
    want[1], want[2] = 1 / 3, 2 / 3
    assert_close(lay.embed_dict["s"].weight.grad, want, 1e-6)
    fe = [R.RefSequenceFeature("s", 6, 2, pooling="concat", padding_idx=0)]
    lay = R.RefRechubEmbeddingLayer(fe)
    o = lay({"s": torch.tensor([[1, 2, 2, 0]])}, fe)
    assert tuple(o.shape) == (1, 1, 4, 2)
    o.sum().backward()
    want = torch.zeros(6, 2)
    want[0], want[1], want[2] = 1, 1, 2
    assert_close(lay.embed_dict["s"].weight.grad, want, 1e-6)


class _MFM(object):
    def __init__(self, specs):
        self.feature_specs = OrderedDict(specs)


def matching_specs():
    s = OrderedDict()
    s["user_id"] = {"source": "user", "type": "categorical", "vocab_size": 13}
    s["user_hist"] = {"source": "user", "type": "sequence", "vocab_size": 20, "padding_idx": 19, "max_len": 6,
                      "embedding_callback": "layers.MaskedAveragePooling()"}
    s["age"] = {"source": "user", "type": "numeric"}
    s["item_id"] = {"source": "item", "type": "categorical", "vocab_size": 20, "padding_idx": 19,
                    "share_embedding": "user_hist"}
    return s


def test_matching_embedding():
    fx = Fixture("matching_embedding")
    layer = load_params(R.RefEmbeddingLayer(_MFM(matching_specs()), 8), fx["p"])
    X = fx.tensors("in")
    u = layer(X, feature_source="user")
    i = layer(X, feature_source="item")
    assert i.dim() == 2  # single feature -> no field axis
    assert_close(u, fx["out"]["user"], TOL)
    assert_close(i, fx["out"]["item"], TOL)
    ((u * X["Ru"]).sum() + (i * X["Ri"]).sum()).backward()
    assert_grads_close(layer, fx["g"], TOL)


def rechub_embedding_features():
    Sp, Sq, De = R.RefSparseFeature, R.RefSequenceFeature, R.RefDenseFeature
    D = 8
    return [Sp("uid", 13, D), Sq("hist_mean", 21, D, pooling="mean", shared_with="iid", padding_idx=0),
            Sq("hist_sum", 21, D, pooling="sum", shared_with="iid", padding_idx=0),
            Sp("iid", 21, D), De("price"), De("age"), Sq("tags", 9, D, pooling="mean")]


def test_rechub_embedding():
    fx = Fixture("rechub_embedding")
    feats = rechub_embedding_features()
    layer = load_params(R.RefRechubEmbeddingLayer(feats), fx["p"])
    X = fx.tensors("in")
    sq = layer(X, feats, squeeze_dim=True)
    ns = layer(X, [f for f in feats if f.kind != "dense"], squeeze_dim=False)
    assert_close(sq, fx["out"]["squeezed"], TOL)
    assert_close(ns, fx["out"]["stacked"], TOL)
    ((sq * X["Rq"]).sum() + (ns * X["Rn"]).sum()).backward()
    assert_grads_close(layer, fx["g"], TOL)
    cfe = [R.RefSequenceFeature("seq", 11, 8, pooling="concat"),
           R.RefSequenceFeature("pos", 11, 8, pooling="concat", shared_with="seq")]
    cl = load_params(R.RefRechubEmbeddingLayer(cfe), fx["pc"])
    co = cl({"seq": X["c_seq"], "pos": X["c_pos"]}, cfe)
    assert_close(co, fx["out"]["concat"], TOL)
    (co * X["Rc"]).sum().backward()
    assert_grads_close(cl, fx["gc"], TOL)


def dssm_features(D=16):
    Sp, Sq = R.RefSparseFeature, R.RefSequenceFeature
    uf = [Sp("user_id", 61, D), Sp("gender", 3, D),
          Sq("hist_movie_id", 38, D, pooling="mean", shared_with="movie_id", padding_idx=0)]
    itf = [Sp("movie_id", 38, D), Sp("cate_id", 7, D)]
    return uf, itf


def test_rechub_dssm():
    fx = Fixture("rechub_dssm")
    uf, itf = dssm_features()
    model = R.RefDSSM(uf, itf, {"dims": [32, 16], "activation": "prelu"},
                      {"dims": [32, 16], "activation": "prelu"}, temperature=0.02)
    load_params(model, fx["p"]).train()
    X = fx.tensors("in")
    p = model(X)
    assert_close(p, fx["out"]["y"], TOL)
    loss = F.binary_cross_entropy(p, X["label"])
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def youtubednn_features(D=16):
    Sp, Sq = R.RefSparseFeature, R.RefSequenceFeature
    uf = [Sp("user_id", 61, D), Sq("hist_movie_id", 38, D, pooling="mean", shared_with="movie_id", padding_idx=0)]
    itf = [Sp("movie_id", 38, D)]
    ngf = [Sq("neg_items", 38, D, pooling="concat", shared_with="movie_id")]
    return uf, itf, ngf


def test_rechub_youtubednn():
    fx = Fixture("rechub_youtubednn")
    uf, itf, ngf = youtubednn_features()
    model = load_params(R.RefYoutubeDNN(uf, itf, ngf, {"dims": [32, 16]}, temperature=0.02), fx["p"]).train()
    X = fx.tensors("in")
    y = model(X)
    assert tuple(y.shape) == (64, 4)
    assert_close(y, fx["out"]["y"], 5e-5)   # logits/0.02 amplify rounding
    loss = F.cross_entropy(y, torch.zeros(64, dtype=torch.long))
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], 5e-5)


def deepfm_features(D=16):
    dense = [R.RefDenseFeature("I%d" % i) for i in range(1, 4)]
    sparse = [R.RefSparseFeature("C%d" % (i + 1), v + 1, D) for i, v in enumerate(CRITEO_SMALL_VOCABS[:8])]
    return dense, sparse


def test_rechub_deepfm():
    fx = Fixture("rechub_deepfm")
    dense, sparse = deepfm_features()
    model = R.RefDeepFM(sparse + dense, sparse, {"dims": [32, 16], "dropout": 0.0, "activation": "relu"})
    load_params(model, fx["p"]).train()
    X = fx.tensors("in")
    p = model(X)
    assert_close(p, fx["out"]["y"], TOL)
    loss = F.binary_cross_entropy(p, X["label"])
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def sasrec_features(V=31, D=8):
    Sq = R.RefSequenceFeature
    return [Sq("seq", V, D, pooling="concat"), Sq("pos", V, D, pooling="concat", shared_with="seq"),
            Sq("neg", V, D, pooling="concat", shared_with="seq")]


def test_rechub_sasrec():
    fx = Fixture("rechub_sasrec")
    model = load_params(R.RefSASRec(sasrec_features(), max_len=12, dropout_rate=0.0, num_blocks=2, num_heads=1),
                        fx["p"]).train()
    X = fx.tensors("in")
    pl, nl = model(X)
    assert_close(pl, fx["out"]["pos_logits"], TOL)
    assert_close(nl, fx["out"]["neg_logits"], TOL)
    m = (X["pos"] != 0).float()
    loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_rechub_sasrec_d64():
    """cfg-5 shape (D = 64, one head, L = 200) from the live reference: pins the oracle at the shape whose attention runs
    on the MFMA kernel on the GPU."""
    fx = Fixture("rechub_sasrec_d64")
    model = load_params(R.RefSASRec(sasrec_features(97, 64), max_len=200, dropout_rate=0.0, num_blocks=2, num_heads=1),
                        fx["p"]).train()
    X = fx.tensors("in")
    pl, nl = model(X)
    assert_close(pl, fx["out"]["pos_logits"], TOL)
    assert_close(nl, fx["out"]["neg_logits"], TOL)
    m = (X["pos"] != 0).float()
    loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_mlp():
    fx = Fixture("mlp")
    x = fx.tensors("in")["x"]
    mods = {"core": R.RefMLP(12, [16, 8], "ReLU", output_dim=3, batch_norm=True),
            "block": R.RefMLP(12, [16, 8], "ReLU", output_dim=1, batch_norm=False),
            "rechub": R.RefRechubMLP(12, True, [16, 8], 0, "relu")}
    for key, m in mods.items():
        load_params(m, {k[len(key) + 1:]: v for k, v in fx["p"].items() if k.startswith(key + ".")}).train()
        xi = x.clone().requires_grad_(True)
        o = m(xi)
        assert_close(o, fx["out"][key], TOL, key)
        (o * torch.from_numpy(fx["out"]["R_" + key])).sum().backward()
        assert_close(xi.grad, fx["g"][key + ".x"], TOL)
        for n, p in m.named_parameters():
            assert_close(p.grad, fx["g"][key + "." + n], TOL, key + "." + n)


def test_attention_and_losses():
    fx = Fixture("attention_losses")
    t = fx.tensors("in")
    q, k, v = (t[n].clone().requires_grad_(True) for n in ("Q", "K", "V"))
    o, a = R.scaled_dot_product_attention(q, k, v, scale=8 ** 0.5, mask=t["mask"])
    assert_close(o, fx["out"]["attn_out"], TOL)
    assert_close(a, fx["out"]["attn"], TOL)
    (o * t["R"]).sum().backward()
    for n, g in (("Q", q.grad), ("K", k.grad), ("V", v.grad)):
        assert_close(g, fx["g"][n], TOL)
    yp = t["y_pred"].clone().requires_grad_(True)
    l = R.softmax_cross_entropy_loss(yp)
    assert_close(l, fx["out"]["softmax_ce"], TOL)
    l.backward()
    assert_close(yp.grad, fx["g"]["softmax_ce"], TOL)
    yp = t["y_pred"].clone().requires_grad_(True)
    l = R.sigmoid_cross_entropy_loss(yp, t["y_true"])
    assert_close(l, fx["out"]["sigmoid_ce"], TOL)
    l.backward()
    assert_close(yp.grad, fx["g"]["sigmoid_ce"], TOL)


def test_second_oracle_recbole_fm_and_attention_layers():
    """the oracle against a SECOND statement of the same arithmetic -- RecBole's ``FMEmbedding`` +
    ``BaseFactorizationMachine`` and the ``MultiHeadAttention`` layer of its ``TransformerEncoder``
    (third_party/recbole/model/layers.py:127-205,380-470; fixture: oracle/gen_golden_recbole.py from the live reference).
    The FM term of both of the oracle's interaction functions, and a plain-torch restatement of the attention layer in
    the op sequence the HIP path composes (projections, causal softmax attention per head, output projection, residual,
    LayerNorm eps 1e-12), equal the fixture: outputs and every gradient."""
    import torch
    from oracle import torch_ref as R
    fx = Fixture("recbole_layers")
    fm = fx.tensors("fm")
    table = fm["p.table"].clone().requires_grad_()
    e = table[fm["in.ids"] + fm["in.offsets"].unsqueeze(0)]
    for got in (R.inner_product_interaction(e, "product_sum"), R.rechub_fm(e, True)):
        assert_close(got, fm["out.sum"], 1e-5, "FM term")
    assert_close(R.inner_product_interaction(e, "bi_interaction"), fm["out.vec"], 1e-5, "FM vector")
    (R.rechub_fm(e, True) * fm["in.R"]).sum().backward()
    assert_close(table.grad, fm["g.table"], 1e-5, "FM table gradient")
    m = fx.tensors("mha")
    heads = int(m["in.heads"])
    x = m["in.x"].clone().requires_grad_()
    p = dict((k[2:], v.clone().requires_grad_()) for k, v in m.items() if k.startswith("p."))
    B, L, H = x.shape
    hd = H // heads

    def split(t):
        return t.view(B, L, heads, hd).transpose(1, 2)
    q = split(torch.nn.functional.linear(x, p["query.weight"], p["query.bias"]))
    k = split(torch.nn.functional.linear(x, p["key.weight"], p["key.bias"]))
    v = split(torch.nn.functional.linear(x, p["value.weight"], p["value.bias"]))
    s = (q @ k.transpose(-1, -2)) / hd ** 0.5
    s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), -1e9)
    ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, H)
    y = torch.nn.functional.layer_norm(torch.nn.functional.linear(ctx, p["dense.weight"], p["dense.bias"]) + x, (H,),
                                       p["LayerNorm.weight"], p["LayerNorm.bias"], 1e-12)
    assert_close(y, m["out.y"], 1e-5, "attention layer output")
    (y * m["in.R"]).sum().backward()
    assert_close(x.grad, m["g.x"], 1e-5, "attention layer dx")
    for name, t in p.items():
        assert_close(t.grad, m["g." + name], 1e-5, "attention layer grad " + name)
