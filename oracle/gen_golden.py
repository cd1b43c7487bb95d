"""Generate tests/golden/*.npz by running the LIVE reference (dev container only).

TEST INFRASTRUCTURE.  Usage:  python oracle/gen_golden.py

Imports reczoo/RecBox from /root/reference through ``oracle/ref_shim.py``,
drives its own modules (not restatements) with seeded inputs and writes small
input/weight/output/gradient vectors.  The fixtures are DATA ONLY; no reference
source text is stored.  Recipe: torch.manual_seed(0) for
weights re-initialised to N(0, 0.1) (the default 1e-4 init would make a 1e-4
tolerance vacuous), ids from a Generator seeded with 1, B in {1, 7, 64}.
Each fixture stores: inputs ``in.*``, parameters ``p.*`` (state_dict keys),
outputs ``out.*`` and parameter gradients ``g.*`` (keyed like ``p.*``).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def reinit(module, std=0.1, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        seen = set()
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            p.copy_(torch.randn(p.shape, generator=g) * std)
        for m in module.modules():
            if isinstance(m, torch.nn.Embedding) and m.padding_idx is not None:
                m.weight[m.padding_idx].zero_()


def pack(prefix, d):
    return {prefix + "." + k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
            for k, v in d.items()}


def grads_of(module):
    out = {}
    for name, p in module.named_parameters():
        out[name] = p.grad if p.grad is not None else torch.zeros_like(p)
    return out


def save(name, **groups):
    flat = {}
    for prefix, d in groups.items():
        flat.update(pack(prefix, d))
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **flat)
    print("wrote %-34s %6.1f KB  %d arrays" % (name + ".npz", os.path.getsize(path) / 1024.0, len(flat)))


# ---------------------------------------------------------------------------
# schema helpers (build the reference's own FeatureMap objects)
# ---------------------------------------------------------------------------
def ranking_feature_map(fuxictr, features, num_fields=None, emb_dim=None):
    fm = fuxictr.features.FeatureMap("golden", "/tmp")
    fm.features = OrderedDict(features)
    fm.num_fields = num_fields if num_fields is not None else sum(
        1 for s in features.values() if s["type"] != "meta")
    fm.labels = ["label"]
    fm.default_emb_dim = emb_dim
    fm.set_column_index()
    return fm


def gen_ranking_embedding(fuxictr, B=7, D=8):
    feats = OrderedDict()
    feats["n1"] = {"source": "user", "type": "numeric"}
    feats["c1"] = {"source": "user", "type": "categorical", "vocab_size": 11, "padding_idx": 0}
    feats["n2"] = {"source": "item", "type": "numeric"}
    feats["hist"] = {"source": "user", "type": "sequence", "vocab_size": 23, "padding_idx": 0,
                     "max_len": 5, "feature_encoder": "layers.MaskedAveragePooling()"}
    feats["c2"] = {"source": "item", "type": "categorical", "vocab_size": 23, "padding_idx": 0,
                   "share_embedding": "hist"}
    feats["c3"] = {"source": "context", "type": "categorical", "vocab_size": 5}
    feats["meta_id"] = {"source": "user", "type": "meta"}
    fm = ranking_feature_map(fuxictr, feats, emb_dim=D)
    L = fuxictr.pytorch.layers
    layer = L.FeatureEmbedding(fm, D)
    reinit(layer)
    g = torch.Generator().manual_seed(1)
    X = OrderedDict()
    X["n1"] = torch.rand(B, generator=g, dtype=torch.float64)
    X["c1"] = torch.randint(0, 11, (B,), generator=g).double()       # ids arrive as float64 (a-2)
    X["n2"] = torch.rand(B, generator=g, dtype=torch.float64)
    h = torch.randint(1, 23, (B, 5), generator=g)
    lens = torch.randint(0, 6, (B,), generator=g)
    h = h * (torch.arange(5)[None, :] < lens[:, None])
    X["hist"] = h.double()
    X["c2"] = torch.randint(0, 23, (B,), generator=g).double()
    X["c3"] = torch.randint(0, 5, (B,), generator=g)
    out = layer(X)
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    out_user = layer(X, feature_source=["user"])
    out_cat = layer(X, feature_type="categorical")
    out_dyn = layer(X, dynamic_emb_dim=True)
    save("ranking_feature_embedding", **{"in": dict(X, R=R), "p": layer.state_dict(),
                                         "out": {"emb": out, "emb_user": out_user, "emb_cat": out_cat,
                                                 "emb_dyn": out_dyn},
                                         "g": grads_of(layer)})


CRITEO_SMALL_VOCABS = [37, 21, 101, 97, 13, 7, 53, 29, 3, 61, 43, 89, 31, 11, 47, 83, 5, 41, 19, 4, 71,
                       9, 8, 67, 17, 59]


def gen_ranking_fm(fuxictr, B=64, D=16):
    feats = OrderedDict()
    for i in range(13):
        feats["I%d" % (i + 1)] = {"source": "", "type": "numeric"}
    for i, v in enumerate(CRITEO_SMALL_VOCABS):
        feats["C%d" % (i + 1)] = {"source": "", "type": "categorical", "vocab_size": v + 1, "padding_idx": 0}
    fm = ranking_feature_map(fuxictr, feats, emb_dim=D)
    L = fuxictr.pytorch.layers
    emb = L.FeatureEmbedding(fm, D)
    fmb = L.FactorizationMachine(fm)
    model = torch.nn.ModuleDict({"embedding_layer": emb, "fm": fmb})
    reinit(model)
    g = torch.Generator().manual_seed(1)
    X = OrderedDict()
    for i in range(13):
        X["I%d" % (i + 1)] = torch.rand(B, generator=g, dtype=torch.float64)
    for i, v in enumerate(CRITEO_SMALL_VOCABS):
        X["C%d" % (i + 1)] = torch.randint(1, v + 1, (B,), generator=g).double()
    X["C3"][:3] = 0.0  # a few padding ids
    y = (torch.rand(B, 1, generator=g) < 0.25).float()
    feature_emb = emb(X)
    logit = fmb(X, feature_emb)
    prob = torch.sigmoid(logit)
    loss = torch.nn.functional.binary_cross_entropy(prob, y, reduction="mean")
    loss.backward()
    save("ranking_fm", **{"in": dict(X, label=y), "p": model.state_dict(),
                          "out": {"feature_emb": feature_emb, "logit": logit, "prob": prob, "loss": loss},
                          "g": grads_of(model)})


def gen_inner_product(fuxictr, B=7, F=6, D=8):
    L = fuxictr.pytorch.layers
    g = torch.Generator().manual_seed(1)
    E = torch.randn(B, F, D, generator=g)
    outs, grads, Rs = {}, {}, {}
    for mode in ("product_sum", "bi_interaction", "inner_product", "elementwise_product"):
        e = E.clone().requires_grad_(True)
        o = L.InnerProductInteraction(F, output=mode)(e)
        R = torch.randn(o.shape, generator=g)
        (o * R).sum().backward()
        outs[mode], grads[mode], Rs["R_" + mode] = o, e.grad, R
    from recbox.third_party.rechub.basic.layers import FM
    for rs in (True, False):
        e = E.clone().requires_grad_(True)
        o = FM(reduce_sum=rs)(e)
        R = torch.randn(o.shape, generator=g)
        (o * R).sum().backward()
        key = "rechub_fm_%d" % int(rs)
        outs[key], grads[key], Rs["R_" + key] = o, e.grad, R
    save("inner_product", **{"in": dict(Rs, E=E), "out": outs, "g": grads})


def gen_pooling(recbox, fuxictr, B=7, Lh=6, D=8):
    g = torch.Generator().manual_seed(1)
    E = torch.randn(B, Lh, D, generator=g)
    lens = torch.tensor([0, 1, 6, 3, 2, 5, 4])
    keep = (torch.arange(Lh)[None, :] < lens[:, None])
    E = E * keep[:, :, None]
    E[2, 1, :] = 0.0
    E[2, 1, 0], E[2, 1, 1] = 1.5, -1.5   # non-zero row that sums to zero (value-mask corner)
    outs, grads = {}, {}
    R = torch.randn(B, D, generator=g)
    core = recbox.core.pytorch.layers
    rk = fuxictr.pytorch.layers
    from recbox.third_party.rechub.basic import layers as rh
    cases = [("core_avg", lambda e: core.MaskedAveragePooling()(e)),
             ("core_sum", lambda e: core.MaskedSumPooling()(e)),
             ("rank_avg", lambda e: rk.MaskedAveragePooling()(e)),
             ("rank_avg_mask", lambda e: rk.MaskedAveragePooling()(e, mask=keep)),
             ("rank_sum", lambda e: rk.MaskedSumPooling()(e)),
             ("rechub_avg", lambda e: rh.AveragePooling()(e, keep.unsqueeze(1).float())),
             ("rechub_sum", lambda e: rh.SumPooling()(e, keep.unsqueeze(1).float())),
             ("rechub_avg_nomask", lambda e: rh.AveragePooling()(e)),
             ("rechub_sum_nomask", lambda e: rh.SumPooling()(e))]
    for key, fn in cases:
        e = E.clone().requires_grad_(True)
        o = fn(e)
        (o * R).sum().backward()
        outs[key], grads[key] = o, e.grad
    save("pooling", **{"in": {"E": E, "keep": keep, "R": R}, "out": outs, "g": grads})


def gen_matching_embedding(recbox, B=7, D=8):
    from recbox.matching.features import FeatureMap
    fm = FeatureMap("golden", "/tmp", None, None, "label")
    specs = OrderedDict()
    specs["user_id"] = {"source": "user", "type": "categorical", "vocab_size": 13}
    specs["user_hist"] = {"source": "user", "type": "sequence", "vocab_size": 20, "padding_idx": 19,
                          "max_len": 6, "embedding_callback": "layers.MaskedAveragePooling()"}
    specs["age"] = {"source": "user", "type": "numeric"}
    specs["item_id"] = {"source": "item", "type": "categorical", "vocab_size": 20, "padding_idx": 19,
                        "share_embedding": "user_hist"}
    fm.feature_specs = specs
    fm.num_fields = 4
    layer = recbox.core.pytorch.layers.EmbeddingLayer(fm, D)
    reinit(layer)
    g = torch.Generator().manual_seed(1)
    X = OrderedDict()
    X["user_id"] = torch.randint(0, 13, (B,), generator=g)
    h = torch.randint(0, 19, (B, 6), generator=g).int()
    lens = torch.randint(0, 7, (B,), generator=g)
    h = torch.where(torch.arange(6)[None, :] < lens[:, None], h, torch.full_like(h, 19))
    X["user_hist"] = h                                   # int32 sequences (a-2)
    X["age"] = torch.rand(B, generator=g)
    X["item_id"] = torch.randint(0, 19, (B,), generator=g)
    u = layer(X, feature_source="user")
    i = layer(X, feature_source="item")
    Ru, Ri = torch.randn(u.shape, generator=g), torch.randn(i.shape, generator=g)
    ((u * Ru).sum() + (i * Ri).sum()).backward()
    save("matching_embedding", **{"in": dict(X, Ru=Ru, Ri=Ri), "p": layer.state_dict(),
                                  "out": {"user": u, "item": i}, "g": grads_of(layer)})


def _rechub_feats(rh):
    from recbox.third_party.rechub.basic.features import SparseFeature, SequenceFeature, DenseFeature
    return SparseFeature, SequenceFeature, DenseFeature


def gen_rechub_embedding(B=7, D=8):
    from recbox.third_party.rechub.basic import layers as rh
    Sp, Sq, De = _rechub_feats(rh)
    feats = [Sp("uid", 13, D), Sq("hist_mean", 21, D, pooling="mean", shared_with="iid", padding_idx=0),
             Sq("hist_sum", 21, D, pooling="sum", shared_with="iid", padding_idx=0),
             Sp("iid", 21, D), De("price"), De("age"), Sq("tags", 9, D, pooling="mean")]
    layer = rh.EmbeddingLayer(feats)
    reinit(layer)
    g = torch.Generator().manual_seed(1)
    X = OrderedDict()
    X["uid"] = torch.randint(0, 13, (B,), generator=g)
    h = torch.randint(1, 21, (B, 6), generator=g)
    lens = torch.randint(0, 7, (B,), generator=g)
    X["hist_mean"] = h * (torch.arange(6)[None, :] < lens[:, None])
    X["hist_sum"] = X["hist_mean"].flip(0)
    X["iid"] = torch.randint(1, 21, (B,), generator=g)
    X["price"] = torch.rand(B, generator=g)
    X["age"] = torch.rand(B, generator=g)
    X["tags"] = torch.randint(0, 9, (B, 3), generator=g)     # no padding_idx: every id counts
    sq = layer(X, feats, squeeze_dim=True)
    sparse_only = [f for f in feats if not isinstance(f, De)]
    ns = layer(X, sparse_only, squeeze_dim=False)
    Rq, Rn = torch.randn(sq.shape, generator=g), torch.randn(ns.shape, generator=g)
    ((sq * Rq).sum() + (ns * Rn).sum()).backward()
    # concat pooling (SASRec path): pad row receives gradient
    cfe = [Sq("seq", 11, D, pooling="concat"), Sq("pos", 11, D, pooling="concat", shared_with="seq")]
    clayer = rh.EmbeddingLayer(cfe)
    reinit(clayer, seed=3)
    Xc = {"seq": torch.randint(0, 11, (B, 5), generator=g), "pos": torch.randint(0, 11, (B, 5), generator=g)}
    co = clayer(Xc, cfe)
    Rc = torch.randn(co.shape, generator=g)
    (co * Rc).sum().backward()
    save("rechub_embedding", **{"in": dict(X, Rq=Rq, Rn=Rn, c_seq=Xc["seq"], c_pos=Xc["pos"], Rc=Rc),
                                "p": layer.state_dict(), "pc": clayer.state_dict(),
                                "out": {"squeezed": sq, "stacked": ns, "concat": co},
                                "g": grads_of(layer), "gc": grads_of(clayer)})


def gen_rechub_models(B=64):
    from recbox.third_party.rechub.basic import layers as rh
    from recbox.third_party.rechub.models.matching import DSSM, YoutubeDNN, SASRec
    from recbox.third_party.rechub.models.ranking import DeepFM
    Sp, Sq, De = _rechub_feats(rh)
    g = torch.Generator().manual_seed(1)
    D = 16
    # ---- DSSM (cfg 1 shape in miniature) ----
    uf = [Sp("user_id", 61, D), Sp("gender", 3, D), Sq("hist_movie_id", 38, D, pooling="mean",
                                                          shared_with="movie_id", padding_idx=0)]
    itf = [Sp("movie_id", 38, D), Sp("cate_id", 7, D)]
    model = DSSM(uf, itf, user_params={"dims": [32, 16], "activation": "prelu"},
                 item_params={"dims": [32, 16], "activation": "prelu"}, temperature=0.02)
    reinit(model)
    X = OrderedDict()
    X["user_id"] = torch.randint(0, 61, (B,), generator=g)
    X["gender"] = torch.randint(0, 3, (B,), generator=g)
    h = torch.randint(1, 38, (B, 10), generator=g)
    lens = torch.randint(1, 11, (B,), generator=g)
    X["hist_movie_id"] = h * (torch.arange(10)[None, :] < lens[:, None])
    X["movie_id"] = torch.randint(1, 38, (B,), generator=g)
    X["cate_id"] = torch.randint(0, 7, (B,), generator=g)
    y = (torch.rand(B, generator=g) < 0.5).float()
    model.train()
    p = model(X)
    loss = torch.nn.functional.binary_cross_entropy(p, y)
    loss.backward()
    save("rechub_dssm", **{"in": dict(X, label=y), "p": model.state_dict(),
                           "out": {"y": p, "loss": loss}, "g": grads_of(model)})
    # ---- YoutubeDNN ----
    uf = [Sp("user_id", 61, D), Sq("hist_movie_id", 38, D, pooling="mean", shared_with="movie_id",
                                   padding_idx=0)]
    itf = [Sp("movie_id", 38, D)]
    ngf = [Sq("neg_items", 38, D, pooling="concat", shared_with="movie_id")]
    model = YoutubeDNN(uf, itf, ngf, user_params={"dims": [32, D]}, temperature=0.02)
    reinit(model)
    X2 = OrderedDict((k, X[k]) for k in ("user_id", "hist_movie_id", "movie_id"))
    X2["neg_items"] = torch.randint(1, 38, (B, 3), generator=g)
    model.train()
    logits = model(X2)
    loss = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long))
    loss.backward()
    save("rechub_youtubednn", **{"in": X2, "p": model.state_dict(),
                                 "out": {"y": logits, "loss": loss}, "g": grads_of(model)})
    # ---- DeepFM ----
    dense = [De("I%d" % i) for i in range(1, 4)]
    sparse = [Sp("C%d" % (i + 1), v + 1, D) for i, v in enumerate(CRITEO_SMALL_VOCABS[:8])]
    model = DeepFM(deep_features=sparse + dense, fm_features=sparse,
                   mlp_params={"dims": [32, 16], "dropout": 0.0, "activation": "relu"})
    reinit(model)
    X3 = OrderedDict()
    for i in range(1, 4):
        X3["I%d" % i] = torch.rand(B, generator=g)
    for i, v in enumerate(CRITEO_SMALL_VOCABS[:8]):
        X3["C%d" % (i + 1)] = torch.randint(0, v + 1, (B,), generator=g)
    model.train()
    p = model(X3)
    loss = torch.nn.functional.binary_cross_entropy(p, y)
    loss.backward()
    save("rechub_deepfm", **{"in": dict(X3, label=y), "p": model.state_dict(),
                             "out": {"y": p, "loss": loss}, "g": grads_of(model)})
    # ---- SASRec ----
    Bs, Ls, Ds, V = 7, 12, 8, 31
    fe = [Sq("seq", V, Ds, pooling="concat"), Sq("pos", V, Ds, pooling="concat", shared_with="seq"),
          Sq("neg", V, Ds, pooling="concat", shared_with="seq")]
    model = SASRec(fe, max_len=Ls, dropout_rate=0.0, num_blocks=2, num_heads=1)
    reinit(model)
    lens = torch.randint(2, Ls + 1, (Bs,), generator=g)
    seq = torch.randint(1, V, (Bs, Ls), generator=g) * (torch.arange(Ls)[None, :] < lens[:, None])
    pos = torch.roll(seq, -1, dims=1) * (seq != 0)
    neg = torch.randint(1, V, (Bs, Ls), generator=g) * (seq != 0)
    Xs = {"seq": seq, "pos": pos, "neg": neg}
    model.train()
    pl, nl = model(Xs)
    m = (pos != 0).float()
    loss = -((torch.nn.functional.logsigmoid(pl) + torch.nn.functional.logsigmoid(-nl)) * m).sum() / m.sum()
    loss.backward()
    save("rechub_sasrec", **{"in": Xs, "p": model.state_dict(),
                             "out": {"pos_logits": pl, "neg_logits": nl, "loss": loss},
                             "g": grads_of(model)})


def gen_rechub_sasrec_d64(B=2, L=200, D=64, V=97):
    """BASELINE.json cfg 5's shape (D = 64 with ONE head => head_dim 64, L = 200): the build's MFMA attention kernel is
    the one that runs here (the small rechub_sasrec fixture, head_dim 8, exercises the VALU kernel)."""
    from recbox.third_party.rechub.basic.features import SequenceFeature as Sq
    from recbox.third_party.rechub.models.matching import SASRec
    g = torch.Generator().manual_seed(3)
    fe = [Sq("seq", V, D, pooling="concat"), Sq("pos", V, D, pooling="concat", shared_with="seq"),
          Sq("neg", V, D, pooling="concat", shared_with="seq")]
    model = SASRec(fe, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1)
    reinit(model, seed=5)
    lens = torch.tensor([L, 57])[:B]
    seq = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    pos = torch.roll(seq, -1, dims=1) * (seq != 0)
    neg = torch.randint(1, V, (B, L), generator=g) * (seq != 0)
    Xs = {"seq": seq, "pos": pos, "neg": neg}
    model.train()
    pl, nl = model(Xs)
    m = (pos != 0).float()
    loss = -((torch.nn.functional.logsigmoid(pl) + torch.nn.functional.logsigmoid(-nl)) * m).sum() / m.sum()
    loss.backward()
    save("rechub_sasrec_d64", **{"in": Xs, "p": model.state_dict(),
                                 "out": {"pos_logits": pl, "neg_logits": nl, "loss": loss},
                                 "g": grads_of(model)})


def gen_mlp(recbox, fuxictr, B=7):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 12, generator=g)
    out, grads, params = {}, {}, {}
    # recbox.core MLP_Layer needs recbox.utils.torch_utils.set_activation
    core = recbox.core.pytorch.layers.MLP_Layer(12, output_dim=3, hidden_units=[16, 8],
                                                 hidden_activations="ReLU", final_activation=None,
                                                 dropout_rates=[0, 0], batch_norm=True)
    blk = fuxictr.pytorch.layers.MLP_Block(12, hidden_units=[16, 8], hidden_activations="ReLU",
                                           output_dim=1, output_activation=None, dropout_rates=0.0,
                                           batch_norm=False)
    from recbox.third_party.rechub.basic.layers import MLP
    rh = MLP(12, output_layer=True, dims=[16, 8], dropout=0, activation="relu")
    for key, m in (("core", core), ("block", blk), ("rechub", rh)):
        reinit(m, seed={"core": 0, "block": 1, "rechub": 2}[key])
        m.train()
        xi = x.clone().requires_grad_(True)
        o = m(xi)
        R = torch.randn(o.shape, generator=g)
        (o * R).sum().backward()
        out[key] = o
        out["R_" + key] = R
        grads[key + ".x"] = xi.grad
        for n, p in m.named_parameters():
            grads[key + "." + n] = p.grad
        for n, p in m.state_dict().items():
            params[key + "." + n] = p
    save("mlp", **{"in": {"x": x}, "p": params, "out": out, "g": grads})


def gen_attention_and_losses(recbox, fuxictr, B=3, H=2, L=9, D=8):
    g = torch.Generator().manual_seed(1)
    Q, K, V = (torch.randn(B, H, L, D, generator=g) for _ in range(3))
    mask = torch.tril(torch.ones(L, L)).expand(B, H, L, L)
    att = fuxictr.pytorch.layers.ScaledDotProductAttention(0.0)
    q, k, v = (t.clone().requires_grad_(True) for t in (Q, K, V))
    o, a = att(q, k, v, scale=D ** 0.5, mask=mask)
    R = torch.randn(o.shape, generator=g)
    (o * R).sum().backward()
    from recbox.core.pytorch import losses
    yp = torch.randn(7, 5, generator=g)
    yt = torch.zeros(7, 5)
    yt[:, 0] = 1
    yp1 = yp.clone().requires_grad_(True)
    l_sm = losses.SoftmaxCrossEntropyLoss()(yp1, yt)
    l_sm.backward()
    yp2 = yp.clone().requires_grad_(True)
    l_sg = losses.SigmoidCrossEntropyLoss()(yp2, yt)
    l_sg.backward()
    save("attention_losses", **{"in": {"Q": Q, "K": K, "V": V, "mask": mask, "R": R, "y_pred": yp, "y_true": yt},
                                "out": {"attn_out": o, "attn": a, "softmax_ce": l_sm, "sigmoid_ce": l_sg},
                                "g": {"Q": q.grad, "K": k.grad, "V": v.grad, "softmax_ce": yp1.grad,
                                      "sigmoid_ce": yp2.grad}})


def gen_target_attention_and_listwise_losses(recbox, fuxictr, B=5, L=7, E=16):
    g = torch.Generator().manual_seed(1)
    att = fuxictr.pytorch.layers.MultiHeadTargetAttention(input_dim=E, attention_dim=16, num_heads=2, dropout_rate=0,
                                                           use_scale=True, use_qkvo=True)
    reinit(att, seed=4)
    tgt = torch.randn(B, E, generator=g).requires_grad_(True)
    hist = torch.randn(B, L, E, generator=g).requires_grad_(True)
    lens = torch.tensor([7, 3, 1, 5, 2])
    mask = (torch.arange(L)[None, :] < lens[:, None]).float()
    out = att(tgt, hist, mask)
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    from recbox.core.pytorch import losses
    yp = torch.randn(6, 5, generator=g)
    yt = torch.zeros(6, 5)
    yt[:, 0] = 1
    louts, lgrads = {}, {}
    for name, fn in (("pairwise_logistic", losses.PairwiseLogisticLoss()), ("pairwise_margin", losses.PairwiseMarginLoss(0.5)),
                     ("mse", losses.MSELoss()), ("ccl", losses.CosineContrastiveLoss(0.1)),
                     ("ccl_weighted", losses.CosineContrastiveLoss(0.1, negative_weight=2.0))):
        y = yp.clone().requires_grad_(True)
        v = fn(y, yt)
        v.backward()
        louts[name], lgrads[name] = v, y.grad
    save("target_attention_losses", **{"in": {"target": tgt, "history": hist, "mask": mask, "R": R, "y_pred": yp, "y_true": yt},
                                       "p": att.state_dict(), "out": dict(louts, attn=out),
                                       "g": dict(lgrads, target=tgt.grad, history=hist.grad,
                                                 **{"p." + n: p.grad for n, p in att.named_parameters()})})


def gen_matching_loader(recbox, N=23, I=17, B=8, num_negs=3):
    """TrainDataset.__getitem__ + collate_fn / collate_fn_unique of the live reference on a fixed
    all_item_indexes block (the sampling itself is numpy's MT19937 stream and is not a fixture)."""
    from torch.utils.data.dataloader import default_collate  # noqa: F401  (what the reference's collate uses)
    import importlib
    ref = importlib.import_module("recbox.matching.pytorch.dataloaders.h5_generator")   # (the package re-exports a
    # function of the same name, so `import ... as` would bind that instead of the module)
    rng = np.random.RandomState(3)
    data = OrderedDict([("user_id", rng.randint(1, 40, size=N).astype(np.int64)),
                        ("user_history", rng.randint(0, I, size=(N, 5)).astype(np.int32)),
                        ("user_age", rng.rand(N).astype(np.float32))])
    corpus = OrderedDict([("item_id", np.arange(I, dtype=np.int64) + 100),
                          ("cate_id", rng.randint(1, 6, size=I).astype(np.int64)),
                          ("item_tags", rng.randint(0, 9, size=(I, 3)).astype(np.int32)),
                          ("item_price", rng.rand(I).astype(np.float64))])
    labels = np.ones(N, dtype=np.float64)
    all_idx = rng.randint(0, I, size=(N, 1 + num_negs)).astype(np.int64)
    all_idx[::4, 2] = all_idx[::4, 1]                       # duplicates inside a row and across rows
    ds = object.__new__(ref.TrainDataset)                   # bypass load_h5 (h5py is not in this image)
    ds.data_dict, ds.num_samples = dict(data), N
    ds.item_corpus_dict, ds.num_items = dict(corpus), I
    ds.labels, ds.all_item_indexes = labels, all_idx
    batch_index = rng.permutation(N)[:B]
    samples = [ds[int(i)] for i in batch_index]
    u, it, lab, inv = ref.collate_fn(samples)
    u2, it2, lab2, inv2 = ref.collate_fn_unique([ds[int(i)] for i in batch_index])
    save("matching_loader",
         **{"data": data, "corpus": corpus, "in": {"labels": labels, "all_item_indexes": all_idx, "batch_index": batch_index},
            "user": u, "item": it, "out": {"labels": lab},
            "user_u": u2, "item_u": it2, "out_u": {"labels": lab2, "inverse_indexes": inv2}})


class _ExactIPIndex(object):
    """What faiss.IndexFlatIP is (exact inner-product search on fp32 copies, faiss.py:3-15), in numpy: faiss itself
    is not installed in this image.  Only the SEARCH is replaced; masking, sorting and every metric below run in the
    reference's own evaluate_block / metric classes."""

    class _Idx(object):
        pass

    def __init__(self, corpus_vecs, dim, l2_normalize=False, index_name="IndexFlatIP"):
        self.vecs = corpus_vecs.astype("float32")
        self.index = self._Idx()
        self.index.ntotal = self.vecs.shape[0]

    def search(self, query_vecs, topk=50):
        scores = query_vecs.astype("float32") @ self.vecs.T
        order = np.argsort(-scores, axis=1, kind="stable")[:, :topk]
        return np.take_along_axis(scores, order, axis=1), order


def gen_retrieval_metrics(recbox, U=37, I=640, D=16):
    """evaluate_metrics / evaluate_block of the live reference (core/metrics.py) on seeded embeddings."""
    import importlib
    ref = importlib.import_module("recbox.core.metrics")
    ref.FaissIndex = _ExactIPIndex
    rng = np.random.RandomState(5)
    user = rng.randn(U, D).astype(np.float64)
    item = rng.randn(I, D).astype(np.float64)
    n_q = 41
    query = rng.permutation(n_q)[:U].astype(np.int64)
    train, valid = {}, {}
    for q in range(n_q):
        n_tr = int(rng.randint(0, 60)) if q % 5 else int(rng.randint(300, 520))     # some users mask most of the top 500
        train[q] = rng.choice(I, n_tr, replace=False).tolist()
        v = rng.choice(I, int(rng.randint(1, 30)), replace=False).tolist()
        valid[q] = v + v[:2] if q % 3 == 0 else v                                    # duplicates in the label list
    metrics = ["Recall(k=5)", "Recall(k=50)", "nRecall(k=20)", "Precision(k=10)", "F1(k=10)", "DCG(k=10)", "NDCG(k=20)",
               "NDCG(k=50)", "MRR(k=50)", "HitRate(k=10)", "MAP(k=50)"]
    funcs = [eval(m, vars(ref)) for m in metrics]
    per_user = np.array(ref.evaluate_block(user, _ExactIPIndex(item, D), query, train, valid, funcs, 50))
    avg = ref.evaluate_metrics(user, item, train, valid, query, metrics)
    # top-50 item lists the reference scored (recomputed the same way, for the index-level check)
    sc, ind = _ExactIPIndex(item, D).search(user, topk=500)
    mask = np.zeros((U, I))
    for i, q in enumerate(query):
        mask[i, train[q]] = 1
    sc += -1e9 * np.take_along_axis(mask, ind, axis=1)
    top = np.take_along_axis(ind, np.argsort(-sc, axis=1), axis=1)[:, :50]
    def csr(d):
        off = np.zeros(n_q + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(d[q]) for q in range(n_q)])
        return off, np.array([x for q in range(n_q) for x in d[q]], dtype=np.int64)
    tr_off, tr_items = csr(train)
    va_off, va_items = csr(valid)
    save("retrieval_metrics",
         **{"in": {"user": user, "item": item, "query": query, "train_off": tr_off, "train_items": tr_items,
                   "valid_off": va_off, "valid_items": va_items},
            "out": {"per_user": per_user, "average": np.array([avg[m] for m in metrics]), "top50": top,
                    "metrics": np.array(metrics)}})


def gen_cross_net(fuxictr, B=7, dim=24, layers=3):
    """CrossNet / CrossNetV2 / CrossInteraction of the live reference (cross_net.py:22-59)."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, dim, generator=g)
    R = torch.randn(B, dim, generator=g)
    groups = {"in": {"x": x, "R": R}}
    for name, cls in (("v1", FL.CrossNet), ("v2", FL.CrossNetV2)):
        net = cls(dim, layers)
        reinit(net, std=0.2, seed=3)
        xi = x.clone().requires_grad_(True)
        out = net(xi)
        (out * R).sum().backward()
        groups["p_" + name] = net.state_dict()
        groups["out_" + name] = {"y": out, "dx": xi.grad}
        groups["g_" + name] = grads_of(net)
    save("cross_net", **groups)


def gen_cross_net_mix(fuxictr, B=9, dim=20, layers=2, rank=6, experts=3):
    """CrossNetMix of the live reference (cross_net.py:60-117): mixture of low-rank cross experts."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, dim, generator=g)
    R = torch.randn(B, dim, generator=g)
    net = FL.CrossNetMix(dim, layer_num=layers, low_rank=rank, num_experts=experts)
    reinit(net, std=0.3, seed=13)
    xi = x.clone().requires_grad_(True)
    out = net(xi)
    (out * R).sum().backward()
    save("cross_net_mix", **{"in": {"x": x, "R": R}, "p": net.state_dict(), "out": {"y": out, "dx": xi.grad},
                             "g": grads_of(net)})


def gen_im_hfm(fuxictr, B=6, F=5, D=8):
    """InteractionMachine (orders 2 and 5, with and without BatchNorm) and HolographicInteraction (three types) of the
    live reference (interaction_machine.py:21-71, holographic_interaction.py:22-52)."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, F, D, generator=g) * 0.7
    groups = {"in": {"x": x}}
    for tag, order, bn in (("o2", 2, False), ("o5bn", 5, True)):
        net = FL.InteractionMachine(D, order=order, batch_norm=bn)
        reinit(net, std=0.3, seed=order)
        net.train()
        xi = x.clone().requires_grad_(True)
        out = net(xi)
        Rm = torch.randn(out.shape, generator=g)
        (out * Rm).sum().backward()
        groups["p_" + tag] = {k: v for k, v in net.state_dict().items() if "running" not in k and "num_batches" not in k}
        groups["out_" + tag] = {"y": out, "dx": xi.grad, "R": Rm}
        groups["g_" + tag] = grads_of(net)
    for kind in ("hadamard_product", "circular_convolution", "circular_correlation"):
        layer = FL.HolographicInteraction(F, interaction_type=kind)
        xi = x.clone().requires_grad_(True)
        out = layer(xi)
        Rh = torch.randn(out.shape, generator=g)
        (out * Rh).sum().backward()
        groups["hfm_" + kind] = {"y": out, "dx": xi.grad, "R": Rh}
    save("im_hfm", **groups)


def gen_senet_din(fuxictr, B=6, F=7, D=8, L=5):
    """SqueezeExcitation (both activations) and DIN_Attention (ReLU / Dice units, with and without softmax) of the live
    reference (squeeze_excitation.py:21-41, target_attention.py:25-66, activations.py:23-32)."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, F, D, generator=g)
    target = torch.randn(B, D, generator=g)
    hist = torch.randn(B, L, D, generator=g)
    mask = (torch.rand(B, L, generator=g) < 0.7).float()
    mask[:, 0] = 1.0
    groups = {"in": {"x": x, "target": target, "hist": hist, "mask": mask}}
    for act in ("ReLU", "Sigmoid"):
        net = FL.SqueezeExcitation(F, reduction_ratio=3, excitation_activation=act)
        reinit(net, std=0.4, seed=3)
        xi = x.clone().requires_grad_(True)
        out = net(xi)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        groups["p_se_" + act] = net.state_dict()
        groups["out_se_" + act] = {"y": out, "dx": xi.grad, "R": R}
        groups["g_se_" + act] = grads_of(net)
    for tag, acts, soft in (("relu", "ReLU", False), ("dice_soft", "Dice", True)):
        net = FL.DIN_Attention(embedding_dim=D, attention_units=[12, 6], hidden_activations=acts, use_softmax=soft)
        reinit(net, std=0.3, seed=7)
        net.train()
        t, h = target.clone().requires_grad_(True), hist.clone().requires_grad_(True)
        out = net(t, h, mask)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        groups["p_din_" + tag] = {k: v for k, v in net.state_dict().items() if "running" not in k and "num_batches" not in k}
        groups["out_din_" + tag] = {"y": out, "dt": t.grad, "dh": h.grad, "R": R}
        groups["g_din_" + tag] = grads_of(net)
    save("senet_din", **groups)


def gen_bilinear(fuxictr, B=7, F=5, D=4):
    """BilinearInteraction (pair loop) and BilinearInteractionV2 (index_select) of the live reference, three W layouts."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, F, D, generator=g)
    R = torch.randn(B, F * (F - 1) // 2, D, generator=g)
    groups = {"in": {"x": x, "R": R}}
    for kind in ("field_all", "field_each", "field_interaction"):
        for ver, cls in (("v1", FL.BilinearInteraction), ("v2", FL.BilinearInteractionV2)):
            layer = cls(F, D, bilinear_type=kind)
            with torch.no_grad():
                layer.bilinear_W.copy_(torch.randn(layer.bilinear_W.shape, generator=torch.Generator().manual_seed(5)) * 0.3)
            xi = x.clone().requires_grad_(True)
            out = layer(xi)
            (out * R).sum().backward()
            groups["out_%s_%s" % (kind, ver)] = {"y": out, "dx": xi.grad, "W": layer.bilinear_W.detach(),
                                                 "dW": layer.bilinear_W.grad}
    save("bilinear", **groups)


def gen_cin(fuxictr, B=7, F=5, D=4):
    """CompressedInteractionNet of the live reference (compressed_interaction_net.py:21-48), two layers."""
    import fuxictr.pytorch.layers as FL
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, F, D, generator=g)
    net = FL.CompressedInteractionNet(F, [6, 3], output_dim=2)
    reinit(net, std=0.3, seed=6)
    xi = x.clone().requires_grad_(True)
    out = net(xi)
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    save("cin", **{"in": {"x": x, "R": R}, "p": net.state_dict(), "out": {"y": out, "dx": xi.grad}, "g": grads_of(net)})


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    recbox = ref_shim.import_reference()
    import fuxictr
    import fuxictr.features
    import fuxictr.pytorch.layers
    import recbox.core.pytorch.layers
    import recbox.matching.features
    if len(sys.argv) > 1:                     # regenerate only the named fixtures: python oracle/gen_golden.py rechub_sasrec_d64
        for name in sys.argv[1:]:
            {"rechub_sasrec_d64": gen_rechub_sasrec_d64}[name]()
        return
    gen_rechub_sasrec_d64()
    gen_ranking_embedding(fuxictr)
    gen_ranking_fm(fuxictr)
    gen_inner_product(fuxictr)
    gen_pooling(recbox, fuxictr)
    gen_matching_embedding(recbox)
    gen_rechub_embedding()
    gen_rechub_models()
    gen_mlp(recbox, fuxictr)
    gen_attention_and_losses(recbox, fuxictr)
    gen_target_attention_and_listwise_losses(recbox, fuxictr)
    gen_matching_loader(recbox)
    gen_retrieval_metrics(recbox)
    gen_cross_net(fuxictr)
    gen_bilinear(fuxictr)
    gen_cin(fuxictr)
    gen_cross_net_mix(fuxictr)
    gen_im_hfm(fuxictr)
    gen_senet_din(fuxictr)


if __name__ == "__main__":
    main()
