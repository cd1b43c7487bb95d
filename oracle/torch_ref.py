"""CPU restatement (PyTorch fp32, plain ATen ops) of the reference hot path.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file.  The product
(``recbox_amd``) never does: it runs hand-written HIP kernels and raises when
its extension is missing.

What is restated, with the reference location each item follows (paths are
relative to /root/reference/recbox):

  pooling            core/pytorch/layers/sequence.py:4-20,
                     ranking/pytorch/layers/pooling.py:22-40,
                     third_party/rechub/basic/layers.py:176-230
  embedding layers   core/pytorch/layers/embedding.py:10-138,
                     ranking/pytorch/layers/embeddings/feature_embedding.py:28-214,
                     third_party/rechub/basic/layers.py:29-148
  interactions       ranking/pytorch/layers/interactions/inner_product.py:22-56,
                     third_party/rechub/basic/layers.py:269-292
  LR / FM blocks     ranking/pytorch/layers/blocks/logistic_regression.py:23-35,
                     ranking/pytorch/layers/blocks/factorization_machine.py:24-34
  MLP towers         core/pytorch/layers/mlp.py:7-39,
                     ranking/pytorch/layers/blocks/mlp_block.py:23-61,
                     third_party/rechub/basic/layers.py:233-266
  attention          ranking/pytorch/layers/attentions/dot_product_attention.py:23-43,
                     third_party/rechub/models/matching/sasrec.py:65-124
  two-tower models   third_party/rechub/models/matching/dssm.py:15-66,
                     third_party/rechub/models/matching/youtube_dnn.py:14-71,
                     third_party/rechub/models/ranking/deepfm.py:14-42
  losses             core/pytorch/losses/softmax_crossentropy_loss.py:14-22,
                     core/pytorch/losses/sigmoid_crossentropy_loss.py:14-22

Parity pinning: the reference ships no tests, so this
restatement is pinned against the *live reference imported in the dev
container* -- ``oracle/gen_golden.py`` runs the real modules and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays every fixture
through this file.  Parameter names equal the reference's ``state_dict`` keys
so one set of weights drives reference, oracle and product.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------
# pooling
# --------------------------------------------------------------------------
def value_masked_mean(emb, mask=None, eps=1e-12):
    """recbox MaskedAveragePooling: the mask is VALUE based (row sum != 0)."""
    total = emb.sum(dim=1)
    if mask is None:
        mask = emb.sum(dim=-1) != 0
    return total / (mask.float().sum(-1, keepdim=True) + eps)


def plain_sum(emb):
    """recbox MaskedSumPooling: pad rows are zero rows, so a plain sum."""
    return emb.sum(dim=1)


def id_masked_pool(emb, mask, mode):
    """rechub Average/Sum/ConcatPooling; ``mask`` is [B,1,L] float, id based."""
    if mode == "concat":
        return emb
    if mask is None:
        return emb.mean(dim=1) if mode == "mean" else emb.sum(dim=1)
    pooled = torch.bmm(mask, emb).squeeze(1)
    if mode == "sum":
        return pooled
    return pooled / (mask.sum(dim=-1).float() + 1e-16)


class ValueMaskedMean(nn.Module):
    def forward(self, emb, mask=None):
        return value_masked_mean(emb, mask)


class PlainSum(nn.Module):
    def forward(self, emb):
        return plain_sum(emb)


_POOL_BY_NAME = {"MaskedAveragePooling": ValueMaskedMean, "MaskedSumPooling": PlainSum}


def _pool_from_spec(text):
    """'layers.MaskedAveragePooling()' -> module (the reference ``eval``s it)."""
    if text is None:
        return None
    if isinstance(text, (list, tuple)):
        return nn.Sequential(*[_pool_from_spec(t) for t in text])
    name = text.strip().split(".")[-1].split("(")[0]
    if name not in _POOL_BY_NAME:
        raise ValueError("feature_encoder={} is not supported.".format(text))
    return _POOL_BY_NAME[name]()


# --------------------------------------------------------------------------
# interactions
# --------------------------------------------------------------------------
def inner_product_interaction(emb, output="product_sum"):
    """InnerProductInteraction over emb[B,F,D] (inner_product.py:40-56)."""
    if output in ("product_sum", "bi_interaction"):
        sq_of_sum = emb.sum(dim=1) ** 2
        sum_of_sq = (emb ** 2).sum(dim=1)
        bi = (sq_of_sum - sum_of_sq) * 0.5
        return bi if output == "bi_interaction" else bi.sum(dim=-1, keepdim=True)
    nf = emb.shape[1]
    if output == "inner_product":
        gram = torch.bmm(emb, emb.transpose(1, 2))
        keep = torch.triu(torch.ones(nf, nf), 1).bool()
        return torch.masked_select(gram, keep).view(-1, nf * (nf - 1) // 2)
    if output == "elementwise_product":
        iu = torch.triu_indices(nf, nf, offset=1)
        return emb.index_select(1, iu[0]) * emb.index_select(1, iu[1])
    raise ValueError("InnerProductInteraction output={} is not supported.".format(output))


def rechub_fm(emb, reduce_sum=True):
    ix = emb.sum(dim=1) ** 2 - (emb ** 2).sum(dim=1)
    if reduce_sum:
        ix = ix.sum(dim=1, keepdim=True)
    return 0.5 * ix


# --------------------------------------------------------------------------
# embedding layers (recbox flavours)
# --------------------------------------------------------------------------
def _specs_of(feature_map):
    return feature_map.features if hasattr(feature_map, "features") else feature_map.feature_specs


class RefRankingEmbeddingDict(nn.Module):
    """FeatureEmbeddingDict (feature_embedding.py:52-214)."""

    def __init__(self, feature_map, embedding_dim, required=None, not_required=None,
                 use_pretrain=True, use_sharing=True, init_std=1e-4):
        super().__init__()
        self._fm = feature_map
        self.embedding_layers = nn.ModuleDict()
        self.feature_encoders = nn.ModuleDict()
        lr_mode = (not (use_pretrain and use_sharing)) and embedding_dim == 1
        for name, spec in _specs_of(feature_map).items():
            if spec["type"] == "meta":
                continue
            if required and name not in required:
                continue
            if not_required and name in not_required:
                continue
            if lr_mode:
                dim = 1
                if spec["type"] == "sequence":
                    self.feature_encoders[name] = PlainSum()
            else:
                dim = spec.get("embedding_dim", embedding_dim)
                if spec.get("feature_encoder", None):
                    self.feature_encoders[name] = _pool_from_spec(spec["feature_encoder"])
            if use_sharing and spec.get("share_embedding") in self.embedding_layers:
                self.embedding_layers[name] = self.embedding_layers[spec["share_embedding"]]
                continue
            if spec["type"] == "numeric":
                self.embedding_layers[name] = nn.Linear(1, dim, bias=False)
            elif spec["type"] in ("categorical", "sequence"):
                self.embedding_layers[name] = nn.Embedding(spec["vocab_size"], dim,
                                                           padding_idx=spec.get("padding_idx", None))
        for mod in self.embedding_layers.values():
            if type(mod) == nn.Embedding:
                if mod.padding_idx is not None:
                    nn.init.normal_(mod.weight[1:, :], std=init_std)
                else:
                    nn.init.normal_(mod.weight, std=init_std)

    def forward(self, X, feature_source=(), feature_type=()):
        src = [feature_source] if isinstance(feature_source, str) else list(feature_source or [])
        typ = [feature_type] if isinstance(feature_type, str) else list(feature_type or [])
        out = OrderedDict()
        for name, spec in _specs_of(self._fm).items():
            if src and spec["source"] not in src:
                continue
            if typ and spec["type"] not in typ:
                continue
            if name not in self.embedding_layers:
                continue
            if spec["type"] == "numeric":
                e = self.embedding_layers[name](X[name].float().view(-1, 1))
            elif spec["type"] in ("categorical", "sequence"):
                e = self.embedding_layers[name](X[name].long())
            else:
                raise NotImplementedError
            if name in self.feature_encoders:
                e = self.feature_encoders[name](e)
            out[name] = e
        return out

    def dict2tensor(self, d, dynamic_emb_dim=False):
        vals = [d[k] for k in _specs_of(self._fm) if k in d]
        return torch.cat(vals, dim=-1) if dynamic_emb_dim else torch.stack(vals, dim=1)


class RefFeatureEmbedding(nn.Module):
    """FeatureEmbedding (feature_embedding.py:28-49)."""

    def __init__(self, feature_map, embedding_dim, required_feature_columns=None,
                 not_required_feature_columns=None, use_pretrain=True, use_sharing=True):
        super().__init__()
        self.embedding_layer = RefRankingEmbeddingDict(feature_map, embedding_dim,
                                                       required_feature_columns,
                                                       not_required_feature_columns,
                                                       use_pretrain, use_sharing)

    def forward(self, X, feature_source=(), feature_type=(), dynamic_emb_dim=False):
        d = self.embedding_layer(X, feature_source, feature_type)
        return self.embedding_layer.dict2tensor(d, dynamic_emb_dim)


class RefMatchingEmbeddingDict(nn.Module):
    """EmbeddingDictLayer (core/pytorch/layers/embedding.py:30-138)."""

    def __init__(self, feature_map, embedding_dim, disable_sharing_pretrain=False,
                 required=None, not_required=None):
        super().__init__()
        self._fm = feature_map
        self.embedding_layers = nn.ModuleDict()
        self.embedding_callbacks = nn.ModuleDict()
        for name, spec in _specs_of(feature_map).items():
            if required and name not in required:
                continue
            if not_required and name in not_required:
                continue
            if disable_sharing_pretrain:
                assert embedding_dim == 1
                dim = embedding_dim
            else:
                dim = spec.get("embedding_dim", embedding_dim)
            if (not disable_sharing_pretrain) and "embedding_callback" in spec:
                self.embedding_callbacks[name] = _pool_from_spec(spec["embedding_callback"])
            if (not disable_sharing_pretrain) and "share_embedding" in spec:
                self.embedding_layers[name] = self.embedding_layers[spec["share_embedding"]]
                continue
            if spec["type"] == "numeric":
                self.embedding_layers[name] = nn.Linear(1, dim, bias=False)
            elif spec["type"] in ("categorical", "sequence"):
                self.embedding_layers[name] = nn.Embedding(spec["vocab_size"], dim,
                                                           padding_idx=spec.get("padding_idx", None))

    def forward(self, X, feature_source=None, feature_type=None):
        out = OrderedDict()
        for name, spec in _specs_of(self._fm).items():
            if feature_source and spec["source"] != feature_source:
                continue
            if feature_type and spec["type"] != feature_type:
                continue
            if name not in self.embedding_layers:
                continue
            if spec["type"] == "numeric":
                e = self.embedding_layers[name](X[name].float().view(-1, 1))
            elif spec["type"] in ("categorical", "sequence"):
                e = self.embedding_layers[name](X[name].long())
            else:
                raise NotImplementedError
            if name in self.embedding_callbacks:
                e = self.embedding_callbacks[name](e)
            out[name] = e
        return out

    def dict2tensor(self, d):
        vals = list(d.values())
        return vals[0] if len(vals) == 1 else torch.stack(vals, dim=1)


class RefEmbeddingLayer(nn.Module):
    """EmbeddingLayer (core/pytorch/layers/embedding.py:10-27)."""

    def __init__(self, feature_map, embedding_dim, disable_sharing_pretrain=False,
                 required_feature_columns=None, not_required_feature_columns=None):
        super().__init__()
        self.embedding_layer = RefMatchingEmbeddingDict(feature_map, embedding_dim,
                                                        disable_sharing_pretrain,
                                                        required_feature_columns,
                                                        not_required_feature_columns)

    def forward(self, X, feature_source=None):
        return self.embedding_layer.dict2tensor(self.embedding_layer(X, feature_source=feature_source))


# --------------------------------------------------------------------------
# LR / FM blocks (ranking)
# --------------------------------------------------------------------------
class RefLogisticRegression(nn.Module):
    def __init__(self, feature_map, use_bias=True):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1)) if use_bias else None
        self.embedding_layer = RefFeatureEmbedding(feature_map, 1, use_pretrain=False, use_sharing=False)

    def forward(self, X):
        out = self.embedding_layer(X).sum(dim=1)
        if self.bias is not None:
            out = out + self.bias
        return out


class RefInnerProductInteraction(nn.Module):
    def __init__(self, num_fields, output="product_sum"):
        super().__init__()
        if output not in ("product_sum", "bi_interaction", "inner_product", "elementwise_product"):
            raise ValueError("InnerProductInteraction output={} is not supported.".format(output))
        self._output_type = output

    def forward(self, emb):
        return inner_product_interaction(emb, self._output_type)


class RefFactorizationMachine(nn.Module):
    def __init__(self, feature_map):
        super().__init__()
        self.fm_layer = RefInnerProductInteraction(feature_map.num_fields, "product_sum")
        self.lr_layer = RefLogisticRegression(feature_map, use_bias=True)

    def forward(self, X, feature_emb):
        return self.fm_layer(feature_emb) + self.lr_layer(X)


class RefFMModel(nn.Module):
    """FuxiCTR-convention FM model body (the class itself is not in the
    reference tree; RankingModel subclasses do: emb -> fm -> sigmoid,
    ranking_model.py:66-70 applies F.binary_cross_entropy on the sigmoid)."""

    def __init__(self, feature_map, embedding_dim):
        super().__init__()
        self.embedding_layer = RefFeatureEmbedding(feature_map, embedding_dim)
        self.fm = RefFactorizationMachine(feature_map)

    def forward(self, X):
        return self.fm(X, self.embedding_layer(X))


# --------------------------------------------------------------------------
# MLP towers
# --------------------------------------------------------------------------
def _activation(name):
    if name is None:
        return None
    if isinstance(name, nn.Module):
        return name
    low = name.lower()
    table = {"relu": nn.ReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "prelu": nn.PReLU,
             "gelu": nn.GELU, "leakyrelu": nn.LeakyReLU}
    if low == "softmax":
        return nn.Softmax(dim=-1)
    if low in table:
        return table[low]()
    return getattr(nn, name)()


class RefMLP(nn.Module):
    """MLP_Layer (mlp.py:7-39) and MLP_Block (mlp_block.py:23-61); attribute
    ``mlp`` is an nn.Sequential with the reference's child order."""

    def __init__(self, input_dim, hidden_units=(), hidden_activations="ReLU", output_dim=None,
                 output_activation=None, dropout_rates=0.0, batch_norm=False,
                 norm_before_activation=True, use_bias=True):
        super().__init__()
        hidden_units = list(hidden_units)
        if not isinstance(dropout_rates, list):
            dropout_rates = [dropout_rates] * len(hidden_units)
        if not isinstance(hidden_activations, list):
            hidden_activations = [hidden_activations] * len(hidden_units)
        dims = [input_dim] + hidden_units
        mods = []
        for i in range(len(hidden_units)):
            mods.append(nn.Linear(dims[i], dims[i + 1], bias=use_bias))
            if batch_norm and norm_before_activation:
                mods.append(nn.BatchNorm1d(dims[i + 1]))
            act = _activation(hidden_activations[i])
            if act is not None:
                mods.append(act)
            if batch_norm and not norm_before_activation:
                mods.append(nn.BatchNorm1d(dims[i + 1]))
            if dropout_rates[i] > 0:
                mods.append(nn.Dropout(p=dropout_rates[i]))
        if output_dim is not None:
            mods.append(nn.Linear(dims[-1], output_dim, bias=use_bias))
        if output_activation is not None:
            mods.append(_activation(output_activation))
        self.mlp = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp(x)


class RefRechubMLP(nn.Module):
    """rechub MLP: Linear -> BatchNorm1d -> act -> Dropout per layer (layers.py:250-266)."""

    def __init__(self, input_dim, output_layer=True, dims=None, dropout=0, activation="relu"):
        super().__init__()
        mods = []
        for d in (dims or []):
            mods += [nn.Linear(input_dim, d), nn.BatchNorm1d(d), _activation(activation), nn.Dropout(p=dropout)]
            input_dim = d
        if output_layer:
            mods.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp(x)


# --------------------------------------------------------------------------
# rechub feature classes + embedding layer
# --------------------------------------------------------------------------
class RefSparseFeature(object):
    kind = "sparse"

    def __init__(self, name, vocab_size, embed_dim, shared_with=None, padding_idx=None):
        self.name, self.vocab_size, self.embed_dim = name, vocab_size, embed_dim
        self.shared_with, self.padding_idx = shared_with, padding_idx


class RefSequenceFeature(RefSparseFeature):
    kind = "sequence"

    def __init__(self, name, vocab_size, embed_dim, pooling="mean", shared_with=None, padding_idx=None):
        super().__init__(name, vocab_size, embed_dim, shared_with, padding_idx)
        self.pooling = pooling


class RefDenseFeature(object):
    kind = "dense"

    def __init__(self, name):
        self.name, self.embed_dim = name, 1


def _kind(fea):
    if hasattr(fea, "kind"):
        return fea.kind
    n = type(fea).__name__
    return {"SparseFeature": "sparse", "SequenceFeature": "sequence", "DenseFeature": "dense"}[n]


class RefRechubEmbeddingLayer(nn.Module):
    """rechub EmbeddingLayer + InputMask (layers.py:29-148): tables live in
    ``embed_dict``, WITHOUT padding_idx (initializers.py:17)."""

    def __init__(self, features):
        super().__init__()
        self.features = features
        self.embed_dict = nn.ModuleDict()
        self.n_dense = 0
        for fea in features:
            if fea.name in self.embed_dict:
                continue
            k = _kind(fea)
            if k in ("sparse", "sequence") and fea.shared_with is None:
                self.embed_dict[fea.name] = nn.Embedding(fea.vocab_size, fea.embed_dim)
            elif k == "dense":
                self.n_dense += 1

    def forward(self, x, features, squeeze_dim=False):
        sparse, dense = [], []
        for fea in features:
            k = _kind(fea)
            if k == "dense":
                dense.append(x[fea.name].float().unsqueeze(1))
                continue
            table = self.embed_dict[fea.name if fea.shared_with is None else fea.shared_with]
            ids = x[fea.name].long()
            if k == "sparse":
                sparse.append(table(ids).unsqueeze(1))
            else:
                if fea.pooling not in ("sum", "mean", "concat"):
                    raise ValueError("Sequence pooling method supports only pooling in %s, got %s." %
                                     (["sum", "mean"], fea.pooling))
                pad = fea.padding_idx if fea.padding_idx is not None else -1
                mask = (ids != pad).unsqueeze(1).float()
                sparse.append(id_masked_pool(table(ids), mask, fea.pooling).unsqueeze(1))
        have_d, have_s = len(dense) > 0, len(sparse) > 0
        if have_d:
            dense = torch.cat(dense, dim=1)
        if have_s:
            sparse = torch.cat(sparse, dim=1)
        if squeeze_dim:
            if have_d and not have_s:
                return dense
            if have_s and not have_d:
                return sparse.flatten(start_dim=1)
            if have_s and have_d:
                return torch.cat((sparse.flatten(start_dim=1), dense), dim=1)
            raise ValueError("The input features can note be empty")
        if have_s:
            return sparse
        raise ValueError("If keep the original shape:[batch_size, num_features, embed_dim], expected %s "
                         "in feature list, got %s" % ("SparseFeatures", features))


# --------------------------------------------------------------------------
# rechub models
# --------------------------------------------------------------------------
class RefDSSM(nn.Module):
    def __init__(self, user_features, item_features, user_params, item_params, temperature=1.0):
        super().__init__()
        self.user_features, self.item_features = user_features, item_features
        self.temperature = temperature
        self.embedding = RefRechubEmbeddingLayer(user_features + item_features)
        self.user_mlp = RefRechubMLP(sum(f.embed_dim for f in user_features), output_layer=False, **user_params)
        self.item_mlp = RefRechubMLP(sum(f.embed_dim for f in item_features), output_layer=False, **item_params)
        self.mode = None

    def user_tower(self, x):
        if self.mode == "item":
            return None
        u = self.user_mlp(self.embedding(x, self.user_features, squeeze_dim=True))
        return F.normalize(u, p=2, dim=1)

    def item_tower(self, x):
        if self.mode == "user":
            return None
        i = self.item_mlp(self.embedding(x, self.item_features, squeeze_dim=True))
        return F.normalize(i, p=2, dim=1)

    def forward(self, x):
        u, i = self.user_tower(x), self.item_tower(x)
        if self.mode == "user":
            return u
        if self.mode == "item":
            return i
        return torch.sigmoid(torch.mul(u, i).sum(dim=1))


class RefYoutubeDNN(nn.Module):
    def __init__(self, user_features, item_features, neg_item_feature, user_params, temperature=1.0):
        super().__init__()
        self.user_features, self.item_features = user_features, item_features
        self.neg_item_feature, self.temperature = neg_item_feature, temperature
        self.embedding = RefRechubEmbeddingLayer(user_features + item_features)
        self.user_mlp = RefRechubMLP(sum(f.embed_dim for f in user_features), output_layer=False, **user_params)
        self.mode = None

    def user_tower(self, x):
        if self.mode == "item":
            return None
        u = self.user_mlp(self.embedding(x, self.user_features, squeeze_dim=True)).unsqueeze(1)
        u = F.normalize(u, p=2, dim=2)
        return u.squeeze(1) if self.mode == "user" else u

    def item_tower(self, x):
        if self.mode == "user":
            return None
        pos = F.normalize(self.embedding(x, self.item_features, squeeze_dim=False), p=2, dim=2)
        if self.mode == "item":
            return pos.squeeze(1)
        neg = self.embedding(x, self.neg_item_feature, squeeze_dim=False).squeeze(1)
        neg = F.normalize(neg, p=2, dim=2)
        return torch.cat((pos, neg), dim=1)

    def forward(self, x):
        u, i = self.user_tower(x), self.item_tower(x)
        if self.mode == "user":
            return u
        if self.mode == "item":
            return i
        return torch.mul(u, i).sum(dim=2) / self.temperature


class RefRechubLR(nn.Module):
    def __init__(self, input_dim, sigmoid=False):
        super().__init__()
        self.sigmoid = sigmoid
        self.fc = nn.Linear(input_dim, 1, bias=True)

    def forward(self, x):
        y = self.fc(x)
        return torch.sigmoid(y) if self.sigmoid else y


class RefDeepFM(nn.Module):
    def __init__(self, deep_features, fm_features, mlp_params):
        super().__init__()
        self.deep_features, self.fm_features = deep_features, fm_features
        self.linear = RefRechubLR(sum(f.embed_dim for f in fm_features))
        self.embedding = RefRechubEmbeddingLayer(deep_features + fm_features)
        self.mlp = RefRechubMLP(sum(f.embed_dim for f in deep_features), **mlp_params)

    def forward(self, x):
        deep_in = self.embedding(x, self.deep_features, squeeze_dim=True)
        fm_in = self.embedding(x, self.fm_features, squeeze_dim=False)
        y = self.linear(fm_in.flatten(start_dim=1)) + rechub_fm(fm_in, True) + self.mlp(deep_in)
        return torch.sigmoid(y.squeeze(1))


class RefPointWiseFeedForward(nn.Module):
    def __init__(self, hidden_units, dropout_rate):
        super().__init__()
        self.conv1 = nn.Conv1d(hidden_units, hidden_units, kernel_size=1)
        self.dropout1 = nn.Dropout(p=dropout_rate)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv1d(hidden_units, hidden_units, kernel_size=1)
        self.dropout2 = nn.Dropout(p=dropout_rate)

    def forward(self, inputs):
        h = self.dropout1(self.conv1(inputs.transpose(-1, -2)))
        h = self.dropout2(self.conv2(self.relu(h))).transpose(-1, -2)
        return h + inputs


class RefSASRec(nn.Module):
    """rechub SASRec (sasrec.py:17-107).  The reference builds positions and
    masks with CPU-only constructors; this restatement is device neutral but
    arithmetically identical (pad id 0; LayerNorm eps 1e-8; nn.MultiheadAttention
    with a boolean causal mask; dropout applied as written)."""

    def __init__(self, features, max_len=50, dropout_rate=0.5, num_blocks=2, num_heads=1):
        super().__init__()
        self.features = features
        self.embed_dim = features[0].embed_dim
        self.item_emb = RefRechubEmbeddingLayer(features)
        self.position_emb = nn.Embedding(max_len, self.embed_dim)
        self.emb_dropout = nn.Dropout(p=dropout_rate)
        self.attention_layernorms = nn.ModuleList()
        self.attention_layers = nn.ModuleList()
        self.forward_layernorms = nn.ModuleList()
        self.forward_layers = nn.ModuleList()
        self.last_layernorm = nn.LayerNorm(self.embed_dim, eps=1e-8)
        for _ in range(num_blocks):
            self.attention_layernorms.append(nn.LayerNorm(self.embed_dim, eps=1e-8))
            self.attention_layers.append(nn.MultiheadAttention(self.embed_dim, num_heads, dropout_rate))
            self.forward_layernorms.append(nn.LayerNorm(self.embed_dim, eps=1e-8))
            self.forward_layers.append(RefPointWiseFeedForward(self.embed_dim, dropout_rate))

    def seq_forward(self, x, e):
        ids = x["seq"]
        e = e * (self.embed_dim ** 0.5)
        pos = torch.arange(ids.shape[1], device=ids.device).unsqueeze(0).expand(ids.shape[0], -1)
        e = self.emb_dropout(e + self.position_emb(pos))
        keep = (ids != 0).unsqueeze(-1)
        e = e * keep
        L = e.shape[1]
        causal = ~torch.tril(torch.ones((L, L), dtype=torch.bool, device=e.device))
        for i in range(len(self.attention_layers)):
            et = e.transpose(0, 1)
            q = self.attention_layernorms[i](et)
            att, _ = self.attention_layers[i](q, et, et, attn_mask=causal)
            e = (q + att).transpose(0, 1)
            e = self.forward_layers[i](self.forward_layernorms[i](e))
            e = e * keep
        return self.last_layernorm(e)

    def forward(self, x):
        emb = self.item_emb(x, self.features)
        seq, pos, neg = emb[:, 0], emb[:, 1], emb[:, 2]
        out = self.seq_forward(x, seq)
        return (out * pos).sum(dim=-1), (out * neg).sum(dim=-1)


# --------------------------------------------------------------------------
# attention core + losses
# --------------------------------------------------------------------------
def scaled_dot_product_attention(Q, K, V, scale=None, mask=None):
    """ScaledDotProductAttention without dropout (dot_product_attention.py:31-43)."""
    scores = torch.matmul(Q, K.transpose(-1, -2))
    if scale:
        scores = scores / scale
    if mask is not None:
        scores = scores.masked_fill(mask.view_as(scores).float() == 0, -1.e9)
    att = scores.softmax(dim=-1)
    return torch.matmul(att, V), att


def softmax_cross_entropy_loss(y_pred):
    """-log softmax(y)[:,0] averaged (softmax_crossentropy_loss.py:19-22)."""
    return -torch.log(F.softmax(y_pred, dim=1)[:, 0]).mean()


def sigmoid_cross_entropy_loss(y_pred, y_true):
    return F.binary_cross_entropy_with_logits(y_pred.flatten(), y_true.flatten(), reduction="sum")


def ranking_bce_loss(y_prob, y_true):
    """ranking_model.py:66-70: F.binary_cross_entropy on sigmoid outputs, mean."""
    return F.binary_cross_entropy(y_prob, y_true, reduction="mean")
