"""rechub-style layers (drop-in for ``torch_rechub.basic.layers``,
/root/reference/recbox/third_party/rechub/basic/layers.py): ``EmbeddingLayer`` (+``InputMask``
and the three pooling modules), ``MLP``, ``FM``, ``LR``, ``PredictionLayer`` with the same
constructors, outputs, exceptions and parameter names (``embed_dict.<feature>.weight``,
``mlp.<i>.*``, ``fc.*``).  ``EmbeddingLayer.forward`` is ONE ``rbx_embed_fwd`` launch for all
requested features: one-hot lookups, id-masked mean/sum pooling (mask = ``id != padding_idx``,
eps 1e-16, the pad row is a normal trainable row that just gets weight 0), concat pooling and
dense pass-through all land in one ``[B, width]`` row.
"""
import torch
import torch.nn as nn

from ... import _embed_host as host
from ... import dense, ops
from ..._lib import (FIELD_CATEGORICAL, FIELD_DENSE, POOL_CONCAT, POOL_MEAN_ID, POOL_NONE, POOL_SUM_ID)
from .features import DenseFeature, SequenceFeature, SparseFeature


class PredictionLayer(nn.Module):
    def __init__(self, task_type='classification'):
        super(PredictionLayer, self).__init__()
        if task_type not in ["classification", "regression"]:
            raise ValueError("task_type must be classification or regression")
        self.task_type = task_type

    def forward(self, x):
        return torch.sigmoid(x) if self.task_type == "classification" else x


class EmbeddingLayer(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features
        self.embed_dict = nn.ModuleDict()
        self.n_dense = 0
        self._plans = {}
        for fea in features:
            if fea.name in self.embed_dict:
                continue
            if isinstance(fea, SparseFeature) and fea.shared_with == None:  # noqa: E711
                self.embed_dict[fea.name] = fea.get_embedding_layer()
            elif isinstance(fea, SequenceFeature) and fea.shared_with == None:  # noqa: E711
                self.embed_dict[fea.name] = fea.get_embedding_layer()
            elif isinstance(fea, DenseFeature):
                self.n_dense += 1

    def _lookups(self, x, features):
        sparse, dense_l = [], []
        for fea in features:
            if isinstance(fea, DenseFeature):
                dense_l.append(host.Lookup(fea.name, FIELD_DENSE, None, 1))
                continue
            table = self.embed_dict[fea.name if fea.shared_with is None else fea.shared_with]
            if isinstance(fea, SparseFeature):
                sparse.append(host.Lookup(fea.name, FIELD_CATEGORICAL, table, table.embedding_dim))
            elif isinstance(fea, SequenceFeature):
                if fea.pooling not in ("sum", "mean", "concat"):
                    raise ValueError("Sequence pooling method supports only pooling in %s, got %s." %
                                     (["sum", "mean"], fea.pooling))
                pool = {"sum": POOL_SUM_ID, "mean": POOL_MEAN_ID, "concat": POOL_CONCAT}[fea.pooling]
                mask_id = fea.padding_idx if fea.padding_idx is not None else -1   # InputMask: id != -1
                sparse.append(host.Lookup(fea.name, FIELD_CATEGORICAL, table, table.embedding_dim, pool=pool,
                                          seq_len=x[fea.name].shape[1], mask_id=mask_id, eps=1e-16))
            else:
                raise ValueError("unknown feature class %s" % type(fea).__name__)
        return sparse, dense_l

    def forward(self, x, features, squeeze_dim=False):
        key = (tuple(id(f) for f in features), squeeze_dim,
               tuple(x[f.name].shape[1] if isinstance(f, SequenceFeature) else 0 for f in features))
        cached = self._plans.get(key)
        if cached is None:
            sparse, dense_l = self._lookups(x, features)
            if squeeze_dim:
                if not sparse and not dense_l:
                    raise ValueError("The input features can note be empty")
                lookups = sparse + dense_l                  # cat(sparse.flatten(1), dense_values)
            else:
                if not sparse:
                    raise ValueError(
                        "If keep the original shape:[batch_size, num_features, embed_dim], expected %s in feature "
                        "list, got %s" % ("SparseFeatures", features))
                lookups = sparse                            # dense values are dropped, as in the reference
            cached = (host.Plan(lookups), len(sparse))
            self._plans[key] = cached
        plan, n_sparse = cached
        # the flattened [B, sum(dims) + n_dense] layout rarely has 16-byte aligned rows (26 * 64 + 13 = 1677 floats):
        # it is produced with a padded row stride and consumed in place by the tower's first GEMM
        out = plan.run([x[lk.name] for lk in plan.lookups], pad_rows=squeeze_dim)
        if squeeze_dim:
            return out
        B = out.shape[0]
        specs = plan.specs
        if plan.uniform_dim is None:
            raise RuntimeError("Sizes of tensors must match except in dimension 1 (embed_dim differs across features)")
        concat = [s.pool == POOL_CONCAT for s in specs]
        if any(concat):
            if not all(concat) or len(set(s.seq_len for s in specs)) != 1:
                raise RuntimeError("Tensors must have same number of dimensions (mixing pooling='concat' "
                                   "with pooled/sparse features)")
            return out.view(B, n_sparse, specs[0].seq_len, plan.uniform_dim)
        return out.view(B, n_sparse, plan.uniform_dim)


class InputMask(nn.Module):
    """[B, n_features(, L)] float mask: id != padding_idx (or != -1 when no padding_idx)."""

    def forward(self, x, features):
        mask = []
        if not isinstance(features, list):
            features = [features]
        for fea in features:
            if isinstance(fea, SparseFeature) or isinstance(fea, SequenceFeature):
                pad = fea.padding_idx if fea.padding_idx != None else -1  # noqa: E711
                mask.append((x[fea.name].long() != pad).unsqueeze(1).float())
            else:
                raise ValueError("Only SparseFeature or SequenceFeature support to get mask.")
        return torch.cat(mask, dim=1)


class LR(nn.Module):
    def __init__(self, input_dim, sigmoid=False):
        super().__init__()
        self.sigmoid = sigmoid
        self.fc = nn.Linear(input_dim, 1, bias=True)

    def forward(self, x):
        y = ops.linear(x, self.fc.weight, self.fc.bias)
        return torch.sigmoid(y) if self.sigmoid else y


class ConcatPooling(nn.Module):
    def forward(self, x, mask=None):
        return x


class AveragePooling(nn.Module):
    """bmm(mask, x) / (mask.sum + 1e-16); plain mean over L when mask is None."""

    def forward(self, x, mask=None):
        if mask == None:  # noqa: E711
            return ops.pool(x, None, False, ops.DENOM_LEN, 0.0)
        return ops.pool(x, mask, True, ops.DENOM_MASK, 1e-16)


class SumPooling(nn.Module):
    def forward(self, x, mask=None):
        if mask == None:  # noqa: E711
            return ops.pool(x, None, False, ops.DENOM_NONE, 0.0)
        return ops.pool(x, mask, True, ops.DENOM_NONE, 0.0)


class Dice(nn.Module):
    """Dice activation (rechub/basic/activation.py:5-26), kept as written there."""

    def __init__(self, epsilon=1e-3):
        super(Dice, self).__init__()
        self.epsilon = epsilon
        self.alpha = nn.Parameter(torch.randn(1))

    def forward(self, x):
        avg = x.mean(dim=1).unsqueeze(dim=1)
        var = (torch.pow(x - avg, 2) + self.epsilon).sum(dim=1).unsqueeze(dim=1)
        ps = torch.sigmoid((x - avg) / torch.sqrt(var))
        return ps * x + (1 - ps) * self.alpha * x


def activation_layer(act_name):
    if isinstance(act_name, str):
        low = act_name.lower()
        if low == 'sigmoid':
            return nn.Sigmoid()
        if low == 'relu':
            return nn.ReLU(inplace=True)
        if low == 'dice':
            return Dice()
        if low == 'prelu':
            return nn.PReLU()
        if low == "softmax":
            return nn.Softmax(dim=1)
        raise NotImplementedError
    if issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError


class MLP(nn.Module):
    """Linear -> BatchNorm1d -> activation -> Dropout per layer (always BatchNorm), optional Linear(*,1)."""

    def __init__(self, input_dim, output_layer=True, dims=None, dropout=0, activation="relu"):
        super().__init__()
        if dims is None:
            dims = []
        layers = list()
        for i_dim in dims:
            layers.append(nn.Linear(input_dim, i_dim))
            layers.append(nn.BatchNorm1d(i_dim))
            layers.append(activation_layer(activation))
            layers.append(nn.Dropout(p=dropout))
            input_dim = i_dim
        if output_layer:
            layers.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        return dense.run_sequential(self.mlp, x)


class FM(nn.Module):
    """0.5 * [(sum_f x)^2 - sum_f x^2], summed over embed_dim when reduce_sum."""

    def __init__(self, reduce_sum=True):
        super().__init__()
        self.reduce_sum = reduce_sum

    def forward(self, x):
        return ops.interaction(x, "product_sum" if self.reduce_sum else "bi_interaction")
