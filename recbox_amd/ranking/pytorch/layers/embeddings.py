"""Ranking-side embedding layer (drop-in for ``recbox.ranking.pytorch.layers.
{FeatureEmbedding, FeatureEmbeddingDict}``, /root/reference/recbox/ranking/pytorch/
layers/embeddings/feature_embedding.py:28-214).

Kept contract: constructor / forward signatures; parameter holders
are real ``nn.Embedding`` / ``nn.Linear(1, D, bias=False)`` modules registered as
``embedding_layer.embedding_layers.<feature>`` (checkpoint keys, ``reset_parameters``
type test, ``save/load_init_embs`` regex, ``share_embedding`` aliasing); init is
``normal_(std=1e-4)`` on rows ``1:`` when ``padding_idx`` is set; LR mode
(``embedding_dim == 1`` without pretrain+sharing) forces dim 1 and sum pooling.
Compute: one ``rbx_embed_fwd`` launch for all selected features -- ids are read in
whatever dtype they arrive (float64 columns of the hstacked batch included), the
``[B, F, D]`` stack is produced directly, pooling encoders are fused.
"""
from collections import OrderedDict
from functools import partial  # noqa: F401  (initializer strings use it)

import torch
from torch import nn

from .... import _embed_host as host
from ...._lib import FIELD_CATEGORICAL, FIELD_NUMERIC, POOL_CONCAT, POOL_NONE
from . import pooling as layers  # noqa: F401  ("layers.MaskedAveragePooling()" in feature maps)
from .pooling import MaskedAveragePooling, MaskedSumPooling  # noqa: F401

__all__ = ["FeatureEmbedding", "FeatureEmbeddingDict"]


def _as_list(x):
    return x if isinstance(x, list) else [x]


def get_initializer(initializer):
    """String -> callable, as fuxictr.pytorch.torch_utils.get_initializer (torch_utils.py:111-119)."""
    if isinstance(initializer, str):
        try:
            initializer = eval(initializer)
        except Exception:
            raise ValueError("initializer={} is not supported.".format(initializer))
    return initializer


_FusedDict = host.FusedDict      # feature -> views; drops its fused block as soon as the caller edits the dict


class FeatureEmbedding(nn.Module):
    def __init__(self, feature_map, embedding_dim, embedding_initializer="partial(nn.init.normal_, std=1e-4)",
                 required_feature_columns=None, not_required_feature_columns=None, use_pretrain=True,
                 use_sharing=True):
        super(FeatureEmbedding, self).__init__()
        self.embedding_layer = FeatureEmbeddingDict(feature_map, embedding_dim,
                                                    embedding_initializer=embedding_initializer,
                                                    required_feature_columns=required_feature_columns,
                                                    not_required_feature_columns=not_required_feature_columns,
                                                    use_pretrain=use_pretrain, use_sharing=use_sharing)

    def forward(self, X, feature_source=[], feature_type=[], dynamic_emb_dim=False):
        feature_emb_dict = self.embedding_layer(X, feature_source=feature_source, feature_type=feature_type)
        return self.embedding_layer.dict2tensor(feature_emb_dict, dynamic_emb_dim=dynamic_emb_dim)


class FeatureEmbeddingDict(nn.Module):
    def __init__(self, feature_map, embedding_dim, embedding_initializer="partial(nn.init.normal_, std=1e-4)",
                 required_feature_columns=None, not_required_feature_columns=None, use_pretrain=True,
                 use_sharing=True):
        super(FeatureEmbeddingDict, self).__init__()
        self._feature_map = feature_map
        self.required_feature_columns = required_feature_columns
        self.not_required_feature_columns = not_required_feature_columns
        self.use_pretrain = use_pretrain
        self.embedding_initializer = embedding_initializer
        self.embedding_layers = nn.ModuleDict()
        self.feature_encoders = nn.ModuleDict()
        self._plans = {}
        # "LR mode": a width-1 layer that neither loads pretrained tables nor shares them is the first-order term of
        # LogisticRegression -- every table is [V, 1] and sequences are summed
        first_order = embedding_dim == 1 and not (use_pretrain and use_sharing)
        for feature, spec in self._feature_map.features.items():
            if not self.is_required(feature):
                continue
            encoder = self._encoder_for(spec, first_order)
            if encoder is not None:
                self.feature_encoders[feature] = encoder
            target = spec.get("share_embedding") if use_sharing else None
            if target in self.embedding_layers:          # an alias; a target that is not registered (yet) gets its own table
                self.embedding_layers[feature] = self.embedding_layers[target]
                continue
            holder = self._new_holder(feature, spec, 1 if first_order else spec.get("embedding_dim", embedding_dim))
            if holder is not None:
                self.embedding_layers[feature] = holder
        self.reset_parameters()

    def _encoder_for(self, spec, first_order):
        if first_order:
            return layers.MaskedSumPooling() if spec["type"] == "sequence" else None
        text = spec.get("feature_encoder", None)
        return self.get_feature_encoder(text) if text else None

    def _new_holder(self, feature, spec, width):
        kind = spec["type"]
        if kind == "numeric":
            return nn.Linear(1, width, bias=False)
        if kind not in ("categorical", "sequence"):
            return None
        pad = spec.get("padding_idx", None)
        table = nn.Embedding(spec["vocab_size"], width, padding_idx=pad)
        if self.use_pretrain and "pretrained_emb" in spec:
            table = self.load_pretrained_embedding(table, self._feature_map, feature, freeze=spec["freeze_emb"],
                                                   padding_idx=pad)
        return table

    def get_feature_encoder(self, encoder):
        texts = encoder if isinstance(encoder, list) else None
        try:
            return nn.Sequential(*[eval(t) for t in texts]) if texts is not None else eval(encoder)
        except Exception:
            raise ValueError("feature_encoder={} is not supported.".format(encoder))

    def reset_parameters(self):
        """Initialiser on every trainable table of this layer (rows 1: when the table has a padding row, which stays
        zero); pretrained tables and frozen shared tables are left alone."""
        self.embedding_initializer = get_initializer(self.embedding_initializer)
        for name, module in self.embedding_layers.items():
            spec = self._feature_map.features[name]
            loaded = self.use_pretrain and "pretrained_emb" in spec
            frozen_alias = "share_embedding" in spec and module.weight.requires_grad is False
            if loaded or frozen_alias or type(module) != nn.Embedding:
                continue
            self.embedding_initializer(module.weight if module.padding_idx is None else module.weight[1:, :])

    def is_required(self, feature):
        if self._feature_map.features[feature]["type"] == "meta":
            return False
        wanted, unwanted = self.required_feature_columns, self.not_required_feature_columns
        return not ((wanted and feature not in wanted) or (unwanted and feature in unwanted))

    def get_pretrained_embedding(self, pretrained_path, feature_name):
        import h5py  # only needed for pretrained tables (imported on use)
        with h5py.File(pretrained_path, 'r') as store:
            return store[feature_name][:]

    def load_pretrained_embedding(self, embedding_matrix, feature_map, feature_name, freeze=False, padding_idx=None):
        import os
        rel = feature_map.features[feature_name]["pretrained_emb"]
        rows = torch.as_tensor(self.get_pretrained_embedding(os.path.join(feature_map.data_dir, rel), feature_name)).float()
        if rows.shape[-1] != embedding_matrix.embedding_dim:
            raise AssertionError("{}'s embedding_dim is not correctly set to match its pretrained_emb shape"
                                 .format(feature_name))
        if padding_idx is not None:
            rows[padding_idx].zero_()
        embedding_matrix.weight = torch.nn.Parameter(rows, requires_grad=not freeze)
        return embedding_matrix

    def dict2tensor(self, embedding_dict, feature_source=[], feature_type=[], dynamic_emb_dim=False):
        sources, types = _as_list(feature_source), _as_list(feature_type)
        picked = []
        for feature, spec in self._feature_map.features.items():
            if sources and spec["source"] not in sources:
                continue
            if types and spec["type"] not in types:
                continue
            if feature in embedding_dict:
                picked.append(feature)
        fused = getattr(embedding_dict, "fused", None)
        if fused is not None and tuple(picked) == tuple(embedding_dict.names) \
                and all(s.pool != POOL_CONCAT for s in embedding_dict.plan.specs):
            if dynamic_emb_dim:
                return fused                                   # == torch.cat(dim=-1)
            if embedding_dict.plan.uniform_dim is not None:
                return fused.view(fused.shape[0], len(picked), embedding_dict.plan.uniform_dim)
        values = [embedding_dict[f] for f in picked]
        return torch.cat(values, dim=-1) if dynamic_emb_dim else torch.stack(values, dim=1)

    def _lookup_for(self, feature, spec, value):
        module = self.embedding_layers[feature]
        if spec["type"] == "numeric":
            return host.Lookup(feature, FIELD_NUMERIC, module, module.out_features), None
        dim = module.embedding_dim
        encoder = self.feature_encoders[feature] if feature in self.feature_encoders else None
        if value.dim() == 1:
            return host.Lookup(feature, FIELD_CATEGORICAL, module, dim), encoder
        seq_len = value.shape[1]
        pool = getattr(encoder, "fused_pool", None)
        if pool is not None:
            return host.Lookup(feature, FIELD_CATEGORICAL, module, dim, pool=pool, seq_len=seq_len,
                               eps=encoder.fused_eps), None
        return host.Lookup(feature, FIELD_CATEGORICAL, module, dim, pool=POOL_CONCAT, seq_len=seq_len), encoder

    def plan_for(self, inputs, feature_source=[], feature_type=[]):
        """(names, values, plan, posts) of one call signature; plans are cached."""
        sources, types = _as_list(feature_source), _as_list(feature_type)
        names, values = [], []
        for feature, spec in self._feature_map.features.items():
            if sources and spec["source"] not in sources:
                continue
            if types and spec["type"] not in types:
                continue
            if feature in self.embedding_layers:
                if spec["type"] not in ("numeric", "categorical", "sequence"):
                    raise NotImplementedError
                names.append(feature)
                values.append(inputs[feature])
        if not names:
            return names, values, None, None
        key = (tuple(names), tuple(v.shape[1] if v.dim() > 1 else 0 for v in values))
        cached = self._plans.get(key)
        if cached is None:
            lookups, posts = [], []
            for feature, value in zip(names, values):
                lk, post = self._lookup_for(feature, self._feature_map.features[feature], value)
                lookups.append(lk)
                posts.append(post)
            cached = (host.Plan(lookups), posts)
            self._plans[key] = cached
        return names, values, cached[0], cached[1]

    def fusable(self, plan, posts):
        """True when the FM / LR body can be fused: one id per sample, no encoders, one dim."""
        return (plan is not None and plan.uniform_dim is not None and all(p is None for p in posts)
                and all(s.seq_len == 1 and s.pool == POOL_NONE for s in plan.specs))

    def forward(self, inputs, feature_source=[], feature_type=[]):
        names, values, plan, posts = self.plan_for(inputs, feature_source, feature_type)
        out = _FusedDict()
        if not names:
            return out
        fused = plan.run(values)
        clean = True
        for i, feature in enumerate(names):
            emb = plan.slot(fused, i)
            if posts[i] is not None:
                emb = posts[i](emb)
                clean = False
            out[feature] = emb
        if clean:
            out.seal(fused, plan, names)
        return out
