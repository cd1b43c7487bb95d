"""Matching-side embedding layer (drop-in for ``recbox.core.pytorch.layers.
{EmbeddingLayer, EmbeddingDictLayer}``, /root/reference/recbox/core/pytorch/layers/
embedding.py:10-138).

Same constructor / forward signatures, same parameter holders (real
``nn.Embedding`` / ``nn.Linear(1, D, bias=False)`` in ``embedding_layers``, so
``MatchingModel.init_weights`` type tests, name-keyed regularisers and
``state_dict`` checkpoints keep working), but the compute is one
``rbx_embed_fwd`` launch over all selected features with pooling callbacks fused,
and the backward is the sorted segmented scatter-add of ``rbx_embed_bwd``.
"""
from collections import OrderedDict

import torch
from torch import nn

from .... import _embed_host as host
from ...._lib import FIELD_CATEGORICAL, FIELD_NUMERIC, POOL_CONCAT, POOL_NONE
from . import sequence as layers  # noqa: F401  (feature maps say "layers.MaskedAveragePooling()")
from .sequence import *  # noqa: F401,F403

__all__ = ["EmbeddingLayer", "EmbeddingDictLayer"]


_FusedDict = host.FusedDict      # feature -> views; drops its fused block as soon as the caller edits the dict


class EmbeddingLayer(nn.Module):
    def __init__(self, feature_map, embedding_dim, disable_sharing_pretrain=False,
                 required_feature_columns=[], not_required_feature_columns=[]):
        super(EmbeddingLayer, self).__init__()
        self.embedding_layer = EmbeddingDictLayer(feature_map, embedding_dim,
                                                  disable_sharing_pretrain=disable_sharing_pretrain,
                                                  required_feature_columns=required_feature_columns,
                                                  not_required_feature_columns=not_required_feature_columns)

    def forward(self, X, feature_source=None):
        feature_emb_dict = self.embedding_layer(X, feature_source=feature_source)
        return self.embedding_layer.dict2tensor(feature_emb_dict)


class EmbeddingDictLayer(nn.Module):
    def __init__(self, feature_map, embedding_dim, disable_sharing_pretrain=False,
                 required_feature_columns=None, not_required_feature_columns=None):
        super(EmbeddingDictLayer, self).__init__()
        self._feature_map = feature_map
        self.required_feature_columns = required_feature_columns
        self.not_required_feature_columns = not_required_feature_columns
        self.embedding_layers = nn.ModuleDict()
        self.embedding_callbacks = nn.ModuleDict()
        self._plans = {}
        plain = not disable_sharing_pretrain          # the LR flavour (width 1) ignores sharing / callbacks / pretrained tables
        if not plain and embedding_dim != 1:
            raise AssertionError("disable_sharing_pretrain is the width-1 (LR) flavour")
        for feature, spec in self._feature_map.feature_specs.items():
            if not self.is_required(feature):
                continue
            if plain and "embedding_callback" in spec:
                self.embedding_callbacks[feature] = eval(spec["embedding_callback"])
            if plain and "share_embedding" in spec:
                # the SAME module object is registered twice; a target that is not registered yet is a KeyError, as in
                # the reference
                self.embedding_layers[feature] = self.embedding_layers[spec["share_embedding"]]
                continue
            holder = self._new_holder(feature, spec, spec.get("embedding_dim", embedding_dim) if plain else 1, plain)
            if holder is not None:
                self.embedding_layers[feature] = holder

    def _new_holder(self, feature, spec, width, plain):
        """The parameter holder of one feature: Linear(1, D) for numeric values, Embedding(V, D) for ids."""
        kind = spec["type"]
        if kind == "numeric":
            return nn.Linear(1, width, bias=False)
        if kind not in ("categorical", "sequence"):
            return None
        pad = spec.get("padding_idx", None)
        table = nn.Embedding(spec["vocab_size"], width, padding_idx=pad)
        if plain and "pretrained_emb" in spec:
            table = self.load_pretrained_embedding(table, self._feature_map, feature, freeze=spec["freeze_emb"],
                                                   padding_idx=pad)
        return table

    def is_required(self, feature):
        wanted, unwanted = self.required_feature_columns, self.not_required_feature_columns
        return not ((wanted and feature not in wanted) or (unwanted and feature in unwanted))

    def get_pretrained_embedding(self, pretrained_path, feature_name):
        import h5py  # only needed for pretrained tables (not installed in every image: imported on use)
        with h5py.File(pretrained_path, 'r') as store:
            return store[feature_name][:]

    def load_pretrained_embedding(self, embedding_matrix, feature_map, feature_name, freeze=False, padding_idx=None):
        import os
        rel = feature_map.feature_specs[feature_name]["pretrained_emb"]
        rows = torch.as_tensor(self.get_pretrained_embedding(os.path.join(feature_map.data_dir, rel), feature_name)).float()
        if padding_idx is not None:
            rows[padding_idx].zero_()
        embedding_matrix.weight = torch.nn.Parameter(rows, requires_grad=not freeze)
        return embedding_matrix

    def dict2tensor(self, embedding_dict):
        if len(embedding_dict) == 1:
            return list(embedding_dict.values())[0]
        fused = getattr(embedding_dict, "fused", None)
        if fused is not None and tuple(embedding_dict.keys()) == tuple(embedding_dict.names) \
                and embedding_dict.plan.uniform_dim is not None \
                and all(s.pool != POOL_CONCAT for s in embedding_dict.plan.specs):
            return fused.view(fused.shape[0], len(embedding_dict), embedding_dict.plan.uniform_dim)
        return torch.stack(list(embedding_dict.values()), dim=1)

    # ---- planning ----
    def _lookup_for(self, feature, spec, value):
        module = self.embedding_layers[feature]
        if spec["type"] == "numeric":
            return host.Lookup(feature, FIELD_NUMERIC, module, module.out_features), None
        if spec["type"] not in ("categorical", "sequence"):
            raise NotImplementedError
        dim = module.embedding_dim
        callback = self.embedding_callbacks[feature] if feature in self.embedding_callbacks else None
        if value.dim() == 1:
            # a callback on a [B, D] lookup is applied by torch below (the reference would do the same)
            return host.Lookup(feature, FIELD_CATEGORICAL, module, dim), callback
        seq_len = value.shape[1]
        pool = getattr(callback, "fused_pool", None)
        if pool is not None:
            return host.Lookup(feature, FIELD_CATEGORICAL, module, dim, pool=pool, seq_len=seq_len,
                               eps=callback.fused_eps), None
        return host.Lookup(feature, FIELD_CATEGORICAL, module, dim, pool=POOL_CONCAT, seq_len=seq_len), callback

    def forward(self, inputs, feature_source=None, feature_type=None):
        names, values = [], []
        for feature, spec in self._feature_map.feature_specs.items():
            if feature_source and spec["source"] != feature_source:
                continue
            if feature_type and spec["type"] != feature_type:
                continue
            if feature in self.embedding_layers:
                if spec["type"] not in ("numeric", "categorical", "sequence"):
                    raise NotImplementedError
                names.append(feature)
                values.append(inputs[feature])
        out = _FusedDict()
        if not names:
            return out
        key = (tuple(names), tuple(v.dim() for v in values), tuple(v.shape[-1] if v.dim() > 1 else 1 for v in values))
        cached = self._plans.get(key)
        if cached is None:
            lookups, posts = [], []
            for feature, value in zip(names, values):
                lk, post = self._lookup_for(feature, self._feature_map.feature_specs[feature], value)
                lookups.append(lk)
                posts.append(post)
            cached = (host.Plan(lookups), posts)
            self._plans[key] = cached
        plan, posts = cached
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
