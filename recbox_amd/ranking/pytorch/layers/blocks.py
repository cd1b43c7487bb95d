"""LR / FM blocks (drop-in for ``recbox.ranking.pytorch.layers.{LogisticRegression,
FactorizationMachine}``, /root/reference/recbox/ranking/pytorch/layers/blocks/
logistic_regression.py:23-35 and factorization_machine.py:24-34)."""
import torch
from torch import nn

from .... import ops
from .embeddings import FeatureEmbedding
from .interactions import InnerProductInteraction

__all__ = ["LogisticRegression", "FactorizationMachine"]


class LogisticRegression(nn.Module):
    def __init__(self, feature_map, use_bias=True):
        super(LogisticRegression, self).__init__()
        self.bias = nn.Parameter(torch.zeros(1), requires_grad=True) if use_bias else None
        # dim-1 tables: "a trick for quick one-hot encoding in LR"
        self.embedding_layer = FeatureEmbedding(feature_map, 1, use_pretrain=False, use_sharing=False)

    def forward(self, X):
        table = self.embedding_layer.embedding_layer
        names, values, plan, posts = table.plan_for(X)
        if table.fusable(plan, posts):
            # one kernel: sum_f w_f[id] (+ x * w for numeric features) + bias, no [B, F, 1] tensor
            return ops.fm_fused(None, plan.plan, values, [], [m.weight for m in plan.modules], self.bias)
        embed_weights = self.embedding_layer(X)        # sequence features: gather+sum-pool, then reduce over F
        output = ops.interaction_rowsum(embed_weights)
        if self.bias is not None:
            output = output + self.bias
        return output


class FactorizationMachine(nn.Module):
    def __init__(self, feature_map):
        super(FactorizationMachine, self).__init__()
        self.fm_layer = InnerProductInteraction(feature_map.num_fields, output="product_sum")
        self.lr_layer = LogisticRegression(feature_map, use_bias=True)

    def forward(self, X, feature_emb):
        lr_out = self.lr_layer(X)
        fm_out = self.fm_layer(feature_emb)
        return fm_out + lr_out
