"""LR / FM blocks (drop-in for ``recbox.ranking.pytorch.layers.{LogisticRegression,
FactorizationMachine}``, /root/reference/recbox/ranking/pytorch/layers/blocks/
logistic_regression.py:23-35 and factorization_machine.py:24-34)."""
import torch
from torch import nn

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
        embed_weights = self.embedding_layer(X)        # [B, F, 1] from one gather launch
        output = embed_weights.sum(dim=1)
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
