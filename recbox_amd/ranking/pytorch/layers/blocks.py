"""LR / FM / MLP blocks (drop-in for ``recbox.ranking.pytorch.layers.{LogisticRegression,
FactorizationMachine, MLP_Block}``, /root/reference/recbox/ranking/pytorch/layers/blocks/
logistic_regression.py:23-35, factorization_machine.py:24-34, mlp_block.py:23-61)."""
import torch
from torch import nn

from .... import dense, ops
from .embeddings import FeatureEmbedding
from .interactions import InnerProductInteraction

__all__ = ["LogisticRegression", "FactorizationMachine", "MLP_Block"]


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


class MLP_Block(nn.Module):
    """Same constructor / ``self.mlp`` child order as the reference (real nn.Linear holders);
    every Linear runs on the fp32 matrix cores (rbx_linear_fwd/bwd, ReLU fused)."""

    def __init__(self, input_dim, hidden_units=[], hidden_activations="ReLU", output_dim=None,
                 output_activation=None, dropout_rates=0.0, batch_norm=False, norm_before_activation=True,
                 use_bias=True):
        super(MLP_Block, self).__init__()
        self.mlp = nn.Sequential(*dense.tower_modules(input_dim, hidden_units, hidden_activations, dropout_rates,
                                                      batch_norm, use_bias, out_dim=output_dim,
                                                      out_activation=output_activation,
                                                      norm_after_activation=not norm_before_activation,
                                                      width_aware=True))

    def forward(self, inputs):
        return dense.run_sequential(self.mlp, inputs)
