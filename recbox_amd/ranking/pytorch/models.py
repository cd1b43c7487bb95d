"""Model bodies in the FuxiCTR convention the reference's ``RankingModel`` harness
expects from its subclasses (``forward(inputs) -> {"y_pred": sigmoid(logit)}``,
/root/reference/recbox/ranking/pytorch/models/ranking_model.py:66-70,191-197).
The reference tree ships the harness but no concrete FM/DeepFM class;
these are the bodies a user of the harness writes, built from the drop-in layers.
"""
import torch
from torch import nn

from ... import ops
from .layers import FactorizationMachine, FeatureEmbedding

__all__ = ["FM"]


class FM(nn.Module):
    """y = sigmoid(LR(X) + 0.5 * sum_d[(sum_f e)^2 - sum_f e^2])."""

    def __init__(self, feature_map, embedding_dim=10, fused=True, **kwargs):
        super(FM, self).__init__()
        self.feature_map = feature_map
        self.fused = fused          # False: compose the drop-in layers exactly as the reference does
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.fm = FactorizationMachine(feature_map)

    def logits(self, X):
        emb = self.embedding_layer.embedding_layer
        lr = self.fm.lr_layer.embedding_layer.embedding_layer
        names, values, plan, posts = emb.plan_for(X)
        lnames, _, lplan, lposts = lr.plan_for(X)
        if self.fused and emb.fusable(plan, posts) and lr.fusable(lplan, lposts) and names == lnames:
            # gather + LR + interaction in ONE kernel; [B, F, D] is never written (rbx_fm_fwd / rbx_fm_bwd)
            return ops.fm_fused(plan.plan, lplan.plan, values, [m.weight for m in plan.modules],
                                [m.weight for m in lplan.modules], self.fm.lr_layer.bias)
        return self.fm(X, self.embedding_layer(X))

    def forward(self, X):
        return {"y_pred": torch.sigmoid(self.logits(X))}
