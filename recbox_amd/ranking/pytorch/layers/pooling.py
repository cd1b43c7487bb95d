"""Ranking-side pooling modules (drop-in for ``recbox.ranking.pytorch.layers.
{MaskedAveragePooling, MaskedSumPooling}``, /root/reference/recbox/ranking/pytorch/
layers/pooling.py:22-40).  Registered as ``feature_encoder`` they are fused into the
gather kernel by ``FeatureEmbeddingDict``; called on their own they run
``rbx_pool_fwd/bwd`` on the materialised tensor."""
from torch import nn

from .... import ops
from ...._lib import POOL_MEAN_VALUE, POOL_SUM

__all__ = ["MaskedAveragePooling", "MaskedSumPooling", "KMaxPooling"]


class MaskedAveragePooling(nn.Module):
    """sum_L(E) / (count + 1e-12); count = #rows with sum_d != 0, or an explicit mask's sum."""
    fused_pool = POOL_MEAN_VALUE
    fused_eps = 1e-12

    def forward(self, embedding_matrix, mask=None):
        if mask is None:
            return ops.pool(embedding_matrix, None, False, ops.DENOM_VALUE, 1e-12)
        # the numerator stays the UNMASKED sum in the reference (pooling.py:27-31)
        return ops.pool(embedding_matrix, mask, False, ops.DENOM_MASK, 1e-12)


class MaskedSumPooling(nn.Module):
    fused_pool = POOL_SUM
    fused_eps = 0.0

    def forward(self, embedding_matrix):
        return ops.pool(embedding_matrix, None, False, ops.DENOM_NONE, 0.0)


class KMaxPooling(nn.Module):
    """The k largest entries along ``dim`` in their original order (pooling.py:43-53).  Selection with a gradient
    through the selected positions: ATen's topk / sort / gather (the rbx_topk kernel of the retrieval metrics returns
    values for ranking, it carries no backward)."""

    def __init__(self, k, dim):
        super(KMaxPooling, self).__init__()
        self.k, self.dim = k, dim

    def forward(self, X):
        keep = X.topk(self.k, dim=self.dim).indices.sort(dim=self.dim).values
        return X.gather(self.dim, keep)
