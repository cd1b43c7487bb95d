"""Sequence pooling modules of the matching side (drop-in for
``recbox.core.pytorch.layers.{MaskedAveragePooling, MaskedSumPooling}``,
/root/reference/recbox/core/pytorch/layers/sequence.py:4-20).

Standalone calls run ``rbx_pool_fwd/bwd`` on the materialised ``[B, L, D]``
tensor.  When one of these modules is registered as an ``embedding_callback``,
``EmbeddingDictLayer`` does not call it at all: it recognises the type and asks
the gather kernel to pool in registers (``RBX_POOL_MEAN_VALUE`` / ``RBX_POOL_SUM``),
so ``[B, L, D]`` never reaches HBM.
"""
from torch import nn

from .... import ops
from ...._lib import POOL_MEAN_VALUE, POOL_SUM

__all__ = ["MaskedAveragePooling", "MaskedSumPooling"]


class MaskedAveragePooling(nn.Module):
    """sum_L(E) / (#rows whose sum_d != 0 + 1e-12) -- the mask is VALUE based."""
    fused_pool = POOL_MEAN_VALUE
    fused_eps = 1.e-12

    def forward(self, embedding_matrix):
        return ops.pool(embedding_matrix, None, False, ops.DENOM_VALUE, self.fused_eps)


class MaskedSumPooling(nn.Module):
    fused_pool = POOL_SUM
    fused_eps = 0.0

    def forward(self, embedding_matrix):
        return ops.pool(embedding_matrix, None, False, ops.DENOM_NONE, 0.0)
