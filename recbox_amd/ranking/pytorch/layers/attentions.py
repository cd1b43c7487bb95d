"""First-party attention core (drop-in for ``recbox.ranking.pytorch.layers.ScaledDotProductAttention``,
/root/reference/recbox/ranking/pytorch/layers/attentions/dot_product_attention.py:23-43): same
call signature ``forward(Q, K, V, scale=None, mask=None) -> (output, attention)``; the QK^T ->
mask(-1e9) -> softmax -> .V chain is one fused HIP kernel (``rbx_attn_fwd``) and the score
matrix only reaches HBM because this API returns the attention probabilities."""
from torch import nn

from .... import ops

__all__ = ["ScaledDotProductAttention"]


class ScaledDotProductAttention(nn.Module):
    def __init__(self, dropout_rate=0.):
        super(ScaledDotProductAttention, self).__init__()
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def forward(self, Q, K, V, scale=None, mask=None):
        if self.dropout is not None and self.training:
            raise NotImplementedError("attention dropout inside the fused kernel is not implemented; "
                                      "use dropout_rate=0 (BASELINE.json configs run with dropout 0)")
        s = 1.0 / scale if scale else 1.0           # the reference DIVIDES the scores by `scale`
        if mask is not None:
            mask = mask.view(*Q.shape[:-1], K.shape[-2])
        output, attention = ops.attention(Q, K, V, mask=mask, scale=s, causal=False, fill=-1.0e9, need_probs=True)
        return output, attention
