"""First-party attention core (drop-in for ``recbox.ranking.pytorch.layers.ScaledDotProductAttention``,
/root/reference/recbox/ranking/pytorch/layers/attentions/dot_product_attention.py:23-43): same
call signature ``forward(Q, K, V, scale=None, mask=None) -> (output, attention)``; the QK^T ->
mask(-1e9) -> softmax -> .V chain is one fused HIP kernel (``rbx_attn_fwd``) and the score
matrix only reaches HBM because this API returns the attention probabilities."""
from torch import nn

from .... import ops

__all__ = ["ScaledDotProductAttention", "MultiHeadTargetAttention"]


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


class MultiHeadTargetAttention(nn.Module):
    """Target attention over a behaviour sequence (drop-in for ``MultiHeadTargetAttention``,
    /root/reference/recbox/ranking/pytorch/layers/attentions/target_attention.py:69-121): same
    constructor and parameter names (``W_q/W_k/W_v/W_o``, bias-free ``nn.Linear`` holders); the four
    projections run on the fp32 matrix cores (``rbx_linear_*``) and the per-head attention of the single
    target query over the sequence is the fused kernel behind ``ScaledDotProductAttention``."""

    def __init__(self, input_dim=64, attention_dim=64, num_heads=1, dropout_rate=0, use_scale=True, use_qkvo=True):
        super(MultiHeadTargetAttention, self).__init__()
        if not use_qkvo:
            attention_dim = input_dim
        assert attention_dim % num_heads == 0, \
            "attention_dim={} is not divisible by num_heads={}".format(attention_dim, num_heads)
        self.num_heads = num_heads
        self.head_dim = attention_dim // num_heads
        self.scale = self.head_dim ** 0.5 if use_scale else None
        self.use_qkvo = use_qkvo
        if use_qkvo:
            self.W_q = nn.Linear(input_dim, attention_dim, bias=False)
            self.W_k = nn.Linear(input_dim, attention_dim, bias=False)
            self.W_v = nn.Linear(input_dim, attention_dim, bias=False)
            self.W_o = nn.Linear(attention_dim, input_dim, bias=False)
        self.dot_attention = ScaledDotProductAttention(dropout_rate)

    def forward(self, target_item, history_sequence, mask=None):
        """target_item [B, E]; history_sequence [B, L, E]; mask [B, L] with 0 at masked positions."""
        if self.use_qkvo:
            query = ops.linear(target_item, self.W_q.weight)
            key = ops.linear(history_sequence, self.W_k.weight)
            value = ops.linear(history_sequence, self.W_v.weight)
        else:
            query, key, value = target_item, history_sequence, history_sequence
        B = query.size(0)
        query = query.view(B, 1, self.num_heads, self.head_dim).transpose(1, 2)
        key = key.view(B, -1, self.num_heads, self.head_dim).transpose(1, 2)
        value = value.view(B, -1, self.num_heads, self.head_dim).transpose(1, 2)
        if mask is not None:
            mask = mask.view(B, 1, 1, -1).expand(-1, self.num_heads, -1, -1)
        output, _ = self.dot_attention(query, key, value, scale=self.scale, mask=mask)
        output = output.transpose(1, 2).contiguous().view(-1, self.num_heads * self.head_dim)
        if self.use_qkvo:
            output = ops.linear(output, self.W_o.weight)
        return output
