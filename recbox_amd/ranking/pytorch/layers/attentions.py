"""First-party attention core (drop-in for ``recbox.ranking.pytorch.layers.ScaledDotProductAttention``,
/root/reference/recbox/ranking/pytorch/layers/attentions/dot_product_attention.py:23-43): same
call signature ``forward(Q, K, V, scale=None, mask=None) -> (output, attention)``; the QK^T ->
mask(-1e9) -> softmax -> .V chain is one fused HIP kernel (``rbx_attn_fwd``) and the score
matrix only reaches HBM because this API returns the attention probabilities."""
import math

import torch
from torch import nn

from .... import dense, ops

__all__ = ["ScaledDotProductAttention", "MultiHeadTargetAttention", "SqueezeExcitation", "DIN_Attention", "Dice", "GELU"]


class ScaledDotProductAttention(nn.Module):
    def __init__(self, dropout_rate=0.):
        super(ScaledDotProductAttention, self).__init__()
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def forward(self, Q, K, V, scale=None, mask=None):
        # `attention = self.dropout(attention)` (dot_product_attention.py:40-41): applied to the probabilities inside
        # the fused kernel; the returned attention is the dropped one, as in the reference
        p_drop = self.dropout.p if (self.dropout is not None and self.training) else 0.0
        s = 1.0 / scale if scale else 1.0           # the reference DIVIDES the scores by `scale`
        if mask is not None:
            mask = mask.view(*Q.shape[:-1], K.shape[-2])
        output, attention = ops.attention(Q, K, V, mask=mask, scale=s, causal=False, fill=-1.0e9, need_probs=True,
                                          dropout_p=p_drop)
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


class Dice(nn.Module):
    """DIN's data-adaptive activation (activations.py:23-32): p = sigmoid(BN(x)) with a non-affine BatchNorm
    (eps 1e-9, momentum 0.01), y = p x + alpha (1 - p) x: rbx_dice_fwd/bwd on 2-D GPU input (statistics, gate and both
    gradients in the library's passes), the reference's composition elsewhere."""

    def __init__(self, input_dim, eps=1e-9):
        super(Dice, self).__init__()
        self.bn = nn.BatchNorm1d(input_dim, affine=False, eps=eps, momentum=0.01)
        self.alpha = nn.Parameter(torch.zeros(input_dim))

    def forward(self, X):
        if X.dim() == 2 and X.is_cuda and X.shape[0] > 0:
            return ops.dice(X, self.bn, self.alpha)
        p = torch.sigmoid(self.bn(X))
        return p * X + self.alpha * (1 - p) * X


class GELU(nn.Module):
    """The tanh form the reference spells out (activations.py:35-40)."""

    def forward(self, x):
        return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


class SqueezeExcitation(nn.Module):
    """FiBiNET's SENET re-weighting of the fields (squeeze_excitation.py:21-41): per-field mean over the embedding
    dimension -> two bias-free Linears (``excitation``, the reference's nn.Sequential: both on the fp32-MFMA GEMM, ReLU
    fused) -> field weights."""

    def __init__(self, num_fields, reduction_ratio=3, excitation_activation="ReLU"):
        super(SqueezeExcitation, self).__init__()
        reduced = max(1, int(num_fields / reduction_ratio))
        kind = excitation_activation.lower()
        if kind not in ("relu", "sigmoid"):
            raise NotImplementedError
        self.excitation = nn.Sequential(nn.Linear(num_fields, reduced, bias=False), nn.ReLU(),
                                        nn.Linear(reduced, num_fields, bias=False),
                                        nn.ReLU() if kind == "relu" else nn.Sigmoid())

    def forward(self, feature_emb):
        A = dense.run_sequential(self.excitation, feature_emb.mean(dim=-1))
        return feature_emb * A.unsqueeze(-1)


class DIN_Attention(nn.Module):
    """DIN's local activation unit (target_attention.py:25-66): an MLP over [target, history, target - history,
    target * history] scores every position of the behaviour sequence (``attention_layer`` = MLP_Block, i.e. the
    [B L, 4E] x [4E, units] products on the fp32 matrix cores); masked, optionally soft-maxed, weighted sum."""

    def __init__(self, embedding_dim=64, attention_units=[32], hidden_activations="ReLU", output_activation=None,
                 dropout_rate=0, batch_norm=False, use_softmax=False):
        super(DIN_Attention, self).__init__()
        from .blocks import MLP_Block
        self.embedding_dim = embedding_dim
        self.use_softmax = use_softmax
        if isinstance(hidden_activations, str) and hidden_activations.lower() == "dice":
            hidden_activations = [Dice(units) for units in attention_units]
        self.attention_layer = MLP_Block(input_dim=4 * embedding_dim, output_dim=1, hidden_units=attention_units,
                                         hidden_activations=hidden_activations, output_activation=output_activation,
                                         dropout_rates=dropout_rate, batch_norm=batch_norm)

    def forward(self, target_item, history_sequence, mask=None):
        seq_len = history_sequence.size(1)
        target = target_item.unsqueeze(1).expand(-1, seq_len, -1)
        pairs = torch.cat([target, history_sequence, target - history_sequence, target * history_sequence], dim=-1)
        weight = self.attention_layer(pairs.view(-1, 4 * self.embedding_dim)).view(-1, seq_len)
        if mask is not None:
            weight = weight * mask.float()
        if self.use_softmax:
            if mask is not None:
                weight = weight + -1.e9 * (1 - mask.float())
            weight = weight.softmax(dim=-1)
        return (weight.unsqueeze(-1) * history_sequence).sum(dim=1)
