"""Two-tower and sequential matching models (drop-in for ``torch_rechub.models.matching.
{DSSM, YoutubeDNN, SASRec}``, /root/reference/recbox/third_party/rechub/models/matching/
dssm.py:15-66, youtube_dnn.py:14-71, sasrec.py:17-124): same constructors, ``mode`` switch,
outputs and parameter names.  Compute: embedding + pooling in one gather launch, MLP Linear
layers on the fp32 matrix cores, ``F.normalize`` and the pairwise dot as HIP kernels
(``rbx_l2norm_*``, ``rbx_pairdot_*``), SASRec's causal attention as the fused LDS-resident
kernel (``rbx_attn_*``)."""
import torch
import torch.nn as nn

from ... import ops
from ..basic.features import DenseFeature
from ..basic.layers import MLP, EmbeddingLayer


class DSSM(torch.nn.Module):
    def __init__(self, user_features, item_features, user_params, item_params, temperature=1.0):
        super().__init__()
        self.user_features = user_features
        self.item_features = item_features
        self.temperature = temperature
        self.user_dims = sum([fea.embed_dim for fea in user_features])
        self.item_dims = sum([fea.embed_dim for fea in item_features])
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.user_mlp = MLP(self.user_dims, output_layer=False, **user_params)
        self.item_mlp = MLP(self.item_dims, output_layer=False, **item_params)
        self.mode = None

    def forward(self, x):
        user_embedding = self.user_tower(x)
        item_embedding = self.item_tower(x)
        if self.mode == "user":
            return user_embedding
        if self.mode == "item":
            return item_embedding
        y = ops.pair_dot(user_embedding, item_embedding).squeeze(1)      # cosine score [B]
        return torch.sigmoid(y)                                          # (the reference leaves /temperature out)

    def user_tower(self, x):
        if self.mode == "item":
            return None
        input_user = self.embedding(x, self.user_features, squeeze_dim=True)
        return ops.l2_normalize(self.user_mlp(input_user))

    def item_tower(self, x):
        if self.mode == "user":
            return None
        input_item = self.embedding(x, self.item_features, squeeze_dim=True)
        return ops.l2_normalize(self.item_mlp(input_item))


class YoutubeDNN(torch.nn.Module):
    def __init__(self, user_features, item_features, neg_item_feature, user_params, temperature=1.0):
        super().__init__()
        self.user_features = user_features
        self.item_features = item_features
        self.neg_item_feature = neg_item_feature
        self.temperature = temperature
        self.user_dims = sum([fea.embed_dim for fea in user_features])
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.user_mlp = MLP(self.user_dims, output_layer=False, **user_params)
        self.mode = None

    def forward(self, x):
        if self.mode is None and not any(isinstance(f, DenseFeature) for f in self.user_features):
            # training: ONE gather launch (and one backward scatter-add) for the user side, the positive item and
            # the negatives.  The history, the positive and the negatives share one table; looking them up in two
            # calls, as youtube_dnn.py:52-70 does, would build two dense [V, D] gradients of that table and add them.
            dim = self.item_features[0].embed_dim
            both = self.embedding(x, self.user_features + self.item_features + self.neg_item_feature, squeeze_dim=True)
            # the two column blocks in place (the tower's first GEMM and the normalisation take a row stride); backward = one
            # concatenation
            user_in, items = ops.split_last(both, self.user_dims, views=True)
            user_embedding = ops.l2_normalize(self.user_mlp(user_in))
            # F.normalize of the item rows + the inner product in one pass over them (rbx_cosdot_*): the normalised
            # [B, 1 + n_neg, D] block is never written, and its gradient lands in a block of `both`'s shape
            return ops.cos_dot(user_embedding, items.unflatten(1, (-1, dim)), eps=1e-12,
                               scale=1.0 / self.temperature)                                    # [B, 1 + n_neg]
        # (a squeezed gather lays DenseFeature values out AFTER every embedding of the call -- layers.py:109-114 --, so
        #  with dense user features the single gather above would put them behind the item / negative rows: the towers
        #  are then looked up separately, as youtube_dnn.py:46-70 does)
        user_embedding = self.user_tower(x)
        item_embedding = self.item_tower(x)
        if self.mode == "user":
            return user_embedding
        if self.mode == "item":
            return item_embedding
        return ops.pair_dot(user_embedding, item_embedding, scale=1.0 / self.temperature)       # [B, 1 + n_neg]

    def user_tower(self, x):
        if self.mode == "item":
            return None
        input_user = self.embedding(x, self.user_features, squeeze_dim=True)
        user_embedding = ops.l2_normalize(self.user_mlp(input_user)).unsqueeze(1)           # [B, 1, D]
        if self.mode == "user":
            return user_embedding.squeeze(1)
        return user_embedding

    def item_tower(self, x):
        if self.mode == "user":
            return None
        if self.mode == "item":
            pos = self.embedding(x, self.item_features, squeeze_dim=False)                  # [B, 1, D]
            return ops.l2_normalize(pos).squeeze(1)
        # positive id and the n_neg negative ids in ONE gather launch: [B, (1 + n_neg) * D]
        both = self.embedding(x, self.item_features + self.neg_item_feature, squeeze_dim=True)
        dim = self.item_features[0].embed_dim
        return ops.l2_normalize(both.reshape(both.shape[0], -1, dim))                          # [B, 1 + n_neg, D]


class PointWiseFeedForward(torch.nn.Module):
    def __init__(self, hidden_units, dropout_rate):
        super(PointWiseFeedForward, self).__init__()
        self.conv1 = torch.nn.Conv1d(hidden_units, hidden_units, kernel_size=1)
        self.dropout1 = torch.nn.Dropout(p=dropout_rate)
        self.relu = torch.nn.ReLU()
        self.conv2 = torch.nn.Conv1d(hidden_units, hidden_units, kernel_size=1)
        self.dropout2 = torch.nn.Dropout(p=dropout_rate)

    def branch(self, inputs):
        """The two 1x1 convolutions without the residual connection."""
        # a kernel-size-1 Conv1d over [B, C, L] is a Linear over the channel axis of [B, L, C]
        if self.dropout1.p > 0 and self.training:
            h = self.relu(self.dropout1(ops.linear(inputs, self.conv1.weight.squeeze(-1), self.conv1.bias)))
        else:
            h = ops.linear(inputs, self.conv1.weight.squeeze(-1), self.conv1.bias, act="relu")
        return self.dropout2(ops.linear(h, self.conv2.weight.squeeze(-1), self.conv2.bias))

    def forward(self, inputs):
        return self.branch(inputs) + inputs


class SASRec(torch.nn.Module):
    """features = [seq, pos, neg] SequenceFeatures (pooling='concat', pos/neg shared_with seq)."""

    def __init__(self, features, max_len=50, dropout_rate=0.5, num_blocks=2, num_heads=1):
        super(SASRec, self).__init__()
        self.features = features
        self.item_num = self.features[0].vocab_size
        self.embed_dim = self.features[0].embed_dim
        self.item_emb = EmbeddingLayer(self.features)
        self.position_emb = torch.nn.Embedding(max_len, self.embed_dim)
        self.emb_dropout = torch.nn.Dropout(p=dropout_rate)
        self.attention_layernorms = torch.nn.ModuleList()
        self.attention_layers = torch.nn.ModuleList()
        self.forward_layernorms = torch.nn.ModuleList()
        self.forward_layers = torch.nn.ModuleList()
        self.last_layernorm = torch.nn.LayerNorm(self.embed_dim, eps=1e-8)
        for _ in range(num_blocks):
            self.attention_layernorms.append(torch.nn.LayerNorm(self.embed_dim, eps=1e-8))
            self.attention_layers.append(torch.nn.MultiheadAttention(self.embed_dim, num_heads, dropout_rate))
            self.forward_layernorms.append(torch.nn.LayerNorm(self.embed_dim, eps=1e-8))
            self.forward_layers.append(PointWiseFeedForward(self.embed_dim, dropout_rate))

    def _mha(self, layer, query, keyval):
        """nn.MultiheadAttention(query, keyval, keyval, attn_mask=causal) on [B, L, E] tensors."""
        p_drop = layer.dropout if self.training else 0.0      # dropout on the probabilities, inside the fused kernel
        E, H = layer.embed_dim, layer.num_heads
        hd = E // H
        w, b = layer.in_proj_weight, layer.in_proj_bias
        q = ops.linear(query, w[:E], b[:E] if b is not None else None)
        kv = ops.linear(keyval, w[E:], b[E:] if b is not None else None)          # one GEMM for K and V
        B, L = query.shape[0], query.shape[1]
        if ops.attention_packed_supported(L, hd):
            # the kernels read Q [B, L, E] and K | V [B, L, 2 E] where the projections left them and write O [B, L, E],
            # dQ and dK | dV in place: no head transposes, no K / V split copies, no concatenation in the backward
            o = ops.attention_packed(q, kv, H, hd ** -0.5, causal=True, dropout_p=p_drop)
            return ops.linear(o, layer.out_proj.weight, layer.out_proj.bias)
        q = q.view(B, L, H, hd).transpose(1, 2)
        k, v = ops.split_last(kv, E)                            # contiguous halves; backward = one concatenation
        k = k.view(B, L, H, hd).transpose(1, 2)
        v = v.view(B, L, H, hd).transpose(1, 2)
        o, _ = ops.attention(q, k, v, mask=None, scale=hd ** -0.5, causal=True, fill=float("-inf"), dropout_p=p_drop)
        o = o.transpose(1, 2).reshape(B, L, E)
        return ops.linear(o, layer.out_proj.weight, layer.out_proj.bias)

    def seq_forward(self, x, embed_x_feature):
        ids = x['seq']
        e = embed_x_feature
        if e.dim() == 4:
            e = e.squeeze(1)
        positions = torch.arange(ids.shape[1], device=ids.device).unsqueeze(0).expand(ids.shape[0], -1)
        keep = (ids != 0).to(e.dtype)                         # ~timeline_mask, one value per (sample, position)
        scale = self.features[0].embed_dim ** 0.5
        input_stage = None
        if isinstance(self.emb_dropout, torch.nn.Dropout) and self.emb_dropout.p > 0 and self.training:
            e = self.emb_dropout(e * scale + ops_position(self.position_emb, positions))
            e = ops.row_scale(e, keep)
        elif (ops.config.seq_positions_in_place and e.dim() == 3 and self.position_emb.padding_idx is None
              and self.position_emb.max_norm is None and ids.shape[1] <= self.position_emb.num_embeddings):
            # positions are arange(L) for every sequence: the table's first L rows read in place, their gradient a column sum
            first = self.attention_layers[0] if len(self.attention_layers) else None
            ffn0 = self.forward_layers[0] if len(self.forward_layers) else None
            if (first is not None and ops.config.fuse_sublayers and ops.config.seqblock_bwd
                    and not (self.training and (ffn0.dropout1.p > 0 or ffn0.dropout2.p > 0))
                    and ops.seqblock_supported(e, first, False)):
                # ... and the stage itself belongs to the first block's node: its backward (de * keep * sqrt(D)) rides in
                # the store of the block's last backward pass instead of a pass of its own over [B L, D]
                input_stage = (self.position_emb.weight[:ids.shape[1]], scale)
            else:
                e = ops.sasrec_input(e, self.position_emb.weight[:ids.shape[1]], keep, alpha=scale)
        else:      # (e * sqrt(D) + position) * ~mask in one pass (rbx_rowscale) instead of three element-wise kernels
            e = ops.row_scale(e, keep, add=ops_position(self.position_emb, positions), alpha=scale)
        for i in range(len(self.attention_layers)):
            mha, ffn = self.attention_layers[i], self.forward_layers[i]
            stage, input_stage = input_stage, None
            E, H = mha.embed_dim, mha.num_heads
            ffn_drop = self.training and (ffn.dropout1.p > 0 or ffn.dropout2.p > 0)
            if ops.config.fuse_sublayers and ops.seqblock_supported(e, mha, ffn_drop):
                # the whole block as one autograd node: LayerNorm + in-projections, attention, out-projection + residual +
                # LayerNorm + FFN + residual + mask -- three launches forward (csrc/rbx_seqblock.hip)
                e = ops.sasrec_block(e, self.attention_layernorms[i], mha, self.forward_layernorms[i],
                                     ffn.conv1.weight.squeeze(-1), ffn.conv1.bias, ffn.conv2.weight.squeeze(-1), ffn.conv2.bias,
                                     keep, dropout_p=mha.dropout if self.training else 0.0, input_stage=stage)
                continue
            if (ops.config.fuse_sublayers and e.dim() == 3 and mha.in_proj_weight is not None
                    and ops.attention_packed_supported(e.shape[1], E // H)):
                # LayerNorm, the three projections, the attention and the residual as ONE autograd node: the residual add
                # and the two gradient sums of the backward live in GEMM epilogues
                e = ops.sasrec_attention_sublayer(e, self.attention_layernorms[i], mha,
                                                  dropout_p=mha.dropout if self.training else 0.0)
            else:
                q = ops.layer_norm(e, self.attention_layernorms[i])
                e = q + self._mha(mha, q, e)
            if ops.config.fuse_sublayers and e.dim() == 3 and not (self.training and (ffn.dropout1.p > 0 or ffn.dropout2.p > 0)):
                e = ops.sasrec_ffn_sublayer(e, self.forward_layernorms[i], ffn.conv1.weight.squeeze(-1), ffn.conv1.bias,
                                            ffn.conv2.weight.squeeze(-1), ffn.conv2.bias, keep, keep_is_mask=True)
            else:
                e = ops.layer_norm(e, self.forward_layernorms[i])
                # (ffn(e) + e) * ~mask: the residual add and the timeline mask in one pass
                e = ops.row_scale(ffn.branch(e), keep, add=e)
        return ops.layer_norm(e, self.last_layernorm)

    def forward(self, x):
        """sasrec.py:96-107.  The reference embeds seq, pos and neg ([B, 3, L, D]) and multiplies; here only the
        input sequence is materialised: the pos / neg logits come from rbx_gatherdot (K7), which reads each
        candidate row once and never writes the two [B, L, D] candidate tensors (nor their gradients)."""
        seq_f, cand_f = self.features[0], self.features[1:]
        seq_embed = self.item_emb(x, [seq_f])                 # [B, 1, L, D]
        # (a view, not seq_embed[:, 0]: the backward of a select is a zero fill + a copy of the whole [B, L, D] gradient --
        #  117 us per step at cfg 5, profiles/r04/INDEX.md; a view's backward is a view)
        seq_output = self.seq_forward(x, seq_embed.view(seq_embed.shape[0], seq_embed.shape[2], seq_embed.shape[3])
                                      if seq_embed.dim() == 4 and seq_embed.shape[1] == 1 else seq_embed[:, 0])
        B, L, D = seq_output.shape
        tables = [self.item_emb.embed_dict[f.name if f.shared_with is None else f.shared_with].weight for f in cand_f]
        logits = ops.gather_dot(seq_output.reshape(B * L, D), [x[f.name].reshape(-1) for f in cand_f], tables)
        return tuple(logits[:, c].reshape(B, L) for c in range(len(cand_f)))


def ops_position(position_emb, positions):
    """position_emb(positions) through the gather kernel (one-table lookup with a [B, L] id block)."""
    from ... import _embed_host as host
    from ..._lib import FIELD_CATEGORICAL, POOL_CONCAT
    plan = getattr(position_emb, "_rbx_plan", None)
    L = positions.shape[1]
    if plan is None or plan.specs[0].seq_len != L:
        plan = host.Plan([host.Lookup("position", FIELD_CATEGORICAL, position_emb, position_emb.embedding_dim,
                                      pool=POOL_CONCAT, seq_len=L)])
        position_emb._rbx_plan = plan
    out = plan.run([positions])
    return out.view(positions.shape[0], L, position_emb.embedding_dim)
