"""Feature-interaction layer (drop-in for ``recbox.ranking.pytorch.layers.
InnerProductInteraction``, /root/reference/recbox/ranking/pytorch/layers/interactions/
inner_product.py:22-56): same constructor, same four output modes, same ValueError
for an unknown mode; forward/backward run ``rbx_interaction_fwd/bwd``."""
import torch
from torch import nn

from .... import ops

__all__ = ["InnerProductInteraction", "CrossInteraction", "CrossNet", "CrossNetV2", "BilinearInteraction",
           "BilinearInteractionV2", "CompressedInteractionNet", "CrossNetMix", "InteractionMachine",
           "HolographicInteraction"]


class InnerProductInteraction(nn.Module):
    """output: product_sum (bs x 1), bi_interaction (bs x dim), inner_product (bs x f(f-1)/2),
    elementwise_product (bs x f(f-1)/2 x dim)"""

    def __init__(self, num_fields, output="product_sum"):
        super(InnerProductInteraction, self).__init__()
        self._output_type = output
        if output not in ["product_sum", "bi_interaction", "inner_product", "elementwise_product"]:
            raise ValueError("InnerProductInteraction output={} is not supported.".format(output))
        # the reference registers these (non-trainable) parameters; keep them so state_dicts line up
        if output == "inner_product":
            self.interaction_units = int(num_fields * (num_fields - 1) / 2)
            self.triu_mask = nn.Parameter(torch.triu(torch.ones(num_fields, num_fields), 1).bool(),
                                          requires_grad=False)
        elif output == "elementwise_product":
            self.triu_index = nn.Parameter(torch.triu_indices(num_fields, num_fields, offset=1),
                                           requires_grad=False)

    def forward(self, feature_emb):
        return ops.interaction(feature_emb, self._output_type)


class CrossInteraction(nn.Module):
    """``weight(X_i) * X_0 + bias`` (cross_net.py:22-31): parameters as in the reference (``weight`` =
    nn.Linear(input_dim, 1, bias=False), ``bias`` [input_dim])."""

    def __init__(self, input_dim):
        super(CrossInteraction, self).__init__()
        self.weight = nn.Linear(input_dim, 1, bias=False)
        self.bias = nn.Parameter(torch.zeros(input_dim))

    def forward(self, X_0, X_i):
        return ops.cross(X_0, torch.zeros_like(X_i), ops.linear(X_i, self.weight.weight), self.bias)


class CrossNet(nn.Module):
    """X_{i+1} = X_i + (X_i w_i) * X_0 + b_i (cross_net.py:34-46).  The [B, 1] projection runs on the narrow fp32-MFMA
    tile, the rest is one fused pass (rbx_cross_fwd) per layer."""

    def __init__(self, input_dim, num_layers):
        super(CrossNet, self).__init__()
        self.num_layers = num_layers
        self.cross_net = nn.ModuleList(CrossInteraction(input_dim) for _ in range(self.num_layers))

    def forward(self, X_0):
        X_i = X_0
        for i in range(self.num_layers):
            layer = self.cross_net[i]
            X_i = ops.cross(X_0, X_i, ops.linear(X_i, layer.weight.weight), layer.bias)
        return X_i


class CrossNetV2(nn.Module):
    """X_{i+1} = X_i + X_0 * Linear_i(X_i) (cross_net.py:48-59): the Linear is rbx_linear_fwd (fp32 MFMA), the
    element-wise tail rbx_cross_fwd."""

    def __init__(self, input_dim, num_layers):
        super(CrossNetV2, self).__init__()
        self.num_layers = num_layers
        self.cross_layers = nn.ModuleList(nn.Linear(input_dim, input_dim) for _ in range(self.num_layers))

    def forward(self, X_0):
        X_i = X_0
        for i in range(self.num_layers):
            layer = self.cross_layers[i]
            X_i = ops.cross(X_0, X_i, ops.linear(X_i, layer.weight, layer.bias))
        return X_i


class CrossNetMix(nn.Module):
    """DCN-M's mixture of low-rank cross experts (cross_net.py:60-117), parameter holders as in the reference
    (``U_list`` / ``V_list`` / ``C_list``: per layer [experts, in, r] / [experts, in, r] / [experts, r, r], ``gating``:
    one bias-free Linear(in, 1) per expert, ``bias``: per layer [in, 1]).

    Per layer and expert e the reference computes x_0 * (U_e tanh(C_e tanh(V_e^T x_l)) + b), and mixes the experts with
    softmax(gates).  Because the softmax weights sum to one, the mixture is x_0 * (sum_e p_e U_e h_e + b): the experts
    are batched into THREE fp32-MFMA GEMMs per layer -- [B, in] x [in, E r] for all V_e, [B, E r] x [E r, in] for all U_e
    applied to the gate-scaled h_e, [B, in] x [in, E] for the gates -- plus one small [B, r] x [r, r] per expert, and the
    layer ends in rbx_cross_fwd (x_l + x_0 * h, the bias riding in the last GEMM's epilogue), instead of 4 matmuls and a Hadamard product per expert."""

    def __init__(self, in_features, layer_num=2, low_rank=32, num_experts=4):
        super(CrossNetMix, self).__init__()
        self.layer_num, self.num_experts = layer_num, num_experts

        def stack(*shape):
            return nn.ParameterList(nn.Parameter(nn.init.xavier_normal_(torch.empty(num_experts, *shape)))
                                    for _ in range(layer_num))

        self.U_list = stack(in_features, low_rank)
        self.V_list = stack(in_features, low_rank)
        self.C_list = stack(low_rank, low_rank)
        self.gating = nn.ModuleList(nn.Linear(in_features, 1, bias=False) for _ in range(num_experts))
        self.bias = nn.ParameterList(nn.Parameter(torch.zeros(in_features, 1)) for _ in range(layer_num))

    def forward(self, inputs):
        E = self.num_experts
        x_0 = inputs
        x_l = x_0
        gate_w = torch.cat([g.weight for g in self.gating], dim=0)                       # [E, in]
        for i in range(self.layer_num):
            U, V, C = self.U_list[i], self.V_list[i], self.C_list[i]
            n_in, r = V.shape[1], V.shape[2]
            p = torch.softmax(ops.linear(x_l, gate_w), dim=1)                            # [B, E]
            h = torch.tanh(ops.linear(x_l, V.permute(0, 2, 1).reshape(E * r, n_in)))     # every V_e^T x_l: [B, E r]
            h = torch.stack([ops.linear(h[:, e * r:(e + 1) * r], C[e]) for e in range(E)], dim=1)
            h = torch.tanh(h) * p.unsqueeze(2)                                           # p_e tanh(C_e .): [B, E, r]
            mix = ops.linear(h.reshape(-1, E * r), U.permute(1, 0, 2).reshape(n_in, E * r),
                             self.bias[i].reshape(-1))                                   # sum_e p_e U_e h_e + b
            x_l = ops.cross(x_0, x_l, mix)
        return x_l.unsqueeze(2).squeeze()


class BilinearInteractionV2(nn.Module):
    """FiBiNET's bilinear interaction (bilinear_interaction.py:24-90; V1 and V2 of the reference compute the same
    [B, F(F-1)/2, D] tensor, V1 pair by pair in Python): p_ij = (e_i W) * e_j over the pairs i < j with
    ``bilinear_type`` choosing W: one matrix ("field_all"), one per left field ("field_each") or one per pair
    ("field_interaction", the default).  ``bilinear_W`` has the reference's shape and xavier-normal init.

    The products with W are fp32-MFMA GEMMs (rbx_linear_fwd): one [B*F, D] x [D, D] for field_all, one
    [B, F*D] x block_diag(W_0..W_{F-1}) for field_each, and for field_interaction one [B, D] x [D, (F-1-i) D] per left
    field i (its pairs' matrices side by side; the field's [B, D] slice is read in place).  The pairing is
    rbx_pairmul_fwd/bwd."""

    def __init__(self, num_fields, embedding_dim, bilinear_type="field_interaction"):
        super(BilinearInteractionV2, self).__init__()
        self.bilinear_type = bilinear_type
        self.num_fields, self.embedding_dim = num_fields, embedding_dim
        self.interact_dim = int(num_fields * (num_fields - 1) / 2)
        lead = {"field_all": (), "field_each": (num_fields,), "field_interaction": (self.interact_dim,)}
        if bilinear_type not in lead:
            raise NotImplementedError
        self.bilinear_W = nn.Parameter(torch.Tensor(*lead[bilinear_type], embedding_dim, embedding_dim))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_normal_(self.bilinear_W)

    def forward(self, feature_emb):
        B, F, D = feature_emb.shape
        W = self.bilinear_W
        if self.bilinear_type == "field_all":
            hidden = ops.linear(feature_emb.reshape(B * F, D), W.t()).view(B, F, D)
            return ops.pair_mul(hidden, feature_emb)
        if self.bilinear_type == "field_each":
            hidden = ops.linear(feature_emb.reshape(B, F * D), torch.block_diag(*W.unbind(0)).t()).view(B, F, D)
            return ops.pair_mul(hidden, feature_emb)
        lefts, p0 = [], 0
        for i in range(F - 1):
            n = F - 1 - i                                       # pairs (i, i+1) .. (i, F-1) are consecutive in triu order
            # [e_i W_p for the n pairs] = e_i [W_p0 | W_p0+1 | ...]: as an nn.Linear weight that is [(n D), D]
            weight = W[p0:p0 + n].transpose(1, 2).reshape(n * D, D)
            lefts.append(ops.linear(feature_emb[:, i, :], weight))
            p0 += n
        left = torch.cat(lefts, dim=1).view(B, self.interact_dim, D)
        return ops.pair_mul(left, feature_emb, per_pair=True)


class BilinearInteraction(BilinearInteractionV2):
    """Same layer (the reference keeps both spellings; this one loops over pairs in Python there)."""


class CompressedInteractionNet(nn.Module):
    """xDeepFM's CIN (compressed_interaction_net.py:21-48) with the reference's parameter holders (``cin_layer.layer_<k>``
    = nn.Conv1d(kernel_size=1), ``fc`` = nn.Linear) so checkpoints load unchanged.

    Layer k: Z[b, (h, m), d] = X_0[b, h, d] * X_k[b, m, d], X_{k+1}[b, o, d] = sum_c W[o, c] Z[b, c, d] + bias[o].  The 1x1
    convolution is a GEMM over the channel axis for every (b, d): it runs as ONE fp32-MFMA product
    [B*D, F*H_k] x [F*H_k, H_{k+1}] (rbx_linear_fwd) on the outer-product tensor laid out with d next to b, which
    rbx_cin_outer_fwd/bwd (csrc/rbx_cin.hip) writes straight from the embedding layer's [B, F, D] block and the previous
    layer's GEMM output -- materialised once, as in the reference, but with no transposes, broadcasts or reshapes of its own."""

    def __init__(self, num_fields, cin_hidden_units, output_dim=1):
        super(CompressedInteractionNet, self).__init__()
        self.cin_hidden_units = cin_hidden_units
        self.fc = nn.Linear(sum(cin_hidden_units), output_dim)
        self.cin_layer = nn.ModuleDict()
        width = num_fields
        for k, unit in enumerate(cin_hidden_units):
            self.cin_layer["layer_" + str(k + 1)] = nn.Conv1d(num_fields * width, unit, kernel_size=1)
            width = unit

    def forward(self, feature_emb):
        B, F, D = feature_emb.shape
        hip = feature_emb.is_cuda and B > 0
        x0 = feature_emb.transpose(1, 2)                        # [B, D, F]: the GEMM wants channels last
        xk = x0
        flat = None                                             # the previous layer's GEMM output as it is: [(b, d), H_k]
        pooled = []
        for k in range(len(self.cin_hidden_units)):
            conv = self.cin_layer["layer_" + str(k + 1)]
            if hip:
                z = ops.cin_outer(feature_emb, flat)                                         # [(b, d), (h, m)]
            else:
                z = (x0.unsqueeze(3) * xk.unsqueeze(2)).reshape(B * D, F * xk.shape[2])
            flat = ops.linear(z, conv.weight.squeeze(-1), conv.bias)
            xk = flat.view(B, D, -1)                                                      # [B, D, H_{k+1}]
            pooled.append(xk.sum(dim=1))                                                  # sum over d -> [B, H_{k+1}]
        return ops.linear(torch.cat(pooled, dim=-1), self.fc.weight, self.fc.bias)


class InteractionMachine(nn.Module):
    """Interaction Machine (interaction_machine.py:21-71): the order-k elementary symmetric polynomials of the field
    embeddings, from the power sums p_k = sum_f e_f^k (Newton's identities), concatenated, optionally batch-normalised,
    then ``fc``.  Holders ``bn`` / ``fc`` as in the reference; the BatchNorm runs on rbx_batchnorm_fwd/bwd, ``fc`` on the
    fp32-MFMA GEMM; the power sums are element-wise ATen passes over [B, F, D] (memory-bound, nothing to fuse them into)."""

    def __init__(self, embedding_dim, order=2, batch_norm=False):
        super(InteractionMachine, self).__init__()
        assert order < 6, "order={} is not supported.".format(order)
        self.order = order
        self.bn = nn.BatchNorm1d(embedding_dim * order) if batch_norm else None
        self.fc = nn.Linear(order * embedding_dim, 1)

    @staticmethod
    def _elementary(k, p):
        p1, p2, p3, p4, p5 = (p + [None] * 5)[:5]
        if k == 1:
            return p1
        if k == 2:
            return (p1.pow(2) - p2) / 2
        if k == 3:
            return (p1.pow(3) - 3 * p1 * p2 + 2 * p3) / 6
        if k == 4:
            return (p1.pow(4) - 6 * p1.pow(2) * p2 + 3 * p2.pow(2) + 8 * p1 * p3 - 6 * p4) / 24
        return (p1.pow(5) - 10 * p1.pow(3) * p2 + 20 * p1.pow(2) * p3 - 30 * p1 * p4 - 20 * p2 * p3
                + 15 * p1 * p2.pow(2) + 24 * p5) / 120

    def forward(self, X):
        sums, Q = [], X
        for k in range(1, self.order + 1):
            if k > 1:
                Q = Q * X
            sums.append(Q.sum(dim=1))
        out = torch.cat([self._elementary(k, sums) for k in range(1, self.order + 1)], dim=-1)
        if self.bn is not None:
            out = ops.batch_norm(out, self.bn)
        return ops.linear(out, self.fc.weight, self.fc.bias)


class HolographicInteraction(nn.Module):
    """HFM's pairwise compression (holographic_interaction.py:22-52) over the pairs i < j: Hadamard product
    (rbx_pairmul_fwd/bwd), circular convolution or circular correlation of the two embeddings.  The circular forms are
    evaluated through the FFT exactly as the reference writes them (torch.fft, i.e. rocFFT: D = 16..64 points per row);
    ``triu_index`` / ``conj_sign`` stay frozen parameters so that state_dicts match."""

    def __init__(self, num_fields, interaction_type="circular_convolution"):
        super(HolographicInteraction, self).__init__()
        self.interaction_type = interaction_type
        if self.interaction_type == "circular_correlation":
            self.conj_sign = nn.Parameter(torch.tensor([1., -1.]), requires_grad=False)
        self.triu_index = nn.Parameter(torch.triu_indices(num_fields, num_fields, offset=1), requires_grad=False)

    def forward(self, feature_emb):
        if self.interaction_type == "hadamard_product":
            return ops.pair_mul(feature_emb, feature_emb)
        if self.interaction_type not in ("circular_convolution", "circular_correlation"):
            raise ValueError("interaction_type={} not supported.".format(self.interaction_type))
        left = torch.fft.fft(torch.index_select(feature_emb, 1, self.triu_index[0]))
        right = torch.fft.fft(torch.index_select(feature_emb, 1, self.triu_index[1]))
        if self.interaction_type == "circular_correlation":
            left = torch.conj(left)                       # the reference flips the sign of the imaginary part
        return torch.fft.ifft(left * right).real
