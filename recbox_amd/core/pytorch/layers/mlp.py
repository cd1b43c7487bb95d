"""Dense tower of the matching side (drop-in for ``recbox.core.pytorch.layers.MLP_Layer``,
/root/reference/recbox/core/pytorch/layers/mlp.py:7-39): same constructor and the same
``self.mlp`` nn.Sequential child order with real ``nn.Linear`` holders (checkpoints and
``MatchingModel.init_weights`` type tests keep working); ``forward`` runs every Linear on the
fp32 matrix cores through ``rbx_linear_fwd/bwd`` (ReLU fused into the epilogue)."""
from torch import nn

from .... import dense

__all__ = ["MLP_Layer"]


class MLP_Layer(nn.Module):
    def __init__(self, input_dim, output_dim=None, hidden_units=[], hidden_activations="ReLU",
                 final_activation=None, dropout_rates=[], batch_norm=False, use_bias=True):
        super(MLP_Layer, self).__init__()
        layers = []
        if not isinstance(dropout_rates, list):
            dropout_rates = [dropout_rates] * len(hidden_units)
        if not isinstance(hidden_activations, list):
            hidden_activations = [hidden_activations] * len(hidden_units)
        acts = [dense.activation_by_name(a) for a in hidden_activations]
        dims = [input_dim] + list(hidden_units)
        for i in range(len(dims) - 1):
            layers.append(nn.Linear(dims[i], dims[i + 1], bias=use_bias))
            if batch_norm:
                layers.append(nn.BatchNorm1d(dims[i + 1]))
            if acts[i]:
                layers.append(acts[i])
            if dropout_rates[i] > 0:
                layers.append(nn.Dropout(p=dropout_rates[i]))
        if output_dim is not None:
            layers.append(nn.Linear(dims[-1], output_dim, bias=use_bias))
        if final_activation is not None:
            layers.append(dense.activation_by_name(final_activation))
        self.mlp = nn.Sequential(*layers)

    def forward(self, inputs):
        return dense.run_sequential(self.mlp, inputs)
