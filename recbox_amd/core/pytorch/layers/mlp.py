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
        self.mlp = nn.Sequential(*dense.tower_modules(input_dim, hidden_units, hidden_activations, dropout_rates,
                                                      batch_norm, use_bias, out_dim=output_dim,
                                                      out_activation=final_activation))

    def forward(self, inputs):
        return dense.run_sequential(self.mlp, inputs)
