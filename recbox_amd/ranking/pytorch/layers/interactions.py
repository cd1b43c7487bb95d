"""Feature-interaction layer (drop-in for ``recbox.ranking.pytorch.layers.
InnerProductInteraction``, /root/reference/recbox/ranking/pytorch/layers/interactions/
inner_product.py:22-56): same constructor, same four output modes, same ValueError
for an unknown mode; forward/backward run ``rbx_interaction_fwd/bwd``."""
import torch
from torch import nn

from .... import ops

__all__ = ["InnerProductInteraction"]


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
