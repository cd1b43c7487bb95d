"""List-wise / pair-wise losses over ``y_pred[B, 1 + num_negs]`` (column 0 = the positive item).
Drop-in names for ``recbox.core.pytorch.losses`` (/root/reference/recbox/core/pytorch/losses/*.py).
They close the forward->backward loop of the two-tower models; each is a scalar reduction over a
tiny ``[B, N]`` tensor.  ``SoftmaxCrossEntropyLoss`` -- the sampled-softmax loss, the second half of SURVEY.md's K7 --
runs on the fused ``rbx_softmax_ce_*`` epilogue; the others stay ATen expressions as SURVEY.md a-13 prescribes (the hot
kernels upstream of them produce ``y_pred``)."""
import torch
import torch.nn.functional as F
from torch import nn

__all__ = ["SoftmaxCrossEntropyLoss", "SigmoidCrossEntropyLoss", "PairwiseLogisticLoss", "PairwiseMarginLoss",
           "MSELoss", "CosineContrastiveLoss"]


def _split(y_pred):
    return y_pred[:, 0], y_pred[:, 1:]


class SoftmaxCrossEntropyLoss(nn.Module):
    """Sampled softmax: -log softmax(y)[:, 0], mean over the batch (softmax_crossentropy_loss.py:19-22)."""

    def forward(self, y_pred, y_true):
        if y_pred.is_cuda and y_pred.dim() == 2:         # K7's loss epilogue: one pass each way (rbx_softmax_ce_*)
            from ... import ops
            return ops.softmax_cross_entropy(y_pred)
        return -torch.log(F.softmax(y_pred, dim=1)[:, 0]).mean()


class SigmoidCrossEntropyLoss(nn.Module):
    """Sum-reduced BCE-with-logits over every (sample, candidate) (sigmoid_crossentropy_loss.py:19-22)."""

    def forward(self, y_pred, y_true):
        return F.binary_cross_entropy_with_logits(y_pred.flatten(), y_true.flatten(), reduction="sum")


class PairwiseLogisticLoss(nn.Module):
    """-log sigmoid(pos - neg), mean over all pairs (pairwise_logistic_loss.py)."""

    def forward(self, y_pred, y_true):
        pos, neg = _split(y_pred)
        return -torch.log(torch.sigmoid(pos.unsqueeze(-1) - neg)).mean()


class PairwiseMarginLoss(nn.Module):
    def __init__(self, margin=1.0):
        super(PairwiseMarginLoss, self).__init__()
        self._margin = margin

    def forward(self, y_pred, y_true):
        pos, neg = _split(y_pred)
        return torch.relu(self._margin + neg - pos.unsqueeze(-1)).mean()


class MSELoss(nn.Module):
    """(pos - 1)^2 / 2 + sum_neg neg^2 / 2, mean over the batch (mse_loss.py)."""

    def forward(self, y_pred, y_true):
        pos, neg = _split(y_pred)
        return (torch.pow(pos - 1, 2) / 2 + torch.pow(neg, 2).sum(dim=-1) / 2).mean()


class CosineContrastiveLoss(nn.Module):
    """relu(1 - pos) + [sum | weighted mean] relu(neg - margin) (cosine_contrastive_loss.py)."""

    def __init__(self, margin=0, negative_weight=None):
        super(CosineContrastiveLoss, self).__init__()
        self._margin = margin
        self._negative_weight = negative_weight

    def forward(self, y_pred, y_true):
        pos, neg = _split(y_pred)
        neg_loss = torch.relu(neg - self._margin)
        if self._negative_weight:
            return (torch.relu(1 - pos) + neg_loss.mean(dim=-1) * self._negative_weight).mean()
        return (torch.relu(1 - pos) + neg_loss.sum(dim=-1)).mean()
