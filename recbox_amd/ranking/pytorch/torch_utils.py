"""``get_loss`` of the ranking harness (/root/reference/recbox/ranking/pytorch/torch_utils.py:54-65) with the same
aliases and error: "bce" / "binary_crossentropy" / "binary_cross_entropy" resolve to the fused mean-reduced BCE of
``recbox_amd.ops`` (one forward pass + one backward pass on the GPU instead of ATen's five kernels); every other name is
looked up in ``torch.nn.functional`` exactly as the reference does."""
import torch

from ... import ops

_BCE_ALIASES = ("bce", "binary_crossentropy", "binary_cross_entropy")


def get_loss(loss):
    if isinstance(loss, str) and loss in _BCE_ALIASES:
        return ops.binary_cross_entropy
    if callable(loss):
        return loss
    fn = getattr(torch.nn.functional, str(loss), None)
    if fn is None:
        raise NotImplementedError("loss={} is not supported.".format(loss))
    return fn
