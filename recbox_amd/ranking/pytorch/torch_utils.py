"""``get_activation`` and ``get_loss`` of the ranking harness (/root/reference/recbox/ranking/pytorch/torch_utils.py:54-65) with the same
aliases and error: "bce" / "binary_crossentropy" / "binary_cross_entropy" resolve to the fused mean-reduced BCE of
``recbox_amd.ops`` (one forward pass + one backward pass on the GPU instead of ATen's five kernels); every other name is
looked up in ``torch.nn.functional`` exactly as the reference does."""
import torch

from ... import ops

_BCE_ALIASES = ("bce", "binary_crossentropy", "binary_cross_entropy")


def get_activation(activation, hidden_units=None):
    """ranking/pytorch/torch_utils.py:85-110: a name (or a list of names, one per entry of ``hidden_units``) -> module(s);
    "prelu" = nn.PReLU(hidden_units, init=0.1), "dice" = Dice(hidden_units), both needing an int ``hidden_units``."""
    from ... import dense
    if isinstance(activation, str):
        if activation.lower() in ("prelu", "dice"):
            assert type(hidden_units) == int                      # noqa: E721 -- the reference's own check
        return dense.activation_by_name(activation, hidden_units if hidden_units is not None else 0)
    if isinstance(activation, list):
        if hidden_units is not None:
            assert len(activation) == len(hidden_units)
            return [get_activation(act, units) for act, units in zip(activation, hidden_units)]
        return [get_activation(act) for act in activation]
    return activation


def get_loss(loss):
    if isinstance(loss, str) and loss in _BCE_ALIASES:
        return ops.binary_cross_entropy
    if callable(loss):
        return loss
    fn = getattr(torch.nn.functional, str(loss), None)
    if fn is None:
        raise NotImplementedError("loss={} is not supported.".format(loss))
    return fn
