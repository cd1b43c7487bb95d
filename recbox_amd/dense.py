"""Dense-tower execution: nn.Sequential children stay the reference's modules (real
``nn.Linear`` so harness type tests / checkpoints keep working); this walker replaces
their compute with the fp32-MFMA GEMM of ``rbx_linear_fwd/bwd`` and fuses a following
ReLU into the GEMM epilogue; nn.BatchNorm1d on 2-D activations runs through
``rbx_batchnorm_fwd/bwd`` (a following ReLU or PReLU fused in; nn.SyncBatchNorm modules
normalise over all ranks).  nn.PReLU standing alone, nn.Dropout in training mode and Dice run
``rbx_prelu_*`` / ``rbx_dropout`` / ``rbx_dice_*`` (csrc/rbx_act.hip); Dropout with p = 0 or in
evaluation launches nothing; Sigmoid / Tanh / Softmax heads are elementwise ATen kernels.
"""
from torch import nn

from . import ops


def activation_by_name(name, width=None):
    """'ReLU' / 'relu' / nn.Module -> module.  ``width`` None: recbox.utils.torch_utils.set_activation (utils/torch_utils.py:
    74-85: relu / sigmoid / tanh by any case, every other name ``getattr(nn, name)()``).  ``width`` = the layer's units:
    fuxictr's get_activation (ranking/pytorch/torch_utils.py:85-110; 0 = called without units), which also knows "softmax" (dim -1), "prelu" -- ONE SLOPE
    PER UNIT, nn.PReLU(units, init=0.1) -- and "dice" (Dice(units))."""
    if name is None or isinstance(name, nn.Module):
        return name
    if isinstance(name, str):
        low = name.lower()
        if low == "relu":
            return nn.ReLU()
        if low == "sigmoid":
            return nn.Sigmoid()
        if low == "tanh":
            return nn.Tanh()
        if width is not None:                     # (0 = fuxictr's flavour without units: an output activation)
            if low == "softmax":
                return nn.Softmax(dim=-1)
            if low in ("prelu", "dice"):
                assert width > 0, "activation=%s needs the layer's units (get_activation(activation, hidden_units))" % name
                if low == "prelu":
                    return nn.PReLU(int(width), init=0.1)
                from .ranking.pytorch.layers.attentions import Dice
                return Dice(int(width))
        if low in ("none", ""):
            return None
        return getattr(nn, name)()
    raise NotImplementedError("activation={} is not supported.".format(name))


def _broadcast(value, n):
    """A per-layer setting given once (scalar / name / None) or as a list of n entries."""
    return list(value) if isinstance(value, (list, tuple)) else [value] * n


def tower_modules(in_dim, widths, activations, dropouts, batch_norm, bias, out_dim=None, out_activation=None,
                  norm_after_activation=False, width_aware=False):
    """The children of a tower's ``nn.Sequential`` in the order the reference registers them (so ``mlp.<i>.weight``
    checkpoint keys line up): per hidden layer Linear, [BatchNorm1d], [activation], [BatchNorm1d when it comes after
    the activation], [Dropout]; then the optional output Linear and output activation."""
    widths = list(widths)
    per_act = _broadcast(activations, len(widths))
    per_drop = _broadcast(dropouts, len(widths))
    if len(per_act) < len(widths) or len(per_drop) < len(widths):
        # the reference indexes ``hidden_activations[idx]`` / ``dropout_rates[idx]`` (mlp.py:25-37, mlp_block.py:42-58):
        # a list shorter than hidden_units is an IndexError there, never a silently shorter tower
        raise IndexError("list index out of range (%d hidden layers, %d activations, %d dropout rates)"
                         % (len(widths), len(per_act), len(per_drop)))
    fan_in = in_dim
    for width, act, drop in zip(widths, per_act, per_drop):
        yield nn.Linear(fan_in, width, bias=bias)
        fan_in = width
        if batch_norm and not norm_after_activation:
            yield nn.BatchNorm1d(width)
        module = activation_by_name(act, width if width_aware else None)
        if module:
            yield module
        if batch_norm and norm_after_activation:
            yield nn.BatchNorm1d(width)
        if drop and drop > 0:
            yield nn.Dropout(p=drop)
    if out_dim is not None:
        yield nn.Linear(fan_in, out_dim, bias=bias)
    if out_activation is not None:
        yield activation_by_name(out_activation, 0 if width_aware else None)


def run_sequential(seq, x):
    """Forward through an nn.Sequential, routing every nn.Linear through the HIP GEMM."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if type(m) is nn.Linear:
            fuse = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
            x = ops.linear(x, m.weight, m.bias, "relu" if fuse else None)
            i += 2 if fuse else 1
        elif type(m) in (nn.BatchNorm1d, nn.SyncBatchNorm) and x.dim() == 2:
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            fuse = type(nxt) is nn.ReLU
            # nn.PReLU (one slope, or one per column) rides in the BatchNorm's passes too: rechub's towers with
            # activation="prelu" (DSSM, BASELINE cfg 1) are Linear -> BatchNorm1d -> PReLU -> Dropout
            prelu = nxt if (type(nxt) is nn.PReLU and nxt.weight.numel() in (1, x.shape[1])) else None
            x = ops.batch_norm(x, m, relu=fuse, prelu=prelu)
            i += 2 if (fuse or prelu is not None) else 1
        elif type(m) is nn.PReLU:                 # standing alone (behind a BatchNorm1d it rode in the BatchNorm's passes)
            x = ops.prelu(x, m)
            i += 1
        elif type(m) is nn.Dropout:
            x = ops.dropout(x, m.p, m.training) if x.is_cuda else m(x)
            i += 1
        else:
            x = m(x)                              # Dice modules route themselves (ops.dice); Sigmoid / Tanh / ...: ATen
            i += 1
    return x
