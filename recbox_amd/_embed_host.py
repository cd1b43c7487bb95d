"""Host-side planner shared by the three embedding-layer flavours.

A layer call is described as an ordered list of ``Lookup`` items (one per output
slot); ``make_plan`` lays the slots out back to back in one ``[B, width]`` row,
dedupes parameter holders (``share_embedding`` / ``shared_with`` alias the same
module object) and returns an ``ops.EmbedPlan`` plus the holder modules in
parameter order.  Plans are cached by the caller per call signature.
"""
from collections import OrderedDict

from torch import nn

from . import ops
from ._lib import FIELD_CATEGORICAL, FIELD_DENSE, FIELD_NUMERIC, POOL_CONCAT, POOL_NONE


class Lookup(object):
    __slots__ = ("name", "kind", "module", "dim", "pool", "seq_len", "mask_id", "eps")

    def __init__(self, name, kind, module=None, dim=1, pool=POOL_NONE, seq_len=1, mask_id=None, eps=0.0):
        self.name, self.kind, self.module, self.dim = name, kind, module, dim
        self.pool, self.seq_len, self.mask_id, self.eps = pool, seq_len, mask_id, eps


class Plan(object):
    """ops.EmbedPlan + the parameter holders + slot geometry."""

    def __init__(self, lookups, offsets=None, width=None):
        """offsets / width: explicit slot offsets inside a wider output row (the row-sharded layers leave holes that
        the exchange fills: recbox_amd.sharded); default = the slots back to back."""
        self.lookups = lookups
        self.modules = []
        specs, off = [], 0
        for k, lk in enumerate(lookups):
            if offsets is not None:
                off = offsets[k]
            param = -1
            vocab, padding_idx = 0, None
            if lk.kind != FIELD_DENSE:
                for i, m in enumerate(self.modules):
                    if m is lk.module:
                        param = i
                if param < 0:
                    param = len(self.modules)
                    self.modules.append(lk.module)
                if lk.kind == FIELD_CATEGORICAL:
                    if type(lk.module) is not nn.Embedding and not isinstance(lk.module, nn.Embedding):
                        raise TypeError("feature '%s': categorical features need an nn.Embedding holder" % lk.name)
                    vocab, padding_idx = lk.module.num_embeddings, lk.module.padding_idx
            spec = ops.FieldSpec(lk.name, lk.kind, lk.dim, off, param=param, pool=lk.pool, seq_len=lk.seq_len,
                                 vocab=vocab, padding_idx=padding_idx, mask_id=lk.mask_id, eps=lk.eps)
            specs.append(spec)
            off += spec.width
        self.specs = specs
        self.width = off if width is None else width
        self.plan = ops.EmbedPlan(specs, self.width)
        dims = set(s.dim for s in specs)
        self.uniform_dim = dims.pop() if len(dims) == 1 else None

    def run(self, inputs, pad_rows=False):
        params = [m.weight for m in self.modules]
        return ops.embed_lookup(self.plan, inputs, params, pad_rows=pad_rows)

    def slot(self, out, i):
        """View of output slot i: [B, dim] or [B, L, dim] for concat pooling."""
        s = self.specs[i]
        v = out[:, s.out_off:s.out_off + s.width]
        if s.pool == POOL_CONCAT:
            v = v.reshape(out.shape[0], s.seq_len, s.dim)
        return v


class FusedDict(OrderedDict):
    """feature -> embedding views of one layer call that also remembers the fused ``[B, width]`` block they are cut
    from, so that ``dict2tensor`` can hand the block out instead of stacking the views again.  The shortcut is only
    valid while the dict still holds exactly what the layer put there: ANY mutation after ``seal`` (the FuxiCTR idiom
    ``feature_emb_dict[f] = pooled_or_gated_emb``, ``pop``, ``del``, ``update`` ...) drops the block, and
    ``dict2tensor`` then stacks the CURRENT values as the reference does (feature_embedding.py:169-186,
    core/pytorch/layers/embedding.py:109-114)."""
    fused = None
    plan = None
    names = ()
    _sealed = False

    def seal(self, fused, plan, names=()):
        self.fused, self.plan, self.names, self._sealed = fused, plan, tuple(names), True

    def _touch(self):
        if self._sealed:
            self.fused, self.plan, self.names, self._sealed = None, None, (), False

    def __setitem__(self, key, value):
        self._touch()
        OrderedDict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._touch()
        OrderedDict.__delitem__(self, key)

    def pop(self, *args):
        self._touch()
        return OrderedDict.pop(self, *args)

    def popitem(self, last=True):
        self._touch()
        return OrderedDict.popitem(self, last)

    def clear(self):
        self._touch()
        OrderedDict.clear(self)

    def update(self, *args, **kwargs):
        self._touch()
        OrderedDict.update(self, *args, **kwargs)

    def setdefault(self, key, default=None):
        if key not in self:
            self._touch()
        return OrderedDict.setdefault(self, key, default)

    def move_to_end(self, key, last=True):
        self._touch()
        OrderedDict.move_to_end(self, key, last)
