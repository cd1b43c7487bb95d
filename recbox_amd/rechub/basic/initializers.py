"""Embedding initialisers with the names and call convention of ``torch_rechub.basic.initializers``
(/root/reference/recbox/third_party/rechub/basic/initializers.py): ``Init(...)(vocab_size, embed_dim)`` returns
a plain ``nn.Embedding`` WITHOUT padding_idx -- which is why rechub's pad row is an ordinary trainable row that
merely gets weight 0 in the pooling mask (SURVEY.md a-5 ii).

One base class holds the "make a table, fill it" step; a subclass only names the ``torch.nn.init`` routine and
maps its constructor arguments onto that routine's."""
import torch
from torch.nn import init as _init


class _TableInit(object):
    fill = None                       # torch.nn.init.<routine>_
    arg_names = ()                    # constructor arguments, in the order the routine takes them

    def __init__(self, *args, **kwargs):
        given = dict(zip(self.arg_names, args))
        given.update(kwargs)
        for name, default in zip(self.arg_names, self.defaults):
            setattr(self, name, given.get(name, default))

    def __call__(self, vocab_size, embed_dim):
        table = torch.nn.Embedding(vocab_size, embed_dim)
        type(self).fill(table.weight, *(getattr(self, n) for n in self.arg_names))
        return table


class RandomNormal(_TableInit):
    fill, arg_names, defaults = staticmethod(_init.normal_), ("mean", "std"), (0.0, 1.0)


class RandomUniform(_TableInit):
    fill, arg_names, defaults = staticmethod(_init.uniform_), ("minval", "maxval"), (0.0, 1.0)


class XavierNormal(_TableInit):
    fill, arg_names, defaults = staticmethod(_init.xavier_normal_), ("gain",), (1.0,)


class XavierUniform(_TableInit):
    fill, arg_names, defaults = staticmethod(_init.xavier_uniform_), ("gain",), (1.0,)


class Pretrained(object):
    """A table taken from a given weight matrix (frozen unless ``freeze=False``)."""

    def __init__(self, embedding_weight, freeze=True):
        self.embedding_weight = torch.FloatTensor(embedding_weight)
        self.freeze = freeze

    def __call__(self, vocab_size, embed_dim):
        if tuple(self.embedding_weight.shape) != (vocab_size, embed_dim):
            raise AssertionError("pretrained weight is %s, the feature asks for (%d, %d)"
                                 % (tuple(self.embedding_weight.shape), vocab_size, embed_dim))
        return torch.nn.Embedding.from_pretrained(self.embedding_weight, freeze=self.freeze)
