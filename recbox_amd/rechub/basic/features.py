"""Feature descriptors of the rechub-style API (same class names, constructor arguments and attributes as
``torch_rechub.basic.features``, /root/reference/recbox/third_party/rechub/basic/features.py:5-94).

A table-backed feature owns at most ONE ``nn.Embedding``: ``get_embedding_layer()`` builds it on first use and
hands the same object out afterwards, so every model that lists the feature trains the same parameters."""
import numpy as np

from .initializers import RandomNormal


def get_auto_embedding_dim(num_classes):
    """Default width of a table: floor(6 * num_classes ** 0.25) (rechub/utils/data.py)."""
    return int(np.floor(6 * np.power(num_classes, 0.25)))


class _TableFeature(object):
    """What SparseFeature and SequenceFeature share: a vocabulary, a width, an optional table to borrow
    (``shared_with``), the id that marks padding, and the lazily built table."""

    def _describe(self, name, vocab_size, embed_dim, shared_with, padding_idx, initializer):
        self.name = name
        self.vocab_size = vocab_size
        self.embed_dim = embed_dim if embed_dim is not None else get_auto_embedding_dim(vocab_size)
        self.shared_with = shared_with
        self.padding_idx = padding_idx
        self.initializer = initializer

    def get_embedding_layer(self):
        try:
            return self.embed
        except AttributeError:
            self.embed = self.initializer(self.vocab_size, self.embed_dim)
            return self.embed

    def __repr__(self):
        return "<%s %s with Embedding shape (%s, %s)>" % (type(self).__name__, self.name, self.vocab_size, self.embed_dim)


class SequenceFeature(_TableFeature):
    """A padded id sequence; ``pooling`` in {"mean", "sum", "concat"} says how its rows are combined."""

    def __init__(self, name, vocab_size, embed_dim=None, pooling="mean", shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001)):
        self._describe(name, vocab_size, embed_dim, shared_with, padding_idx, initializer)
        self.pooling = pooling


class SparseFeature(_TableFeature):
    """One id per sample."""

    def __init__(self, name, vocab_size, embed_dim=None, shared_with=None, padding_idx=None,
                 initializer=RandomNormal(0, 0.0001)):
        self._describe(name, vocab_size, embed_dim, shared_with, padding_idx, initializer)


class DenseFeature(object):
    """A numeric value passed through as is (width 1)."""

    def __init__(self, name):
        self.name, self.embed_dim = name, 1

    def __repr__(self):
        return "<DenseFeature %s>" % self.name
