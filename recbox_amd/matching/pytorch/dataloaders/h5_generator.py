"""GPU-resident training loader of the two-tower models: negative sampling, item-corpus
feature gather and batch collation without a host round trip.

Mirrors /root/reference/recbox/matching/pytorch/dataloaders/h5_generator.py:
  sampling_block :61-84, TrainGenerator.negative_sampling :144-181  -> ``negative_sampling`` (rbx_negsample)
  TrainDataset.__getitem__ :23-28 + collate_fn :49-58               -> ``collate`` (rbx_gather_rows)
  collate_fn_unique :38-47                                          -> ``collate_unique``
  TrainGenerator :97-142 (constructor arguments, ``__iter__`` re-samples every epoch, ``__len__``)

Differences that are deliberate:
  * the data live in device tensors: ``data`` / ``item_corpus`` are dicts of arrays (what ``load_h5`` returns in
    the reference; h5py itself is out of scope), moved to the GPU once;
  * the random stream is Philox4x32-10 keyed by (seed, epoch) instead of numpy's MT19937 -- same distribution
    (uniform with replacement; ``ignore_pos_items`` = uniform over the items the query never interacted with),
    reproducible for a given seed, and drawn in one kernel instead of ``sampling_num_process`` worker processes;
  * batches are produced by index gathers on the device (no DataLoader workers, no H2D copy of
    ``B * (1 + num_negs)`` item rows per step).
"""
import numpy as np
import torch

from .... import ops

__all__ = ["TrainGenerator", "negative_sampling", "collate", "collate_unique"]


def _to_device(array_dict, device):
    out = {}
    for k, v in array_dict.items():
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(device).contiguous()
    return out


def build_exclusion(query_indexes, corpus_indexes):
    """CSR of ``get_user2items_dict`` (h5_generator.py:35-40): for every query index the sorted, distinct items
    it interacted with.  Returns (offsets [n_queries + 1], items) as int64 CPU tensors."""
    q = torch.as_tensor(query_indexes).long().reshape(-1).cpu()
    c = torch.as_tensor(corpus_indexes).long().reshape(-1).cpu()
    n_q = int(q.max()) + 1 if q.numel() else 0
    n_c = int(c.max()) + 1 if c.numel() else 1
    pairs = torch.unique(q * n_c + c)                      # sorted by (query, item), duplicates dropped
    uq, items = pairs // n_c, pairs % n_c
    offsets = torch.zeros(n_q + 1, dtype=torch.long)
    offsets[1:] = torch.cumsum(torch.bincount(uq, minlength=n_q), 0)
    return offsets, items


def negative_sampling(num_items, pos_item_indexes, num_negs, seed, epoch=0, query_indexes=None, exclusion=None):
    """``all_item_indexes`` [N, 1 + num_negs] = hstack([pos, negs]) (h5_generator.py:178-180), on the device of
    ``pos_item_indexes``.  exclusion = (offsets, items) from ``build_exclusion`` enables ``ignore_pos_items``."""
    pos = pos_item_indexes.long().contiguous()
    n = pos.shape[0]
    kw = {}
    if exclusion is not None:
        kw = dict(query=query_indexes.long().contiguous(), excl_offsets=exclusion[0], excl_items=exclusion[1])
    # a fresh, non-overlapping block of the element counter per epoch
    return ops.negsample(num_items, n, num_negs, seed, offset=int(epoch) * n * max(num_negs, 1), pos=pos, **kw)


def collate(user_data, item_corpus, labels, item_indexes, batch_index):
    """One batch as the reference's ``collate_fn`` builds it: (user_dict, item_dict, labels, None).
    user_dict[k] = data[k][batch]; item_dict[k] = item_corpus[k][item_indexes[batch]].flatten(end_dim=1);
    labels = [label, 0, ..., 0] per row."""
    names_u, names_i = list(user_data), list(item_corpus)
    cols = ops.gather_rows([user_data[k] for k in names_u] + [labels, item_indexes], batch_index)
    user_dict = dict(zip(names_u, cols[:len(names_u)]))
    lab, idx = cols[len(names_u)], cols[len(names_u) + 1]
    item_cols = ops.gather_rows([item_corpus[k] for k in names_i], idx.reshape(-1))
    item_dict = dict(zip(names_i, item_cols))
    num_negs = idx.shape[1] - 1
    out_labels = torch.cat([lab.view(-1, 1).float(), torch.zeros((lab.shape[0], num_negs), device=lab.device)], dim=1)
    return user_dict, item_dict, out_labels, None, idx


def collate_unique(user_data, item_corpus, labels, item_indexes, batch_index):
    """``collate_fn_unique`` (h5_generator.py:38-47): every distinct item of the batch is gathered once;
    returns (user_dict, item_dict over the sorted unique items, labels, inverse_indexes)."""
    names_u, names_i = list(user_data), list(item_corpus)
    cols = ops.gather_rows([user_data[k] for k in names_u] + [labels, item_indexes], batch_index)
    user_dict = dict(zip(names_u, cols[:len(names_u)]))
    lab, idx = cols[len(names_u)], cols[len(names_u) + 1]
    unique, inverse = torch.unique(idx.flatten(), return_inverse=True, sorted=True)
    item_dict = dict(zip(names_i, ops.gather_rows([item_corpus[k] for k in names_i], unique)))
    num_negs = idx.shape[1] - 1
    out_labels = torch.cat([lab.view(-1, 1).float(), torch.zeros((lab.shape[0], num_negs), device=lab.device)], dim=1)
    # the reference returns the FLIPPED inverse indexes (it reverses them to emulate np.unique's return_index)
    return user_dict, item_dict, out_labels, inverse.flip([0]), idx


class TrainGenerator(object):
    """Same role and constructor arguments as the reference's ``TrainGenerator`` (h5_generator.py:97-142), with
    in-memory dicts in place of the h5 paths.  Iterating re-samples the negatives (one kernel) and yields
    ``(user_dict, item_dict, labels, inverse_indexes_or_None)`` batches of device tensors."""

    def __init__(self, feature_map, data, item_corpus, batch_size=32, shuffle=True, num_workers=1, num_negs=0,
                 compress_duplicate_items=False, device="cuda", seed=2019, **kwargs):
        self.device = torch.device(device)
        data = _to_device(data, self.device)
        self.item_corpus = _to_device(item_corpus, self.device)
        self.num_items = next(iter(self.item_corpus.values())).shape[0]
        self.labels = data[feature_map.label_name]
        self.pos_item_indexes = data[feature_map.corpus_index].long()
        self.query_indexes = data[feature_map.query_index].long()
        # "delete some columns to speed up batch generator" (h5_generator.py:111-114)
        self.user_data = dict((k, v) for k, v in data.items()
                              if k not in (feature_map.query_index, feature_map.corpus_index, feature_map.label_name))
        self.num_samples = self.labels.shape[0]
        self.batch_size, self.shuffle, self.num_negs = batch_size, shuffle, num_negs
        self.num_batches = int(np.ceil(self.num_samples * 1.0 / batch_size))
        self.ignore_pos_items = kwargs.get("ignore_pos_items", False)
        self.compress_duplicate_items = compress_duplicate_items
        self.seed, self.epoch = seed, 0
        self.exclusion = None
        if self.ignore_pos_items:
            off, items = build_exclusion(self.query_indexes, self.pos_item_indexes)
            self.exclusion = (off.to(self.device), items.to(self.device))
        self.all_item_indexes = self.pos_item_indexes.view(-1, 1)
        self._perm_gen = torch.Generator(device="cpu").manual_seed(seed)

    def __len__(self):
        return self.num_batches

    def negative_sampling(self):
        if self.num_negs > 0:
            self.all_item_indexes = negative_sampling(self.num_items, self.pos_item_indexes, self.num_negs, self.seed,
                                                      self.epoch, self.query_indexes, self.exclusion)
        self.epoch += 1

    def __iter__(self):
        self.negative_sampling()
        n = self.num_samples
        order = (torch.randperm(n, generator=self._perm_gen) if self.shuffle else torch.arange(n)).to(self.device)
        fn = collate_unique if self.compress_duplicate_items else collate
        for b in range(self.num_batches):
            batch_index = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield fn(self.user_data, self.item_corpus, self.labels, self.all_item_indexes, batch_index)[:4]
