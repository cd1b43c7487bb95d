"""Retrieval evaluation of two-tower models on the GPU.

Mirrors /root/reference/recbox/core/metrics.py: ``evaluate_metrics`` (:11-52), ``evaluate_block`` (:54-68) and the
metric classes (:71-190), with the same names, arguments and results.  The reference searches with
``faiss.IndexFlatIP`` (exact inner-product search, utils/ann/faiss.py:3-15) on the host and scores one user at a
time in Python; here

  scores = U I^T          rbx_linear_fwd (fp32 MFMA GEMM, a chunk of users x all items)
  top 500 per user        rbx_topk
  scores += -1e9 * mask   rbx_penalize_members (items the user clicked in the training data)
  argsort(-scores)[:k]    rbx_topk on the 500 survivors
  item in true_items      rbx_membership

and the metrics are closed-form reductions over the [users, k] hit flags (float64, like the reference's Python
floats).  Ties in the scores are broken towards the lower item index (the reference leaves them to faiss/numpy).
"""
import logging

import numpy as np
import torch

from .. import ops

__all__ = ["evaluate_metrics", "evaluate_block", "Recall", "nRecall", "Precision", "F1", "DCG", "NDCG", "MRR", "HitRate",
           "MAP", "build_csr"]

_SEARCH_TOPK = 500          # "set to topk=500 here since the retrieval results may contain clicked items" (metrics.py:55)


class _Metric(object):
    """Host form ``metric(topk_items, true_items)`` (the reference's per-user call) + the vectorised device form
    ``metric.batch(hits[:, :k], n_true)`` used by ``evaluate_block``."""

    def __init__(self, k=1):
        self.topk = k

    def __call__(self, topk_items, true_items):
        topk = list(topk_items)[:self.topk]
        true_set = set(true_items)
        hits = torch.tensor([[item in true_set for item in topk] + [False] * (self.topk - len(topk))])
        n_true = torch.tensor([float(len(true_items))], dtype=torch.float64)
        return float(self.batch(hits, n_true)[0])


class Recall(_Metric):
    def batch(self, hits, n_true):
        return hits[:, :self.topk].sum(1).double() / (n_true + 1e-12)


class nRecall(_Metric):
    def batch(self, hits, n_true):
        return hits[:, :self.topk].sum(1).double() / torch.clamp(n_true + 1e-12, max=float(self.topk))


class Precision(_Metric):
    def batch(self, hits, n_true):
        return hits[:, :self.topk].sum(1).double() / (self.topk + 1e-12)


class F1(_Metric):
    def batch(self, hits, n_true):
        p = Precision(self.topk).batch(hits, n_true)
        r = Recall(self.topk).batch(hits, n_true)
        return 2 * p * r / (p + r + 1e-12)


def _discount(k, device):
    return 1.0 / torch.log(2.0 + torch.arange(k, dtype=torch.float64, device=device))


class DCG(_Metric):
    def batch(self, hits, n_true):
        h = hits[:, :self.topk].double()
        return (h * _discount(h.shape[1], h.device)).sum(1)


class NDCG(_Metric):
    def batch(self, hits, n_true):
        dcg = DCG(self.topk).batch(hits, n_true)
        # idcg = DCG(true_items[:k], true_items): every one of the first min(k, len(true_items)) positions hits
        ideal = torch.cumsum(_discount(self.topk, hits.device), 0)
        n = torch.clamp(n_true, max=float(self.topk)).long()
        idcg = torch.where(n > 0, ideal[torch.clamp(n - 1, min=0)], torch.zeros_like(dcg))
        return dcg / (idcg + 1e-12)


class MRR(_Metric):
    def batch(self, hits, n_true):
        h = hits[:, :self.topk].double()
        return (h / (1.0 + torch.arange(h.shape[1], dtype=torch.float64, device=h.device))).sum(1)


class HitRate(_Metric):
    def batch(self, hits, n_true):
        return hits[:, :self.topk].any(1).double()


class MAP(_Metric):
    def batch(self, hits, n_true):
        h = hits[:, :self.topk].double()
        pos = torch.cumsum(h, 1)
        precision = (h * pos / (1.0 + torch.arange(h.shape[1], dtype=torch.float64, device=h.device))).sum(1)
        return precision / (pos[:, -1] + 1e-12)


def build_csr(user2items, n_queries, device):
    """dict query -> list of items  ->  (offsets [n_queries + 1], sorted distinct items, list lengths) on ``device``."""
    lens = np.zeros(n_queries, dtype=np.int64)
    chunks, counts = [], np.zeros(n_queries, dtype=np.int64)
    for q in range(n_queries):
        items = user2items.get(q, ()) if hasattr(user2items, "get") else user2items[q]
        lens[q] = len(items)
        u = np.unique(np.asarray(items, dtype=np.int64)) if len(items) else np.zeros(0, dtype=np.int64)
        counts[q] = u.size
        chunks.append(u)
    offsets = np.zeros(n_queries + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    items = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
    if items.size == 0:
        items = np.zeros(1, dtype=np.int64)
    return (torch.from_numpy(offsets).to(device), torch.from_numpy(items).to(device),
            torch.from_numpy(lens).to(device).double())


def evaluate_block(user_embs, item_embs, query_indices, train_csr, valid_csr, metric_funcs, max_topk):
    """Device form of metrics.py:54-68 for one chunk of users: returns (results [users, n_metrics] float64,
    topk_items [users, max_topk] int64)."""
    scores = ops.linear(user_embs, item_embs)                        # [users, n_items] = U I^T
    k_search = min(_SEARCH_TOPK, item_embs.shape[0])
    vals, idx = ops.topk(scores, k_search)
    ops.penalize_members_(vals, idx, query_indices, train_csr[0], train_csr[1], -1e9)
    k_out = min(max_topk, k_search)
    _, order = ops.topk(vals, k_out)                                 # argsort(-scores)[:, :max_topk]
    topk_items = torch.gather(idx, 1, order)
    hits = ops.membership(topk_items.contiguous(), query_indices, valid_csr[0], valid_csr[1])
    if k_out < max_topk:
        hits = torch.cat([hits, hits.new_zeros((hits.shape[0], max_topk - k_out))], dim=1)
    n_true = valid_csr[2][query_indices]
    results = torch.stack([f.batch(hits, n_true) for f in metric_funcs], dim=1)
    return results, topk_items


def evaluate_metrics(user_embs, item_embs, train_user2items, valid_user2items, query_indices, metrics, num_workers=1,
                     device="cuda"):
    """Same signature and return value as the reference (dict metric-string -> average over users);
    ``num_workers`` is accepted and ignored (one GPU does the work of the process pool)."""
    logging.info("Evaluating metrics for {} users.".format(len(user_embs)))
    metric_funcs = []
    max_topk = 0
    for metric in metrics:
        try:
            metric_funcs.append(eval(metric))
            max_topk = max(max_topk, int(metric.split("k=")[-1].strip(")")))
        except Exception:
            raise NotImplementedError('metrics={} not implemented.'.format(metric))
    dev = torch.device(device)
    U = torch.as_tensor(np.asarray(user_embs), dtype=torch.float32).to(dev).contiguous()      # .astype("float32") in faiss.py
    V = torch.as_tensor(np.asarray(item_embs), dtype=torch.float32).to(dev).contiguous()
    q = torch.as_tensor(np.asarray(query_indices), dtype=torch.int64).to(dev).contiguous()
    n_q = int(q.max()) + 1 if q.numel() else 0
    train_csr = build_csr(train_user2items, n_q, dev)
    valid_csr = build_csr(valid_user2items, n_q, dev)
    chunk = max(1, min(1000, (1 << 28) // max(V.shape[0], 1)))       # the reference scores 1000 users per block
    parts = []
    for i in range(0, U.shape[0], chunk):
        res, _ = evaluate_block(U[i:i + chunk], V, q[i:i + chunk], train_csr, valid_csr, metric_funcs, max_topk)
        parts.append(res)
    average_result = torch.cat(parts, 0).mean(0).tolist() if parts else [float("nan")] * len(metrics)
    return_dict = dict(zip(metrics, average_result))
    logging.info('[Metrics] ' + ' - '.join('{}: {:.6f}'.format(k, v) for k, v in zip(metrics, average_result)))
    return return_dict
