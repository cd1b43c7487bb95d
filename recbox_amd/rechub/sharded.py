"""Multi-GPU forms of the rechub model mirrors (BASELINE.json configs 3 and 4): one process per GPU,
``torch.distributed`` backend "nccl" (= RCCL over xGMI), every process trains on its own slice of the global batch.

The reference offers ``nn.DataParallel`` only for these models (third_party/rechub/trainers/ctr_trainer.py:41-43,
match_trainer.py:46): every table replicated, full dense gradients gathered on GPU 0 -- a 10 M x 128 table does not
scale that way.  Here

* tables with ``vocab_size >= shard_min_vocab`` live in a ``ShardedStore`` (recbox_amd/sharded.py): ``owner = id % W``,
  ONE all-to-all each way per step, mean/sum-pooled sequences reduced at the owner;
* everything else -- small tables, MLP towers, BatchNorm, the FM / LR heads -- is replicated and data-parallel: the
  dense gradients are all-reduced as ONE flat buffer, issued the moment the last tower / head parameter has received its
  gradient (counted by per-parameter hooks) so that it overlaps the gradient exchange and the scatter-adds;
* BatchNorm statistics are PER REPLICA, as under the reference's ``nn.DataParallel`` (each replica normalises its own
  slice; ctr_trainer.py:43) -- not synchronised as RecBole's DDP path does (recbole/trainer/trainer.py:61).  Running
  statistics therefore differ slightly between ranks; checkpoint rank 0's, as DataParallel does.

``ShardedEmbeddingLayer`` keeps the interface of ``EmbeddingLayer`` (``forward(x, features, squeeze_dim)``), so the
model classes are the single-GPU ones with that layer swapped in.
"""
import torch
import torch.nn as nn

from .. import _embed_host as host
from .. import comm
from ..sharded import ShardCall, ShardedStore
from .basic.features import DenseFeature, SequenceFeature, SparseFeature
from .basic.layers import FM, LR, MLP, EmbeddingLayer
from .models.matching import YoutubeDNN
from .models.ranking import DeepFM


class ShardedEmbeddingLayer(EmbeddingLayer):
    """``EmbeddingLayer`` whose large tables are row-sharded over the process group.

    ``embed_dict`` holds the replicated tables (real ``nn.Embedding``, same names as on one GPU); ``stores[str(dim)]``
    (``store`` = the first) hold this rank's rows of the sharded ones, one store per embedding dimension.  A layer call is one local gather launch for the replicated features plus one
    exchange for the sharded ones, both writing into the same ``[B, width]`` block the single-GPU layer produces."""

    def __init__(self, features, shard_min_vocab=100000, capacity_factor=1.25, process_group=None, local_ops=None):
        nn.Module.__init__(self)
        self.features = features
        self.embed_dict = nn.ModuleDict()
        self.n_dense = 0
        self._plans = {}
        self.group = process_group
        by_name = dict((f.name, f) for f in features if not isinstance(f, DenseFeature))
        self.sharded_tables = []                      # names of the sharded tables, in store order
        for fea in features:
            if isinstance(fea, DenseFeature):
                self.n_dense += 1
                continue
            if fea.shared_with is not None or fea.name in self.embed_dict or fea.name in self.sharded_tables:
                continue
            if fea.vocab_size >= shard_min_vocab:
                self.sharded_tables.append(fea.name)                 # (never materialised as a full table)
            else:
                self.embed_dict[fea.name] = fea.get_embedding_layer()
        # one ShardedStore per embedding dimension (a store keeps the rows of all its tables back to back in one
        # [rows, D] weight).  ``store`` is the first of them -- the only one in the usual case of one dimension, and the
        # name its weight has in checkpoints ("embedding.store.weight"); further dimensions register as ``store_<dim>``.
        self.store = None
        self.stores = {}                              # str(dim) -> store (a plain dict: the modules are registered above)
        self._store_of, self._index_in = {}, {}
        for k, dim in enumerate(sorted(set(by_name[n].embed_dim for n in self.sharded_tables))):
            names = [n for n in self.sharded_tables if by_name[n].embed_dim == dim]
            st = ShardedStore([by_name[n].vocab_size for n in names], dim, capacity_factor=capacity_factor,
                              process_group=process_group, local_ops=local_ops if k == 0 else None)
            if k == 0:
                self.store = st
            else:
                setattr(self, "store_%d" % dim, st)
            self.stores[str(dim)] = st
            for t, n in enumerate(names):
                self._store_of[n], self._index_in[n] = str(dim), t

    def table_of(self, fea):
        return fea.name if fea.shared_with is None else fea.shared_with

    def forward(self, x, features, squeeze_dim=False):
        key = (tuple(id(f) for f in features), squeeze_dim,
               tuple(x[f.name].shape[1] if isinstance(f, SequenceFeature) else 0 for f in features))
        cached = self._plans.get(key)
        if cached is None:
            cached = self._plans[key] = self._build(x, features, squeeze_dim)
        local, calls, n_sparse, width, seq_len, dim = cached
        B = x[features[0].name].shape[0]
        block = None
        if local is not None:
            block = local.run([x[lk.name] for lk in local.lookups], pad_rows=squeeze_dim)
        # one exchange per (store, pooled sequence): the single-row lookups of a store ride with its first call
        for key, call, order in calls:
            rows = []
            for name, col in order["rows"]:
                rows.append(x[name] if col is None else x[name][:, col])
            pool = x[order["pool"]] if order["pool"] is not None else None
            block = self.stores[key].lookup(call, width, rows, pool, block)
        if squeeze_dim:
            return block
        if seq_len is not None:
            return block.view(B, n_sparse, seq_len, dim)
        return block.view(B, n_sparse, dim)

    def _build(self, x, features, squeeze_dim):
        """Lay the call out exactly as EmbeddingLayer does (layers.py:66-116: sparse / sequence slots in feature order,
        dense values behind them when squeezed), then split the slots into the local plan and the exchange."""
        from .._lib import FIELD_CATEGORICAL, FIELD_DENSE, POOL_CONCAT, POOL_MEAN_ID, POOL_SUM_ID
        sparse, dense_l = [], []
        for fea in features:
            if isinstance(fea, DenseFeature):
                dense_l.append(fea)
            elif isinstance(fea, (SparseFeature, SequenceFeature)):
                sparse.append(fea)
            else:
                raise ValueError("unknown feature class %s" % type(fea).__name__)
        if squeeze_dim:
            if not sparse and not dense_l:
                raise ValueError("The input features can note be empty")
        elif not sparse:
            raise ValueError("If keep the original shape:[batch_size, num_features, embed_dim], expected %s in feature "
                             "list, got %s" % ("SparseFeatures", features))
        lookups, offsets, off = [], [], 0
        per_store = {}                 # store key -> {"rows": [(t, off)], "row_src": [...], "pools": [(pool tuple, feature name)]}
        dims, seq_lens, concat = set(), set(), []
        for fea in sparse:
            tname = self.table_of(fea)
            dim = fea.embed_dim
            dims.add(dim)
            is_seq = isinstance(fea, SequenceFeature)
            if is_seq and fea.pooling not in ("sum", "mean", "concat"):
                raise ValueError("Sequence pooling method supports only pooling in %s, got %s." % (["sum", "mean"], fea.pooling))
            L = x[fea.name].shape[1] if is_seq else 1
            is_concat = is_seq and fea.pooling == "concat"
            concat.append(is_concat)
            if is_concat:
                seq_lens.add(L)
            if tname in self.sharded_tables:
                key, t = self._store_of[tname], self._index_in[tname]
                ent = per_store.setdefault(key, {"rows": [], "row_src": [], "pools": []})
                if not is_seq:
                    ent["rows"].append((t, off))
                    ent["row_src"].append((fea.name, None))
                elif is_concat:
                    for c in range(L):                       # every column is a single-row lookup
                        ent["rows"].append((t, off + c * dim))
                        ent["row_src"].append((fea.name, c))
                else:
                    mask_id = fea.padding_idx if fea.padding_idx is not None else -1      # InputMask: id != -1
                    ent["pools"].append(((t, off, L, fea.pooling, mask_id, 1e-16 if fea.pooling == "mean" else 0.0), fea.name))
            else:
                table = self.embed_dict[tname]
                if not is_seq:
                    lookups.append(host.Lookup(fea.name, FIELD_CATEGORICAL, table, dim))
                else:
                    kind = {"sum": POOL_SUM_ID, "mean": POOL_MEAN_ID, "concat": POOL_CONCAT}[fea.pooling]
                    mask_id = fea.padding_idx if fea.padding_idx is not None else -1
                    lookups.append(host.Lookup(fea.name, FIELD_CATEGORICAL, table, dim, pool=kind, seq_len=L,
                                               mask_id=mask_id, eps=1e-16))
                offsets.append(off)
            off += dim * (L if is_concat else 1)
        if squeeze_dim:
            for fea in dense_l:
                lookups.append(host.Lookup(fea.name, FIELD_DENSE, None, 1))
                offsets.append(off)
                off += 1
        width = off
        seq_len, dim = None, None
        if not squeeze_dim:
            if len(dims) != 1:
                raise RuntimeError("Sizes of tensors must match except in dimension 1 (embed_dim differs across features)")
            dim = dims.pop()
            if any(concat):
                if not all(concat) or len(seq_lens) != 1:
                    raise RuntimeError("Tensors must have same number of dimensions (mixing pooling='concat' "
                                       "with pooled/sparse features)")
                seq_len = seq_lens.pop()
        local = host.Plan(lookups, offsets=offsets, width=width) if lookups else None
        # The wire format of one exchange carries any number of single-row lookups and at most ONE pooled sequence
        # (include/recbox_hip.h, rbx_shard_geom_t): a layer call with several pooled sequences over sharded tables (the
        # reference's EmbeddingLayer takes any mix, third_party/rechub/basic/layers.py:66-116) becomes several exchanges
        # into the same output block -- the first one of a store also carries that store's single-row lookups.
        calls = []
        for key in sorted(per_store):
            ent = per_store[key]
            pools = ent["pools"] or [(None, None)]
            for k, (pool, pool_src) in enumerate(pools):
                rows = ent["rows"] if k == 0 else []
                src = ent["row_src"] if k == 0 else []
                if rows or pool is not None:
                    calls.append((key, ShardCall(rows, pool), {"rows": src, "pool": pool_src}))
        return local, calls, len(sparse), width, seq_len, dim


class DenseGradSync(object):
    """Data-parallel gradients of the replicated parameters.

    ``early``: towers and heads.  Their gradients are packed into one flat buffer and all-reduced asynchronously THE MOMENT
    THE LAST ONE OF THEM HAS ITS GRADIENT OF THIS BACKWARD (a post-accumulate hook per parameter counts arrivals), which is
    normally while the embedding backward -- gradient exchange, scatter-adds -- is still under way, so the all-reduce runs
    beside it.  Nothing is reduced on a guess: a gradient left over from an earlier step (``zero_grad(set_to_none=False)``)
    of a parameter whose backward has not run yet never enters the buffer.  ``late``: the replicated tables, whose
    gradients come out of the embedding backward itself: one more flat all-reduce in ``finish()``.  ``finish()`` also
    un-flattens and picks up whatever did not go early (a head that took no part in this loss); call it after every
    ``loss.backward()`` -- a second backward before it is refused (the first one's gradients are already summed over the
    ranks in place).

    Gradients that survive a step (``zero_grad(set_to_none=False)``, accumulation over several backward + ``finish()``
    rounds without ``zero_grad``): autograd adds this backward's LOCAL gradient to what ``p.grad`` holds -- the sum over
    the ranks of the earlier rounds, identical on every rank -- and the all-reduce would count that W times.  The first
    gradient of a backward therefore divides every surviving tower / head gradient by W (one fused launch; exact for the
    power-of-two worlds of a node, one rounding otherwise), so that sum_r (prior / W + g_r) = prior + sum_r g_r.  The
    replicated tables (``late``) are divided in the same launch: their hooks fire the division too, so it runs before any
    gradient of this backward is added whichever parameter comes first."""

    def __init__(self, early, late, group=None):
        self.early = [p for p in early if p.requires_grad]
        self.late = [p for p in late if p.requires_grad]
        self.group = group
        self.world = comm.world(group)[1]
        self.active = comm.multi(group)         # (also in a world of one under comm.force_world_of_one)
        self._pending = None
        self._arrived = set()
        self._hooks = []
        self.bucket = None
        self._views = []
        self._prescaled = False
        if self.active:
            for p in self.early:
                self._hooks.append(p.register_hook(self._before_grad))
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                p._rbx_joins_beside = True      # (_start joins the side stream before it reads gradients: ops._hooked)
            for p in self.late:
                # a table gradient that survived the last round is divided like the towers' before this backward adds to it
                self._hooks.append(p.register_hook(self._before_grad))
            self._make_bucket()

    def _make_bucket(self):
        """ONE flat buffer for the tower / head gradients; every parameter's slot is registered with the ops as the place
        its gradient is to be WRITTEN (ops._grad_views), so that the all-reduce runs over the buffer in place.  A gradient
        that arrives elsewhere (an op without the hook-up, a second use of the parameter) is copied in and back."""
        from .. import ops
        ps = [p for p in self.early if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()]
        if not ps:
            return
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]                     # 16-byte aligned slots
        self.bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=ps[0].device)
        o = 0
        for p, n in zip(ps, sizes):
            view = self.bucket[o:o + p.numel()].view(p.shape)
            ops._grad_views[p.data_ptr()] = ops._GradView(view, p)
            self._views.append((p, view))
            o += n

    def _prescale(self):
        if self._prescaled:
            return
        self._prescaled = True
        if self.world > 1:
            left = [p.grad for p in self.early + self.late if p.grad is not None]
            if left:
                with torch.no_grad():
                    torch._foreach_div_(left, float(self.world))

    def _before_grad(self, grad):
        """Tensor hook of every tower / head parameter: runs when its gradient of this backward exists, before autograd
        accumulates it.  The first one of a backward pre-divides the surviving gradients (class docstring)."""
        if self._pending is None:
            self._prescale()
        return None

    def close(self):
        """Remove the hooks and the registered gradient destinations (a model that moves to another device re-installs)."""
        from .. import ops
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.early:
            if getattr(p, "_rbx_joins_beside", False):
                del p._rbx_joins_beside
        for p, _ in self._views:
            ops._grad_views.pop(p.data_ptr(), None)
        self._views, self.bucket = [], None

    def _on_grad(self, p):
        if id(p) in self._arrived or self._pending is not None:
            raise RuntimeError("DenseGradSync: a second backward reached a replicated parameter before sync_grads() "
                               "finished the first one (call model.sync_grads() after every loss.backward())")
        self._arrived.add(id(p))
        if len(self._arrived) == len(self.early):
            self._pending = self._start(self.early)

    def _start(self, params):
        """Asynchronous all-reduce of the gradients of ``params`` (zeros for a missing one); returns what finish() needs."""
        from .. import ops
        if params and params[0].is_cuda:
            ops.join_beside()                       # weight gradients still being written on the side stream (dw_beside_lookup)
        copied, loose = [], []
        in_bucket = dict((id(p), v) for p, v in self._views)
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            v = in_bucket.get(id(p))
            if v is None:
                loose.append(p.grad)
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)                     # did not land in its slot: carried by the bucket, copied back afterwards
                copied.append((p, v))
        handles = []
        if self.bucket is not None and any(id(p) in in_bucket for p in params):
            handles.append(comm.all_reduce_sum_(self.bucket, self.group, async_op=True))
        if loose:
            handles.append(comm.all_reduce_coalesced_(loose, self.group, async_op=True))
        return handles, copied

    def start(self):
        """(kept for callers of the earlier interface: the all-reduce now starts from the parameters' own hooks)"""
        return

    def abandon(self):
        """Drop the state of a backward that will not be finished (a refused second backward, an exception in the step):
        waits for an all-reduce already under way, forgets the arrivals."""
        if self._pending is not None:
            for h in self._pending[0]:
                h.wait()
            self._pending = None
        self._arrived.clear()
        self._prescaled = False

    def finish(self):
        if not self.active:
            return
        # Every rank issues the SAME collectives per step -- the early layout, then the late layout -- whatever arrived
        # where: a rank on which some tower parameter received no gradient (it then never fired from the hooks) reduces the
        # early layout here, zeros standing in for what is missing.
        self._prescale()                        # (no gradient at all arrived in this backward: nothing divided them yet)
        if self._pending is None and self.early:
            self._pending = self._start(self.early)
        if self._pending is not None:
            handles, copied = self._pending
            for h in handles:
                h.wait()
            for p, v in copied:
                p.grad.copy_(v)
            self._pending = None
        self._arrived.clear()
        self._prescaled = False
        from .. import ops
        for p, _ in self._views:
            ent = ops._grad_views.get(p.data_ptr())
            if ent is not None:
                ent.taken = False
        if self.late:
            # the replicated tables: their dense gradients are views of the lookup's one flat buffer -- reduced as a span
            for p in self.late:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            comm.all_reduce_coalesced_([p.grad for p in self.late], self.group).wait()


class _ShardedModelMixin(object):
    def _apply(self, fn, *args, **kwargs):
        # .cuda() / .to(): the parameters' storage changes -- the gradient bucket and its registered views are per device
        # (built on the CPU there is no bucket at all: the tower gradients would then be packed and copied back every step)
        out = super(_ShardedModelMixin, self)._apply(fn, *args, **kwargs)
        if getattr(self, "grad_sync", None) is not None:
            self._install_sync()
        return out

    def _install_sync(self):
        old = getattr(self, "grad_sync", None)
        if old is not None:
            old.close()
        emb = self.embedding
        tables = list(emb.embed_dict.parameters())
        skip = set(id(p) for p in tables)
        for st in emb.stores.values():
            skip.add(id(st.weight))
        towers = [p for p in self.parameters() if id(p) not in skip]
        self.grad_sync = DenseGradSync(towers, tables, emb.group)

    @property
    def world_size(self):
        return comm.world(self.embedding.group)[1]

    def sync_grads(self):
        """After ``(loss / world_size).backward()``: finish the all-reduces of the replicated gradients."""
        self.grad_sync.finish()
        for st in self.embedding.stores.values():           # step boundary for the owners' persistent gradient buffers
            end = getattr(st.local_ops, "end_step", None)
            if end is not None:
                end()

    def replicated_parameters(self):
        skip = set(id(st.weight) for st in self.embedding.stores.values())
        return [p for p in self.parameters() if id(p) not in skip]

    def raise_if_overflowed(self):
        """Collective (call it from every rank, once per step or every N steps): RuntimeError on every rank when any
        rank's exchange ran out of slots since the last call (recbox_amd.sharded.raise_if_overflowed)."""
        for st in self.embedding.stores.values():
            st.raise_if_overflowed()


class ShardedYoutubeDNN(_ShardedModelMixin, YoutubeDNN):
    """``YoutubeDNN`` (third_party/rechub/models/matching/youtube_dnn.py:14-71) with the item table row-sharded:
    the history is mean-pooled AT THE OWNERS (one partial sum per (sample, shard) comes back), the positive and the
    negative items travel as rows; user tower replicated, gradients all-reduced.  BASELINE.json cfg 3."""

    def __init__(self, user_features, item_features, neg_item_feature, user_params, temperature=1.0,
                 shard_min_vocab=100000, capacity_factor=1.25, process_group=None, local_ops=None):
        torch.nn.Module.__init__(self)
        self.user_features = user_features
        self.item_features = item_features
        self.neg_item_feature = neg_item_feature
        self.temperature = temperature
        self.user_dims = sum([fea.embed_dim for fea in user_features])
        self.embedding = ShardedEmbeddingLayer(user_features + item_features, shard_min_vocab=shard_min_vocab,
                                               capacity_factor=capacity_factor, process_group=process_group,
                                               local_ops=local_ops)
        self.user_mlp = MLP(self.user_dims, output_layer=False, **user_params)
        self.mode = None
        self._install_sync()


class ShardedDeepFM(_ShardedModelMixin, DeepFM):
    """``DeepFM`` (third_party/rechub/models/ranking/deepfm.py:14-42) data-parallel over the ranks, its large tables
    row-sharded.  BASELINE.json cfg 4 ("1 vs 8 GPU data-parallel all-reduce")."""

    def __init__(self, deep_features, fm_features, mlp_params, shard_min_vocab=100000, capacity_factor=1.25,
                 process_group=None, local_ops=None):
        torch.nn.Module.__init__(self)
        self.deep_features = deep_features
        self.fm_features = fm_features
        self.deep_dims = sum([fea.embed_dim for fea in deep_features])
        self.fm_dims = sum([fea.embed_dim for fea in fm_features])
        self.linear = LR(self.fm_dims)
        self.fm = FM(reduce_sum=True)
        self.embedding = ShardedEmbeddingLayer(deep_features + fm_features, shard_min_vocab=shard_min_vocab,
                                               capacity_factor=capacity_factor, process_group=process_group,
                                               local_ops=local_ops)
        self.mlp = MLP(self.deep_dims, **mlp_params)
        self._install_sync()
