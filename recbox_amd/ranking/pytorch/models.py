"""Model bodies in the FuxiCTR convention the reference's ``RankingModel`` harness
expects from its subclasses (``forward(inputs) -> {"y_pred": sigmoid(logit)}``,
/root/reference/recbox/ranking/pytorch/models/ranking_model.py:66-70,191-197).
The reference tree ships the harness but no concrete FM/DeepFM class;
these are the bodies a user of the harness writes, built from the drop-in layers.
"""
import torch
from torch import nn

from ... import ops
from .layers import FactorizationMachine, FeatureEmbedding

__all__ = ["FM", "ShardedFM", "inputs_from_batch"]


def inputs_from_batch(feature_map, batch, feature_source=None):
    """``RankingModel.get_inputs`` for a batch that is ALREADY on the GPU (ranking_model.py:106-116): the
    flat ``[B, cols]`` tensor the reference's loaders deliver (float64 as soon as one column is float,
    h5_dataloader.py:36-47) is cut into per-feature COLUMN VIEWS -- no slice copies, no ``.long()`` casts: the
    kernels read strided float64/float32/int columns in place and cast in registers.  A narrower wire format
    (int32 ids, fp32 dense values) is therefore just a different dtype of ``batch``."""
    from collections import OrderedDict
    if feature_source and isinstance(feature_source, str):
        feature_source = [feature_source]
    X = OrderedDict()
    for name, spec in feature_map.features.items():
        if feature_source is not None and spec["source"] not in feature_source:
            continue
        if spec["type"] == "meta":
            continue
        col = feature_map.get_column_index(name)
        X[name] = batch[:, col[0]:col[-1] + 1] if isinstance(col, list) else batch[:, col]
    return X


class FM(nn.Module):
    """y = sigmoid(LR(X) + 0.5 * sum_d[(sum_f e)^2 - sum_f e^2])."""

    def __init__(self, feature_map, embedding_dim=10, fused=True, **kwargs):
        super(FM, self).__init__()
        self.feature_map = feature_map
        self.fused = fused          # False: compose the drop-in layers exactly as the reference does
        self.embedding_layer = FeatureEmbedding(feature_map, embedding_dim)
        self.fm = FactorizationMachine(feature_map)

    def logits(self, X, with_prob=False, presorted=None):
        emb = self.embedding_layer.embedding_layer
        lr = self.fm.lr_layer.embedding_layer.embedding_layer
        names, values, plan, posts = emb.plan_for(X)
        lnames, _, lplan, lposts = lr.plan_for(X)
        if self.fused and emb.fusable(plan, posts) and lr.fusable(lplan, lposts) and names == lnames:
            # gather + LR + interaction (+ the output sigmoid) in ONE kernel; [B, F, D] is never written (rbx_fm_fwd / rbx_fm_bwd)
            return ops.fm_fused(plan.plan, lplan.plan, values, [m.weight for m in plan.modules],
                                [m.weight for m in lplan.modules], self.fm.lr_layer.bias, with_prob=with_prob,
                                presorted=presorted)
        if presorted is not None:
            raise ValueError("FM: presorted ids belong to the fused path")
        if getattr(self, "_packed", False):
            raise RuntimeError("FM: the tables of this model were re-homed by pack_tables() (rows of %s floats): only the fused "
                               "forward reads that layout.  Use fused=True with the model's own feature set, or rebuild the "
                               "model and load_state_dict() the checkpoint (it holds plain contiguous tables)" % "row_floats")
        logit = self.fm(X, self.embedding_layer(X))
        return (logit, None) if with_prob else logit

    def presort(self, X, into=None):
        """The id sort of this model's backward for batch ``X``, on the current stream (ops.fm_presort): it needs the ids only,
        so a loop that has batch i + 1 at hand while step i runs sorts it then -- on a second stream, beside step i -- and
        passes the result to ``forward(X, presorted=...)`` one step later.  ``into``: re-use the workspace of an earlier result."""
        if torch.is_tensor(X):
            X = inputs_from_batch(self.feature_map, X)
        emb = self.embedding_layer.embedding_layer
        lr = self.fm.lr_layer.embedding_layer.embedding_layer
        names, values, plan, posts = emb.plan_for(X)
        lnames, _, lplan, lposts = lr.plan_for(X)
        if not (self.fused and emb.fusable(plan, posts) and lr.fusable(lplan, lposts) and names == lnames):
            raise ValueError("FM.presort: only the fused path sorts ahead of its step")
        return ops.fm_presort(plan.plan, lplan.plan, values, [m.weight for m in plan.modules],
                              [m.weight for m in lplan.modules], into=into)

    def forward(self, X, presorted=None):
        if torch.is_tensor(X):                      # the reference's harness hands over the flat batch tensor
            X = inputs_from_batch(self.feature_map, X)
        logit, prob = self.logits(X, with_prob=True, presorted=presorted)
        return {"y_pred": ops.sigmoid_output(logit, prob)}

    @torch.no_grad()
    def pack_tables(self, row_floats=None, min_vocab=0):
        """Re-home every (embedding table, LR table) pair of a categorical feature in ONE packed storage
        ``[vocab, row_floats]`` -- floats ``[0, D)`` the embedding row, float ``D`` the dim-1 LR weight, the rest padding;
        ``row_floats`` defaults to the next multiple of 32 floats (128-byte rows) -- so that the fused forward issues one
        128-byte request per lookup instead of a 64-byte row plus a 4-byte weight from two different lines.

        The holders stay the reference's modules and keys: ``embedding_layers.<f>`` is still an ``nn.Embedding`` whose
        ``weight`` is now the column view ``packed[:, :D]`` (a Parameter with row stride ``row_floats``), the LR holder's
        ``weight`` is ``packed[:, D:D+1]``; ``state_dict()`` / ``load_state_dict()`` / optimisers see the same names and
        shapes.  Gradients stay dense contiguous ``[vocab, D]`` / ``[vocab, 1]`` tensors.  Call it AFTER moving the model
        to its device (``.to()`` / ``.cuda()`` re-allocate every parameter on its own, which silently un-packs: the
        kernels then simply run on the separate tables again).  Tables shared by several features, pretrained / frozen
        tables and sequence features are left alone.  ``min_vocab``: only tables with at least that many rows -- a table
        that does not fit the caches pays one 128-byte line for the embedding row and ANOTHER for its 4-byte LR weight on
        every lookup, which packing halves; a small, cache-resident table only doubles its footprint.  Returns the number of
        packed pairs."""
        emb = self.embedding_layer.embedding_layer.embedding_layers
        lr = self.fm.lr_layer.embedding_layer.embedding_layer.embedding_layers
        seen, packed_pairs = {}, 0
        for name in emb:
            seen[id(emb[name])] = seen.get(id(emb[name]), 0) + 1
        for name, table in emb.items():
            if name not in lr or type(table) is not nn.Embedding or type(lr[name]) is not nn.Embedding:
                continue
            w, l = table.weight, lr[name].weight
            if seen[id(table)] != 1 or l.shape != (w.shape[0], 1) or w.requires_grad != l.requires_grad:
                continue
            V, D = w.shape
            if V < min_vocab:
                continue
            stride = row_floats or (D + 1 + 31) // 32 * 32
            if stride < D + 1 or stride % 4:
                raise ValueError("pack_tables: row_floats must be a multiple of 4 and >= embedding_dim + 1")
            packed = torch.zeros((V, stride), dtype=torch.float32, device=w.device)
            packed[:, :D].copy_(w)
            packed[:, D:D + 1].copy_(l)
            table.weight = nn.Parameter(packed[:, :D], requires_grad=w.requires_grad)
            lr[name].weight = nn.Parameter(packed[:, D:D + 1], requires_grad=l.requires_grad)
            packed_pairs += 1
        if packed_pairs and not getattr(self, "_packed_state_hook", None):
            # a packed parameter is a column view of the [V, row_floats] storage: torch.save of a view writes the WHOLE
            # storage (twice per pair).  Checkpoints keep the reference's format -- plain contiguous [V, D] / [V, 1] tensors
            # under the reference's keys -- whatever the in-memory layout is.
            def contiguous_state(module, state, prefix, local_metadata):
                for k, v in list(state.items()):
                    if torch.is_tensor(v) and not v.is_contiguous():
                        state[k] = v.contiguous()
                return state
            self._packed_state_hook = self._register_state_dict_hook(contiguous_state)
        self._packed = bool(packed_pairs) or getattr(self, "_packed", False)
        return packed_pairs


class ShardedFM(nn.Module):
    """FM over the GPUs of one node: one process per GPU, every process trains on its own
    slice of the global batch.

    * categorical tables with ``vocab_size >= shard_min_vocab`` are ROW-SHARDED (``ShardedTables``: embedding rows
      and LR weights of all such tables behind one all-to-all-v exchange each way over RCCL/xGMI);
    * the remaining (small) tables, numeric weights and the bias are replicated; ``sync_grads()`` all-reduces
      their dense gradients in one flat buffer.

    The fetched remote rows enter the fused FM kernel as ``extra`` rows, so the local compute stays the same
    single forward kernel + fused segmented backward as on one GPU."""

    def __init__(self, feature_map, embedding_dim=10, shard_min_vocab=100000, capacity_factor=None,
                 process_group=None, local_ops=None):
        super(ShardedFM, self).__init__()
        from collections import OrderedDict
        from ...sharded import ShardedTables
        from ... import comm
        self.group = process_group
        self.world_size = comm.world(process_group)[1]
        self.sharded_names = [n for n, s in feature_map.features.items()
                              if s["type"] == "categorical" and s.get("vocab_size", 0) >= shard_min_vocab]
        local_map = _SubFeatureMap(feature_map, OrderedDict((n, s) for n, s in feature_map.features.items()
                                                            if n not in self.sharded_names))
        self.embedding_layer = FeatureEmbedding(local_map, embedding_dim)
        self.fm = FactorizationMachine(local_map)
        self.tables = None
        if self.sharded_names:
            self.tables = ShardedTables([feature_map.features[n]["vocab_size"] for n in self.sharded_names],
                                        embedding_dim, with_lr=True, capacity_factor=capacity_factor,
                                        process_group=process_group, local_ops=local_ops,
                                        padding_idx=[feature_map.features[n].get("padding_idx") for n in self.sharded_names])

    def replicated_parameters(self):
        return list(self.embedding_layer.parameters()) + list(self.fm.parameters())

    def sharded_ids(self, X):
        return torch.stack([X[n].long() for n in self.sharded_names], dim=1)           # [B, T]

    def _fused_plans(self, X):
        emb = self.embedding_layer.embedding_layer
        lr = self.fm.lr_layer.embedding_layer.embedding_layer
        names, values, plan, posts = emb.plan_for(X)
        lnames, _, lplan, lposts = lr.plan_for(X)
        if not (emb.fusable(plan, posts) and lr.fusable(lplan, lposts) and names == lnames):
            raise NotImplementedError("ShardedFM fuses one-id-per-sample categorical and numeric features only")
        return values, plan, lplan

    def presort_local(self, X):
        """The id sort of the replicated tables' backward, on the current stream (ops.fm_presort): it needs ``X`` only,
        so a step can run it while the remote rows are still on their way; pass the result to ``logits``."""
        values, plan, lplan = self._fused_plans(X)
        return ops.fm_presort(plan.plan, lplan.plan, values, [m.weight for m in plan.modules],
                              [m.weight for m in lplan.modules])

    def logits(self, X, packed=None, packed_index=None, presorted=None):
        """``packed`` [B, T, row_width]: rows of the sharded tables already fetched from their owners (the
        piecewise-graphed step of recbox_amd.graph does the exchange itself); None = fetch them here.
        With ``packed_index`` [B, T] int32, ``packed`` is the exchange buffer [slots, row_width] and row (b, t)
        sits at wire slot packed_index[b, t].  ``presorted``: result of ``presort_local(X)``."""
        values, plan, lplan = self._fused_plans(X)
        lr_off = -1
        if self.tables is not None:
            if packed is None:
                packed = self.tables(self.sharded_ids(X))                              # [B, T, D + 4] from the owners
            lr_off = self.tables.lr_off
        # sync_grads() writes all-reduced rows (other ranks' batches) into the replicated tables' gradients in place:
        # they must be fresh tensors, not the row-re-zeroed persistent buffer of ops.config.reuse_grad_buffers
        from ... import comm
        reuse = ops.config.reuse_grad_buffers
        ops.config.reuse_grad_buffers = reuse and not comm.multi(self.group)
        try:
            return ops.fm_fused(plan.plan, lplan.plan, values, [m.weight for m in plan.modules],
                                [m.weight for m in lplan.modules], self.fm.lr_layer.bias, extra=packed,
                                extra_lr_off=lr_off, extra_index=packed_index, presorted=presorted)
        finally:
            ops.config.reuse_grad_buffers = reuse

    def forward(self, X):
        return {"y_pred": ops.sigmoid_output(self.logits(X))}

    def sync_grads(self):
        """All-reduce (sum) the dense gradients of the replicated parameters as ONE flat buffer."""
        from ... import comm
        end = getattr(getattr(self.tables, "local_ops", None), "end_step", None)
        if end is not None:
            end()                                   # step boundary of the owners' persistent gradient buffer
        if not comm.multi(self.group):
            return
        # every rank reduces the SAME layout: a parameter that received no gradient on this rank (an empty local batch, a
        # feature no sample of this rank used) contributes zeros -- ranks whose non-None sets differ would otherwise issue
        # all-reduces of different sizes and hang
        params = [p for p in self.replicated_parameters() if p.requires_grad]
        if not params:
            return
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        # (the fused backward's gradients are views of one flat buffer: reduced in place as one span, no copies)
        comm.all_reduce_coalesced_([p.grad for p in params], self.group).wait()


class _SubFeatureMap(object):
    """A view of a ranking FeatureMap restricted to some features (same attribute surface)."""

    def __init__(self, parent, features):
        self.features = features
        self.num_fields = sum(1 for s in features.values() if s["type"] != "meta")
        self.data_dir = getattr(parent, "data_dir", None)
        self.dataset_id = getattr(parent, "dataset_id", None)
        self.default_emb_dim = getattr(parent, "default_emb_dim", None)
        self.labels = getattr(parent, "labels", [])
