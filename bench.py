#!/usr/bin/env python
"""bench.py -- the embedding + feature-interaction hot path, forward+backward, on synthetic Criteo-shaped batches.

    python bench.py [--config fm|youtubednn|deepfm|sasrec] --gpus N --steps K --warmup W

--config fm (the default: BASELINE.json configs[1], the configuration the headline metric is quoted on) is described
below; youtubednn / deepfm / sasrec are configs[2..4] (see `MODEL_CONFIGS`): with --gpus N > 1 the first two run in
their multi-GPU form (item table / large tables row-sharded over the ranks with one RCCL all-to-all each way, the rest
data-parallel with one flat all-reduce), sasrec runs N independent replicas' worth of data-parallel steps.

FM (recbox.ranking) forward+backward on a Criteo-shaped batch.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line from rank 0.  A "step" is one pass of the hot path over one
synthetic batch: zero grads -> FeatureEmbedding gather -> LR + FM interaction ->
sigmoid + BCE -> backward to DENSE gradients of every table (the reference's
autograd contract).  No optimiser step: BASELINE.json's metric is fwd+bwd.
Workload = BASELINE.json configs[1]: 13 dense + 26 sparse fields, dim 16, batch
65 536 per GPU (weak scaling), cardinalities of SURVEY.md 8(d).
"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CRITEO_VOCABS = [1460, 583, 1000000, 1000000, 305, 24, 12517, 633, 3, 93145, 5683, 1000000, 3194, 27, 14992,
                 1000000, 10, 5652, 2173, 4, 1000000, 18, 15, 286181, 105, 142572]
N_DENSE = 13


class CriteoFeatureMap(object):
    """Minimal ranking FeatureMap carrying the Criteo-shaped schema."""

    def __init__(self, dim):
        from recbox_amd.ranking.features import FeatureMap
        self.fm = FeatureMap("criteo_synth", "/tmp")
        feats = OrderedDict()
        for i in range(N_DENSE):
            feats["I%d" % (i + 1)] = {"source": "", "type": "numeric"}
        for i, v in enumerate(CRITEO_VOCABS):
            feats["C%d" % (i + 1)] = {"source": "", "type": "categorical", "vocab_size": v + 1, "padding_idx": 0}
        self.fm.features = feats
        self.fm.num_fields = len(feats)
        self.fm.labels = ["label"]
        self.fm.default_emb_dim = dim
        self.fm.set_column_index()


def synthetic_batch(B, seed, dist, device):
    """Seeded Criteo-shaped batch.  ids arrive as float64 columns, exactly as the
    reference's ranking loader delivers them (one hstacked float64 [B, cols] tensor)."""
    g = torch.Generator().manual_seed(seed)
    cols = []
    for _ in range(N_DENSE):
        cols.append(torch.rand(B, generator=g, dtype=torch.float64))
    for v in CRITEO_VOCABS:
        if dist == "zipf":
            u = torch.rand(B, generator=g, dtype=torch.float64)
            ids = torch.floor(float(v) ** u).clamp(1, v)
        else:
            ids = torch.randint(1, v + 1, (B,), generator=g).double()
        cols.append(ids)
    cols.append((torch.rand(B, generator=g) < 0.25).double())
    batch = torch.stack(cols, dim=1).to(device)          # [B, 40] float64
    return batch


def slice_inputs(fm, batch):
    X = OrderedDict()
    for name, spec in fm.features.items():
        X[name] = batch[:, fm.get_column_index(name)]   # strided column views; kernels read them in place
    y = batch[:, fm.get_column_index("label")].float().view(-1, 1)
    return X, y


def init_weights(model, seed=0, std=0.1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))
        for m in model.modules():
            if isinstance(m, torch.nn.Embedding) and m.padding_idx is not None:
                m.weight[m.padding_idx].zero_()


def cpu_baseline(dim, B, dist, budget_s):
    """The oracle (a restatement of the reference's op sequence in plain PyTorch CPU ops)
    timed on this box's host cores, same shapes, fwd+bwd, no optimiser step.  ATen's CPU
    embedding/backward kernels do not scale to hundreds of threads, so the thread count is
    picked by a short calibration (one step each at 1/8 batch) and reported as `cores`."""
    from oracle import torch_ref as R
    ncpu = os.cpu_count() or 1
    fmw = CriteoFeatureMap(dim)
    model = R.RefFMModel(fmw.fm, dim)
    init_weights(model)
    batch = synthetic_batch(B, 1, dist, "cpu")
    X, y = slice_inputs(fmw.fm, batch)

    def step(Xs, ys):
        for p in model.parameters():
            p.grad = None
        loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(model(Xs)), ys, reduction="mean")
        loss.backward()

    small = max(B // 8, 1)
    Xs = OrderedDict((k, v[:small]) for k, v in X.items())
    ys = y[:small]
    best, best_t = ncpu, None
    for threads in sorted(set(t for t in (ncpu, 64, 32, 16, 8) if t <= ncpu), reverse=True):
        torch.set_num_threads(threads)
        step(Xs, ys)
        t0 = time.perf_counter()
        step(Xs, ys)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = threads, dt
    torch.set_num_threads(best)
    step(X, y)                                   # warm-up at full size
    t0, n = time.perf_counter(), 0
    while True:
        step(X, y)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20:
            break
    return {"value": B * n / el, "unit": "samples/s", "cores": best, "kind": "port",
            "sample": "%d full steps of the same workload (B=%d, dim=%d, %s ids) on torch CPU ops, %d of %d "
                      "hardware threads (fastest of a 1/8-batch calibration)" % (n, B, dim, dist, best, ncpu)}


def measured_traffic(path, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/rNN/traffic_rNN.json, the newest round on file; FETCH_SIZE / WRITE_SIZE are collected in separate runs --
    counters cannot ride in a timed run -- and corrected as MI355X_MICROARCH.md prescribes).  None when no measurement
    is on file."""
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "traffic_r[0-9][0-9].json")))
        with open(files[-1]) as fh:
            rec = json.load(fh)
        ent = rec.get({"fused": "fm"}.get(path, path), {})
        if kernel is not None and ent.get("kernel") != kernel:
            return None
        return ent.get("hbm_bytes_per_launch")
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2..4]: the rechub model mirrors (SURVEY.md 8d shapes)
# ---------------------------------------------------------------------------------------------------------------
MODEL_CONFIGS = {
    "youtubednn": "YoutubeDNN two-tower (rechub), ONE shared item table 10 M x 128 (history <= 50 mean-pooled + 1 positive + 4 "
                  "negatives per sample), user MLP 256-128 with BatchNorm, temperature 0.02, CE over [B, 5]",
    "deepfm": "DeepFM (rechub) on Criteo-shaped 26 sparse + 13 dense fields, dim 64, MLP 3 x 400 with BatchNorm, dropout 0",
    "sasrec": "SASRec (rechub), 1 M items, dim 64, seq_len 200, 2 blocks, 1 head, dropout 0, pos/neg log-sigmoid loss",
}
BF16_MFMA_PEAK = 2500.0       # TFLOP/s dense, v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md)
FP32_MFMA_PEAK = 157.3        # TFLOP/s dense, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md: no TF32 on gfx950; bf16 would break 1e-4)


def init_weights_device(model, dev, seed, rank, std=0.1, big=1 << 22):
    """Replicated parameters from a CPU generator (identical on every rank); tables / shards with more than `big`
    elements from a device generator (seeded per rank: every rank holds different rows)."""
    g = torch.Generator().manual_seed(seed)
    gd = torch.Generator(device=dev).manual_seed(seed * 1000 + rank)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters()):
            if p.numel() > big:
                p.copy_(torch.randn(p.shape, generator=gd, device=dev) * std)
            else:
                p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))


def _youtube_features(V, D):
    from recbox_amd.rechub.basic.features import SequenceFeature, SparseFeature
    return ([SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)],
            [SparseFeature("item", V, D)],
            [SequenceFeature("neg_items", V, D, pooling="concat", shared_with="item")])


def _youtube_batch(B, V, L, n_neg, seed, dist, dev):
    g = torch.Generator().manual_seed(seed)

    def ids(shape):
        if dist == "zipf":
            return torch.floor(float(V - 1) ** torch.rand(shape, generator=g, dtype=torch.float64)).long().clamp(1, V - 1)
        return torch.randint(1, V, shape, generator=g)

    lens = torch.randint(1, L + 1, (B,), generator=g)                     # U{1..50}, padded with id 0
    hist = ids((B, L)) * (torch.arange(L)[None, :] < lens[:, None])
    return {"hist": hist.to(dev), "item": ids((B,)).to(dev), "neg_items": ids((B, n_neg)).to(dev)}


def _deepfm_features(D):
    from recbox_amd.rechub.basic.features import DenseFeature, SparseFeature
    dense = [DenseFeature("I%d" % i) for i in range(N_DENSE)]
    sparse = [SparseFeature("C%d" % i, v + 1, D) for i, v in enumerate(CRITEO_VOCABS)]
    return dense, sparse


def _deepfm_batch(B, seed, dist, dev):
    g = torch.Generator().manual_seed(seed)
    x = {}
    for i in range(N_DENSE):
        x["I%d" % i] = torch.rand(B, generator=g).to(dev)
    for i, v in enumerate(CRITEO_VOCABS):
        if dist == "zipf":
            x["C%d" % i] = torch.floor(float(v) ** torch.rand(B, generator=g, dtype=torch.float64)).long().clamp(1, v).to(dev)
        else:
            x["C%d" % i] = torch.randint(1, v + 1, (B,), generator=g).to(dev)
    x["label"] = (torch.rand(B, generator=g) < 0.25).float().to(dev)
    return x


def _sasrec_features(V, D):
    from recbox_amd.rechub.basic.features import SequenceFeature
    return [SequenceFeature("seq", V, D, pooling="concat"),
            SequenceFeature("pos", V, D, pooling="concat", shared_with="seq"),
            SequenceFeature("neg", V, D, pooling="concat", shared_with="seq")]


def _sasrec_batch(B, V, L, seed, dev):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(20, L + 1, (B,), generator=g)                    # U{20..200}, left-aligned, 0 pad
    keep = torch.arange(L)[None, :] < lens[:, None]
    return {"seq": (torch.randint(1, V, (B, L), generator=g) * keep).to(dev),
            "pos": (torch.randint(1, V, (B, L), generator=g) * keep).to(dev),
            "neg": (torch.randint(1, V, (B, L), generator=g) * keep).to(dev),
            "weight": (keep.float() / keep.sum().float()).to(dev)}          # masked mean without boolean indexing


def _sasrec_loss(model, x):
    """-mean over the real positions of log sigmoid(pos) + log(1 - sigmoid(neg)) (sasrec.py:100-107 + the BPR-style
    pos/neg objective of the rechub trainer), as a weighted sum so that the step has static shapes (graph capture)."""
    import torch.nn.functional as F
    pos, neg = model(x)
    w = x["weight"]
    if pos.is_cuda:                      # K7's loss epilogue: one pass each way (rbx_pair_logsigmoid_*)
        from recbox_amd import ops
        return ops.pair_logsigmoid_loss(pos, neg, w)
    return -((F.logsigmoid(pos) + F.logsigmoid(-neg)) * w).sum()


def model_cpu_baseline(cfg, args, budget_s):
    """The oracle's restatement of the same model (ATen CPU ops, the reference's op sequence) on this box's host cores:
    a bounded sample of the same workload."""
    import torch.nn.functional as F
    from oracle import torch_ref as R
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    torch.set_num_threads(threads)
    B = args.batch
    if cfg == "youtubednn":
        V = args.items or 10_000_000
        uf, itf, negf = _youtube_features(V, 128)
        model = R.RefYoutubeDNN(uf, itf, negf, {"dims": [256, 128], "activation": "relu"}, temperature=0.02)
        x = _youtube_batch(B, V, 50, 4, 1, args.dist, "cpu")
        tgt = torch.zeros(B, dtype=torch.long)
        loss_of = lambda: F.cross_entropy(model(x), tgt)                                  # noqa: E731
        what = "B=%d, table %d x 128" % (B, V)
    elif cfg == "deepfm":
        dense, sparse = _deepfm_features(64)
        model = R.RefDeepFM(dense + sparse, sparse, {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"})
        x = _deepfm_batch(B, 1, args.dist, "cpu")
        loss_of = lambda: F.binary_cross_entropy(model(x), x["label"])                    # noqa: E731
        what = "B=%d, dim 64" % B
    else:
        B = min(B, 512)
        V = args.items or 1_000_000
        model = R.RefSASRec(_sasrec_features(V, 64), max_len=200, dropout_rate=0.0, num_blocks=2, num_heads=1)
        x = _sasrec_batch(B, V, 200, 1, "cpu")
        loss_of = lambda: _sasrec_loss(model, x)                                          # noqa: E731
        what = "B=%d (of %d), 1 M items, L=200" % (B, args.batch)
    with torch.no_grad():
        for p in model.parameters():
            p.normal_(0.0, 0.1)

    def step():
        for p in model.parameters():
            p.grad = None
        loss_of().backward()

    step()
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 10:
            break
    return {"value": B * n / el, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d steps of the same model (%s) on torch CPU ops, %d of %d hardware threads" % (n, what, threads, ncpu)}


def shutdown_distributed(*holders):
    """Leave the process group: drop every captured graph first (a graph that holds RCCL kernel nodes keeps the
    communicator busy: destroy_process_group() hung behind one in rounds 2-3), synchronise, then destroy the group under a
    watchdog -- a teardown that still hangs must not cost the line this process has already printed."""
    import gc
    import threading
    from recbox_amd import comm
    for h in holders:
        rel = getattr(h, "release", None)
        if rel is not None:
            rel()
    comm.direct.shutdown()
    gc.collect()
    torch.cuda.synchronize()
    sys.stdout.flush()
    sys.stderr.flush()
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return
    guard = threading.Timer(float(os.environ.get("RECBOX_BENCH_TEARDOWN_SECONDS", "20")), lambda: os._exit(0))
    guard.daemon = True
    guard.start()
    try:
        torch.distributed.destroy_process_group()
    finally:
        guard.cancel()


def run_model_config(args, rank, world, dev):
    """configs[2..4].  N = 1: the single-GPU mirror, whole step replayed as one hipGraph (persistent dense gradients).
    N > 1 (or --force-sharded): youtubednn / deepfm in their sharded + data-parallel form (eager launches: the
    collectives sit inside the autograd node), sasrec data-parallel with one flat all-reduce."""
    import torch.nn.functional as F
    from recbox_amd import comm, ops
    cfg = args.config
    ops.config.check_ids = False
    sharded = world > 1 or args.force_sharded
    B, K = args.batch, max(args.rotate, 1)
    factor = args.capacity_factor or 1.25
    note = {}
    # (parameters are created on the GPU: a 5 GB table built on the host first is slow)
    if cfg == "youtubednn":
        V, D, L, n_neg = args.items or 10_000_000, 128, 50, 4
        feats = _youtube_features(V, D)
        params = {"dims": [256, 128], "activation": "relu"}
        with torch.device(dev):
            if sharded:
                from recbox_amd.rechub.sharded import ShardedYoutubeDNN
                model = ShardedYoutubeDNN(*feats, params, temperature=0.02, capacity_factor=factor)
            else:
                from recbox_amd.rechub.models.matching import YoutubeDNN
                model = YoutubeDNN(*feats, params, temperature=0.02)
        batches = [_youtube_batch(B, V, L, n_neg, 1 + rank + 1000 * k, args.dist, dev) for k in range(K)]
        loss_of = lambda x: ops.softmax_cross_entropy(model(x))       # == F.cross_entropy(., 0): rbx_softmax_ce_*  # noqa: E731
    elif cfg == "deepfm":
        D = 64
        dense, sparse = _deepfm_features(D)
        mlp = {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"}
        with torch.device(dev):
            if sharded:
                from recbox_amd.rechub.sharded import ShardedDeepFM
                model = ShardedDeepFM(dense + sparse, sparse, mlp, shard_min_vocab=args.shard_min_vocab,
                                      capacity_factor=factor)
            else:
                from recbox_amd.rechub.models.ranking import DeepFM
                model = DeepFM(dense + sparse, sparse, mlp)
        batches = [_deepfm_batch(B, 1 + rank + 1000 * k, args.dist, dev) for k in range(K)]
        # the CTR trainer's torch.nn.BCELoss (rechub/trainers/ctr_trainer.py:33) on the model's sigmoid output
        loss_of = lambda x: ops.binary_cross_entropy(model(x), x["label"])             # noqa: E731
    else:
        V, D, L = args.items or 1_000_000, 64, 200
        from recbox_amd.rechub.models.matching import SASRec
        with torch.device(dev):
            model = SASRec(_sasrec_features(V, D), max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1)
        batches = [_sasrec_batch(B, V, L, 1 + rank + 1000 * k, dev) for k in range(K)]
        loss_of = lambda x: _sasrec_loss(model, x)                                     # noqa: E731
    model.to(dev).train()
    init_weights_device(model, dev, 0, rank)
    params = list(model.parameters())
    shard_store = getattr(getattr(model, "embedding", None), "store", None)
    if sharded and shard_store is not None and not args.fresh_grads and hasattr(shard_store.local_ops, "persistent"):
        # the shard's dense gradient: one buffer cleared by the previous step's rows instead of a fresh zero-filled
        # [rows, D] tensor per step (5 GB at cfg 3 in a world of one, 640 MB per rank at W = 8)
        shard_store.local_ops.persistent(shard_store.weight)
    if sharded and shard_store is not None:
        # The padded exchange has `capacity_factor` x (a perfectly balanced share) slots per peer; a lookup that finds none
        # raises the store's overflow flag.  Every resident batch is routed once before anything is captured: if any rank
        # overflowed, the factor of every store doubles (the wire sizes derive from it at each call) and the probe repeats.
        stores = list(model.embedding.stores.values())
        for _ in range(4):
            with torch.no_grad():
                for b in batches:
                    model(b)
            flag = torch.stack([st.overflow.float().reshape(()) for st in stores]).max().reshape(1).clone()
            if world > 1:
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            for st in stores:
                st.overflow.zero_()
            if float(flag.item()) == 0:
                break
            factor *= 2
            for st in stores:
                st.capacity_factor = float(factor)
        note["capacity_factor"] = factor
    x = dict((k, v.clone()) for k, v in batches[0].items())

    def refill(i):
        if K > 1:
            for k, v in batches[i % K].items():
                x[k].copy_(v)

    dp_flat = None
    if sharded and not hasattr(model, "sync_grads") and world > 1:
        dp_flat = [p for p in params if p.requires_grad]

    def step_over(xb):
        def one_step():
            return _model_step(xb)
        return one_step

    # --optimizer: the step continues into an optimiser step (not part of BASELINE's fwd+bwd metric: reported beside it).
    # sparse_adam: recbox_amd.optim.SparseAdam on the embedding tables (only the rows the batch touched are read and
    # written) + torch.optim.Adam on the rest; dense_adam: torch.optim.Adam on everything, the reference's loop
    # (ranking_model.py:191-197) -- every row of every table and of its two moments per step.
    opt_steps = []
    if args.optimizer != "none" and not sharded:
        from recbox_amd import optim as rb_optim
        tables, rest = rb_optim.split_parameters(model)
        if args.optimizer == "sparse_adam":
            opt_steps = [rb_optim.SparseAdam(tables, lr=1e-3, capturable=True).step]
            if rest:
                opt_steps.append(torch.optim.Adam(rest, lr=1e-3, capturable=True).step)
        else:
            opt_steps = [torch.optim.Adam(params, lr=1e-3, capturable=True).step]

    def _model_step(xb):
        for p in params:
            p.grad = None
        loss = loss_of(xb)
        if sharded:
            (loss / world).backward()                     # global-mean loss: shard owners sum every rank's contributions
            if hasattr(model, "sync_grads"):
                model.sync_grads()
            elif dp_flat is not None:                     # plain data parallel: ONE flat all-reduce of every gradient
                flat = torch._utils._flatten_dense_tensors([p.grad for p in dp_flat])
                comm.all_reduce_sum_(flat)
                for p, r in zip(dp_flat, torch._utils._unflatten_dense_tensors(flat, [p.grad for p in dp_flat])):
                    p.grad.copy_(r)
        else:
            ops.backward(loss)
        for st in opt_steps:
            st()
        return loss

    eager_step = step_over(x)                 # reads the static buffers `x` (refill(i) copies batch i % K into them)
    step, graph_note = eager_step, "eager launches"
    rotating_graphs = None
    persistent = (not args.fresh_grads) and not sharded and cfg in ("youtubednn", "deepfm")
    if not args.eager and not sharded:
        from recbox_amd.graph import GraphedStep
        try:
            # every table of youtubednn / deepfm feeds exactly ONE lookup per step: the dense gradients may stay in a
            # persistent buffer of which only the previous step's rows are cleared (ops.config.reuse_grad_buffers = "all")
            reuse = "all" if persistent else False
            if K > 1 and not args.rotate_by_copy:
                # one captured step per resident batch (shared intermediate memory): nothing is copied in the timed loop --
                # a refill is one copy_ launch per input tensor, 40 of them for DeepFM's feature dict
                rotating_graphs = []
                for k in range(K):
                    rotating_graphs.append(GraphedStep(step_over(batches[k]), warmup=3 if k == 0 else 2, reuse_grads=reuse,
                                                       params=params,
                                                       pool=rotating_graphs[0].pool() if rotating_graphs else None))
                step = rotating_graphs[0]
                graph_note = "hipGraph replay (one captured step per resident batch)"
            else:
                step = GraphedStep(eager_step, warmup=3, reuse_grads=reuse)
                graph_note = "hipGraph replay"
        except Exception as exc:
            print("[bench] hipGraph capture failed (%s: %s); launching the step eagerly" % (type(exc).__name__, exc),
                  file=sys.stderr)
            torch.cuda.synchronize()
            rotating_graphs = None
            step = eager_step
    elif (sharded and not args.eager and args.sharded_graph in ("auto", "whole") and comm.direct.on
          and comm.direct.capturable):
        # the N > 1 step -- exchanges, all-reduces, the autograd nodes around them -- as ONE hipGraph per resident batch:
        # every collective is an RCCL call on the capturing stream (comm.direct), so nothing is launched from Python
        # inside the timed region (round 3 launched ~130 kernels per step eagerly, each collective on RCCL's own stream)
        from recbox_amd.graph import GraphedStep
        try:
            rotating_graphs = []
            for k in range(K):
                rotating_graphs.append(GraphedStep(step_over(batches[k]), warmup=3 if k == 0 else 2, reuse_grads=False,
                                                   params=params, capture_error_mode="thread_local",
                                                   pool=rotating_graphs[0].pool() if rotating_graphs else None))
            step = rotating_graphs[0]
            graph_note = ("ONE hipGraph per resident batch, RCCL collectives inside (rbx_all_to_all / rbx_all_reduce on the "
                          "step's stream)")
        except Exception as exc:
            print("[bench] hipGraph capture of the sharded step failed (%s: %s); launching it eagerly"
                  % (type(exc).__name__, exc), file=sys.stderr)
            torch.cuda.synchronize()
            rotating_graphs = None
            step = eager_step

    def run_step(i):
        if rotating_graphs is not None:
            rotating_graphs[i % K]()
        else:
            refill(i)
            step()

    for i in range(args.warmup):
        run_step(i)
    # ---- the dominant kernel of the config and its algorithmic work per launch ----
    if cfg == "youtubednn":
        nnz = sum(int((b["hist"] != 0).sum()) for b in batches) / float(K)
        lookups = nnz + B * (1 + n_neg)
        if sharded:
            want = lambda m: m[0] == "shard_serve"                                        # noqa: E731
            kname = "shard_serve_kernel<32,1> (owner-side gather + pooling of the received lookups)"
            # rows read + int32 row numbers + offsets + partial sums and item rows written
            work = lookups * (D * 4 + 4) + world * (B + 1) * 4 + (world * B + B * (1 + n_neg)) * D * 4
        else:
            want = lambda m: m[0] == "embed_fwd" and m[3] == B                            # noqa: E731
            kname = "embed_seq_kernel<64,2,1,true> + embed_fwd_kernel (rbx_embed_fwd: history pooled in the gather)"
            work = lookups * (D * 4 + 8) + B * (2 + n_neg) * D * 4
        roof = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s"}
    elif cfg == "deepfm":
        K1 = len(CRITEO_VOCABS) * 64 + N_DENSE
        want = lambda m: m[0] == "linear_fwd" and m[3] == K1                              # noqa: E731
        bx6 = ops.config.gemm_bx6 and os.environ.get("RBX_GEMM_BX6", "1") != "0"
        kname = (("gemm_bxp_kernel: tower layer 1 forward, [B, %d] x [400, %d]^T as six v_mfma_f32_32x32x16_bf16 products of "
                  "three-way split operands per f32 product, f32 accumulation" % (K1, K1)) if bx6 else
                 ("gemm_f32_kernel (+ narrow tail): tower layer 1 forward, [B, %d] x [400, %d]^T on v_mfma_f32_32x32x2_f32"
                  % (K1, K1)))
        work = 2.0 * B * 400 * K1
        roof = {"bound": "mfma", "peak": FP32_MFMA_PEAK * 1e3, "unit": "GFLOP/s"}
        if bx6:
            # `achieved` / `frac` stay what the contract defines: ALGORITHMIC f32 FLOPs over the f32 MFMA peak (the dtype the
            # path computes in).  The kernel issues 6x that on the bf16 pipes: `pipe` prices the executed FLOPs against THEIR peak.
            roof["pipe"] = {"executed_per_algorithmic": 6, "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s (bf16 MFMA, dense)"}
    else:
        want = lambda m: m[0] == "attn_bwd"                                               # noqa: E731
        kname = "attn_mfma_bwd_q/kv_kernel<64> (causal attention backward, L=200, d=64)"
        work = 5.0 * B * 200 * 200 * 64                                                   # 5 causal-halved L x L x d GEMMs
        roof = {"bound": "mfma", "peak": FP32_MFMA_PEAK * 1e3, "unit": "GFLOP/s"}
    timer = ops.KernelTimer(want)
    if step is eager_step:
        ops.kernel_timer = timer
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if step is not eager_step:
        # the replay cannot be bracketed: time the dominant kernel with HIP events over eager steps queued behind fills
        # (on the stream the captured steps' autograd nodes live on: a backward on another stream than a parameter's
        #  AccumulateGrad node makes the engine sync the two and warn)
        import contextlib
        first = rotating_graphs[0] if rotating_graphs else step
        on_stream = torch.cuda.stream(first.stream) if hasattr(first, "stream") else contextlib.nullcontext()
        # (these eager steps run AFTER the timed region, only to put HIP events around the dominant kernel; the captured
        #  graphs still hold the autograd nodes of their capture, so the engine reports a stream mismatch for them: expected
        #  here, and silenced here only)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        with on_stream:
            n_timed = min(args.steps, 10)
            ballast = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            for _ in range(40):
                ballast.zero_()
            ops.kernel_timer = timer
            for i in range(n_timed):
                refill(args.warmup + args.steps + i)
                eager_step()
            torch.cuda.synchronize()
        del ballast
    ops.kernel_timer = None
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    overflow = False
    store = getattr(getattr(model, "embedding", None), "store", None)
    if store is not None:
        flag = store.overflow.float().reshape(1).clone()
        if world > 1:
            torch.distributed.all_reduce(flag)
        overflow = bool(flag.item() > 0)
    held = tuple(rotating_graphs or ()) + ((step,) if hasattr(step, "graph") else ())
    if rank != 0:
        return held
    kms = timer.mean_ms()
    if kms:
        traffic = None
        if cfg == "youtubednn" and B == 65536 and (args.items or 10_000_000) == 10_000_000 and args.dist == "uniform":
            traffic = measured_traffic("youtubednn_sharded" if sharded else "youtubednn", None)
        roof.update({"achieved": work / (kms * 1e-3) / 1e9, "kernel": kname, "kernel_ms": kms, "traffic": traffic,
                     ("algorithmic_bytes_per_launch" if roof["bound"] == "hbm" else "algorithmic_flop_per_launch"): work,
                     "inputs": ("%d distinct batches in rotation" % K) if K > 1 else "one batch replayed"})
        roof["frac"] = roof["achieved"] / roof["peak"]
        if roof["unit"] == "GFLOP/s":                       # report TFLOP/s as the contract asks
            roof.update({"unit": "TFLOP/s", "achieved": roof["achieved"] / 1e3, "peak": roof["peak"] / 1e3})
        if "pipe" in roof:
            # the kernel runs on the bf16 matrix cores (six products per f32 product): `frac` prices the EXECUTED FLOPs against
            # the pipe they execute on; the algorithmic-f32 rate and its ratio to the f32 MFMA peak (which can exceed 1: it is a
            # speed-up over the f32 pipe, not a roofline) are side fields
            pipe = roof.pop("pipe")
            roof["f32_equivalent"] = {"achieved": roof["achieved"], "unit": "TFLOP/s of algorithmic f32 work",
                                      "over_f32_mfma_peak": roof["achieved"] / roof["peak"], "f32_mfma_peak": roof["peak"]}
            roof["achieved"] = roof["achieved"] * pipe["executed_per_algorithmic"]
            roof["peak"] = pipe["peak"]
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["executed_per_algorithmic"] = pipe["executed_per_algorithmic"]
            roof["unit"] = pipe["unit"]
    else:
        roof = None
    par = "dp1"
    if sharded and cfg == "youtubednn":
        par = ("item table row-sharded over %d ranks (one all-to-all each way, history pooled at the owners) + dp%d tower"
               % (world, world))
    elif sharded and cfg == "deepfm":
        par = ("dp%d, tables >= %d rows row-sharded over the ranks (one all-to-all each way), flat all-reduce of the rest"
               % (world, args.shard_min_vocab))
    elif sharded:
        par = "dp%d (replicated model, one flat all-reduce of every gradient)" % world
    out = {"metric": "samples/sec fwd+bwd, Criteo-shaped batch 65 536; embedding HBM GB/s vs roofline",
           "value": B * world * args.steps / el, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s; batch %d per GPU, %s ids (%s), %s, dense-grad autograd contract (%s), no optimiser step"
                                  % (MODEL_CONFIGS[cfg], B, args.dist,
                                     (("%d distinct batches resident in HBM, replayed in rotation" % K)
                                      if rotating_graphs is not None else ("%d distinct batches rotated by copy" % K))
                                     if K > 1 else "one batch replayed", graph_note,
                                     "persistent grad buffers, rows of the previous step re-zeroed" if persistent
                                     else "fresh zero-filled grads every step"),
                      "global_batch": B * world, "batch_per_gpu": B, "parallelism": par},
           "roofline": roof}
    if cfg in ("deepfm", "youtubednn") and ops.config.gemm_bx6 and os.environ.get("RBX_GEMM_BX6", "1") != "0":
        # f32 in, f32 out, f32 accumulation, f32-level error (tests: the f32 kernel's tolerances): NOT a bf16 run
        out["dtype_note"] = ("towers: y = x W^T and dx = dy W as six bf16-MFMA products of three-way split f32 operands "
                             "(x = h + m + l exact to 2^-24), f32 accumulate; dW and everything else on f32 arithmetic")
    if opt_steps:
        out["config"]["workload"] = out["config"]["workload"].replace("no optimiser step", "+ optimiser step (%s)" % args.optimizer)
        out["metric"] = "samples/sec fwd+bwd+update (beside the fwd+bwd metric of BASELINE.json)"
    if store is not None:
        out["config"]["exchange"] = "padded capacity_factor=%g, overflow=%s" % (factor, overflow)
    if world == 1 and not args.no_cpu_baseline and not sharded:
        out["cpu_baseline"] = model_cpu_baseline(cfg, args, args.cpu_seconds)
    print(json.dumps(out))
    return held


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["fm"] + sorted(MODEL_CONFIGS), default="fm",
                    help="fm = BASELINE.json configs[1] (the headline metric); the others are configs[2..4]")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)      # (a multiple of 4 and 8: --steps-per-graph)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 65536; sasrec 4096)")
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every step from Python instead of replaying a hipGraph")
    ap.add_argument("--fresh-grads", action="store_true",
                    help="allocate and zero-fill new dense gradients every step (autograd's default) instead of re-zeroing "
                         "the rows the previous step wrote in a persistent buffer")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded model even at world size 1 (exercises the RCCL exchange path)")
    ap.add_argument("--shard-min-vocab", type=int, default=100000)
    ap.add_argument("--direct-rccl", action="store_true",
                    help="N>1: exchange through rbx_all_to_all (grouped ncclSend/ncclRecv on the step's stream) instead "
                         "of torch.distributed.all_to_all_single")
    ap.add_argument("--capacity-factor", type=float, default=1.25,
                    help="slots per peer of the sync-free padded exchange, relative to a perfectly balanced batch "
                         "(doubled automatically if the warm-up overflows); 0 = exact all-to-all-v (host sync per "
                         "call, eager launches only)")
    ap.add_argument("--rotate", type=int, default=8,
                    help="number of DISTINCT device-resident batches the timed loop cycles through (a training loop sees "
                         "new ids every step: replaying one batch keeps its rows in the 256 MB Infinity Cache); "
                         "0/1 = replay one batch.  FM on one GPU: one captured step per resident batch; elsewhere (and "
                         "with --rotate-by-copy) batch i %% K is copied into the static input buffer before each step, "
                         "inside the timed region")
    ap.add_argument("--steps-per-graph", type=int, default=None,
                    help="FM, one GPU: capture this many consecutive steps (one per resident batch) in each hipGraph.  Default: 4 "
                         "when --rotate and --steps are multiples of 4 (a replay then costs its ~12 us of graph-to-graph "
                         "latency once per four steps: profiles/r06/fm_steps_per_graph.txt), else 1; the line of 1 is always "
                         "reported beside it (configs.fm_one_step_per_graph)")
    ap.add_argument("--prefetch-sort", action="store_true",
                    help="FM, one GPU: while step i runs, the ids of batch i + 1 are sorted on the side stream (FM.presort: a "
                         "loop whose loader is one batch ahead), so that the sort no longer sits in front of the backward's "
                         "reduce: 0.261 vs 0.267 ms.  Off by default: every step of the headline line sorts its own ids")
    ap.add_argument("--pack-min-vocab", type=int, default=0,
                    help="with --pack-tables: only tables with at least this many rows share one packed row with their LR weight")
    ap.add_argument("--rotate-by-copy", action="store_true",
                    help="FM, one GPU: rotate by copy_ into one static buffer (one captured step) instead of one graph per batch")
    ap.add_argument("--sort-after-forward", action="store_true",
                    help="fm: enqueue the backward's id sort / re-zero after the forward kernel instead of beside it "
                         "(recbox_amd.ops.config.sort_before_forward = False)")
    ap.add_argument("--contiguous-ids", action="store_true",
                    help="fm (experiment): hand every feature its own contiguous [B] column instead of a strided view of the "
                         "[B, 40] float64 batch (what the reference's loader delivers)")
    ap.add_argument("--pack-tables", action="store_true",
                    help="fm: FM.pack_tables() -- every (embedding, LR) table pair in one packed [V, 32] storage, one "
                         "128-byte request per lookup.  Measured: no gain (fm_fused_fwd 48.0 vs 47.5 us; the dim-1 LR "
                         "tables are 22 MB and cache-resident anyway), so it is off by default")
    ap.add_argument("--path", choices=["fused", "layers"], default="fused",
                    help="fused: FM model body in rbx_fm_fwd/bwd; layers: drop-in layers composed as the reference does")
    ap.add_argument("--items", type=int, default=None, help="youtubednn: rows of the item table (10 M); sasrec: items (1 M)")
    ap.add_argument("--optimizer", choices=["none", "sparse_adam", "dense_adam"], default="none",
                    help="model configs, one GPU: append an optimiser step to every step (sparse_adam: only the touched rows of "
                         "the tables, recbox_amd.optim; dense_adam: torch.optim.Adam over every parameter, as the reference does)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="fm, one GPU: do not append the time-boxed youtubednn / deepfm / sasrec sub-runs under \"configs\"")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: weak = --batch samples PER GPU (per-GPU work fixed: the default of fm, deepfm, sasrec); strong "
                         "= --global-batch samples split B/N per GPU (the default of youtubednn: SURVEY.md 8d states cfg 3 as "
                         "GLOBAL B = 65 536, 8 192 per GPU at N = 8)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="--scaling strong: the global batch (default 65536; sasrec 4096)")
    ap.add_argument("--sharded-graph", choices=["auto", "whole", "pieces", "eager"], default="auto",
                    help="N > 1 / --force-sharded: how the step is launched.  whole = ONE hipGraph with the RCCL collectives "
                         "inside (needs them on the step's stream: recbox_amd.comm.direct, checked collectively); pieces = "
                         "hipGraph pieces with the collectives between them (fm only); eager = from Python; auto = whole "
                         "when the check passes, else pieces (fm) / eager")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if (args.config == "youtubednn" and args.gpus > 1 and args.batch is None) else "weak"
    if args.scaling == "strong":
        if args.batch is not None:
            ap.error("--scaling strong takes --global-batch, not --batch")
        gb = args.global_batch or (4096 if args.config == "sasrec" else 65536)
        if gb % max(args.gpus, 1):
            ap.error("--global-batch %d does not split over %d GPUs" % (gb, args.gpus))
        args.batch = gb // max(args.gpus, 1)
    if args.batch is None:
        args.batch = 4096 if args.config == "sasrec" else 65536

    one_gpu = os.environ.get("RECBOX_BENCH_ONE_GPU", "0") != "0"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` from a bare shell: start the N ranks the way the driver does (one process per GPU)
        import socket
        import subprocess
        if not one_gpu and torch.cuda.device_count() < args.gpus:
            raise SystemExit("bench.py: --gpus %d but this box has %d GPU(s) (RECBOX_BENCH_ONE_GPU=1 puts every rank on "
                             "cuda:0 over gloo: control flow only, its timings mean nothing)"
                             % (args.gpus, torch.cuda.device_count()))
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: pass the same N to both (the line's n_gpus "
                         "is the number of ranks that ran)" % (world, args.gpus))
    # RECBOX_BENCH_ONE_GPU=1 (tests only): every rank on cuda:0 over gloo, to run the N>1 control flow of this script
    # on a single-GPU box (RCCL needs one GPU per rank); the numbers of such a run mean nothing
    local_rank = 0 if one_gpu else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from recbox_amd import comm, ops
    if args.force_sharded:
        comm.force_world_of_one = True        # the world-of-one line issues every collective of the N > 1 step
    if (world > 1 or args.force_sharded) and not one_gpu:
        # every collective as an RCCL call on the step's own stream (rbx_all_to_all / rbx_all_reduce) instead of
        # torch.distributed's on RCCL's stream: compared with torch.distributed once, on every rank, eagerly and replayed
        # from a hipGraph, before it is used (comm.direct.self_check); --direct-rccl makes a failed check an error,
        # RECBOX_AMD_DIRECT_RCCL=0 switches it off
        if args.direct_rccl:
            comm.direct.enable(True)
        comm.direct.self_check(device=dev)
    if args.config != "fm":
        held = run_model_config(args, rank, world, dev) or ()
        if world > 1 or args.force_sharded:
            shutdown_distributed(*held)
        return

    from recbox_amd.ranking.pytorch.models import FM, ShardedFM
    ops.config.check_ids = False              # no per-call host sync inside the timed region
    if args.sort_after_forward:
        ops.config.sort_before_forward = False
    # every step starts from zero_grad(set_to_none=True): the dense gradients may live in ONE persistent buffer of which
    # only the rows the previous step wrote are cleared (rbx_fm_rezero) instead of a 379 MB zero fill per step; p.grad
    # after a step is the same dense [V, D] tensor either way (tests/test_gpu_ranking.py: bit-identical)
    # (the N>1 path keeps fresh gradients: its replicated tables are the small ones, there is nothing to save)
    ops.config.reuse_grad_buffers = not (args.fresh_grads or world > 1 or args.force_sharded)
    if ops.config.reuse_grad_buffers and args.path == "layers":
        ops.config.reuse_grad_buffers = "all"       # layer-composed FM: every table feeds exactly one lookup per step
    fmw = CriteoFeatureMap(args.dim)
    sharded = world > 1 or args.force_sharded
    B = args.batch
    K = max(args.rotate, 1)
    # K distinct batches live on the device; `batch` is the static buffer the step reads (X are column views of it):
    # refill(i) copies batch i % K into it before every step, inside the timed region (21 MB, ~10 us)
    batches = [synthetic_batch(B, 1 + rank + 1000 * k, args.dist, dev) for k in range(K)]
    batch = batches[0].clone()
    X, y = slice_inputs(fmw.fm, batch)
    labels = [slice_inputs(fmw.fm, b)[1] for b in batches]
    if args.contiguous_ids:
        batch = batch.t().contiguous()                    # [40, B]: every column of the batch is now a contiguous row
        batches = [b.t().contiguous() for b in batches]
        X = OrderedDict((name, batch[fmw.fm.get_column_index(name)]) for name in fmw.fm.features)

    def refill(i):
        if K > 1:
            batch.copy_(batches[i % K])
            y.copy_(labels[i % K])
    n_fields = len(fmw.fm.features)
    cap_factor = args.capacity_factor

    def build_model():
        if sharded:
            # big tables row-sharded over the ranks (all-to-all over RCCL/xGMI), small ones replicated
            m = ShardedFM(fmw.fm, args.dim, shard_min_vocab=args.shard_min_vocab,
                          capacity_factor=(cap_factor or None)).to(dev)
        else:
            m = FM(fmw.fm, args.dim, fused=(args.path == "fused")).to(dev)
        init_weights(m)                       # same seed on every rank: replicated parameters start identical
        if not sharded and args.path == "fused" and args.pack_tables:
            # each (embedding, LR) table pair behind ONE packed [V, 32] storage: one 128-byte request per lookup in the
            # fused forward (FM.pack_tables; the parameters, their names and the dense gradients are unchanged)
            m.pack_tables(min_vocab=args.pack_min_vocab)
        return m

    model = build_model()
    from recbox_amd.ranking.pytorch.torch_utils import get_loss
    loss_fn = get_loss("binary_crossentropy")

    params = list(model.parameters())

    def step_over(Xk, yk):
        def one_step():
            for p in params:                  # == model.zero_grad(set_to_none=True) without walking the module tree
                p.grad = None
            prob = model(Xk)["y_pred"]
            loss = loss_fn(prob, yk, reduction="mean")    # the harness's get_loss("binary_crossentropy")
            if sharded:
                (loss / world).backward()     # global-mean loss: shard owners sum contributions of every rank
                model.sync_grads()            # replicated small tables / numeric weights / bias: one all-reduce
            else:
                ops.backward(loss)            # == loss.backward() with the constant 1 as its gradient (no ones_like fill)
            return loss
        return one_step

    eager_step = step_over(X, y)              # reads the static buffer `batch` (refill(i) puts batch i % K there)

    def prefetching_step(Xk, yk, mine, X_next, nxt):
        """The same step with its id sort made one step AHEAD: `mine` holds the sorted ids of this batch (made while the
        previous step ran), and while this step runs the ids of the NEXT batch are sorted into `nxt` on the side stream --
        what a training loop whose loader is one batch ahead does.  One sort per step, as before; it has left the chain
        rezero -> sort -> reduce that the step otherwise waits for."""
        side = ops.side_stream(dev)

        def one_step():
            for p in params:
                p.grad = None
            cur = torch.cuda.current_stream(dev)
            start = cur.record_event()
            prob = model(Xk, presorted=mine)["y_pred"]        # (the re-zero of the previous step's rows goes to `side` first)
            side.wait_event(start)
            with torch.cuda.stream(side):
                model.presort(X_next, into=nxt)
            loss = loss_fn(prob, yk, reduction="mean")
            ops.backward(loss)
            cur.wait_stream(side)
            return loss
        return one_step

    def overflowed():
        flag = model.tables.overflow.float().reshape(1).clone()
        if world > 1:
            torch.distributed.all_reduce(flag)
        return bool(flag.item() > 0)

    step = eager_step
    graph_note = "eager launches"
    rotating_graphs = None
    steps_per_launch = 1
    if not args.eager and not sharded:
        # one hipGraph holds the whole step (same kernels, same C ABI); the batch lives in static buffers
        from recbox_amd.graph import GraphedStep
        try:
            if K > 1 and not args.rotate_by_copy:
                # K distinct batches stay resident in HBM, each with a captured step of its own (the graphs share their
                # intermediate memory and the model's persistent gradient buffer); the timed loop replays them in rotation:
                # every step gathers rows of another batch and nothing is copied inside the timed region
                rotating_graphs = []
                inputs_of = []
                for k in range(K):
                    if args.contiguous_ids:               # batches[k] is [40, B]: a column of the batch is a row here
                        Xk = OrderedDict((name, batches[k][fmw.fm.get_column_index(name)]) for name in fmw.fm.features)
                        yk = labels[k]
                    else:
                        Xk, yk = slice_inputs(fmw.fm, batches[k])
                    inputs_of.append((Xk, yk))
                prefetch = args.path == "fused" and K >= 3 and args.prefetch_sort and ops.config.reuse_grad_buffers
                if prefetch:
                    sorted_ids = [model.presort(Xk) for Xk, _ in inputs_of]      # (also the primer of the first step)
                    for k in range(K):                    # the captured re-zero of step k clears the rows step k - 1 wrote
                        sorted_ids[k].previous = sorted_ids[k - 1]
                if args.steps_per_graph is None:
                    args.steps_per_graph = 4 if (not prefetch and K % 4 == 0 and args.steps % 4 == 0) else 1
                S = max(1, int(args.steps_per_graph))
                if S > 1 and (prefetch or K % S or args.steps % S):
                    raise SystemExit("--steps-per-graph must divide --rotate and --steps (and excludes --prefetch-sort)")
                steps_per_launch = S
                for k in range(0, K, S):
                    if S > 1:
                        fns = [step_over(*inputs_of[k + j]) for j in range(S)]

                        def fn(fns=fns):
                            out = None
                            for f in fns:
                                out = f()
                            return out
                    else:
                        Xk, yk = inputs_of[k]
                        fn = (prefetching_step(Xk, yk, sorted_ids[k], inputs_of[(k + 1) % K][0], sorted_ids[(k + 1) % K])
                              if prefetch else step_over(Xk, yk))
                    rotating_graphs.append(GraphedStep(fn, warmup=3 if k == 0 else 2,
                                                       reuse_grads=ops.config.reuse_grad_buffers, params=params,
                                                       pool=rotating_graphs[0].pool() if rotating_graphs else None))
                step = rotating_graphs[0]
                graph_note = "hipGraph replay (one captured step per resident batch%s)" % (
                    "; the id sort of batch i + 1 runs beside step i, as with a loader one batch ahead" if prefetch else "")
                if S > 1:
                    graph_note = "hipGraph replay (%d consecutive steps, one per resident batch, per captured graph)" % S
            else:
                step = GraphedStep(eager_step, warmup=3, reuse_grads=ops.config.reuse_grad_buffers)
                graph_note = "hipGraph replay"
        except Exception as exc:               # a capture this stack refuses: the same step, launched from Python
            rotating_graphs = None
            print("[bench] hipGraph capture failed (%s: %s); launching the step eagerly" % (type(exc).__name__, exc),
                  file=sys.stderr)
            torch.cuda.synchronize()
            step = eager_step
    elif sharded and cap_factor and model.tables is not None:
        # padded sync-free exchange: the step is eight hipGraph pieces with the RCCL collectives between them
        from recbox_amd.graph import ShardedFMStep
        use_graphs = {"auto": "auto", "whole": "whole", "pieces": True, "eager": False}[args.sharded_graph]
        if args.eager:
            use_graphs = False
        for _ in range(4):
            try:
                step = ShardedFMStep(model, X, y, graphs=use_graphs)
            except Exception as exc:           # a capture that this stack refuses: same pieces, launched eagerly
                if not use_graphs:
                    raise
                if rank == 0:
                    print("[bench] graph capture of the sharded step failed (%s: %s); running the pieces eagerly"
                          % (type(exc).__name__, exc), file=sys.stderr)
                torch.cuda.synchronize()
                use_graphs = False
                step = ShardedFMStep(model, X, y, graphs=False)
            step()
            if not overflowed():
                break
            cap_factor *= 2                   # skewed ids: some owner received more than its slots; start over
            step.release()
            del step
            model = build_model()
        where = "rbx_all_to_all / rbx_all_reduce on the step's stream" if comm.direct.on else "torch.distributed"
        if step.whole is not None:
            graph_note = "ONE hipGraph: every piece and the four RCCL collectives (%s)" % where
        elif step.graphs is not None:
            graph_note = "8 hipGraph pieces + RCCL collectives between them (%s)" % where
        else:
            graph_note = "eager launches"

    def run_step(i):
        if rotating_graphs is not None:
            if i % steps_per_launch == 0:
                rotating_graphs[(i // steps_per_launch) % len(rotating_graphs)]()
        else:
            refill(i)
            step()

    warmup_done = args.warmup
    if steps_per_launch > 1:                  # (whole launches only: the warm-up rounds up)
        warmup_done = (args.warmup + steps_per_launch - 1) // steps_per_launch * steps_per_launch
    for i in range(warmup_done):
        run_step(i)
    # dominant kernel = the embedding gather: fm_fused_fwd (fused path) or the [B, 39, 16] embed_fwd (layer path)
    if args.path == "fused" or sharded:
        timer = ops.KernelTimer(lambda m: m[0] == "fm_fwd")
    else:
        timer = ops.KernelTimer(lambda m: m[0] == "embed_fwd" and m[2] == n_fields * args.dim)
    warm_timer = ops.KernelTimer(timer.want)
    alone_timer = ops.KernelTimer(timer.want)
    plain_eager = step is eager_step or (sharded and graph_note == "eager launches")
    if plain_eager:
        ops.kernel_timer = timer
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(warmup_done + i)
    host_el = time.perf_counter() - t0         # when the host has enqueued everything (graph launches run ahead of the GPU)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if not plain_eager:
        # a graph replay cannot be bracketed from Python: time the dominant kernel with HIP events
        # on the launch stream over the same steps launched eagerly right after the timed region
        # Launched eagerly the step is host-bound (the GPU idles between kernels and the kernel would be timed on a
        # drained, down-clocked device), so the stream is first loaded with ~15 ms of fills: the eager launches then
        # queue up behind them and run back to back, as in the replay.  Events are in-stream: they bracket the kernel only.
        # (on the stream the captured steps' autograd nodes live on: a backward on another stream than a parameter's
        #  AccumulateGrad node makes the engine sync the two and warn)
        import contextlib
        first = rotating_graphs[0] if rotating_graphs else step
        on_stream = torch.cuda.stream(first.stream) if hasattr(first, "stream") else contextlib.nullcontext()
        # (these eager steps run AFTER the timed region, only to put HIP events around the dominant kernel; the captured
        #  graphs still hold the autograd nodes of their capture, so the engine reports a stream mismatch for them: expected
        #  here, and silenced here only)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        with on_stream:
            n_timed = min(args.steps, 10)
            ballast = torch.empty(1 << 28, dtype=torch.float32, device=dev)          # 1 GiB
            for _ in range(80):
                ballast.zero_()
            ops.kernel_timer = timer
            for i in range(n_timed):
                refill(args.warmup + args.steps + i)
                eager_step()
            torch.cuda.synchronize()
            if ops.config.sort_before_forward and not sharded:
                # the same kernel WITHOUT the id sort / re-zero of the backward running beside it (they are enqueued after
                # the forward instead): frac_alone, the kernel's own rate
                for _ in range(80):
                    ballast.zero_()
                ops.config.sort_before_forward = False
                ops.kernel_timer = alone_timer
                for i in range(n_timed):
                    refill(args.warmup + args.steps + n_timed + i)
                    eager_step()
                torch.cuda.synchronize()
                ops.config.sort_before_forward = True
            if K > 1:
                # the same kernel on ONE batch replayed (rows of the previous launch still in the Infinity Cache): frac_warm
                for _ in range(80):
                    ballast.zero_()
                ops.kernel_timer = warm_timer
                for i in range(n_timed):
                    eager_step()
                torch.cuda.synchronize()
        del ballast
    ops.kernel_timer = None
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    overflow = overflowed() if (sharded and model.tables is not None) else False
    if rank == 0:
        ms = el / args.steps * 1e3
        # algorithmic bytes of the gather per sample: 26 rows x 64 B
        # + 26 ids x 8 B (float64 columns) + 13 dense values x 8 B + the [39,16] fp32 slot written
        n_sparse = len(CRITEO_VOCABS)
        row_stream = n_sparse * args.dim * 4                # the table rows alone: what north_star's 60 % target is quoted on
        per_sample_incl_s = None
        if args.path == "fused" or sharded:
            # rows + ids (8 B each) + LR rows (26 x 4) + dense inputs (13 x 4) + the 4-byte logit = 2 032 B at
            # D = 16.  What the kernel really moves on top of that -- the dense columns arrive as float64 (+52 B) and the S row
            # it keeps for the backward (+64 B) -- is reported beside it as frac_incl_S, not in `frac`.
            per_sample = n_sparse * args.dim * 4 + n_sparse * 8 + n_sparse * 4 + N_DENSE * 4 + 4
            per_sample_incl_s = per_sample + N_DENSE * 4 + args.dim * 4
            # the flat float64 batch tensor at dim 16 runs rbx_fm_quad.hip's kernel (10 column slots per lane for the 39
            # features, 20 rows in flight, 3 = RBX_F64); anything else -- or ops.fm_quad_kernel(False) -- the general one
            quad = args.dim == 16 and not sharded and ops.fm_quad_kernel()
            kname = ("fm_quad_fwd_kernel<%d,20,3>" % ((n_fields + 3) // 4)) if quad else \
                    ("fm_fused_fwd_kernel<%d,1,true,3>" % (args.dim // 4))
        else:
            per_sample = n_sparse * args.dim * 4 + n_sparse * 8 + N_DENSE * 4 + n_fields * args.dim * 4
            kname = "embed_fwd_kernel<%d,1,true>" % (args.dim // 4)
        kms = timer.mean_ms()
        roof = None
        if kms:
            achieved = per_sample * B / (kms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0,
                    "traffic": measured_traffic(args.path, kname) if (B == 65536 and args.dim == 16) else None,
                    "kernel": kname,
                    "kernel_ms": kms, "algorithmic_bytes_per_launch": per_sample * B,
                    "algorithmic_bytes_per_sample": per_sample,
                    "frac_row_stream": row_stream * B / (kms * 1e-3) / 1e9 / 8000.0,
                    "inputs": ("%d distinct batches in rotation" % K) if K > 1 else "one batch replayed"}
            if per_sample_incl_s is not None:
                roof["frac_incl_S"] = per_sample_incl_s * B / (kms * 1e-3) / 1e9 / 8000.0
            wms = warm_timer.mean_ms()
            if wms:
                roof["frac_warm"] = per_sample * B / (wms * 1e-3) / 1e9 / 8000.0
                roof["kernel_ms_warm"] = wms
            if args.dim == 16 and (args.path == "fused" or sharded):
                # what this GPU delivers on random 64-byte rows at all (a kernel that only gathers them): the ceiling the
                # D = 16 forward can be held against; 128-byte rows and wider reach 5.8-6.0 TB/s
                # NOT a ceiling of this workload: 14 of the 26 tables are cache-resident.  What bounds the kernel
                # is the CU's vector memory path -- profiles/r05/fm_fwd_limiter.md
                roof["limiter"] = ("per-CU vector memory path: ~64 L1 misses in flight x 550 cycles of L2 / fabric latency "
                                   "(TCP_TCC_READ_REQ / _LATENCY), one tag lookup per cycle, issue slots: "
                                   "profiles/r05/fm_fwd_limiter.md; with every lookup hitting L1 the kernel takes 16 us; "
                                   "a scalar-cache prefetch path, balanced occupancy forms and the translation counters "
                                   "change nothing: profiles/r06/fm_fwd_round6.md")
            ams = alone_timer.mean_ms()
            if ams:
                roof["frac_alone"] = per_sample * B / (ams * 1e-3) / 1e9 / 8000.0
                roof["kernel_ms_alone"] = ams
                roof["note"] = ("frac / kernel_ms: the kernel as it runs in the step, beside the large tables' id sort on a "
                                "second stream (recbox_amd.ops.config.sort_before_forward; since round 3 the re-zero of the "
                                "previous step's rows runs in front of the forward, not beside it); frac_alone / "
                                "kernel_ms_alone: the same launches with the sort enqueued after the forward")
        out = {"metric": "samples/sec fwd+bwd, Criteo-shaped batch 65 536; embedding HBM GB/s vs roofline",
               "value": B * world * args.steps / el, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "warmup_done": warmup_done, "steps_per_graph": steps_per_launch,
               "ms_per_step": ms, "host_enqueue_ms_per_step": host_el / args.steps * 1e3, "higher_is_better": True,
               "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "FM (recbox.ranking) Criteo-shaped 26 sparse + 13 dense, dim %d, batch %d per GPU, "
                                      "%s ids (%s), %s path%s, %s, dense-grad autograd contract (%s), no optimiser step"
                                      % (args.dim, B, args.dist,
                                         (("%d distinct batches resident in HBM, replayed in rotation" % K)
                                          if rotating_graphs is not None else
                                          ("%d distinct batches rotated, one copy_ per step in the timed region" % K))
                                         if K > 1 else "one batch replayed", args.path,
                                         " (emb | LR rows packed in one [V, 32] storage per table pair)"
                                         if (not sharded and args.path == "fused" and args.pack_tables) else "", graph_note,
                                         "persistent grad buffer, rows of the previous step re-zeroed"
                                         if ops.config.reuse_grad_buffers else "fresh zero-filled grads every step"),
                          "global_batch": B * world, "batch_per_gpu": B,
                          "parallelism": ("dp%d + row-sharded tables (all-to-all-v)" % world) if sharded else "dp1"},
               "roofline": roof}
        if sharded:
            out["config"]["exchange"] = ("padded capacity_factor=%g, overflow=%s" % (cap_factor, overflow)
                                         if cap_factor else "exact all-to-all-v")
            if overflow:
                out["config"]["warning"] = "exchange capacity overflowed: rerun with a larger --capacity-factor"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.dim, B, args.dist, args.cpu_seconds)
        if world == 1 and not sharded and not args.no_cpu_baseline and not args.no_extra_configs:
            out["configs"] = extra_configs(args)
        print(json.dumps(out))
    if world > 1 or args.force_sharded:
        shutdown_distributed(step)


def extra_configs(args):
    """BASELINE.json configs[2..4] beside the headline line: each one is `python bench.py --config C` in a process of its
    own (its tables -- 5 GB for the YoutubeDNN item table, plus the same again of dense gradient -- do not share this
    process's allocator), time-boxed; the full JSON line of the sub-run (ms_per_step, value, roofline of ITS dominant kernel,
    cpu_baseline) goes under its name.  A sub-run that fails or runs out of time leaves {"skipped": reason}."""
    import subprocess
    res = {}
    budget = float(os.environ.get("RECBOX_BENCH_EXTRA_SECONDS", "75"))
    # "fm_fresh_grads": the headline workload under autograd's literal contract -- new zero-filled dense gradients every step
    # ("report with and without" the 357 MB fill) -- beside the persistent-buffer headline
    # "fm_zipf": SURVEY.md 8(d) asks for both id distributions; the headline is the worst-case one (uniform)
    runs = [("fm_one_step_per_graph", ["--config", "fm", "--steps-per-graph", "1", "--no-cpu-baseline", "--no-extra-configs"]),
            ("fm_zipf", ["--config", "fm", "--no-cpu-baseline", "--no-extra-configs", "--dist", "zipf"]),
            ("fm_fresh_grads", ["--config", "fm", "--fresh-grads", "--no-cpu-baseline", "--no-extra-configs"]),
            ("youtubednn", ["--config", "youtubednn"]), ("deepfm", ["--config", "deepfm"]), ("sasrec", ["--config", "sasrec"])]
    for cfg, extra in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(min(args.steps, 20)),
               "--warmup", str(min(args.warmup, 5)), "--cpu-seconds", "6", "--dist", args.dist] + extra
        t0 = time.perf_counter()
        try:
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=budget)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
            if proc.returncode != 0 or not line:
                res[cfg] = {"skipped": "exit code %d: %s" % (proc.returncode, proc.stderr.strip().splitlines()[-1:] or "")}
            else:
                d = json.loads(line[-1])
                res[cfg] = {k: d[k] for k in ("ms_per_step", "value", "unit", "steps", "warmup", "dtype", "dtype_note", "config",
                                              "roofline", "cpu_baseline") if k in d}
                res[cfg]["wall_s"] = round(time.perf_counter() - t0, 1)
        except subprocess.TimeoutExpired:
            res[cfg] = {"skipped": "did not finish within %.0f s" % budget}
        except Exception as e:                                      # noqa: BLE001 (a broken sub-run must not cost the headline line)
            res[cfg] = {"skipped": "%s: %s" % (type(e).__name__, e)}
    return res


if __name__ == "__main__":
    main()
