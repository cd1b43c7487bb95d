"""Two / three ranks of the row-sharded FM step on ONE GPU (processes sharing cuda:0, rendezvous over gloo with the rows
staged through the host): the HIP pieces of the N>1 path -- rbx_route wire slots for world > 1, owner-side gather,
the fused FM over exchanged rows, dY to the owners, owner-side sorted scatter-add, flat all-reduce of the replicated
gradients -- against the single-GPU FM trained on the GLOBAL batch.  RCCL itself needs one GPU per rank and is
covered by the driver's multi-GPU bench; everything around the collective is exercised here."""
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

VOCABS = [50, 7, 400, 31, 301, 9]
D, B = 16, 257                                    # per-rank batch (odd: ragged last blocks everywhere)
E_PRE = "embedding_layer.embedding_layer.embedding_layers.%s.weight"
L_PRE = "fm.lr_layer.embedding_layer.embedding_layer.embedding_layers.%s.weight"


def _worker(rank, world, port, mode, result):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from conftest import assert_close
        from recbox_amd import ops
        from recbox_amd.graph import ShardedFMStep
        from recbox_amd.ranking.pytorch.models import FM, ShardedFM
        from test_gpu_ranking import _criteo_like, _cuda
        torch.cuda.set_device(0)
        fm, X, y = _criteo_like(world * B, VOCABS, D, seed=31, zipf=True)       # the GLOBAL batch, same on every rank
        torch.manual_seed(3)
        ref = FM(fm, D).cuda()
        with torch.no_grad():
            for p in ref.parameters():
                p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.1)
        factor = None if mode == "exact" else 2.0
        dut = ShardedFM(fm, D, shard_min_vocab=300, capacity_factor=factor).cuda()
        assert dut.sharded_names == ["C3", "C5"] and dut.world_size == world
        sd = ref.state_dict()
        dut.load_state_dict({k: v for k, v in sd.items() if not any(("." + n + ".") in k for n in dut.sharded_names)},
                            strict=False)
        dut.tables.load_full_tables([sd[E_PRE % n] for n in dut.sharded_names], [sd[L_PRE % n] for n in dut.sharded_names])
        Xg, yg = _cuda(X), y.cuda()
        loss_ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(ref.logits(Xg)), yg, reduction="mean")
        loss_ref.backward()
        mine = slice(rank * B, (rank + 1) * B)
        Xr = type(Xg)((k, v[mine].contiguous()) for k, v in Xg.items())
        yr = yg[mine].contiguous()
        if mode == "exact":                      # autograd path, all-to-all-v with exchanged counts
            loss = torch.nn.functional.binary_cross_entropy(dut(Xr)["y_pred"], yr, reduction="mean")
            (loss / world).backward()
            dut.sync_grads()
        else:                                    # the bench's step: padded sync-free exchange; eager pieces, or hipGraph
                                                 # pieces with the collectives between them ("padded_graphs")
            old = ops.config.check_ids
            ops.config.check_ids = False
            try:
                step = ShardedFMStep(dut, Xr, yr, graphs=(mode == "padded_graphs"))
                for _ in range(2):
                    loss = step()
            finally:
                ops.config.check_ids = old
            assert not bool(dut.tables.overflow)
        torch.cuda.synchronize()
        mean_loss = loss.detach().reshape(1).clone()
        dist.all_reduce(mean_loss)
        assert_close(mean_loss / world, loss_ref.reshape(1), 1e-6, "loss")
        ref_g = dict((n, p.grad) for n, p in ref.named_parameters())
        for n, p in dut.named_parameters():
            if n.startswith("tables."):
                continue
            assert p.grad is not None, n
            assert_close(p.grad, ref_g[n], 1e-6, n)
        for t, name in enumerate(dut.sharded_names):
            sl, owned = dut.tables.local_rows_of(t)
            got = dut.tables.weight.grad[sl]
            keep = (owned != 0).cuda()           # row 0 is padding_idx in FM, an ordinary row in the shards
            assert_close(got[keep][:, :D], ref_g[E_PRE % name][owned.cuda()][keep], 1e-6, "shard emb " + name)
            assert_close(got[keep][:, D], ref_g[L_PRE % name][owned.cuda()][keep][:, 0], 1e-6, "shard lr " + name)
        result.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def _spawn(worker, world, *extra):
    """mp.spawn with a fresh rendezvous port; ONE retry when the rendezvous itself failed (port taken between the probe
    and the bind, peer not reachable yet) -- assertion failures of the workers are never retried."""
    from test_distributed_gloo import _free_port
    for attempt in (0, 1):
        result = mp.get_context("spawn").SimpleQueue()
        try:
            mp.spawn(worker, args=(world, _free_port()) + extra + (result,), nprocs=world, join=True)
        except Exception as exc:                       # noqa: BLE001
            text = str(exc)
            net = any(k in text for k in ("Address already in use", "EADDRINUSE", "Connection refused", "Connection reset",
                                          "connect() timed out", "DistNetworkError", "DistStoreError"))
            if attempt == 0 and net and "AssertionError" not in text:
                continue
            raise
        got = {}
        while not result.empty():
            rank, status = result.get()
            got[rank] = status
        return got


@pytest.mark.parametrize("mode,world", [("exact", 2), ("padded", 2), ("padded", 3), ("padded_graphs", 2)])
def test_ranks_on_one_gpu_equal_single_gpu_fm(mode, world):
    assert _spawn(_worker, world, mode) == {r: "ok" for r in range(world)}


def _rccl_worker(rank, world, port, result):
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    try:
        from conftest import assert_close
        from recbox_amd import comm, ops
        from recbox_amd.graph import ShardedFMStep
        from recbox_amd.ranking.pytorch.models import ShardedFM
        from test_gpu_ranking import _criteo_like, _cuda
        torch.cuda.set_device(0)
        comm.direct.enable(True)
        assert comm.direct.self_check() is True            # == torch.distributed's all_to_all_single, sync and async
        x = torch.randn(4096, 20, device="cuda")
        out = torch.empty_like(x)
        comm.all_to_all_equal_into(out, x)
        assert torch.equal(out, x)                         # a world of one: block 0 comes back
        fm, X, y = _criteo_like(513, VOCABS, D, seed=23, zipf=True)
        Xc, yc = _cuda(X), y.cuda()
        grads = []
        ops.config.check_ids = False
        comm.force_world_of_one = True                     # the all-reduce of the replicated gradients is issued too
        steps = []
        for use_direct, mode in ((False, True), (True, True), (True, "whole")):
            comm.direct.enable(use_direct)
            torch.manual_seed(1)
            model = ShardedFM(fm, D, shard_min_vocab=300, capacity_factor=1.5).cuda()
            with torch.no_grad():
                for p in model.parameters():
                    p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.1)
            step = ShardedFMStep(model, Xc, yc, graphs=mode)
            if mode == "whole":
                assert comm.direct.capturable and step.whole is not None      # ONE hipGraph, the four collectives inside
            for _ in range(3):
                loss = step()
            torch.cuda.synchronize()
            grads.append([loss.clone()] + [p.grad.clone() for p in model.parameters()])
            steps.append(step)
        for other in grads[1:]:
            for a, b in zip(grads[0], other):
                assert torch.equal(a, b)
        # the sharded rechub model, whole step (exchanges + all-reduces inside) as one hipGraph == launched eagerly
        import torch.nn.functional as F
        from recbox_amd.graph import GraphedStep
        from recbox_amd.rechub.sharded import ShardedYoutubeDNN
        from test_gpu_shard import _rh, _seed_params, _youtube_batch, _youtube_feats
        Fe = _rh()
        V, Dm, B, L, n_neg = 1501, 16, 96, 7, 3
        got = []
        for graphed in (False, True):
            model = ShardedYoutubeDNN(*_youtube_feats(Fe, V, Dm), {"dims": [32, Dm]}, temperature=0.1, shard_min_vocab=500,
                                      capacity_factor=2.0).cuda()
            _seed_params(model)
            model.embedding.store.local_ops.persistent(model.embedding.store.weight)
            params = list(model.parameters())
            batches = [{k: v.cuda() for k, v in _youtube_batch(B, V, L, n_neg, 60 + k).items()} for k in range(2)]
            tgt = torch.zeros(B, dtype=torch.long, device="cuda")

            def step_over(x):
                def fn():
                    for p in params:
                        p.grad = None
                    loss = F.cross_entropy(model(x), tgt)
                    loss.backward()
                    model.sync_grads()
                    return loss
                return fn
            fns = [step_over(x) for x in batches]
            if graphed:
                fns = [GraphedStep(f, warmup=2, reuse_grads=False, params=params, capture_error_mode="thread_local")
                       for f in fns]
                steps.extend(fns)
            else:
                for f in fns:                                  # (the graphed model has taken its warm-up steps: same here)
                    f(); f()
            res = []
            for k in (0, 1, 0):
                loss = fns[k]()
                torch.cuda.synchronize()
                res.append([loss.detach().clone()] + [p.grad.clone() for p in params])
            got.append(res)
        for ra, rb in zip(*got):
            for a, b in zip(ra, rb):
                assert_close(a, b, 1e-6, "sharded YoutubeDNN: whole-step graph vs eager")
        for st in steps:                                       # graphs that hold RCCL nodes go before the communicator does
            st.release()
        torch.cuda.synchronize()
        result.put((rank, "ok"))
    finally:
        comm.force_world_of_one = False
        comm.direct.shutdown()
        dist.destroy_process_group()


def test_direct_rccl_exchange_equals_torch_distributed():
    """rbx_all_to_all / rbx_all_reduce (RCCL calls on the step's own stream through the communicator of the process group)
    in a world of one THROUGH RCCL: the same bytes as torch.distributed's collectives, eagerly and replayed from a hipGraph
    (comm.direct.self_check); a ShardedFMStep leaves bit-identical loss and gradients as hipGraph pieces on either path
    and as ONE hipGraph holding the collectives; the sharded YoutubeDNN step captured whole == launched eagerly; and the
    process group is destroyed cleanly once the graphs are released.  More than one rank needs one GPU each."""
    assert _spawn(_rccl_worker, 1) == {0: "ok"}


def _keepgrad_worker(rank, world, port, result):
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    try:
        import torch.nn.functional as F
        from conftest import assert_close
        from recbox_amd import comm
        from recbox_amd.rechub.sharded import ShardedYoutubeDNN
        from test_gpu_shard import _rh, _seed_params, _youtube_batch, _youtube_feats
        torch.cuda.set_device(0)
        comm.force_world_of_one = True
        Fe = _rh()
        V, Dm, B, L, n_neg = 1501, 16, 96, 7, 3
        model = ShardedYoutubeDNN(*_youtube_feats(Fe, V, Dm), {"dims": [32, Dm]}, temperature=0.1, shard_min_vocab=500,
                                  capacity_factor=2.0).cuda()
        _seed_params(model)
        assert model.grad_sync.bucket is not None              # the CUDA path: tower gradients are written into bucket views
        towers = model.grad_sync.early
        batches = [{k: v.cuda() for k, v in _youtube_batch(B, V, L, n_neg, 80 + k).items()} for k in range(2)]
        tgt = torch.zeros(B, dtype=torch.long, device="cuda")

        def backward(x):
            F.cross_entropy(model(x), tgt).backward()
            model.sync_grads()

        want = []
        for x in batches:                                      # reference: every step starts without gradients
            model.zero_grad(set_to_none=True)
            backward(x)
            torch.cuda.synchronize()
            want.append([p.grad.clone() for p in towers])
        # (a) zero_grad(set_to_none=False): p.grad stays the bucket view of the first step, zeroed in place
        model.zero_grad(set_to_none=True)
        for k, x in enumerate(batches + batches):
            backward(x)
            torch.cuda.synchronize()
            for p, w in zip(towers, want[k % 2]):
                assert_close(p.grad, w, 1e-6, "set_to_none=False, step %d" % k)
            model.zero_grad(set_to_none=False)
        # (b) accumulation: two backward + sync rounds without zero_grad
        model.zero_grad(set_to_none=True)
        backward(batches[0])
        backward(batches[1])
        torch.cuda.synchronize()
        for p, a, b in zip(towers, *want):
            assert_close(p.grad, a + b, 1e-6, "two backward + sync rounds without zero_grad")
        # (c) a new default group WITHOUT comm.direct.shutdown(): the collectives must find the new communicator
        x = torch.arange(64, dtype=torch.float32, device="cuda")
        comm.all_reduce_sum_(x, force=True)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (port + 1), rank=rank, world_size=world,
                                device_id=torch.device("cuda", 0))
        y = torch.arange(64, dtype=torch.float32, device="cuda")
        comm.all_reduce_sum_(y, force=True)
        out = torch.empty_like(y)
        comm.all_to_all_equal_into(out, y)
        torch.cuda.synchronize()
        assert torch.equal(y, x) and torch.equal(out, y)
        result.put((rank, "ok"))
    finally:
        comm.force_world_of_one = False
        comm.direct.shutdown()
        dist.destroy_process_group()


def test_surviving_gradients_and_a_recreated_group():
    """tower gradients written into DenseGradSync's bucket views stay right when ``p.grad`` survives a step
    (zero_grad(set_to_none=False): no 2 g from the second step on; accumulation over two backward + sync rounds = the
    sum), and the direct RCCL path follows a destroyed + re-created default process group without an explicit shutdown."""
    assert _spawn(_keepgrad_worker, 1) == {0: "ok"}


@pytest.mark.parametrize("how", ["driver", "bare_shell_strong"])
def test_bench_script_runs_its_two_rank_path(how):
    """bench.py with N = 2, both ranks on this box's single GPU over gloo (RECBOX_BENCH_ONE_GPU=1): the N>1 control flow of
    the script -- sharded model, piecewise-graphed step, overflow check, barrier + max-over-ranks timing, one JSON line
    from rank 0.  "driver": launched as the driver launches it (torch.distributed.run, one process per rank), weak
    scaling.  "bare_shell_strong": ``python bench.py --gpus 2 --scaling strong`` from a bare shell -- the script starts its
    own ranks, and the global batch is split B / N per GPU; ``n_gpus`` is the number of ranks that ran."""
    import json
    import os
    import subprocess
    import sys
    from test_distributed_gloo import _free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RECBOX_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if how == "driver":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
               "--warmup", "1", "--batch", "8192"]
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling",
               "strong", "--global-batch", "16384"]
    out = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["scaling"] == ("weak" if how == "driver" else "strong")
    assert rec["value"] > 0 and rec["config"]["global_batch"] == 2 * 8192 and rec["config"]["batch_per_gpu"] == 8192
    assert "row-sharded" in rec["config"]["parallelism"] and "overflow=False" in rec["config"]["exchange"]
    # a launch whose WORLD_SIZE disagrees with --gpus is refused instead of printing a line for another N
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=120) if how == "driver" else None
    if bad is not None:
        assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


# ---- BASELINE.json configs 3 and 4 in their multi-GPU form: ranks sharing ONE GPU over gloo ------------------------------
def _bn_eval(model):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.eval()


def _model_worker(rank, world, port, which, bn_mode, result):
    """``which`` in {"youtubednn", "deepfm"}: the sharded / data-parallel model on this rank's slice against the
    single-GPU mirror on the GLOBAL batch.  bn_mode "eval": BatchNorm uses its running statistics, so the towers do not
    depend on how the batch is split and the two must agree to 1e-5 (loss, every gradient, the shard's rows of the table
    gradients).  bn_mode "replica": training-mode BatchNorm with PER-REPLICA statistics (the reference's nn.DataParallel
    semantics, ctr_trainer.py:43): the reference is then the single-GPU model run on every slice in turn, gradients
    averaged."""
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from conftest import assert_close
        from test_gpu_shard import _rh, _seed_params, _youtube_batch, _youtube_feats
        torch.cuda.set_device(0)
        Fe = _rh()
        B = 96                                           # per rank
        if which == "youtubednn":
            from recbox_amd.rechub.models.matching import YoutubeDNN
            from recbox_amd.rechub.sharded import ShardedYoutubeDNN
            V, D, L, n_neg = 1501, 16, 7, 3
            ref = YoutubeDNN(*_youtube_feats(Fe, V, D), {"dims": [32, D]}, temperature=0.1).cuda()
            _seed_params(ref)
            dut = ShardedYoutubeDNN(*_youtube_feats(Fe, V, D), {"dims": [32, D]}, temperature=0.1, shard_min_vocab=500,
                                    capacity_factor=2.0).cuda()
            sharded = ["item"]
            xg = {k: v.cuda() for k, v in _youtube_batch(world * B, V, L, n_neg, 11).items()}
            target = torch.zeros(world * B, dtype=torch.long, device="cuda")

            def loss_of(model, x, sl):
                return F.cross_entropy(model(x), target[sl])
        else:
            from recbox_amd.rechub.models.ranking import DeepFM
            from recbox_amd.rechub.sharded import ShardedDeepFM
            vocabs, D = [7, 900, 31, 1200, 5, 640], 16

            def feats():
                dense = [Fe.DenseFeature("I%d" % i) for i in range(3)]
                sparse = [Fe.SparseFeature("C%d" % i, v, D) for i, v in enumerate(vocabs)]
                return dense + sparse, sparse

            mlp = {"dims": [24, 16], "dropout": 0.0, "activation": "relu"}
            ref = DeepFM(*feats(), mlp).cuda()
            _seed_params(ref)
            dut = ShardedDeepFM(*feats(), mlp, shard_min_vocab=600, capacity_factor=2.0).cuda()
            sharded = ["C1", "C3", "C5"]
            g = torch.Generator().manual_seed(13)
            xg = {"I%d" % i: torch.rand(world * B, generator=g).cuda() for i in range(3)}
            for i, v in enumerate(vocabs):
                xg["C%d" % i] = torch.randint(0, v, (world * B,), generator=g).cuda()
            target = (torch.rand(world * B, generator=g) < 0.3).float().cuda()

            def loss_of(model, x, sl):
                return F.binary_cross_entropy(model(x), target[sl])
        assert dut.embedding.sharded_tables == sharded and dut.world_size == world
        sd = ref.state_dict()
        full = [sd.pop("embedding.embed_dict.%s.weight" % n) for n in sharded]
        missing, unexpected = dut.load_state_dict(sd, strict=False)
        assert missing == ["embedding.store.weight"] and not unexpected
        dut.embedding.store.load_full_tables(full)
        if bn_mode == "eval":
            _bn_eval(ref)
            _bn_eval(dut)
            loss_ref = loss_of(ref, xg, slice(None))
            loss_ref.backward()
        else:
            loss_ref = 0.0
            for r in range(world):
                sl = slice(r * B, (r + 1) * B)
                part = loss_of(ref, {k: v[sl] for k, v in xg.items()}, sl) / world
                part.backward()
                loss_ref = loss_ref + part.detach()
        mine = slice(rank * B, (rank + 1) * B)
        loss = loss_of(dut, {k: v[mine].contiguous() for k, v in xg.items()}, mine)
        (loss / world).backward()
        dut.sync_grads()
        torch.cuda.synchronize()
        assert not bool(dut.embedding.store.overflow)
        mean_loss = (loss.detach() / world).reshape(1).clone()
        dist.all_reduce(mean_loss)
        assert_close(mean_loss, loss_ref.reshape(1), 1e-5, "loss")
        want = dict(ref.named_parameters())
        store = dut.embedding.store
        for n, p in dut.named_parameters():
            if n == "embedding.store.weight":
                for t, name in enumerate(sharded):
                    sl, owned = store.local_rows_of(t)
                    assert_close(p.grad[sl], want["embedding.embed_dict.%s.weight" % name].grad[owned.cuda()], 1e-5,
                                 "shard of " + name)
            else:
                assert p.grad is not None, n
                assert_close(p.grad, want[n].grad, 1e-5, "grad " + n)
        result.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which,bn_mode,world", [("youtubednn", "eval", 2), ("youtubednn", "replica", 2),
                                                 ("youtubednn", "eval", 3), ("deepfm", "eval", 2),
                                                 ("deepfm", "replica", 2), ("deepfm", "eval", 3)])
def test_sharded_models_on_one_gpu_equal_single_gpu_models(which, bn_mode, world):
    """cfg 3 (YoutubeDNN, item table row-sharded, history pooled at the owners) and cfg 4 (DeepFM data-parallel, large
    tables sharded) with 2 / 3 ranks: every HIP piece of the N>1 step with a real exchange (rows staged through the host
    over gloo) == the single-GPU model on the global batch."""
    assert _spawn(_model_worker, world, which, bn_mode) == {r: "ok" for r in range(world)}


# ---- SyncBatchNorm: the statistics of one normalisation come from the batches of all ranks ------------------------
def _syncbn_worker(rank, world, port, result):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from conftest import assert_close
        from recbox_amd.rechub.basic.layers import MLP
        torch.cuda.set_device(0)
        rows, cols = 96, 40                                    # per-rank rows
        g = torch.Generator().manual_seed(5)
        X = (torch.randn(world * rows, cols, generator=g) * 2 + 1).cuda()              # the GLOBAL batch, same on every rank
        R = torch.randn(world * rows, 1, generator=g).cuda()
        torch.manual_seed(1)
        ref = MLP(cols, output_layer=True, dims=[24, 16], activation="relu").cuda().train()          # BatchNorm1d: global batch
        dut = MLP(cols, output_layer=True, dims=[24, 16], activation="relu").cuda()
        dut.load_state_dict(ref.state_dict())
        dut = torch.nn.SyncBatchNorm.convert_sync_batchnorm(dut).train()          # RecBole's conversion (trainer.py:60-64)
        assert sum(isinstance(m, torch.nn.SyncBatchNorm) for m in dut.modules()) == 2
        xg = X.clone().requires_grad_()
        out_ref = ref(xg)                                       # (one training forward each: the running statistics move once)
        (out_ref * R).sum().backward()
        mine = slice(rank * rows, (rank + 1) * rows)
        xr = X[mine].clone().requires_grad_()
        out = dut(xr)
        assert_close(out, out_ref[mine].detach(), 1e-5, "SyncBatchNorm output == BatchNorm over the global batch")
        (out * R[mine]).sum().backward()
        assert_close(xr.grad, xg.grad[mine], 1e-5, "dx")
        for (n, p), (_, q) in zip(dut.named_parameters(), ref.named_parameters()):
            gsum = p.grad.clone()
            dist.all_reduce(gsum)                               # the job's gradient all-reduce (sum)
            assert_close(gsum, q.grad, 2e-5, "grad " + n)
        ref_b, dut_b = dict(ref.named_buffers()), dict(dut.named_buffers())
        for n, b in ref_b.items():                              # running statistics: those of the global batch, on every rank
            if b.dtype.is_floating_point:
                assert_close(dut_b[n], b, 1e-5, n)
        dut.eval()                                              # eval mode: running statistics, no collective
        assert_close(dut(X[mine]), ref.eval()(X[mine]), 1e-5, "eval")
        result.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sync_batch_norm_over_ranks_equals_batch_norm_over_the_global_batch(world):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) on a rechub MLP: outputs, dx, parameter gradients (summed over the
    ranks) and running statistics equal BatchNorm1d on the global batch; rbx_batchnorm_stats / _apply / _bwd_reduce / _bwd_dx."""
    assert _spawn(_syncbn_worker, world) == {r: "ok" for r in range(world)}
