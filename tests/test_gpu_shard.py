"""GPU parity of the sharded exchange's kernels (csrc/rbx_shard.hip, rbx_embed_bwd_indexed) against their plain-torch
restatement (tests/shard_oracle.py): wire slots, offsets and row numbers bit for bit, partial sums / placed rows /
gradients to 1e-6; then ShardedStore / ShardedEmbeddingLayer / the sharded model mirrors in a world of one against
the single-GPU layers, and property tests at BASELINE.json's cfg-3 size (10 M x 128 table, B = 65 536)."""
import pytest
import torch

from conftest import assert_close
from shard_oracle import OracleShardOps

pytestmark = pytest.mark.gpu


def _case(B, L, T, W, D, vocabs, seed, fill=1.0, dtype=torch.int64):
    from recbox_amd.sharded import ShardCall
    g = torch.Generator().manual_seed(seed)
    tables = [int(torch.randint(0, len(vocabs), (1,), generator=g)) for _ in range(T)]
    offs, off = [], D if L else 0
    for _ in range(T):
        offs.append(off)
        off += D
    call = ShardCall(list(zip(tables, offs)), pool=(0, 0, L, "mean", 0, 1e-16) if L else None)
    rows = [torch.randint(0, vocabs[t], (B,), generator=g).to(dtype) for t in tables]
    hist = None
    if L:
        lens = (torch.rand(B, generator=g) * fill * (L + 1)).long().clamp(max=L)
        hist = (torch.randint(1, vocabs[0], (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])).to(dtype)
    return call, rows, hist, off


@pytest.mark.parametrize("B,L,T,W,D,dtype", [(257, 7, 3, 2, 16, torch.int64), (1, 5, 1, 3, 8, torch.int32),
                                            (1000, 50, 5, 8, 128, torch.int64), (513, 0, 8, 4, 64, torch.float64),
                                            (300, 9, 0, 5, 32, torch.int64), (2049, 3, 2, 1, 4, torch.float32),
                                            (64, 200, 2, 8, 256, torch.int64)])
def test_route_serve_combine_match_restatement(B, L, T, W, D, dtype):
    from recbox_amd.sharded import HipShardOps, ShardedStore
    vocabs = [5003, 97, 1201]
    call, rows, hist, width = _case(B, L, T, W, D, vocabs, seed=B + L + T + W, dtype=dtype)
    geom = call.geometry(W, D, B, 1.5)
    hip, ora = HipShardOps(), OracleShardOps()
    counts = torch.tensor([[(v - r + W - 1) // W for v in vocabs] for r in range(W)])
    base_all = torch.zeros_like(counts)
    base_all[:, 1:] = counts.cumsum(1)[:, :-1]
    tbl = [t for t, _ in call.rows] + ([call.pool[0]] if call.pool else [])
    base = base_all[:, tbl].contiguous().cuda()
    rows_c = [r.cuda() for r in rows]
    hist_c = hist.cuda() if hist is not None else None
    ov0, ov1 = torch.zeros((), dtype=torch.uint8, device="cuda"), torch.zeros((), dtype=torch.uint8, device="cuda")
    st0, st1 = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    send, slot, inv = hip.route(geom, call, rows_c, hist_c, vocabs, base, ov0, st0)
    send_w, slot_w, inv_w = ora.route(geom, call, rows_c, hist_c, vocabs, base, ov1, st1)
    torch.cuda.synchronize()
    assert int(st0) == 0 and int(st1) == 0 and int(ov0) == int(ov1) == 0
    assert torch.equal(send, send_w), "wire requests differ"
    if T:
        assert torch.equal(slot, slot_w)
    if L:
        assert torch.equal(inv, inv_w)
    # the owner: pretend chunk w of `send` came from source w (every chunk is a valid request list)
    n_local = int(counts.sum(dim=1).max())
    weight = torch.randn(n_local, D, generator=torch.Generator().manual_seed(1)).cuda()
    back, keys, src = hip.serve(geom, send, weight, st0)
    back_w, keys_w, src_w = ora.serve(geom, send_w, weight, st1)
    torch.cuda.synchronize()
    assert torch.equal(keys, keys_w)
    live = keys_w >= 0
    assert torch.equal(src[live], src_w[live])
    used = torch.zeros(W * geom.frows, dtype=torch.bool, device="cuda")
    used[src_w[live].long()] = True
    if L:
        used.view(W, geom.frows)[:, :B] = True                    # every partial sum is written (zeros when empty)
    assert_close(back[used], back_w[used], 1e-6, "served rows")
    # requester: place into the block / pack the gradient
    out = torch.full((B, width), 7.0, device="cuda")
    out_w = out.clone()
    hip.combine_fwd(geom, call, back_w, slot_w, inv_w, out)
    ora.combine_fwd(geom, call, back_w, slot_w, inv_w, out_w)
    assert_close(out, out_w, 1e-6, "placed rows")
    dout = torch.randn(B, width, generator=torch.Generator().manual_seed(2)).cuda()
    gs = hip.combine_bwd(geom, call, dout, slot_w, inv_w)
    gs_w = ora.combine_bwd(geom, call, dout, slot_w, inv_w)
    sent = torch.zeros(W * geom.frows, dtype=torch.bool, device="cuda")
    if L:
        sent.view(W, geom.frows)[:, :B] = True
    if T:
        s = slot_w.reshape(-1).long()
        s = s[s < W * geom.cap_rows]
        w = s // geom.cap_rows
        sent[w * geom.frows + (1 if L else 0) * B + (s - w * geom.cap_rows)] = True
    assert_close(gs[sent], gs_w[sent], 1e-6, "packed gradient")
    # owner-side scatter-add through the sorted segmented reduce, with and without the early sort
    grecv = torch.randn(W * geom.frows, D, generator=torch.Generator().manual_seed(3)).cuda()
    want = ora.scatter(weight, keys_w, src_w, grecv)
    got = hip.scatter(weight, keys_w, src_w, grecv)
    got2 = hip.scatter(weight, keys_w, src_w, grecv, sorted_ws=hip.presort(weight, keys_w))
    assert torch.equal(got, got2), "early sort must not change a bit"
    assert_close(got, want, 1e-5, "shard gradient")
    del ShardedStore


def test_route_flags_overflow_and_bad_ids():
    from recbox_amd.sharded import HipShardOps, ShardCall
    hip = HipShardOps()
    W, B, D, L = 4, 256, 8, 6
    call = ShardCall([(0, D)], pool=(0, 0, L, "sum", 0, 0.0))
    geom = call.geometry(W, D, B, 1.0)
    base = torch.zeros((W, 2), dtype=torch.long, device="cuda")
    ov, st = torch.zeros((), dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    rows = [torch.full((B,), 8, device="cuda")]                          # every id -> owner 0: 256 > capacity 64
    hist = torch.full((B, L), 4, device="cuda")
    send, slot, inv = hip.route(geom, call, rows, hist, [100], base, ov, st)
    assert int(ov) == 1 and int(st) == 0
    assert int((slot == W * geom.cap_rows).sum()) == B - geom.cap_rows   # the rest went to the dump slot
    offs = send.view(W, geom.ichunk)[0, geom.off_offs:geom.off_offs + B + 1]
    assert int(offs.max()) == geom.cap_pool and torch.equal(inv, torch.ones_like(inv))
    ov.zero_()
    rows = [torch.arange(B, device="cuda") % 100]
    rows[0][5] = 100                                                    # == vocab: out of range
    hist = torch.arange(B * L, device="cuda").view(B, L) % 100
    hist[7, 2] = -3
    send, slot, inv = hip.route(geom, call, rows, hist, [100], base, ov, st)
    assert int(st) == 1 and int(slot[5, 0]) == W * geom.cap_rows


def _rh():
    from recbox_amd.rechub.basic import features as Fe
    return Fe


def _youtube_feats(Fe, V, D, with_user_id=True):
    uf = ([Fe.SparseFeature("user_id", 61, 8)] if with_user_id else []) + \
        [Fe.SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)]
    return uf, [Fe.SparseFeature("item", V, D)], [Fe.SequenceFeature("neg_items", V, D, pooling="concat", shared_with="item")]


def _youtube_batch(B, V, L, n_neg, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, L + 1, (B,), generator=g)
    return {"user_id": torch.randint(0, 61, (B,), generator=g),
            "hist": torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None]),
            "item": torch.randint(1, V, (B,), generator=g), "neg_items": torch.randint(1, V, (B, n_neg), generator=g)}


def _seed_params(model, seed=0, std=0.2):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in sorted(model.named_parameters()):
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))


@pytest.mark.parametrize("with_user_id", [True, False])
def test_sharded_youtubednn_world_of_one_equals_youtubednn(with_user_id):
    """The sharded model in a world of one (the exchange degenerates to copies) == the single-GPU mirror: logits, loss,
    every gradient -- the item table's against the store's shard."""
    import torch.nn.functional as F
    Fe = _rh()
    from recbox_amd.rechub.models.matching import YoutubeDNN
    from recbox_amd.rechub.sharded import ShardedYoutubeDNN
    V, D, B, L, n_neg = 997, 16, 129, 9, 3
    ref = YoutubeDNN(*_youtube_feats(Fe, V, D, with_user_id), {"dims": [32, D]}, temperature=0.1).cuda()
    _seed_params(ref)
    dut = ShardedYoutubeDNN(*_youtube_feats(Fe, V, D, with_user_id), {"dims": [32, D]}, temperature=0.1,
                            shard_min_vocab=500).cuda()
    assert dut.embedding.sharded_tables == ["item"] and ("user_id" in dut.embedding.embed_dict) == with_user_id
    sd = ref.state_dict()
    full = sd.pop("embedding.embed_dict.item.weight")
    missing, unexpected = dut.load_state_dict(sd, strict=False)
    assert missing == ["embedding.store.weight"] and not unexpected
    dut.embedding.store.load_full_tables([full])
    x = {k: v.cuda() for k, v in _youtube_batch(B, V, L, n_neg, 3).items()}
    tgt = torch.zeros(B, dtype=torch.long, device="cuda")
    y0, y1 = ref(x), dut(x)
    assert_close(y1, y0, 1e-5, "logits")
    F.cross_entropy(y0, tgt).backward()
    F.cross_entropy(y1, tgt).backward()
    dut.sync_grads()
    want = dict(ref.named_parameters())
    for n, p in dut.named_parameters():
        if n == "embedding.store.weight":
            assert_close(p.grad, want["embedding.embed_dict.item.weight"].grad, 1e-5, "item table grad")
        else:
            assert_close(p.grad, want[n].grad, 1e-5, "grad " + n)


def test_sharded_deepfm_world_of_one_equals_deepfm():
    import torch.nn.functional as F
    Fe = _rh()
    from recbox_amd.rechub.models.ranking import DeepFM
    from recbox_amd.rechub.sharded import ShardedDeepFM
    vocabs, D, B = [7, 900, 31, 1200, 5, 640], 16, 200

    def feats():
        dense = [Fe.DenseFeature("I%d" % i) for i in range(3)]
        sparse = [Fe.SparseFeature("C%d" % i, v, D) for i, v in enumerate(vocabs)]
        return dense + sparse, sparse

    ref = DeepFM(*feats(), {"dims": [24, 16], "dropout": 0.0, "activation": "relu"}).cuda()
    _seed_params(ref)
    dut = ShardedDeepFM(*feats(), {"dims": [24, 16], "dropout": 0.0, "activation": "relu"}, shard_min_vocab=600).cuda()
    assert dut.embedding.sharded_tables == ["C1", "C3", "C5"]
    sd = ref.state_dict()
    full = [sd.pop("embedding.embed_dict.%s.weight" % n) for n in dut.embedding.sharded_tables]
    missing, unexpected = dut.load_state_dict(sd, strict=False)
    assert missing == ["embedding.store.weight"] and not unexpected
    dut.embedding.store.load_full_tables(full)
    g = torch.Generator().manual_seed(9)
    x = {"I%d" % i: torch.rand(B, generator=g).cuda() for i in range(3)}
    for i, v in enumerate(vocabs):
        x["C%d" % i] = torch.randint(0, v, (B,), generator=g).cuda()
    y = (torch.rand(B, generator=g) < 0.3).float().cuda()
    p0, p1 = ref(x), dut(x)
    assert_close(p1, p0, 1e-5, "prediction")
    F.binary_cross_entropy(p0, y).backward()
    F.binary_cross_entropy(p1, y).backward()
    dut.sync_grads()
    want = dict(ref.named_parameters())
    store = dut.embedding.store
    for n, p in dut.named_parameters():
        if n == "embedding.store.weight":
            for t, name in enumerate(dut.embedding.sharded_tables):
                sl, _ = store.local_rows_of(t)
                assert_close(p.grad[sl], want["embedding.embed_dict.%s.weight" % name].grad, 1e-5, "table " + name)
        else:
            assert_close(p.grad, want[n].grad, 1e-5, "grad " + n)
    # running statistics of the towers' BatchNorm moved identically
    mine = dict(dut.named_buffers())
    for n0, b0 in ref.named_buffers():
        assert_close(mine[n0], b0, 1e-5, n0)


def test_sharded_layer_with_several_pooled_sequences_and_two_dims_equals_embedding_layer():
    """The reference's EmbeddingLayer takes any mix of features (third_party/rechub/basic/layers.py:66-116): a layer call
    with TWO pooled sequences over sharded tables (mean and sum pooled, sharing the id table) and sharded tables of two
    embedding dimensions -- one exchange per (store, pooled sequence), all into one output block -- equals the single-GPU
    layer: the [B, width] block bit for bit on the single rows and to 1e-6 on the pooled ones, every table gradient."""
    Fe = _rh()
    from recbox_amd.rechub.basic.layers import EmbeddingLayer
    from recbox_amd.rechub.sharded import ShardedEmbeddingLayer

    def feats():
        return [Fe.SparseFeature("a", 1000, 16), Fe.SequenceFeature("h1", 1000, 16, pooling="mean", shared_with="a"),
                Fe.SequenceFeature("h2", 1000, 16, pooling="sum", shared_with="a"), Fe.SparseFeature("b", 2000, 32),
                Fe.SparseFeature("s", 10, 16), Fe.DenseFeature("d")]

    B = 133
    g = torch.Generator().manual_seed(11)
    fr, fd = feats(), feats()
    ref = EmbeddingLayer(fr).cuda()
    dut = ShardedEmbeddingLayer(fd, shard_min_vocab=500).cuda()
    assert dut.sharded_tables == ["a", "b"] and sorted(dut.stores) == ["16", "32"]
    with torch.no_grad():
        for n, p in ref.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        dut.embed_dict["s"].weight.copy_(ref.embed_dict["s"].weight)
    dut.stores["16"].load_full_tables([ref.embed_dict["a"].weight.detach()])
    dut.stores["32"].load_full_tables([ref.embed_dict["b"].weight.detach()])
    lens1, lens2 = torch.randint(0, 6, (B,), generator=g), torch.randint(1, 8, (B,), generator=g)
    x = {"a": torch.randint(0, 1000, (B,), generator=g), "b": torch.randint(0, 2000, (B,), generator=g),
         "s": torch.randint(0, 10, (B,), generator=g), "d": torch.rand(B, generator=g),
         "h1": torch.randint(1, 1000, (B, 5), generator=g) * (torch.arange(5)[None, :] < lens1[:, None]),
         "h2": torch.randint(1, 1000, (B, 7), generator=g) * (torch.arange(7)[None, :] < lens2[:, None])}
    x = {k: v.cuda() for k, v in x.items()}
    for f in fr + fd:                                    # rechub masks pooled sequences with the feature's padding_idx
        if isinstance(f, Fe.SequenceFeature):
            f.padding_idx = 0
    want = ref(x, fr, squeeze_dim=True)
    got = dut(x, fd, squeeze_dim=True)
    assert got.shape == want.shape
    assert_close(got, want, 1e-6, "block")
    R = torch.randn(want.shape, generator=g).cuda()
    (want * R).sum().backward()
    (got * R).sum().backward()
    assert_close(dut.stores["16"].weight.grad, ref.embed_dict["a"].weight.grad, 1e-5, "table a (single rows + two pooled sequences)")
    assert_close(dut.stores["32"].weight.grad, ref.embed_dict["b"].weight.grad, 1e-5, "table b")
    assert_close(dut.embed_dict["s"].weight.grad, ref.embed_dict["s"].weight.grad, 1e-5, "replicated table")


def test_persistent_shard_gradient_equals_fresh_over_steps():
    """HipShardOps.persistent(): the shard's dense gradient lives in ONE buffer whose rows are cleared by the previous
    step's sorted keys (rbx_embed_rezero) instead of a fresh zero-filled [rows, D] tensor per step -- three steps over
    different batches leave bit-identical gradients to the fresh path; a step that does not start from grad None is refused."""
    import torch.nn.functional as F
    Fe = _rh()
    from recbox_amd.rechub.sharded import ShardedYoutubeDNN
    V, D, B, L, n_neg = 2003, 16, 257, 7, 3
    models = []
    for _ in range(2):
        m = ShardedYoutubeDNN(*_youtube_feats(Fe, V, D, True), {"dims": [32, D]}, temperature=0.1, shard_min_vocab=500).cuda()
        _seed_params(m)
        models.append(m)
    fresh, kept = models
    kept.embedding.store.local_ops.persistent(kept.embedding.store.weight)
    tgt = torch.zeros(B, dtype=torch.long, device="cuda")
    for step in range(3):
        x = {k: v.cuda() for k, v in _youtube_batch(B, V, L, n_neg, 20 + step).items()}
        for m in models:
            m.zero_grad(set_to_none=True)
            F.cross_entropy(m(x), tgt).backward()
            m.sync_grads()
        for (n, p0), (_, p1) in zip(fresh.named_parameters(), kept.named_parameters()):
            assert torch.equal(p1.grad, p0.grad), "step %d: %s" % (step, n)
    with pytest.raises(RuntimeError, match="persistent shard gradient"):
        F.cross_entropy(kept(x), tgt).backward()           # weight.grad still set


def test_store_at_cfg3_size_properties():
    """BASELINE.json cfg 3 at full size on one GPU (10 M x 128 table, B = 65 536, history <= 50, 1 + 4 items): the pooled
    output of 200 sampled users against an index_select restatement on the device, the item rows exactly, and the
    shard gradient of an all-ones upstream gradient == integer lookup multiplicities."""
    from recbox_amd import ops
    from recbox_amd.sharded import ShardCall, ShardedStore
    V, D, B, L, T = 10_000_000, 128, 65536, 50, 5
    g = torch.Generator(device="cuda").manual_seed(4)
    store = ShardedStore([V], D, capacity_factor=1.05).cuda()
    with torch.no_grad():
        store.weight.copy_(torch.randn(V, D, device="cuda", generator=g) * 0.1)
    lens = torch.randint(1, L + 1, (B,), device="cuda", generator=g)
    hist = torch.randint(1, V, (B, L), device="cuda", generator=g) * (torch.arange(L, device="cuda")[None, :] < lens[:, None])
    items = torch.randint(1, V, (B, T), device="cuda", generator=g)
    call = ShardCall([(0, D * (1 + t)) for t in range(T)], pool=(0, 0, L, "mean", 0, 1e-16))
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        out = store.lookup(call, D * (1 + T), [items[:, t] for t in range(T)], hist)
        assert not bool(store.overflow)
        assert torch.equal(out[:, D:].reshape(B, T, D), store.weight.detach()[items])
        pick = torch.randint(0, B, (200,), device="cuda", generator=g)
        rows = store.weight.detach()[hist[pick]]                                    # [200, L, D]
        m = (hist[pick] != 0).float()
        want = (rows * m.unsqueeze(-1)).sum(1) / (m.sum(1, keepdim=True) + 1e-16)
        assert_close(out[pick, :D], want, 1e-5, "pooled history")
        up = torch.ones_like(out)
        up[:, :D] = 0                                                               # items only: integer multiplicities
        out.backward(up)
        mult = torch.bincount(items.reshape(-1), minlength=V).float()
        assert torch.equal(store.weight.grad[:, 0], mult) and torch.equal(store.weight.grad[:, D - 1], mult)
        store.weight.grad = None
        call_sum = ShardCall([], pool=(0, 0, L, "sum", 0, 0.0))
        out2 = store.lookup(call_sum, D, [], hist)
        out2.backward(torch.ones_like(out2))
        ids = hist[hist != 0]
        assert torch.equal(store.weight.grad[:, 3], torch.bincount(ids, minlength=V).float())
    finally:
        ops.config.check_ids = old
