"""world_size-2 gloo tests (CPU) of the row-sharded embedding exchange: ids out, rows back, dY to
the owners, owner-side scatter-add.  The local gather / scatter-add kernels need a GPU, so the
test injects an oracle-backed local backend; everything else (bucketing, all-to-all-v emulation,
un-permute, shard bookkeeping, dense all-reduce) is the product code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def padded_route(owner, row, capacity, W, overflow):
    """TEST INFRASTRUCTURE: torch restatement of ``rbx_route`` (slot assignment of the padded exchange), used by the
    gloo tests below and as the checker of the HIP kernel in test_gpu_ranking.py.  owner/row: flat [n].  Returns (slot, send): ``slot[i]`` is the wire slot of
    lookup i = owner * capacity + (number of EARLIER lookups with the same owner); ``W * capacity`` = the dump
    slot of lookups that do not fit (sets ``overflow``); ``send`` [W * capacity] holds the row numbers, -1 = empty."""
    n = owner.numel()
    dev = owner.device
    order = torch.argsort(owner, stable=True)
    owner_s = owner[order]
    counts = torch.zeros(W, dtype=torch.long, device=dev).scatter_add_(0, owner, torch.ones_like(owner))
    starts = torch.cumsum(counts, 0) - counts
    rank_in = torch.arange(n, device=dev) - starts[owner_s]
    fits = rank_in < capacity
    overflow.logical_or_((~fits).any())                       # reported, never silently dropped
    dump = W * capacity                                       # one spare slot swallows what does not fit
    slot_s = torch.where(fits, owner_s * capacity + rank_in, torch.full_like(rank_in, dump))
    send = torch.full((dump + 1,), -1, dtype=torch.long, device=dev)
    send[slot_s] = row[order]
    slot = torch.empty_like(slot_s)
    slot[order] = slot_s
    return slot, send[:dump]


class OracleLocalOps(object):
    """CPU stand-in for HipLocalOps (test infrastructure): plain index_select / index_add_;
    row -1 is an empty slot of the padded exchange (zero vector, no gradient)."""

    def gather(self, weight, rows):
        out = weight.detach().index_select(0, rows.clamp(min=0))
        return out * (rows >= 0).unsqueeze(1)

    def scatter_add(self, weight, rows, dy, sorted_ws=None):
        g = torch.zeros_like(weight)
        keep = rows >= 0
        g.index_add_(0, rows[keep], dy[keep])
        return g

    def route(self, ids, world, capacity, base, overflow):
        if not torch.is_tensor(ids):
            ids = torch.stack([c.long() for c in ids], dim=1)
        ids = ids.long()
        owner = ids % world
        t_index = torch.arange(ids.shape[1]).unsqueeze(0).expand_as(ids)
        row = base[owner, t_index] + ids // world
        slot, send = padded_route(owner.reshape(-1), row.reshape(-1), capacity, world, overflow)
        return slot.view_as(ids), send


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recbox_amd import comm
        from recbox_amd.sharded import ShardedEmbedding
        V, D, B, Lh = 37, 4, 11, 3
        g = torch.Generator().manual_seed(0)
        full = torch.randn(V, D, generator=g)
        full[0].zero_()
        gi = torch.Generator().manual_seed(100 + rank)               # every rank has its own batch
        ids = torch.randint(0, V, (B, Lh), generator=gi)
        R = torch.randn(B, Lh, D, generator=gi)
        emb = ShardedEmbedding(V, D, padding_idx=0, local_ops=OracleLocalOps())
        emb.load_full_table(full)
        out = emb(ids)
        assert torch.equal(out, full[ids]), "rows returned by the exchange differ"
        (out * R).sum().backward()
        emb.zero_pad_grad()
        # reference: dense grad of the FULL table from ALL ranks' batches, then this rank's rows
        all_ids = [torch.empty_like(ids) for _ in range(world)]
        all_R = [torch.empty_like(R) for _ in range(world)]
        dist.all_gather(all_ids, ids)
        dist.all_gather(all_R, R)
        want = torch.zeros(V, D)
        for i, r in zip(all_ids, all_R):
            want.index_add_(0, i.reshape(-1), r.reshape(-1, D))
        want[0].zero_()
        mine = torch.arange(rank, V, world)
        got = emb.local.weight.grad[:mine.numel()]
        assert torch.allclose(got, want[mine], atol=1e-6), "shard gradient differs"
        # empty and one-sided exchanges
        e2 = emb(torch.zeros(0, dtype=torch.long))
        assert tuple(e2.shape) == (0, D)
        one_sided = torch.full((5,), 2 * 3 + (1 if world > 1 else 0))   # all ids owned by rank 1 (or 0 if W=1)
        assert torch.equal(emb(one_sided), full[one_sided])
        # several tables behind ONE exchange, embedding + LR weight packed in one row; both exchange modes
        from recbox_amd.sharded import ShardedTables
        vocabs = [13, 29, 7]
        gt = torch.Generator().manual_seed(5)
        full_e = [torch.randn(v, D, generator=gt) for v in vocabs]
        full_l = [torch.randn(v, 1, generator=gt) for v in vocabs]
        tid = torch.stack([torch.randint(0, v, (B,), generator=gi) for v in vocabs], dim=1)      # [B, T]
        RE, RL = torch.randn(B, 3, D, generator=gi), torch.randn(B, 3, generator=gi)
        g_ids = [torch.empty_like(tid) for _ in range(world)]
        g_RE = [torch.empty_like(RE) for _ in range(world)]
        g_RL = [torch.empty_like(RL) for _ in range(world)]
        dist.all_gather(g_ids, tid)
        dist.all_gather(g_RE, RE)
        dist.all_gather(g_RL, RL)
        for factor in (None, 3.0):
            # table 1 has a padding row (id 2, owned by rank 2 % world): a zero row that receives no gradient
            tabs = ShardedTables(vocabs, D, with_lr=True, capacity_factor=factor, local_ops=OracleLocalOps(),
                                 padding_idx=[None, 2, None])
            assert tabs.row_width == 8 and tabs.lr_off == D
            assert tabs.pad_rows.numel() == (1 if rank == 2 % world else 0)
            full_e[1][2].zero_()
            full_l[1][2].zero_()
            tabs.load_full_tables(full_e, full_l)
            E, Lw = tabs.split(tabs(tid))
            for t in range(3):
                assert torch.equal(E[:, t], full_e[t][tid[:, t]]) and torch.equal(Lw[:, t], full_l[t][tid[:, t], 0])
            ((E * RE).sum() + (Lw * RL).sum()).backward()
            assert not bool(tabs.overflow)
            for t, v in enumerate(vocabs):
                we, wl = torch.zeros(v, D), torch.zeros(v)
                for i, re, rl in zip(g_ids, g_RE, g_RL):
                    we.index_add_(0, i[:, t], re[:, t])
                    wl.index_add_(0, i[:, t], rl[:, t])
                if t == 1:
                    we[2].zero_()
                    wl[2].zero_()
                sl, owned = tabs.local_rows_of(t)
                assert torch.allclose(tabs.weight.grad[sl, :D], we[owned], atol=1e-6)
                assert torch.allclose(tabs.weight.grad[sl, D], wl[owned], atol=1e-6)
                assert float(tabs.weight.grad[sl, D + 1:].abs().max()) == 0.0
        # a capacity that cannot hold a one-sided batch raises the overflow flag instead of dropping silently
        tight = ShardedTables([64], D, with_lr=False, capacity_factor=1.0, local_ops=OracleLocalOps())
        tight(torch.zeros(256, 1, dtype=torch.long))         # every lookup goes to rank 0: 256 > capacity 128
        flag = torch.tensor([float(bool(tight.overflow))])
        dist.all_reduce(flag)
        assert (flag.item() > 0) == (world > 1)
        # ... and the library reports it on EVERY rank (collective), then starts from a clean flag
        if world > 1:
            with pytest.raises(RuntimeError, match="capacity_factor"):
                tight.raise_if_overflowed()
        tight.raise_if_overflowed()
        tabs.raise_if_overflowed()
        # dense tower grads: all-reduce
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(1.0)
            lin.bias.fill_(0.0)
        lin(torch.full((1, 3), float(rank + 1))).sum().backward()
        comm.all_reduce_grads(lin.parameters())
        assert torch.allclose(lin.weight.grad, torch.full((2, 3), float(sum(range(1, world + 1)))))
        # a rank that produced NO gradient (an empty local batch) still takes part with zeros: same collectives on every rank
        lin2 = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin2.weight.fill_(1.0)
        if rank != 0:
            lin2(torch.full((1, 3), float(rank + 1))).sum().backward()
        comm.all_reduce_grads(lin2.parameters())
        assert torch.allclose(lin2.weight.grad, torch.full((2, 3), float(sum(range(2, world + 1)))))
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_embedding_exchange_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_worker, args=(world, port, result), nprocs=world, join=True)
    assert dict(result) == {r: "ok" for r in range(world)}


def test_single_process_is_identity_exchange():
    from recbox_amd.sharded import ShardedEmbedding
    emb = ShardedEmbedding(10, 3, local_ops=OracleLocalOps())
    full = torch.arange(30.0).reshape(10, 3)
    emb.load_full_table(full)
    ids = torch.tensor([[1, 9], [0, 4]])
    assert torch.equal(emb(ids), full[ids])


# ---- second generation exchange: ShardedStore (one exchange each way, pooled lookups reduced at the owner) -------------
def _store_case(rank, world):
    """A layer call over two sharded tables: three single-row lookups (two of table 0, one of table 1) and a mean-pooled
    history over table 0 -- the shape of YoutubeDNN's item side (youtube_dnn.py:46-70) plus a second table."""
    from recbox_amd.sharded import ShardCall
    vocabs, D, B, L = [37, 13], 8, 11, 5
    call = ShardCall([(0, 8), (0, 16), (1, 24)], pool=(0, 0, L, "mean", 0, 1e-16))
    g = torch.Generator().manual_seed(7)
    full = [torch.randn(v, D, generator=g) for v in vocabs]
    gi = torch.Generator().manual_seed(200 + rank)
    lens = torch.randint(0, L + 1, (B,), generator=gi)
    hist = torch.randint(1, vocabs[0], (B, L), generator=gi) * (torch.arange(L)[None, :] < lens[:, None])
    rows = [torch.randint(0, vocabs[0], (B,), generator=gi), torch.randint(0, vocabs[0], (B,), generator=gi),
            torch.randint(0, vocabs[1], (B,), generator=gi)]
    R = torch.randn(B, 32, generator=gi)
    return vocabs, D, B, L, call, full, hist, rows, R


def _store_reference(full, hist, rows, D):
    """What the single-GPU layer computes on full tables (rechub EmbeddingLayer: id-masked mean, eps 1e-16)."""
    mask = (hist != 0).float()
    pooled = (full[0][hist] * mask.unsqueeze(-1)).sum(dim=1) / (mask.sum(dim=1, keepdim=True) + 1e-16)
    return torch.cat([pooled, full[0][rows[0]], full[0][rows[1]], full[1][rows[2]]], dim=1)


def _store_worker(rank, world, port, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from shard_oracle import OracleShardOps
        from recbox_amd import ops
        from recbox_amd.sharded import ShardedStore
        vocabs, D, B, L, call, full, hist, rows, R = _store_case(rank, world)
        store = ShardedStore(vocabs, D, capacity_factor=3.0, local_ops=OracleShardOps())
        store.load_full_tables(full)
        old = ops.config.check_ids
        ops.config.check_ids = True
        try:
            out = store.lookup(call, 32, rows, hist)
            want = _store_reference(full, hist, rows, D)
            assert torch.allclose(out, want, atol=1e-6), "block differs: %g" % (out - want).abs().max()
            assert torch.equal(out[:, 8:], want[:, 8:]), "single rows must be exact"
            (out * R).sum().backward()
            assert not bool(store.overflow)
            # reference: dense grads of the FULL tables from ALL ranks' batches, then this rank's rows
            leaves = [f.clone().requires_grad_() for f in full]
            for r in range(world):
                _, _, _, _, _, _, h_r, rows_r, R_r = _store_case(r, world)
                (_store_reference(leaves, h_r, rows_r, D) * R_r).sum().backward()
            for t in range(len(vocabs)):
                sl, owned = store.local_rows_of(t)
                assert torch.allclose(store.weight.grad[sl], leaves[t].grad[owned], atol=1e-5), "shard grad, table %d" % t
            # an id outside [0, vocab) is an IndexError (the reference's nn.Embedding raises), on every rank or none
            # -- also when only ONE rank holds the bad id (the others must not walk into the next collective alone)
            bad = [r.clone() for r in rows]
            if rank == 0:
                bad[2][0] = vocabs[1]
            try:
                store.lookup(call, 32, bad, hist)
                raised = False
            except IndexError:
                raised = True
            assert raised
            # ranks with different batch sizes: a ValueError on every rank, not a hang in the all-to-all
            # ("once", the default, checks a batch size the first time a rank sees it -- a set-up error shows at the first
            #  step; "always" also catches a size that changes on one rank only, at one tiny all-reduce + host sync per call)
            n = B - 1 if rank == world - 1 else B
            for mode, st in (("once", ShardedStore(vocabs, D, capacity_factor=3.0, local_ops=OracleShardOps())), ("always", store)):
                st.check_batch_size = mode
                try:
                    st.lookup(call, 32, [r[:n] for r in rows], hist[:n])
                    raised = False
                except ValueError as e:
                    raised = "different batch sizes" in str(e)
                assert raised, mode
            # a history with more ids for one owner than its capacity raises the overflow flag (never silently dropped)
            tight = ShardedStore(vocabs, D, capacity_factor=1.0, local_ops=OracleShardOps())
            one_owner = torch.full((64, L), world if world < vocabs[0] else 0)      # every id -> owner 0
            tight.lookup(call, 32, [r.repeat(6)[:64] for r in rows], one_owner)
            flag = torch.tensor([float(bool(tight.overflow))])
            dist.all_reduce(flag)
            assert (flag.item() > 0) == (world > 1)
        finally:
            ops.config.check_ids = old
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_store_exchange_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_store_worker, args=(world, port, result), nprocs=world, join=True)
    assert dict(result) == {r: "ok" for r in range(world)}


def test_sharded_store_single_process():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from shard_oracle import OracleShardOps
    from recbox_amd.sharded import ShardedStore
    vocabs, D, B, L, call, full, hist, rows, R = _store_case(0, 1)
    store = ShardedStore(vocabs, D, local_ops=OracleShardOps())
    store.load_full_tables(full)
    out = store.lookup(call, 32, rows, hist)
    assert torch.allclose(out, _store_reference(full, hist, rows, D), atol=1e-6)
    (out * R).sum().backward()
    leaves = [f.clone().requires_grad_() for f in full]
    (_store_reference(leaves, hist, rows, D) * R).sum().backward()
    assert torch.allclose(store.weight.grad[:vocabs[0]], leaves[0].grad, atol=1e-5)
    assert torch.allclose(store.weight.grad[vocabs[0]:], leaves[1].grad, atol=1e-5)


# ---- DenseGradSync: the data-parallel all-reduce of tower / head gradients starts from the parameters' own hooks --------
def _sync_worker(rank, world, port, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recbox_amd.rechub.sharded import DenseGradSync
        torch.manual_seed(0)
        item, user, table = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3), torch.nn.Embedding(5, 3)
        towers = list(item.parameters()) + list(user.parameters())
        sync = DenseGradSync(towers, list(table.parameters()))
        for step in range(3):
            g = torch.Generator().manual_seed(10 * step + rank)
            xi, xu = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
            ids = torch.randint(0, 5, (6,), generator=g)
            # gradients of an earlier step stay in place as zeros (zero_grad(set_to_none=False)): ADVICE r2's stale-gradient case
            for p in towers + list(table.parameters()):
                if p.grad is not None:
                    p.grad.zero_()
            # the item branch's backward completes long before the user branch's (detach + a second backward inside one
            # autograd pass is not needed: autograd orders the two Linear nodes by their position in the graph)
            loss = ((item(xi) * table(ids)).sum() + (user(xu) ** 2).sum()) / world
            loss.backward()
            if step == 1:
                # a second backward before sync_grads() is refused instead of summing already-reduced gradients again
                try:
                    (user(xu).sum()).backward()
                    refused = False
                except RuntimeError as e:
                    refused = "second backward" in str(e)
                assert refused
                sync.abandon()                          # (restore: the refused backward left nothing behind but this)
                for p in towers + list(table.parameters()):
                    p.grad = None
                loss = ((item(xi) * table(ids)).sum() + (user(xu) ** 2).sum()) / world
                loss.backward()
            sync.finish()
            # reference: the same loss summed over every rank's batch, on one process
            ri, ru, rt = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3), torch.nn.Embedding(5, 3)
            ri.load_state_dict(item.state_dict()); ru.load_state_dict(user.state_dict()); rt.load_state_dict(table.state_dict())
            for r in range(world):
                g = torch.Generator().manual_seed(10 * step + r)
                xi, xu = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
                ids = torch.randint(0, 5, (6,), generator=g)
                (((ri(xi) * rt(ids)).sum() + (ru(xu) ** 2).sum()) / world).backward()
            for a, b in zip(towers + list(table.parameters()), list(ri.parameters()) + list(ru.parameters()) + list(rt.parameters())):
                assert torch.allclose(a.grad, b.grad, atol=1e-6), "step %d" % step
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _accum_worker(rank, world, port, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recbox_amd.rechub.sharded import DenseGradSync
        torch.manual_seed(0)
        item, user, table = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3), torch.nn.Embedding(5, 3)
        towers = list(item.parameters()) + list(user.parameters())
        sync = DenseGradSync(towers, list(table.parameters()))      # the table: a replicated ("late") parameter, ADVICE r5
        ri, ru, rt = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3), torch.nn.Embedding(5, 3)
        ri.load_state_dict(item.state_dict()); ru.load_state_dict(user.state_dict()); rt.load_state_dict(table.state_dict())

        def loss_of(it, us, tb, xi, xu, ids, step):
            if step == 1:
                return (it(xi) ** 2).sum() / world          # neither the user tower nor the table takes part in this round
            if step == 3:
                return (tb(ids) ** 2).sum() / world         # only the table: no tower hook fires before its gradient
            return ((it(xi) * tb(ids)).sum() + (us(xu) ** 2).sum()) / world
        for step in range(5):                               # five backward + finish rounds, no zero_grad in between
            g = torch.Generator().manual_seed(10 * step + rank)
            xi, xu = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
            ids = torch.randint(0, 5, (6,), generator=g)
            loss_of(item, user, table, xi, xu, ids, step).backward()
            sync.finish()
            for r in range(world):
                g = torch.Generator().manual_seed(10 * step + r)
                xi, xu = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
                ids = torch.randint(0, 5, (6,), generator=g)
                loss_of(ri, ru, rt, xi, xu, ids, step).backward()
            for a, b in zip(towers + [table.weight], list(ri.parameters()) + list(ru.parameters()) + [rt.weight]):
                assert torch.allclose(a.grad, b.grad, atol=1e-5), "round %d" % step
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_dense_grad_sync_accumulates_over_rounds_gloo():
    """Gradients that survive a round (no zero_grad between backward + finish rounds) are not counted W times."""
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_accum_worker, args=(2, port, result), nprocs=2, join=True)
    assert dict(result) == {0: "ok", 1: "ok"}


def test_dense_grad_sync_reduces_only_this_steps_gradients_gloo():
    port = _free_port()
    mgr = mp.Manager()
    result = mgr.dict()
    mp.spawn(_sync_worker, args=(2, port, result), nprocs=2, join=True)
    assert dict(result) == {0: "ok", 1: "ok"}
