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


class OracleLocalOps(object):
    """CPU stand-in for HipLocalOps (test infrastructure): plain index_select / index_add_."""

    def gather(self, weight, rows):
        return weight.detach().index_select(0, rows)

    def scatter_add(self, weight, rows, dy):
        g = torch.zeros_like(weight)
        g.index_add_(0, rows, dy)
        return g


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
        # dense tower grads: all-reduce
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(1.0)
            lin.bias.fill_(0.0)
        lin(torch.full((1, 3), float(rank + 1))).sum().backward()
        comm.all_reduce_grads(lin.parameters())
        assert torch.allclose(lin.weight.grad, torch.full((2, 3), float(sum(range(1, world + 1)))))
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
