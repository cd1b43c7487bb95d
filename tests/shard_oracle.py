"""TEST INFRASTRUCTURE: plain-torch restatement of the sharded exchange's local work (csrc/rbx_shard.hip:
rbx_shard_route / rbx_shard_serve / rbx_shard_combine_fwd / rbx_shard_combine_bwd + the owner's indexed scatter-add).

Two uses: (1) the backend of the world_size-2/3 gloo tests on CPU (``ShardedStore(local_ops=OracleShardOps())``), where
everything around it -- wire geometry, the three all-to-alls, autograd plumbing, shard bookkeeping -- is the product code;
(2) the checker of the HIP kernels on the GPU (integers bit for bit, sums to 1e-6).  Never imported by the product."""
import torch


def _stable_ranks(owner, W):
    """owner [n] in [-1, W): rank of every lookup among the EARLIER lookups with the same owner (stable counting sort),
    and the per-owner totals.  -1 = absent (no slot)."""
    n = owner.numel()
    dev = owner.device
    key = torch.where(owner >= 0, owner, torch.full_like(owner, W))
    order = torch.argsort(key, stable=True)
    ks = key[order]
    counts = torch.bincount(key, minlength=W + 1)
    starts = torch.cumsum(counts, 0) - counts
    rank_sorted = torch.arange(n, device=dev) - starts[ks]
    rank = torch.empty_like(rank_sorted)
    rank[order] = rank_sorted
    return rank, counts[:W]


class OracleShardOps(object):
    def route(self, geom, call, row_ids, pool_ids, vocabs, base, overflow, status):
        W, B, T, P = geom.W, geom.B, geom.T, geom.P
        dev = base.device
        send = torch.full((W, geom.ichunk), -1, dtype=torch.int32, device=dev)
        slot = inv = None
        bad = False
        if T:
            ids = torch.stack([c.long().to(dev) for c in row_ids], dim=1)                     # [B, T]
            voc = torch.tensor([vocabs[t] for t, _ in call.rows], device=dev)
            ok = (ids >= 0) & (ids < voc)
            bad = bad or bool((~ok).any())
            owner = torch.where(ok, ids % W, torch.full_like(ids, -1)).reshape(-1)
            rank, counts = _stable_ranks(owner, W)
            if bool((counts > geom.cap_rows).any()):
                overflow.fill_(1)
            k = torch.arange(T, device=dev).repeat(B)
            fits = (owner >= 0) & (rank < geom.cap_rows)
            row = base[owner.clamp(min=0), k] + ids.reshape(-1) // W
            send[owner[fits], geom.off_rows + rank[fits]] = row[fits].int()
            dump = W * geom.cap_rows
            slot = torch.where(fits, owner * geom.cap_rows + rank, torch.full_like(rank, dump)).int().view(B, T)
        if P:
            tbl, _, L, mode, mask_id, eps = call.pool
            ids = pool_ids.long().to(dev)                                                     # [B, L]
            masked = (ids == mask_id) if mask_id is not None else torch.zeros_like(ids, dtype=torch.bool)
            inrange = (ids >= 0) & (ids < vocabs[tbl])
            bad = bad or bool((~masked & ~inrange).any())
            ok = ~masked & inrange
            owner = torch.where(ok, ids % W, torch.full_like(ids, -1))
            rank, counts = _stable_ranks(owner.reshape(-1), W)
            if bool((counts > geom.cap_pool).any()):
                overflow.fill_(1)
            fits = (owner.reshape(-1) >= 0) & (rank < geom.cap_pool)
            row = base[owner.reshape(-1).clamp(min=0), T] + ids.reshape(-1) // W
            send[owner.reshape(-1)[fits], rank[fits]] = row[fits].int()
            per = torch.stack([(owner == w).sum(dim=1) for w in range(W)], dim=0)            # [W, B]
            offs = torch.zeros((W, B + 1), dtype=torch.long, device=dev)
            offs[:, 1:] = per.cumsum(dim=1)
            send[:, geom.off_offs:geom.off_offs + B + 1] = offs.clamp(max=geom.cap_pool).int()
            cnt = ok.sum(dim=1)
            inv = (1.0 / (cnt.float() + eps)) if mode == "mean" else torch.ones(B, device=dev)
            inv = inv.float()
        if bad and status is not None:
            status.fill_(1)
        return send.reshape(-1), slot, inv

    def serve(self, geom, recv, weight, status):
        W, B, P, D = geom.W, geom.B, geom.P, geom.D
        dev = weight.device
        recv = recv.view(W, geom.ichunk).long()
        back = torch.zeros((W * geom.frows, D), dtype=torch.float32, device=dev)
        keys = torch.full((geom.n_keys,), -1, dtype=torch.int32, device=dev)
        src = torch.zeros(geom.n_keys, dtype=torch.int32, device=dev)
        n_local = weight.shape[0]
        for w in range(W):
            if P:
                offs = recv[w, geom.off_offs:geom.off_offs + B + 1].clamp(0, geom.cap_pool)
                total = int(offs[B])
                rows = recv[w, :total]
                seg = torch.repeat_interleave(torch.arange(B, device=dev), offs[1:] - offs[:-1])
                ok = (rows >= 0) & (rows < n_local)
                for j in range(total):                                 # slot order == the kernel's summation order
                    if ok[j]:
                        back[w * geom.frows + seg[j]] += weight[rows[j]]
                keys[w * geom.cap_pool:w * geom.cap_pool + total] = torch.where(ok, rows, torch.full_like(rows, -1)).int()
                src[w * geom.cap_pool:w * geom.cap_pool + total] = (w * geom.frows + seg).int()
            if geom.T:
                rows = recv[w, geom.off_rows:geom.off_rows + geom.cap_rows]
                ok = (rows >= 0) & (rows < n_local)
                f0 = w * geom.frows + P * B
                back[f0:f0 + geom.cap_rows][ok] = weight[rows[ok]]
                k0 = W * geom.cap_pool + w * geom.cap_rows
                keys[k0:k0 + geom.cap_rows] = torch.where(ok, rows, torch.full_like(rows, -1)).int()
                src[k0:k0 + geom.cap_rows] = torch.arange(f0, f0 + geom.cap_rows, device=dev).int()
        return back, keys, src

    def combine_fwd(self, geom, call, back, slot, inv, out):
        W, B, T, P, D = geom.W, geom.B, geom.T, geom.P, geom.D
        for f, (_, off) in enumerate(call.rows):
            s = slot[:, f].long()
            ok = s < W * geom.cap_rows
            w = s.clamp(max=W * geom.cap_rows - 1) // geom.cap_rows
            idx = w * geom.frows + P * B + (s - w * geom.cap_rows)
            rows = back[idx.clamp(0, back.shape[0] - 1)] * ok.unsqueeze(1)
            out[:, off:off + D] = rows
        if P:
            off = call.pool[1]
            acc = torch.zeros((B, D), dtype=torch.float32, device=back.device)
            for w in range(W):                                         # fixed owner order, as the kernel
                acc = acc + back[w * geom.frows:w * geom.frows + B]
            out[:, off:off + D] = acc * inv.unsqueeze(1)

    def combine_bwd(self, geom, call, dout, slot, inv):
        W, B, T, P, D = geom.W, geom.B, geom.T, geom.P, geom.D
        gsend = torch.zeros((W * geom.frows, D), dtype=torch.float32, device=dout.device)
        for f, (_, off) in enumerate(call.rows):
            s = slot[:, f].long()
            ok = s < W * geom.cap_rows
            w = s[ok] // geom.cap_rows
            gsend[w * geom.frows + P * B + (s[ok] - w * geom.cap_rows)] = dout[ok, off:off + D]
        if P:
            off = call.pool[1]
            g = dout[:, off:off + D] * inv.unsqueeze(1)
            for w in range(W):
                gsend[w * geom.frows:w * geom.frows + B] = g
        return gsend

    def scatter(self, weight, keys, src, grecv, sorted_ws=None):
        grad = torch.zeros_like(weight)
        ok = keys >= 0
        grad.index_add_(0, keys[ok].long(), grecv[src[ok].long()])
        return grad
