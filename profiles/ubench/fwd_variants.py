"""Time rbx_fm_fwd (the fused FM forward) under different id layouts / parts, to see what the
55 us at B=65 536 is made of.  Run on the GPU box: python profiles/ubench/fwd_variants.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CRITEO_VOCABS  # noqa: E402
from recbox_amd import _lib as L  # noqa: E402

B, D = 65536, 16
g = torch.Generator().manual_seed(1)
n = len(CRITEO_VOCABS)
emb = [torch.randn(v + 1, D, generator=g).cuda() for v in CRITEO_VOCABS]
lr = [torch.randn(v + 1, 1, generator=g).cuda() for v in CRITEO_VOCABS]
ids64 = [torch.randint(1, v + 1, (B,), generator=g).cuda() for v in CRITEO_VOCABS]
block = torch.stack([t.double() for t in ids64], dim=1).contiguous()          # [B, 26] float64, strided columns
fm_major = torch.stack(ids64, dim=0).contiguous()                             # [26, B] int64
fm_major32 = fm_major.int()
logit = torch.empty(B, device="cuda")
ssum = torch.empty(B, D, device="cuda")


def fields(kind, tables, dim):
    arr = (L.rbx_field_t * n)()
    for i in range(n):
        f = arr[i]
        if kind == "f64_strided":
            f.ids, f.ids_stride_b, f.ids_dtype = block[:, i].data_ptr(), block.stride(0), L.RBX_F64
        elif kind == "i64_contig":
            f.ids, f.ids_stride_b, f.ids_dtype = fm_major[i].data_ptr(), 1, L.RBX_I64
        else:
            f.ids, f.ids_stride_b, f.ids_dtype = fm_major32[i].data_ptr(), 1, L.RBX_I32
        f.table, f.vocab, f.dim = tables[i].data_ptr(), CRITEO_VOCABS[i] + 1, dim
        f.padding_idx, f.mask_id, f.seq_len, f.kind, f.pool = 0, L.RBX_NO_ID, 1, L.FIELD_CATEGORICAL, L.POOL_NONE
    return arr


def time_it(ea, la, tag):
    for _ in range(5):
        L.check(L.lib.rbx_fm_fwd(ea, la, n, B, None, None, 0, 0, -1, None, 0, logit.data_ptr(), ssum.data_ptr() if ea else None, None, None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.check(L.lib.rbx_fm_fwd(ea, la, n, B, None, None, 0, 0, -1, None, 0, logit.data_ptr(), ssum.data_ptr() if ea else None, None, None))
    e1.record()
    torch.cuda.synchronize()
    print("%-44s %7.1f us" % (tag, e0.elapsed_time(e1) * 1000 / 50))


for kind in ("f64_strided", "i64_contig", "i32_contig"):
    time_it(fields(kind, emb, D), fields(kind, lr, 1), "emb+lr, ids " + kind)
    time_it(fields(kind, emb, D), None, "emb only, ids " + kind)
time_it(None, fields("i64_contig", lr, 1), "lr only, ids i64_contig")

# ---- bench-like conditions: 13 numeric fields, [B, 40] float64 batch, cache flushed between launches ----
N_DENSE = 13
batch = torch.cat([torch.rand(B, N_DENSE, dtype=torch.float64, device="cuda"), block,
                   torch.zeros(B, 1, dtype=torch.float64, device="cuda")], dim=1).contiguous()      # [B, 40]
wn = [torch.randn(D, device="cuda") for _ in range(N_DENSE)]
wl = [torch.randn(1, device="cuda") for _ in range(N_DENSE)]


def bench_fields(tables_num, tables_cat, dim):
    arr = (L.rbx_field_t * (N_DENSE + n))()
    for i in range(N_DENSE + n):
        f = arr[i]
        f.ids, f.ids_stride_b, f.ids_dtype = batch[:, i].data_ptr(), batch.stride(0), L.RBX_F64
        f.dim, f.seq_len, f.pool, f.mask_id = dim, 1, L.POOL_NONE, L.RBX_NO_ID
        if i < N_DENSE:
            f.table, f.kind, f.vocab, f.padding_idx = tables_num[i].data_ptr(), L.FIELD_NUMERIC, 0, L.RBX_NO_ID
        else:
            f.table, f.kind, f.vocab, f.padding_idx = tables_cat[i - N_DENSE].data_ptr(), L.FIELD_CATEGORICAL, \
                CRITEO_VOCABS[i - N_DENSE] + 1, 0
    return arr


ea, la = bench_fields(wn, emb, D), bench_fields(wl, lr, 1)
scratch = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for flush in (False, True):
    ts = []
    for _ in range(12):
        if flush:
            scratch.zero_()                       # 512 MB of writes: evicts L2 and the 256 MB Infinity Cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.lib.rbx_fm_fwd(ea, la, N_DENSE + n, B, None, None, 0, 0, -1, None, 0, logit.data_ptr(), ssum.data_ptr(), None, None))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    print("bench layout (39 fields, f64 [B,40]), cache %-7s median %6.1f us  min %6.1f us"
          % ("flushed" if flush else "warm", sorted(ts)[len(ts) // 2], min(ts)))
