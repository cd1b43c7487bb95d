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
        L.check(L.lib.rbx_fm_fwd(ea, la, n, B, None, None, 0, 0, -1, logit.data_ptr(), ssum.data_ptr() if ea else None, None, None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.check(L.lib.rbx_fm_fwd(ea, la, n, B, None, None, 0, 0, -1, logit.data_ptr(), ssum.data_ptr() if ea else None, None, None))
    e1.record()
    torch.cuda.synchronize()
    print("%-44s %7.1f us" % (tag, e0.elapsed_time(e1) * 1000 / 50))


for kind in ("f64_strided", "i64_contig", "i32_contig"):
    time_it(fields(kind, emb, D), fields(kind, lr, 1), "emb+lr, ids " + kind)
    time_it(fields(kind, emb, D), None, "emb only, ids " + kind)
time_it(None, fields("i64_contig", lr, 1), "lr only, ids i64_contig")
