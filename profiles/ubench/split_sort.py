"""Would sorting the small-vocabulary fields (16-bit keys, 2 radix passes) and the large ones (23-bit, 3 passes) as two
sorts on two streams beat the one sort of all 26 fields?   python profiles/ubench/split_sort.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from recbox_amd import _lib as L  # noqa: E402


def fields_for(vocabs, ids, tables):
    n = len(vocabs)
    arr = (L.rbx_field_t * n)()
    for i, v in enumerate(vocabs):
        f = arr[i]
        f.ids, f.table, f.grad = ids[i].data_ptr(), tables[i].data_ptr(), tables[i].data_ptr()
        f.ids_stride_b, f.ids_stride_l, f.vocab = 1, 0, v + 1
        f.padding_idx, f.mask_id, f.out_off = 0, L.RBX_NO_ID, i * 16
        f.dim, f.seq_len, f.ids_dtype, f.kind, f.pool, f.eps = 16, 1, L.RBX_I64, L.FIELD_CATEGORICAL, L.POOL_NONE, 0.0
    return arr, n


def main():
    B = 65536
    g = torch.Generator().manual_seed(0)
    V = bench.CRITEO_VOCABS
    ids = [torch.randint(1, v + 1, (B,), generator=g).cuda() for v in V]
    tables = [torch.zeros(v + 1, 16, device="cuda") for v in V]
    small = [i for i, v in enumerate(V) if v + 1 <= 65536]
    large = [i for i, v in enumerate(V) if v + 1 > 65536]
    print("small fields %d (rows %d), large fields %d (rows %d)" % (len(small), sum(V[i] + 1 for i in small), len(large),
                                                                  sum(V[i] + 1 for i in large)))
    sets = {"all": list(range(len(V))), "small": small, "large": large}
    plans = {}
    for name, idx in sets.items():
        arr, n = fields_for([V[i] for i in idx], [ids[i] for i in idx], [tables[i] for i in idx])
        nbytes = L.lib.rbx_embed_bwd_workspace_size(arr, n, B)
        plans[name] = (arr, n, torch.empty(nbytes, dtype=torch.uint8, device="cuda"), nbytes)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")

    def sort(name, stream):
        arr, n, ws, nbytes = plans[name]
        L.check(L.lib.rbx_embed_sort(arr, n, B, ws.data_ptr(), nbytes, None, ctypes.c_void_p(stream.cuda_stream)))

    def timed(fn, reps=20):
        cur = torch.cuda.current_stream()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        for _ in range(10):
            big.fill_(0.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1000.0 / reps

    cur = torch.cuda.current_stream()

    def one():
        sort("all", cur)

    def seq():
        sort("small", cur)
        sort("large", cur)

    def par():
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        sort("small", s1)
        sort("large", s2)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    for name, fn in (("one sort, 26 fields", one), ("small then large, one stream", seq), ("small || large, two streams", par),
                     ("small only", lambda: sort("small", cur)), ("large only", lambda: sort("large", cur))):
        print("%-32s %7.1f us" % (name, timed(fn)))


if __name__ == "__main__":
    main()
