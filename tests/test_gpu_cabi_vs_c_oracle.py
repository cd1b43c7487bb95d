"""Raw C-ABI calls (ctypes on device pointers, no Python layer in between) against the plain-C
oracle fed with IDENTICAL descriptors on host copies: id dtypes, strided id columns, every pool
mode, shared tables, padding rows, scalar / float4 / multi-unit lane-group paths, and the
full-size (B = 65 536) configuration through size-independent properties."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


def _lib():
    from recbox_amd import _lib
    return _lib


def _dev_field(L, host_field, ids_t, table_t, grad_t):
    f = L.rbx_field_t()
    for name, _ in L.rbx_field_t._fields_:
        setattr(f, name, getattr(host_field, name))
    f.ids = ids_t.data_ptr()
    f.table = table_t.data_ptr() if table_t is not None else None
    f.grad = grad_t.data_ptr() if grad_t is not None else None
    return f


def _mixed_case(B, D, seed, id_dtype, Lh=6):
    """numeric + one-hot + shared table + all pool modes + dense, ids read from a strided block."""
    g = np.random.default_rng(seed)
    V1, V2 = 50, 9
    block = np.zeros((B, 4 + 3 * Lh), dtype=id_dtype)             # one [B, cols] batch tensor, column views
    block[:, 0] = g.random(B).astype(id_dtype) if np.issubdtype(id_dtype, np.floating) else g.integers(0, 3, B)
    block[:, 1] = g.integers(0, V1, B)
    block[:, 2] = g.integers(0, V2, B)
    block[:, 3] = g.integers(0, 5, B)
    lens = g.integers(0, Lh + 1, B)
    for k in range(3):
        h = g.integers(1, V1 if k < 2 else V2, (B, Lh))        # the third block indexes the small table
        h[np.arange(Lh)[None, :] >= lens[:, None]] = 0
        block[:, 4 + k * Lh:4 + (k + 1) * Lh] = h
    W1 = (g.standard_normal((V1, D)) * 0.1).astype(np.float32)
    W1[0] = 0
    W2 = (g.standard_normal((V2, D)) * 0.1).astype(np.float32)
    wn = (g.standard_normal(D) * 0.1).astype(np.float32)
    specs = [dict(col=slice(0, 1), table="wn", kind=C.NUMERIC),
             dict(col=slice(1, 2), table="W1", padding_idx=0),
             dict(col=slice(2, 3), table="W2"),
             dict(col=slice(4, 4 + Lh), table="W1", pool=C.POOL_MEAN_VALUE, padding_idx=0, eps=1e-12),
             dict(col=slice(4 + Lh, 4 + 2 * Lh), table="W1", pool=C.POOL_MEAN_ID, mask_id=0, eps=1e-16, padding_idx=0),
             dict(col=slice(4 + 2 * Lh, 4 + 3 * Lh), table="W2", pool=C.POOL_SUM_ID, mask_id=0),
             dict(col=slice(4, 4 + Lh), table="W1", pool=C.POOL_SUM, padding_idx=0),
             dict(col=slice(4 + 2 * Lh, 4 + 3 * Lh), table="W2", pool=C.POOL_CONCAT),
             dict(col=slice(3, 4), table=None, kind=C.DENSE)]
    return block, {"W1": W1, "W2": W2, "wn": wn}, specs, Lh


@pytest.mark.parametrize("B,D,id_dtype,Lh", [(1, 16, np.int64, 6), (77, 16, np.float64, 6), (300, 7, np.int32, 6),
                                             (129, 1, np.float32, 6), (64, 128, np.int64, 6), (33, 252, np.int64, 6),
                                             # long histories: several LDS compaction chunks per (sample, feature) in
                                             # every lane-group shape of the sequence kernel (cfg 5 has L = 200)
                                             (40, 64, np.int64, 200), (9, 16, np.int32, 300), (20, 128, np.int64, 70),
                                             (12, 4, np.float64, 130), (7, 32, np.int64, 257), (5, 256, np.int64, 65)])
def test_embed_fwd_bwd_raw_cabi(B, D, id_dtype, Lh):
    L = _lib()
    orc = C.load()
    block, W, specs, Lh = _mixed_case(B, D, seed=B + D, id_dtype=id_dtype, Lh=Lh)
    dW = {k: np.zeros_like(v) for k, v in W.items()}
    dblock = torch.from_numpy(block).cuda()
    dWt = {k: torch.from_numpy(v).cuda() for k, v in W.items()}
    dG = {k: torch.zeros_like(v) for k, v in dWt.items()}
    host, dev, off = [], [], 0
    for s in specs:
        ids = block[:, s["col"]] if (s["col"].stop - s["col"].start) > 1 else block[:, s["col"].start]
        ids_d = dblock[:, s["col"]] if ids.ndim > 1 else dblock[:, s["col"].start]
        kind = s.get("kind", C.CATEGORICAL)
        dim = 1 if kind == C.DENSE else D
        t = s["table"]
        f = C.field(ids, W[t] if t else None, dW[t] if t else None, kind=kind, dim=dim, out_off=off,
                    pool=s.get("pool", C.POOL_NONE), padding_idx=s.get("padding_idx"), mask_id=s.get("mask_id"),
                    eps=s.get("eps", 0.0))
        if kind == C.NUMERIC:
            f.vocab = 0
        host.append(f)
        dev.append(_dev_field(L, f, ids_d, dWt[t] if t else None, dG[t] if t else None))
        off += dim * (f.seq_len if f.pool == C.POOL_CONCAT else 1)
    n, width = len(host), off
    harr = C.array_of(host)
    darr = (L.rbx_field_t * n)(*dev)
    out0 = np.zeros((B, width), np.float32)
    sc0 = np.zeros((n, B), np.float32)
    assert orc.orc_embed_fwd(harr, n, B, C.ptr(out0), width, C.ptr(sc0)) == 0
    out1 = torch.empty(B, width, device="cuda")
    sc1 = torch.zeros(n, B, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.check(L.lib.rbx_embed_fwd(darr, n, B, out1.data_ptr(), width, sc1.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert_close(out1, out0, 1e-5, "forward")
    R = np.random.default_rng(1).standard_normal((B, width)).astype(np.float32)
    orc.orc_embed_bwd(harr, n, B, C.ptr(R), width, C.ptr(sc0))
    ws_bytes = L.lib.rbx_embed_bwd_workspace_size(darr, n, B)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    Rd = torch.from_numpy(R).cuda()
    L.check(L.lib.rbx_embed_sort(darr, n, B, ws.data_ptr(), ws_bytes, status.data_ptr(), None))
    L.check(L.lib.rbx_embed_bwd(darr, n, B, Rd.data_ptr(), width, sc1.data_ptr(), 1, ws.data_ptr(), ws_bytes, None))
    torch.cuda.synchronize()
    # a 9-row table receives B * Lh * 3 contributions: the fp32 sums (oracle: left to right, kernel: chunked, fixed
    # order) differ by rounding that grows with sqrt(#terms); 1e-4 holds for the short histories
    gtol = 1e-4 * max(1.0, (B * Lh) ** 0.5 / 8.0)
    for k in W:
        assert_close(dG[k], dW[k], gtol, "grad " + k)
    # accumulate = 1 really accumulates: a second pass doubles the gradients
    L.check(L.lib.rbx_embed_bwd(darr, n, B, Rd.data_ptr(), width, sc1.data_ptr(), 1, ws.data_ptr(), ws_bytes, None))
    torch.cuda.synchronize()
    for k in W:
        assert_close(dG[k], 2 * dW[k], 2 * gtol, "accumulated grad " + k)


def test_bad_descriptors_return_error_codes():
    L = _lib()
    arr = (L.rbx_field_t * 1)()
    assert L.lib.rbx_embed_fwd(arr, 0, 4, None, 4, None, None, None) == L.RBX_ERR_INVALID
    assert b"n_fields" in L.lib.rbx_last_error()
    arr[0].ids, arr[0].kind, arr[0].dim, arr[0].seq_len = 1, 7, 4, 1
    out = torch.empty(4, 4, device="cuda")
    assert L.lib.rbx_embed_fwd(arr, 1, 4, out.data_ptr(), 4, None, None, None) == L.RBX_ERR_UNSUPPORTED
    with pytest.raises(NotImplementedError):
        L.check(L.RBX_ERR_UNSUPPORTED)
    assert L.lib.rbx_interaction_fwd(out.data_ptr(), 4, 4, 2, 2, 9, out.data_ptr(), None) == L.RBX_ERR_INVALID
    assert L.lib.rbx_attn_fwd(out.data_ptr(), out.data_ptr(), out.data_ptr(), None, 1, 2, 2, 48, 1.0, 0, 0.0,
                              out.data_ptr(), None, None, None) == L.RBX_ERR_UNSUPPORTED


def test_full_size_properties_b65536():
    """BASELINE.json configs[1] size: gather -> scatter round trip and linearity of the backward,
    checked without a CPU oracle pass over 1.7 M lookups."""
    from bench import CRITEO_VOCABS
    L = _lib()
    B, D = 65536, 16
    g = torch.Generator().manual_seed(7)
    tables = [torch.randn(v + 1, D, generator=g).cuda() for v in CRITEO_VOCABS]
    ids = [torch.randint(1, v + 1, (B,), generator=g).cuda() for v in CRITEO_VOCABS]
    grads = [torch.zeros_like(t) for t in tables]
    n = len(tables)
    arr = (L.rbx_field_t * n)()
    for i in range(n):
        f = arr[i]
        f.ids, f.table, f.grad = ids[i].data_ptr(), tables[i].data_ptr(), grads[i].data_ptr()
        f.ids_stride_b, f.ids_stride_l, f.vocab = 1, 0, CRITEO_VOCABS[i] + 1
        f.padding_idx, f.mask_id, f.out_off = 0, L.RBX_NO_ID, i * D
        f.dim, f.seq_len, f.ids_dtype, f.kind, f.pool, f.eps = D, 1, L.RBX_I64, L.FIELD_CATEGORICAL, L.POOL_NONE, 0.0
    out = torch.empty(B, n * D, device="cuda")
    L.check(L.lib.rbx_embed_fwd(arr, n, B, out.data_ptr(), n * D, None, None, None))
    for i in (2, 8, 25):                                 # rows are copied bit-exactly
        assert torch.equal(out[:, i * D:(i + 1) * D], tables[i][ids[i]])
    ws_bytes = L.lib.rbx_embed_bwd_workspace_size(arr, n, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    L.check(L.lib.rbx_embed_sort(arr, n, B, ws.data_ptr(), ws_bytes, None, None))
    ones = torch.ones(B, n * D, device="cuda")
    L.check(L.lib.rbx_embed_bwd(arr, n, B, ones.data_ptr(), n * D, None, 0, ws.data_ptr(), ws_bytes, None))
    torch.cuda.synchronize()
    for i in range(n):                                   # dY = 1  =>  dW[r, :] = multiplicity of r (integers: exact)
        counts = torch.bincount(ids[i], minlength=CRITEO_VOCABS[i] + 1).float()
        assert torch.equal(grads[i][:, 0], counts), "field %d" % i
        assert torch.equal(grads[i][:, D - 1], counts)
