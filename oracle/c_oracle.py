"""ctypes/numpy front end of oracle/recbox_oracle.c (TEST INFRASTRUCTURE).

Builds descriptor arrays over numpy buffers (host pointers) with the same ``rbx_field_t``
layout the GPU library uses, so one test can feed identical descriptors to both sides."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liborc.so")
RBX_NO_ID = -(1 << 63)
I32, I64, F32, F64 = 0, 1, 2, 3
CATEGORICAL, NUMERIC, DENSE = 0, 1, 2
POOL_NONE, POOL_SUM, POOL_MEAN_VALUE, POOL_MEAN_ID, POOL_SUM_ID, POOL_CONCAT = range(6)
_DT = {np.dtype("int32"): I32, np.dtype("int64"): I64, np.dtype("float32"): F32, np.dtype("float64"): F64}


class Field(ctypes.Structure):
    _fields_ = [("ids", ctypes.c_void_p), ("table", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("ids_stride_b", ctypes.c_int64), ("ids_stride_l", ctypes.c_int64), ("vocab", ctypes.c_int64),
                ("padding_idx", ctypes.c_int64), ("mask_id", ctypes.c_int64), ("out_off", ctypes.c_int64),
                ("dim", ctypes.c_int32), ("seq_len", ctypes.c_int32), ("ids_dtype", ctypes.c_int32),
                ("kind", ctypes.c_int32), ("pool", ctypes.c_int32), ("eps", ctypes.c_float),
                ("table_stride", ctypes.c_int64)]        # always 0 here: the oracle restates contiguous tables


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "recbox_oracle.c")):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    lib = ctypes.CDLL(LIB)
    P, FP, i32, i64, f32 = ctypes.c_void_p, ctypes.POINTER(Field), ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    lib.orc_embed_fwd.restype = ctypes.c_int
    lib.orc_embed_fwd.argtypes = [FP, i32, i64, P, i64, P]
    lib.orc_embed_bwd.argtypes = [FP, i32, i64, P, i64, P]
    lib.orc_fm_fwd.argtypes = [FP, FP, i32, i64, P, P, P]
    lib.orc_fm_bwd.argtypes = [FP, FP, i32, i64, P, P, P]
    lib.orc_interaction_fwd.argtypes = [P, i64, i32, i32, i32, P]
    lib.orc_l2norm_fwd.argtypes = [P, i64, i32, f32, P]
    lib.orc_pairdot_fwd.argtypes = [P, P, i64, i32, i32, f32, P]
    u64 = ctypes.c_uint64
    lib.orc_philox4x32_10.argtypes = [P, P, P]
    lib.orc_negsample.argtypes = [i64, i64, i32, u64, u64, P, P, P, P, P]
    lib.orc_gather_rows.argtypes = [P, P, i64, P, i64, i64]
    lib.orc_gather_rows.restype = ctypes.c_int
    lib.orc_philox4x32_10.restype = None
    lib.orc_negsample.restype = None
    for fn in (lib.orc_embed_bwd, lib.orc_fm_fwd, lib.orc_fm_bwd, lib.orc_interaction_fwd, lib.orc_l2norm_fwd,
               lib.orc_pairdot_fwd):
        fn.restype = None
    return lib


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def field(ids, table=None, grad=None, kind=CATEGORICAL, dim=1, out_off=0, pool=POOL_NONE, padding_idx=None,
          mask_id=None, eps=0.0):
    """Descriptor over numpy arrays (kept alive by the caller)."""
    f = Field()
    f.ids = ids.ctypes.data
    f.table = table.ctypes.data if table is not None else None
    f.grad = grad.ctypes.data if grad is not None else None
    item = ids.dtype.itemsize
    f.ids_stride_b = ids.strides[0] // item
    f.ids_stride_l = ids.strides[1] // item if ids.ndim > 1 else 0
    f.seq_len = ids.shape[1] if ids.ndim > 1 else 1
    f.vocab = table.shape[0] if (table is not None and kind == CATEGORICAL) else 0
    f.padding_idx = RBX_NO_ID if padding_idx is None else padding_idx
    f.mask_id = RBX_NO_ID if mask_id is None else mask_id
    f.out_off, f.dim, f.kind, f.pool, f.eps = out_off, dim, kind, pool, eps
    f.ids_dtype = _DT[ids.dtype]
    return f


def array_of(fields):
    arr = (Field * len(fields))()
    for i, f in enumerate(fields):
        arr[i] = f
    return arr


def philox4x32_10(ctr, key):
    """Philox4x32-10 block: ctr 4 x uint32, key 2 x uint32 -> 4 x uint32."""
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    load().orc_philox4x32_10(ptr(c), ptr(k), ptr(out))
    return out


def negsample(num_items, rows, num_negs, seed, offset=0, pos=None, query=None, excl_offsets=None, excl_items=None):
    width = num_negs + (1 if pos is not None else 0)
    out = np.zeros((rows, width), dtype=np.int64)
    arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.int64) for a in (pos, query, excl_offsets, excl_items)]
    load().orc_negsample(num_items, rows, num_negs, seed, offset, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]),
                         ptr(out))
    return out


def gather_rows(column, index):
    column = np.ascontiguousarray(column)
    index = np.ascontiguousarray(index, dtype=np.int64).reshape(-1)
    out = np.zeros((index.size,) + column.shape[1:], dtype=column.dtype)
    row_bytes = column.dtype.itemsize * int(np.prod(column.shape[1:], dtype=np.int64))
    bad = load().orc_gather_rows(ptr(column), ptr(out), row_bytes, ptr(index), index.size, column.shape[0])
    return out, bool(bad)
