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
                ("kind", ctypes.c_int32), ("pool", ctypes.c_int32), ("eps", ctypes.c_float)]


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
