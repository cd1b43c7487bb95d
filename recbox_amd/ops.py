"""torch.autograd glue over the C ABI (include/recbox_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the
autograd graph.  Every forward/backward below is one or a few calls into
librecbox_hip.so on raw ``data_ptr()``s.  There is no eager/PyTorch fallback: a
CPU tensor or a missing library raises.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (FIELD_CATEGORICAL, FIELD_DENSE, FIELD_NUMERIC, POOL_CONCAT, POOL_MEAN_ID, POOL_MEAN_VALUE,
                   POOL_NONE, POOL_SUM, POOL_SUM_ID, RBX_NO_ID, check, lib)

_DTYPE_CODE = {torch.int32: _lib.RBX_I32, torch.int64: _lib.RBX_I64,
               torch.float32: _lib.RBX_F32, torch.float64: _lib.RBX_F64}


class config(object):
    """Run-time switches of the host layer."""
    # The reference raises IndexError for an out-of-range id (nn.Embedding on CPU).
    # The kernels flag it on device; checking the flag costs one sync per call.
    check_ids = os.environ.get("RECBOX_AMD_CHECK_IDS", "1") != "0"


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("recbox_amd: %s must live on the GPU (got %s); the hot path has no CPU fallback"
                           % (what, t.device))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class FieldSpec(object):
    """Static description of one feature inside an EmbedPlan."""
    __slots__ = ("name", "kind", "pool", "dim", "seq_len", "vocab", "padding_idx", "mask_id", "eps",
                 "out_off", "param")

    def __init__(self, name, kind, dim, out_off, param=-1, pool=POOL_NONE, seq_len=1, vocab=0,
                 padding_idx=None, mask_id=None, eps=0.0):
        self.name, self.kind, self.dim, self.out_off, self.param = name, kind, dim, out_off, param
        self.pool, self.seq_len, self.vocab = pool, seq_len, vocab
        self.padding_idx, self.mask_id, self.eps = padding_idx, mask_id, eps

    @property
    def width(self):
        return self.dim * (self.seq_len if self.pool == POOL_CONCAT else 1)


class EmbedPlan(object):
    """A reusable, pre-filled rbx_field_t array for one layer call signature.

    Only the per-batch members (ids pointer / strides / dtype, grad pointer) are
    rewritten per call, which keeps the host cost of a 39-field lookup to a few
    microseconds of ctypes stores."""

    def __init__(self, specs, width):
        if not 0 < len(specs) <= _lib.RBX_MAX_FIELDS:
            raise NotImplementedError("a single lookup supports 1..%d features, got %d"
                                      % (_lib.RBX_MAX_FIELDS, len(specs)))
        self.specs = list(specs)
        self.width = int(width)
        self.n = len(specs)
        self.arr = (_lib.rbx_field_t * self.n)()
        self.needs_row_scale = any(s.pool in (POOL_MEAN_VALUE, POOL_MEAN_ID) for s in specs)
        for f, s in zip(self.arr, specs):
            f.kind, f.pool, f.dim, f.seq_len = s.kind, s.pool, s.dim, s.seq_len
            f.vocab = s.vocab
            f.padding_idx = RBX_NO_ID if s.padding_idx is None else int(s.padding_idx)
            f.mask_id = RBX_NO_ID if s.mask_id is None else int(s.mask_id)
            f.out_off = s.out_off
            f.eps = s.eps
            f.ids_stride_l = 0

    def bind_inputs(self, inputs):
        """Point the descriptors at this batch; returns (B, kept tensors)."""
        keep = []
        B = None
        for f, s, t in zip(self.arr, self.specs, inputs):
            _require_cuda(t, "input '%s'" % s.name)
            if t.dtype not in _DTYPE_CODE:
                t = t.float() if (t.is_floating_point() or s.kind != FIELD_CATEGORICAL) else t.long()
            want_dims = 2 if s.seq_len > 1 or (s.kind == FIELD_CATEGORICAL and s.pool != POOL_NONE) else 1
            if want_dims == 1:
                if t.dim() != 1:
                    t = t.reshape(-1)
            else:
                if t.dim() != 2 or t.shape[1] != s.seq_len:
                    raise ValueError("feature '%s': expected ids of shape [B, %d], got %s"
                                     % (s.name, s.seq_len, tuple(t.shape)))
                f.ids_stride_l = t.stride(1)
            if B is None:
                B = t.shape[0]
            elif t.shape[0] != B:
                raise ValueError("feature '%s': batch %d != %d" % (s.name, t.shape[0], B))
            f.ids = t.data_ptr()
            f.ids_stride_b = t.stride(0)
            f.ids_dtype = _DTYPE_CODE[t.dtype]
            keep.append(t)
        return B, keep

    def bind_params(self, params, grads=None):
        for f, s in zip(self.arr, self.specs):
            if s.param < 0:
                f.table = None
                f.grad = None
                continue
            p = params[s.param]
            f.table = p.data_ptr()
            g = None if grads is None else grads[s.param]
            f.grad = g.data_ptr() if g is not None else None


class KernelTimer(object):
    """HIP-event bracket around one C-ABI call on torch's current stream (the stream the
    kernel is enqueued on).  bench.py installs one to time the dominant kernel live."""

    def __init__(self, want):
        self.want = want            # predicate(meta) -> bool
        self.samples = []           # (start_event, end_event)

    def bracket(self, meta, fn):
        if not self.want(meta):
            return fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.samples.append((e0, e1))
        return rc

    def mean_ms(self):
        if not self.samples:
            return None
        return sum(a.elapsed_time(b) for a, b in self.samples) / len(self.samples)


kernel_timer = None   # set by bench.py


def _timed(meta, fn):
    return fn() if kernel_timer is None else kernel_timer.bracket(meta, fn)


def _check_status(status):
    if status is not None and int(status.item()) != 0:
        raise IndexError("index out of range in self")


class _EmbedLookup(torch.autograd.Function):
    """out[B, width] = multi-table gather (+pooling); backward = sorted segmented scatter-add."""

    @staticmethod
    def forward(ctx, plan, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        for p in params:
            _require_cuda(p, "embedding parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("recbox_amd: embedding parameters must be contiguous fp32")
        B, keep = plan.bind_inputs(inputs)
        plan.bind_params(params)
        dev = params[0].device if params else keep[0].device
        out = torch.empty((B, plan.width), dtype=torch.float32, device=dev)
        row_scale = torch.empty((plan.n, B), dtype=torch.float32, device=dev) if plan.needs_row_scale else None
        status = torch.zeros(1, dtype=torch.int32, device=dev) if config.check_ids else None
        check(_timed(("embed_fwd", plan.n, plan.width, B),
                     lambda: lib.rbx_embed_fwd(plan.arr, plan.n, B, _ptr(out), plan.width, _ptr(row_scale),
                                               _ptr(status), _stream())))
        _check_status(status)
        ctx.plan, ctx.inputs, ctx.row_scale, ctx.B = plan, keep, row_scale, B
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        plan, params, B = ctx.plan, ctx.params, ctx.B
        if dout.stride(1) != 1 or dout.dtype != torch.float32:
            dout = dout.contiguous().float()
        need = [i + 2 + len(ctx.inputs) for i in range(len(params))]
        want = [ctx.needs_input_grad[j] for j in need]
        # one zero-filled flat buffer for every dense gradient (single memset), views per parameter
        sizes = [p.numel() if w else 0 for p, w in zip(params, want)]
        padded = [(s + 3) // 4 * 4 for s in sizes]           # keep every view 16-byte aligned
        flat = torch.zeros(sum(padded), dtype=torch.float32, device=dout.device)
        grads, o = [], 0
        for p, w, s, ps in zip(params, want, sizes, padded):
            grads.append(flat[o:o + s].view_as(p) if w else None)
            o += ps
        plan.bind_inputs(ctx.inputs)
        plan.bind_params(params, grads)
        ws_bytes = lib.rbx_embed_bwd_workspace_size(plan.arr, plan.n, B)
        if ws_bytes == 0:
            check(_lib.RBX_ERR_INVALID if _lib.last_error() else _lib.RBX_OK)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dout.device)
        check(lib.rbx_embed_sort(plan.arr, plan.n, B, _ptr(ws), ws_bytes, None, _stream()))
        check(lib.rbx_embed_bwd(plan.arr, plan.n, B, _ptr(dout), dout.stride(0), _ptr(ctx.row_scale),
                                _ptr(ws), ws_bytes, _stream()))
        return (None, None) + (None,) * len(ctx.inputs) + tuple(grads)


def embed_lookup(plan, inputs, params):
    """Run plan over ``inputs`` (one tensor per feature) and ``params`` (distinct tables/weights)."""
    return _EmbedLookup.apply(plan, len(inputs), *inputs, *params)


class _Interaction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, mode):
        _require_cuda(emb, "feature_emb")
        if emb.dim() != 3:
            raise ValueError("feature_emb must be [B, F, D], got %s" % (tuple(emb.shape),))
        emb = emb.contiguous().float()
        B, F, D = emb.shape
        P = F * (F - 1) // 2
        shape = {0: (B, 1), 1: (B, D), 2: (B, P), 3: (B, P, D)}[mode]
        out = torch.empty(shape, dtype=torch.float32, device=emb.device)
        check(lib.rbx_interaction_fwd(_ptr(emb), B, F, D, mode, _ptr(out), _stream()))
        ctx.save_for_backward(emb)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dout):
        (emb,) = ctx.saved_tensors
        B, F, D = emb.shape
        dout = dout.contiguous().float()
        demb = torch.empty_like(emb)
        if F < 2 and ctx.mode >= 2:
            demb.zero_()
        check(lib.rbx_interaction_bwd(_ptr(emb), _ptr(dout), B, F, D, ctx.mode, _ptr(demb), _stream()))
        return demb, None


def interaction(emb, output="product_sum"):
    if output not in _lib.INTERACTION_MODES:
        raise ValueError("InnerProductInteraction output={} is not supported.".format(output))
    return _Interaction.apply(emb, _lib.INTERACTION_MODES[output])


class _Pool(torch.autograd.Function):
    """Pooling of a materialised [B, L, D] tensor (standalone pooling modules)."""

    @staticmethod
    def forward(ctx, emb, mask, numer_masked, denom, eps):
        _require_cuda(emb, "embedding_matrix")
        if emb.dim() != 3:
            raise ValueError("pooling expects [B, L, D], got %s" % (tuple(emb.shape),))
        emb = emb.contiguous().float()
        B, L, D = emb.shape
        if mask is not None:
            mask = mask.reshape(B, L).contiguous().float()
        out = torch.empty((B, D), dtype=torch.float32, device=emb.device)
        inv = torch.empty((B,), dtype=torch.float32, device=emb.device)
        check(lib.rbx_pool_fwd(_ptr(emb), _ptr(mask), B, L, D, int(numer_masked), int(denom), float(eps),
                               _ptr(out), _ptr(inv), _stream()))
        ctx.save_for_backward(mask, inv)
        ctx.shape, ctx.numer_masked = (B, L, D), int(numer_masked)
        return out

    @staticmethod
    def backward(ctx, dout):
        mask, inv = ctx.saved_tensors
        B, L, D = ctx.shape
        dout = dout.contiguous().float()
        demb = torch.empty((B, L, D), dtype=torch.float32, device=dout.device)
        check(lib.rbx_pool_bwd(_ptr(dout), _ptr(mask), _ptr(inv), B, L, D, ctx.numer_masked, _ptr(demb), _stream()))
        return demb, None, None, None, None


DENOM_NONE, DENOM_VALUE, DENOM_MASK, DENOM_LEN = 0, 1, 2, 3


def pool(emb, mask=None, numer_masked=False, denom=DENOM_NONE, eps=0.0):
    return _Pool.apply(emb, mask, numer_masked, denom, eps)
