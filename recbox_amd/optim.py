"""Sparse-row optimisers for embedding tables: one step reads and writes only the rows the batch touched.

The reference trains every table with a dense ``torch.optim`` step over the dense ``[V, D]`` gradient autograd builds
(/root/reference/recbox/ranking/pytorch/models/ranking_model.py:191-197: ``clip_grad_norm_`` + ``optimizer.step()``;
matching/pytorch/models/match_model.py:194-199).  At BASELINE cfg 3 (one 10 M x 128 table) that is 5 GB of gradient read
plus 10-20 GB of optimiser traffic per step around a 1.8 ms forward + backward for ~2 M touched rows.  SURVEY.md 8(b) names
the alternative ("or, opt-in, a sparse-row update path"): the sorted ids that drove the backward's deterministic
scatter-add still sit in its workspace and name every touched row once; ``rbx_embed_sparse_update`` /
``rbx_fm_sparse_update`` apply the optimiser rule to exactly those rows of the table and of its state tensors.

    tables, rest = recbox_amd.optim.split_parameters(model)
    opt_tables = recbox_amd.optim.SparseAdam(tables, lr=1e-3)      # or SparseSGD / SparseAdagrad
    opt_rest = torch.optim.Adam(rest, lr=1e-3)
    ...
    loss.backward(); opt_tables.step(); opt_rest.step(); opt_tables.zero_grad(); opt_rest.zero_grad(set_to_none=True)

Rules (the sparse branches of torch.optim, so that results can be checked against them):
  SparseSGD      w -= lr g                                             (torch.optim.SGD on a sparse gradient)
  SparseAdagrad  s += g^2;  w -= clr g / (sqrt(s) + eps), clr = lr / (1 + (t - 1) lr_decay)      (torch.optim.Adagrad)
  SparseAdam     lazy Adam: the moments of untouched rows do not decay                            (torch.optim.SparseAdam)
Dense stays the default and the parity surface of the layers; nothing here changes what ``p.grad`` is after a backward.
A table that several lookups of one step feed (SASRec's item table: the sequence lookup and gather_dot's candidates) is
stepped with the same rule over the union of their rows (the lookups leave their id tensors: ``ops.touched_ids``); a
parameter that got its gradient from something else than ``embed_lookup`` / ``gather_dot`` / ``fm_fused`` is stepped over
the non-zero rows of its dense gradient.
"""
import ctypes
import math

import torch

from . import _lib, ops
from ._lib import check, lib

__all__ = ["SparseSGD", "SparseAdagrad", "SparseAdam", "split_parameters"]


def split_parameters(model):
    """(weights of nn.Embedding modules, every other parameter) -- the usual split between a sparse-row optimiser and a
    dense ``torch.optim`` one."""
    tables, seen = [], set()
    for m in model.modules():
        if isinstance(m, torch.nn.Embedding) and id(m.weight) not in seen:
            seen.add(id(m.weight))
            tables.append(m.weight)
    rest = [p for p in model.parameters() if id(p) not in seen]
    return tables, rest


class _SparseRows(object):
    kind = None

    def __init__(self, params, lr, weight_decay=0.0, capturable=False):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        self.lr, self.weight_decay = float(lr), float(weight_decay)
        self.state = {}
        self.calls = {"rows": 0, "dense": 0}          # C-ABI sparse-row calls / dense fallbacks so far (tests read it)
        # capturable: the step count and the step size derived from it (Adam's bias correction, Adagrad's lr_decay) live in
        # DEVICE memory and advance inside the step (rbx_opt_advance), so that ``step()`` can be captured into a hipGraph and
        # every replay takes the step size of ITS step -- by value (the default) a capture would freeze the step size of the
        # capture step (at t = 3 about 0.2 of Adam's asymptotic one) for every replay, which is why ``step()`` refuses a
        # capture without it.  One counter per optimiser: every table of it is then on the same step (torch.optim keeps one
        # per parameter; the two agree whenever every table receives a gradient in every step, as in a captured step).
        self.capturable = bool(capturable)
        self._dev = None                              # {"t": float32[1], "step_size": float32[1]} on the tables' device
        ops.config.track_touched_rows = True

    # ---- per-rule pieces ------------------------------------------------------------------------------------------
    n_state = 0

    def _step_size(self, t):
        return self.lr

    def _betas_eps(self):
        return 0.0, 0.0, 0.0

    def _dense_rows(self, p, g, st, rows, t):          # the same rule through torch ops on ``rows`` (fallback, small params)
        raise NotImplementedError

    # ---- the step ---------------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _state_of(self, p):
        st = self.state.get(id(p))
        if st is None:
            st = self.state[id(p)] = {"step": 0, "s": [torch.zeros_like(p, memory_format=torch.contiguous_format)
                                                       for _ in range(self.n_state)]}
        return st

    def _opt_struct(self, t):
        b1, b2, eps = self._betas_eps()
        dptr = self._dev["step_size"].data_ptr() if self.capturable else None
        return _lib.rbx_opt_t(self.kind, self._step_size(t), b1, b2, eps, self.weight_decay, dptr)

    def _lr_decay(self):
        return 0.0

    def _advance(self, device):
        """capturable: t += 1 and the step size of step t, on the device, in stream order (one 1-thread launch)."""
        if self._dev is None:
            self._dev = {"t": torch.zeros(1, dtype=torch.float32, device=device),
                         "step_size": torch.zeros(1, dtype=torch.float32, device=device)}
        b1, b2, _ = self._betas_eps()
        check(lib.rbx_opt_advance(self.kind, self.lr, b1, b2, self._lr_decay(), ops._ptr(self._dev["t"]),
                                  ops._ptr(self._dev["step_size"]), ops._stream()))

    @staticmethod
    def _ptr_array(n, ptrs):
        arr = (ctypes.c_void_p * n)()
        for i, t in ptrs.items():
            arr[i] = t.data_ptr()
        return arr

    @torch.no_grad()
    def step(self):
        mine = dict((id(p), p) for p in self.params if p.grad is not None)
        if not mine:
            return
        if self.capturable:
            self._advance(next(iter(mine.values())).device)
        elif torch.cuda.is_available() and torch.cuda.is_current_stream_capturing() and self._step_size(1) != self._step_size(2):
            raise RuntimeError("%s.step() inside a hipGraph capture: this rule's step size depends on the step count and is "
                               "passed by value, so every replay would reuse the capture step's; build the optimiser with "
                               "capturable=True (step count and step size then live on the device)" % type(self).__name__)
        done, recs = set(), {}
        for pid in mine:
            rec = ops.touched.get(pid)
            if rec is not None and rec.ws is not None:
                recs[rec.serial] = rec
        for rec in sorted(recs.values(), key=lambda r: r.serial):
            done |= self._step_record(rec, mine)
        for pid, p in mine.items():
            if pid not in done:
                self._step_dense(p)
        # the records name gradient tensors and sort workspaces: once stepped they would only keep a [V, D] gradient alive
        # past zero_grad(set_to_none=True) (the next backward allocates its own before overwriting the record)
        for pid in mine:
            ops.touched.pop(pid, None)
            ops.touched_ids.pop(pid, None)

    def _step_dense(self, p):
        if self.capturable:
            raise RuntimeError("%s(capturable=True): the table %s got its gradient outside embed_lookup / fm_fused (or from "
                               "two lookups of one step); the dense fallback reads the step count on the host and cannot "
                               "be captured" % (type(self).__name__, tuple(p.shape)))
        self.calls["dense"] += 1
        st = self._state_of(p)
        st["step"] += 1
        g = p.grad
        if g.dim() >= 2 and g.shape[0] > 1:
            ent = ops.touched_ids.get(id(p))
            if ent is not None and ent[0] == g.data_ptr() and ent[1]:
                # every lookup that wrote into this gradient left its id tensors: the union of their rows, cut to the rows
                # that did receive a gradient (padding / masked ids do not) -- the same set the scan below finds, without
                # reading the [V, D] gradient
                self.calls["union"] = self.calls.get("union", 0) + 1
                rows = torch.unique(torch.cat([t.reshape(-1).long() for t in ent[1]]))
                rows = rows[(rows >= 0) & (rows < g.shape[0])]
                rows = rows[(g.reshape(g.shape[0], -1).index_select(0, rows) != 0).any(dim=1)]
            else:
                rows = (g.reshape(g.shape[0], -1) != 0).any(dim=1).nonzero().reshape(-1)
        else:
            rows = None
        self._dense_rows(p, g, st, rows, st["step"])

    def _step_record(self, rec, mine):
        """One C call for the tables of one lookup; returns the ids of the parameters it stepped (none when the record
        cannot be served as a whole: the caller then steps those parameters densely)."""
        groups = []          # per plan: {field index: parameter}
        for plan, plist, glist in zip(rec.plans, rec.params, rec.grads):
            sel = {}
            if plan is not None:
                for i, sp in enumerate(plan.specs):
                    if sp.kind != _lib.FIELD_CATEGORICAL or sp.param < 0:
                        continue
                    p, g = plist[sp.param], glist[sp.param]
                    if g is None:
                        continue                                   # frozen table: not part of the sort's plan
                    # every table of the sort's plan must be this optimiser's, contiguous, and still hold the gradient
                    # that backward wrote: the kernel walks ALL of the plan's sorted rows
                    if (id(p) not in mine or not p.is_contiguous() or p.grad is None or p.grad.data_ptr() != g.data_ptr()):
                        return set()
                    sel[i] = p
            groups.append(sel)
        stepped, steps = set(), set()
        for sel in groups:
            for p in sel.values():
                if id(p) not in stepped:
                    st = self._state_of(p)
                    st["step"] += 1
                    steps.add(st["step"])
                    stepped.add(id(p))
        if not stepped:
            return stepped
        self.calls["rows"] += 1
        opt = self._opt_struct(max(steps))
        st_arrays = []
        for plan, sel in zip(rec.plans, groups):
            n = plan.n if plan is not None else 1
            st_arrays.append([self._ptr_array(n, dict((i, self._state_of(p)["s"][k]) for i, p in sel.items())
                                              if k < self.n_state else {}) for k in range(2)])
        lead = rec.plans[0] if rec.plans[0] is not None else rec.plans[-1]
        _, keep = lead.bind_inputs(rec.inputs)
        st = ops._stream()
        if rec.kind == "embed":
            plan = rec.plans[0]
            plan.bind_params(rec.params[0], rec.grads[0])
            check(lib.rbx_embed_sparse_update(plan.arr, plan.n, rec.B, ops._ptr(rec.ws), rec.ws_bytes, ctypes.byref(opt),
                                              st_arrays[0][0], st_arrays[0][1], st))
        else:
            emb_plan, lr_plan = rec.plans
            if emb_plan is not None:
                emb_plan.bind_params(rec.params[0], rec.grads[0])
                if lr_plan is not None:
                    lr_plan.bind_inputs(keep)
            if lr_plan is not None:
                lr_plan.bind_params(rec.params[1], rec.grads[1])
            ea = emb_plan.arr if emb_plan is not None else None
            la = lr_plan.arr if lr_plan is not None else None
            e_arr = st_arrays[0] if emb_plan is not None else [None, None]
            l_arr = st_arrays[1] if lr_plan is not None else [None, None]
            check(lib.rbx_fm_sparse_update(ea, la, lead.n, rec.B, ops._ptr(rec.ws), rec.ws_bytes, ctypes.byref(opt),
                                           e_arr[0], e_arr[1], l_arr[0], l_arr[1], st))
        return stepped


class SparseSGD(_SparseRows):
    kind = _lib.OPT_SGD
    n_state = 0

    def __init__(self, params, lr=1e-2, weight_decay=0.0, capturable=False):
        super().__init__(params, lr, weight_decay, capturable)

    def _dense_rows(self, p, g, st, rows, t):
        g = g + self.weight_decay * p if self.weight_decay else g
        if rows is None:
            p.add_(g, alpha=-self.lr)
        else:
            p.index_add_(0, rows, g.index_select(0, rows), alpha=-self.lr)


class SparseAdagrad(_SparseRows):
    kind = _lib.OPT_ADAGRAD
    n_state = 1

    def __init__(self, params, lr=1e-2, lr_decay=0.0, eps=1e-10, weight_decay=0.0, initial_accumulator_value=0.0,
                 capturable=False):
        super().__init__(params, lr, weight_decay, capturable)
        self.lr_decay, self.eps, self.init_acc = float(lr_decay), float(eps), float(initial_accumulator_value)

    def _lr_decay(self):
        return self.lr_decay

    def _state_of(self, p):
        new = id(p) not in self.state
        st = super()._state_of(p)
        if new and self.init_acc:
            st["s"][0].fill_(self.init_acc)
        return st

    def _step_size(self, t):
        return self.lr / (1.0 + (t - 1) * self.lr_decay)

    def _betas_eps(self):
        return 0.0, 0.0, self.eps

    def _dense_rows(self, p, g, st, rows, t):
        clr = self._step_size(t)
        g = g + self.weight_decay * p if self.weight_decay else g
        s = st["s"][0]
        if rows is None:
            s.addcmul_(g, g)
            p.addcdiv_(g, s.sqrt().add_(self.eps), value=-clr)
        else:
            gr = g.index_select(0, rows)
            sr = s.index_select(0, rows).addcmul_(gr, gr)
            s.index_copy_(0, rows, sr)
            p.index_add_(0, rows, gr / (sr.sqrt() + self.eps), alpha=-clr)


class SparseAdam(_SparseRows):
    kind = _lib.OPT_ADAM
    n_state = 2

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        super().__init__(params, lr, weight_decay, capturable)
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)

    def _step_size(self, t):
        return self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def _betas_eps(self):
        return self.b1, self.b2, self.eps

    def _dense_rows(self, p, g, st, rows, t):
        step = self._step_size(t)
        g = g + self.weight_decay * p if self.weight_decay else g
        m, v = st["s"]
        if rows is None:
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-step)
        else:
            gr = g.index_select(0, rows)
            mr = m.index_select(0, rows).mul_(self.b1).add_(gr, alpha=1 - self.b1)
            vr = v.index_select(0, rows).mul_(self.b2).addcmul_(gr, gr, value=1 - self.b2)
            m.index_copy_(0, rows, mr)
            v.index_copy_(0, rows, vr)
            p.index_add_(0, rows, mr / (vr.sqrt() + self.eps), alpha=-step)
