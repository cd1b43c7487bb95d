"""Every piece of the fused FM step ALONE, back to back, at the bench shape (B = 65 536, Criteo tables, D = 16): run it
under `rocprofv3 --kernel-trace --stats` and read the per-kernel averages -- what each kernel costs when nothing runs
beside it (inside the step they overlap and slow each other down).
    python profiles/ubench/fm_bwd_parts.py [reps]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                # noqa: E402
from recbox_amd import ops                                   # noqa: E402
from recbox_amd._lib import check, lib                       # noqa: E402
from recbox_amd.ranking.pytorch.models import FM             # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    B, D = 65536, 16
    fmw = bench.CriteoFeatureMap(D)
    model = FM(fmw.fm, D).to(dev)
    bench.init_weights(model)
    batch = bench.synthetic_batch(B, 1, os.environ.get("DIST", "uniform"), dev)
    X, y = bench.slice_inputs(fmw.fm, batch)
    emb = model.embedding_layer.embedding_layer
    lr = model.fm.lr_layer.embedding_layer.embedding_layer
    names, values, plan, _ = emb.plan_for(X)
    _, _, lplan, _ = lr.plan_for(X)
    ep, lp = plan.plan, lplan.plan
    eparams = [m.weight for m in plan.modules]
    lparams = [m.weight for m in lplan.modules]
    Bn, keep = ep.bind_inputs(values)
    lp.bind_inputs(keep)
    grads_e = [torch.zeros_like(p) for p in eparams]
    grads_l = [torch.zeros_like(p) for p in lparams]
    ep.bind_params(eparams, grads_e)
    lp.bind_params(lparams, grads_l)
    ea, la, n = ep.arr, lp.arr, ep.n
    ws_bytes = lib.rbx_fm_bwd_workspace_size(ea, la, n, B)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    print("workspace %.1f MB" % (ws_bytes / 1e6))
    logit = torch.empty(B, 1, device=dev)
    prob = torch.empty(B, 1, device=dev)
    ssum = torch.empty(B, D, device=dev)
    g = (torch.rand(B, device=dev) - 0.5) / B
    gb = torch.zeros(1, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    bias = model.fm.lr_layer.bias
    flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)

    def fwd():
        check(lib.rbx_fm_fwd(ea, la, n, B, P(bias), None, 0, 0, -1, None, 0, P(logit), P(prob), P(ssum), None, st))
    pieces = [
        ("fwd", fwd),
        ("sort_compact", lambda: check(lib.rbx_fm_sort_phases(ea, la, n, B, P(ws), ws_bytes, None, 1, st))),
        ("sort_blocksort", lambda: check(lib.rbx_fm_sort_phases(ea, la, n, B, P(ws), ws_bytes, None, 4, st))),
        ("sort_tierB", lambda: check(lib.rbx_fm_sort_phases(ea, la, n, B, P(ws), ws_bytes, None, 2, st))),
        ("bwd_tierA", lambda: check(lib.rbx_fm_bwd(ea, la, n, B, P(g), P(ssum), P(gb), 0, 1 | 16, P(ws), ws_bytes, st))),
        ("bwd_tierB", lambda: check(lib.rbx_fm_bwd(ea, la, n, B, P(g), P(ssum), P(gb), 0, 1 | 8, P(ws), ws_bytes, st))),
        ("bwd_numeric", lambda: check(lib.rbx_fm_bwd(ea, la, n, B, P(g), P(ssum), P(gb), 0, 2 | 4, P(ws), ws_bytes, st))),
        ("rezero", lambda: check(lib.rbx_fm_rezero(ea, la, n, B, P(ws), ws_bytes, st))),
    ]
    cold = os.environ.get("COLD", "0") != "0"
    for name, fn in pieces:
        fn()
    torch.cuda.synchronize()
    for name, fn in pieces:
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for r in range(reps):
            if cold:
                flush.zero_()
            e0[r].record()
            fn()
            e1[r].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in zip(e0, e1))
        print("%-16s median %7.1f us  min %7.1f" % (name, ts[len(ts) // 2], ts[0]))


if __name__ == "__main__":
    main()
