"""GPU parity of the sequence-block chains (csrc/rbx_seqblock.hip): one SASRec block of
third_party/rechub/models/matching/sasrec.py:81-92 (+ PointWiseFeedForward :110-124) restated in torch float64 on the CPU
against the C-ABI entry points and against the one-node block of recbox_amd.ops -- outputs, saved statistics, every
gradient; full slabs, a ragged last slab and fewer rows than one slab."""
import copy
import ctypes

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4
E = 64


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(ln1_w=torch.rand(E, generator=g) + 0.5, ln1_b=r(E) * 0.1, in_w=r(3 * E, E) * 0.15, in_b=r(3 * E) * 0.1,
                out_w=r(E, E) * 0.15, out_b=r(E) * 0.1, ln2_w=torch.rand(E, generator=g) + 0.5, ln2_b=r(E) * 0.1,
                w1=r(E, E) * 0.2, b1=r(E) * 0.1, w2=r(E, E) * 0.2, b2=r(E) * 0.1), g


def _ln(x, w, b, eps=1e-8):
    return torch.nn.functional.layer_norm(x, (E,), w, b, eps)


@pytest.mark.parametrize("M", [19, 32, 9000, 20480])
def test_qkv_chain_vs_float64(M):
    """rbx_seqblock_qkv_fwd: q = LayerNorm(x), Q = q Wq^T + bq, K | V = x [Wk; Wv]^T + [bk; bv] (sasrec.py:82-84)."""
    from recbox_amd._lib import lib
    P, g = _params(1)
    x = torch.randn(M, E, generator=g)
    d = {k: v.cuda() for k, v in P.items()}
    xc = x.cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    mean, rstd, q, Q, KV = f(M), f(M), f(M, E), f(M, E), f(M, 2 * E)
    rc = lib.rbx_seqblock_qkv_fwd(_p(xc), M, _p(d["ln1_w"]), _p(d["ln1_b"]), 1e-8, _p(d["in_w"]), _p(d["in_b"]), _p(mean),
                                  _p(rstd), _p(q), _p(Q), _p(KV), None)
    assert rc == 0
    torch.cuda.synchronize()
    x64 = x.double()
    q64 = _ln(x64, P["ln1_w"].double(), P["ln1_b"].double())
    assert_close(q, q64, TOL, "q")
    assert_close(Q, q64 @ P["in_w"][:E].double().t() + P["in_b"][:E].double(), TOL, "Q")
    assert_close(KV, x64 @ P["in_w"][E:].double().t() + P["in_b"][E:].double(), TOL, "K|V")
    assert_close(mean, x64.mean(1), TOL, "mean")
    assert_close(rstd, (x64.var(1, unbiased=False) + 1e-8).rsqrt(), TOL, "rstd")


@pytest.mark.parametrize("M,pro,with_n", [(19, True, True), (9000, True, True), (20480, True, False), (9000, False, True),
                                         (33, False, False)])
def test_ffn_chain_vs_float64(M, pro, with_n):
    """rbx_seqblock_ffn_fwd with and without the out-projection prologue; parameters present or NULL."""
    from recbox_amd._lib import lib
    P, g = _params(2)
    attn, res = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
    keep = (torch.rand(M, generator=g) > 0.3).float()
    d = {k: v.cuda() for k, v in P.items()}
    attn_c, res_c, keep_c = attn.cuda(), res.cuda(), keep.cuda()          # (held: the entry points take raw pointers)
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    mean, rstd, h, out = f(M), f(M), f(M, E), f(M, E)
    n = f(M, E) if with_n else None
    nob = not with_n                                       # the variant without n also runs without biases
    if pro:
        x = f(M, E)
        rc = lib.rbx_seqblock_ffn_fwd(_p(attn_c), _p(res_c), _p(d["out_w"]), None if nob else _p(d["out_b"]), _p(x), M,
                                      _p(d["ln2_w"]), _p(d["ln2_b"]), 1e-8, _p(d["w1"]), None if nob else _p(d["b1"]),
                                      _p(d["w2"]), None if nob else _p(d["b2"]), _p(keep_c), _p(mean), _p(rstd), _p(n),
                                      _p(h), _p(out), None, None, None, None, None)
        x64 = res.double() + attn.double() @ P["out_w"].double().t() + (0 if nob else P["out_b"].double())
        assert_close(x, x64, TOL, "x")
    else:
        x = res_c
        rc = lib.rbx_seqblock_ffn_fwd(None, None, None, None, _p(x), M, None, None, 1e-8, _p(d["w1"]),
                                      None if nob else _p(d["b1"]), _p(d["w2"]), None if nob else _p(d["b2"]), None, _p(mean),
                                      _p(rstd), _p(n), _p(h), _p(out), None, None, None, None, None)
        x64 = res.double()
        keep = torch.ones(M)
    assert rc == 0
    torch.cuda.synchronize()
    lw, lb = (P["ln2_w"].double(), P["ln2_b"].double()) if pro else (None, None)
    n64 = _ln(x64, lw, lb)
    h64 = torch.relu(n64 @ P["w1"].double().t() + (0 if nob else P["b1"].double()))
    o64 = (n64 + h64 @ P["w2"].double().t() + (0 if nob else P["b2"].double())) * keep.double().unsqueeze(1)
    if n is not None:
        assert_close(n, n64, TOL, "n")
    assert_close(h, h64, TOL, "h")
    assert_close(out, o64, TOL, "out")
    assert_close(mean, x64.mean(1), TOL, "mean")


@pytest.mark.parametrize("M", [19, 64, 9000, 40960])
def test_ffn_backward_chain_vs_float64(M):
    """rbx_seqblock_ffn_bwd: one pass from dout to the gradient of the LayerNorm input and all six parameter gradients."""
    from recbox_amd._lib import lib
    P, g = _params(4)
    x = torch.randn(M, E, generator=g)
    dout = torch.randn(M, E, generator=g)
    keep = (torch.rand(M, generator=g) > 0.3).float()
    ps = {k: P[k].double().requires_grad_(True) for k in ("ln2_w", "ln2_b", "w1", "b1", "w2", "b2")}
    x64 = x.double().requires_grad_(True)
    n64 = _ln(x64, ps["ln2_w"], ps["ln2_b"])
    h64 = torch.relu(n64 @ ps["w1"].t() + ps["b1"])
    out = (n64 + h64 @ ps["w2"].t() + ps["b2"]) * keep.double().unsqueeze(1)
    (out * dout.double()).sum().backward()
    d = {k: v.cuda() for k, v in P.items()}
    xc, dc, kc, hc = x.cuda(), dout.cuda(), keep.cuda(), h64.detach().float().cuda()
    mean, rstd = x64.detach().mean(1).float().cuda(), (x64.detach().var(1, unbiased=False) + 1e-8).rsqrt().float().cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    dx, dw1, db1, dw2, db2, dg, db = f(M, E), f(E, E), f(E), f(E, E), f(E), f(E), f(E)
    nbytes = lib.rbx_seqblock_ffn_bwd_workspace_size(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for rep in range(2):                                    # (twice: the sums are formed in a fixed order)
        rc = lib.rbx_seqblock_ffn_bwd(_p(dc), _p(kc), _p(hc), _p(xc), _p(mean), _p(rstd), M, _p(d["ln2_w"]), _p(d["ln2_b"]),
                                      _p(d["w1"]), _p(d["w2"]), _p(dx), _p(dw1), _p(db1), _p(dw2), _p(db2), _p(dg), _p(db),
                                      _p(ws), nbytes, None)
        assert rc == 0
        torch.cuda.synchronize()
        got = [t.clone() for t in (dx, dw1, db1, dw2, db2, dg, db)]
        if rep == 0:
            first = got
    for a, b in zip(first, got):
        assert torch.equal(a, b)
    want = [x64.grad, ps["w1"].grad, ps["b1"].grad, ps["w2"].grad, ps["b2"].grad, ps["ln2_w"].grad, ps["ln2_b"].grad]
    for name, a, b in zip(("dx", "dw1", "db1", "dw2", "db2", "dgamma", "dbeta"), got, want):
        assert_close(a, b, TOL * max(1.0, float(b.abs().max())), name)
    # parameter gradients are optional
    rc = lib.rbx_seqblock_ffn_bwd(_p(dc), _p(kc), _p(hc), _p(xc), _p(mean), _p(rstd), M, _p(d["ln2_w"]), _p(d["ln2_b"]),
                                  _p(d["w1"]), _p(d["w2"]), _p(dx), None, None, None, None, None, None, _p(ws), nbytes, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(dx, got[0])


@pytest.mark.parametrize("M", [19, 64, 9000, 40960])
def test_attention_input_backward_chain_vs_float64(M):
    """rbx_seqblock_attn_in_bwd: dq = dQ Wq + g, LayerNorm backward, + dK Wk + dV Wv, dgamma / dbeta."""
    from recbox_amd._lib import lib
    P, gen = _params(5)
    r = lambda *s: torch.randn(*s, generator=gen)
    x, dQ, dKV, g = r(M, E), r(M, E), r(M, 2 * E), r(M, E)
    lw = P["ln1_w"].double().requires_grad_(True)
    lb = P["ln1_b"].double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    W = P["in_w"].double()
    q = _ln(x64, lw, lb)
    # the scalar whose gradient is the chain: q feeds Q = q Wq^T (gradient dQ) and the residual (gradient g); e feeds K | V
    ((q @ W[:E].t()) * dQ.double()).sum().add((q * g.double()).sum()).add(((x64 @ W[E:].t()) * dKV.double()).sum()).backward()
    d = {k: v.cuda() for k, v in P.items()}
    xc, qc, kvc, gc = x.cuda(), dQ.cuda(), dKV.cuda(), g.cuda()
    mean = x64.detach().mean(1).float().cuda()
    rstd = (x64.detach().var(1, unbiased=False) + 1e-8).rsqrt().float().cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    de, dg, db = f(M, E), f(E), f(E)
    nbytes = lib.rbx_seqblock_attn_in_bwd_workspace_size(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        rc = lib.rbx_seqblock_attn_in_bwd(_p(qc), _p(kvc), _p(gc), _p(xc), _p(mean), _p(rstd), M, _p(d["ln1_w"]), _p(d["in_w"]),
                                          None, 1.0, _p(de), _p(dg), _p(db), _p(ws), nbytes, None)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (de, dg, db)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for name, a, b in zip(("de", "dgamma", "dbeta"), outs[1], (x64.grad, lw.grad, lb.grad)):
        assert_close(a, b, TOL * max(1.0, float(b.abs().max())), name)
    # the input stage's backward folded into the store: de * keep[row] * alpha
    keep = (torch.rand(M, generator=gen) > 0.3).float()
    kc = keep.cuda()
    rc = lib.rbx_seqblock_attn_in_bwd(_p(qc), _p(kvc), _p(gc), _p(xc), _p(mean), _p(rstd), M, _p(d["ln1_w"]), _p(d["in_w"]),
                                      _p(kc), 8.0, _p(de), _p(dg), _p(db), _p(ws), nbytes, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(de, outs[1][0] * kc.unsqueeze(1) * 8.0) and torch.equal(dg, outs[1][1])


@pytest.mark.parametrize("M", [19, 64, 9000, 40960])
def test_out_projection_backward_vs_float64(M):
    """rbx_seqblock_attn_out_bwd: dO = g Wo, dWo = g^T O, dbo = colsum g."""
    from recbox_amd._lib import lib
    P, gen = _params(6)
    g, O = torch.randn(M, E, generator=gen), torch.randn(M, E, generator=gen)
    gc, oc, wc = g.cuda(), O.cuda(), P["out_w"].cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    dO, dw, db = f(M, E), f(E, E), f(E)
    nbytes = lib.rbx_seqblock_attn_out_bwd_workspace_size(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        assert lib.rbx_seqblock_attn_out_bwd(_p(gc), _p(oc), M, _p(wc), _p(dO), _p(dw), _p(db), _p(ws), nbytes, None) == 0
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (dO, dw, db)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    g64, O64, W = g.double(), O.double(), P["out_w"].double()
    for name, a, b in zip(("dO", "dWo", "dbo"), outs[1], (g64 @ W, g64.t() @ O64, g64.sum(0))):
        assert_close(a, b, TOL * max(1.0, float(b.abs().max())), name)


@pytest.mark.parametrize("M", [19, 64, 9000, 40960])
def test_in_projection_weight_gradients_in_one_pass_vs_float64(M):
    """rbx_seqblock_inproj_dw: dWq = dQ^T LayerNorm(x), dWk | dWv = (dK | dV)^T x, and the three bias gradients."""
    from recbox_amd._lib import lib
    P, gen = _params(8)
    r = lambda *s: torch.randn(*s, generator=gen)
    x, dQ, dKV = r(M, E), r(M, E), r(M, 2 * E)
    x64 = x.double()
    q64 = _ln(x64, P["ln1_w"].double(), P["ln1_b"].double())
    want_w = torch.cat([dQ.double().t() @ q64, dKV.double().t() @ x64], 0)
    want_b = torch.cat([dQ.double().sum(0), dKV.double().sum(0)], 0)
    d = {k: v.cuda() for k, v in P.items()}
    xc, qc, kvc = x.cuda(), dQ.cuda(), dKV.cuda()
    mean, rstd = x64.mean(1).float().cuda(), (x64.var(1, unbiased=False) + 1e-8).rsqrt().float().cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    dw, db = f(3 * E, E), f(3 * E)
    nbytes = lib.rbx_seqblock_inproj_dw_workspace_size(M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        assert lib.rbx_seqblock_inproj_dw(_p(qc), _p(kvc), _p(xc), _p(mean), _p(rstd), M, _p(d["ln1_w"]), _p(d["ln1_b"]), _p(dw),
                                          _p(db), _p(ws), nbytes, None) == 0
        torch.cuda.synchronize()
        outs.append([dw.clone(), db.clone()])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert_close(dw, want_w, TOL * max(1.0, float(want_w.abs().max())), "d in_proj_weight")
    assert_close(db, want_b, TOL * max(1.0, float(want_b.abs().max())), "d in_proj_bias")


@pytest.mark.parametrize("M", [19, 9000])
def test_ffn_chain_rebuilds_the_residual_from_the_block_input(M):
    """rbx_seqblock_ffn_fwd with d_res = the block input e and its LayerNorm statistics: the same outputs as with the stored
    q = LayerNorm(e) as the residual; rbx_seqblock_qkv_fwd without a q output."""
    from recbox_amd._lib import lib
    P, g = _params(9)
    e, attn = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g)
    keep = (torch.rand(M, generator=g) > 0.3).float()
    d = {k: v.cuda() for k, v in P.items()}
    ec, ac, kc = e.cuda(), attn.cuda(), keep.cuda()
    f = lambda *s: torch.full(s, float("nan"), device="cuda")
    mean1, rstd1, q, Q, KV = f(M), f(M), f(M, E), f(M, E), f(M, 2 * E)
    assert lib.rbx_seqblock_qkv_fwd(_p(ec), M, _p(d["ln1_w"]), _p(d["ln1_b"]), 1e-8, _p(d["in_w"]), _p(d["in_b"]), _p(mean1),
                                    _p(rstd1), _p(q), _p(Q), _p(KV), None) == 0
    Q2, KV2 = f(M, E), f(M, 2 * E)
    assert lib.rbx_seqblock_qkv_fwd(_p(ec), M, _p(d["ln1_w"]), _p(d["ln1_b"]), 1e-8, _p(d["in_w"]), _p(d["in_b"]), _p(mean1),
                                    _p(rstd1), None, _p(Q2), _p(KV2), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(Q, Q2) and torch.equal(KV, KV2)
    outs = []
    for res, stats in ((q, (None, None, None, None)), (ec, (_p(mean1), _p(rstd1), _p(d["ln1_w"]), _p(d["ln1_b"])))):
        x, mean, rstd, h, out = f(M, E), f(M), f(M), f(M, E), f(M, E)
        assert lib.rbx_seqblock_ffn_fwd(_p(ac), _p(res), _p(d["out_w"]), _p(d["out_b"]), _p(x), M, _p(d["ln2_w"]), _p(d["ln2_b"]),
                                        1e-8, _p(d["w1"]), _p(d["b1"]), _p(d["w2"]), _p(d["b2"]), _p(kc), _p(mean), _p(rstd),
                                        None, _p(h), _p(out), stats[0], stats[1], stats[2], stats[3], None) == 0
        torch.cuda.synchronize()
        outs.append((x, h, out))
    for name, a, b in zip(("x", "h", "out"), outs[1], outs[0]):
        assert_close(a, b, 1e-5, name)


def _block64(e, P, keep, heads):
    """sasrec.py:81-92 in float64 (batch-first; causal mask; no dropout)."""
    B, L, _ = e.shape
    hd = E // heads
    q = _ln(e, P["ln1_w"], P["ln1_b"])
    Q = q @ P["in_w"][:E].t() + P["in_b"][:E]
    K = e @ P["in_w"][E:2 * E].t() + P["in_b"][E:2 * E]
    V = e @ P["in_w"][2 * E:].t() + P["in_b"][2 * E:]
    sp = lambda t: t.view(B, L, heads, hd).transpose(1, 2)
    s = (sp(Q) @ sp(K).transpose(-1, -2)) * hd ** -0.5
    s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    o = (torch.softmax(s, -1) @ sp(V)).transpose(1, 2).reshape(B, L, E)
    x = q + o @ P["out_w"].t() + P["out_b"]
    n = _ln(x, P["ln2_w"], P["ln2_b"])
    return (n + torch.relu(n @ P["w1"].t() + P["b1"]) @ P["w2"].t() + P["b2"]) * keep.unsqueeze(-1)


@pytest.mark.parametrize("B,L,heads,one_pass_bwd", [(64, 200, 1, True), (45, 200, 2, True), (300, 30, 1, True), (45, 200, 1, False)])
def test_block_as_one_node_vs_float64_and_vs_the_sublayer_nodes(B, L, heads, one_pass_bwd):
    """ops.sasrec_block (rbx_seqblock_* forward and backward) against the float64 restatement and against the two sub-layer
    nodes it replaces: output and every gradient (a ragged last slab at B L = 9000)."""
    from recbox_amd import ops
    P, g = _params(3)
    e = torch.randn(B, L, E, generator=g)
    keep = (torch.rand(B, L, generator=g) > 0.3).float()
    R = torch.randn(B, L, E, generator=g)
    names = list(P)

    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    e64 = e.double().requires_grad_(True)
    out64 = _block64(e64, P64, keep.double(), heads)
    (out64 * R.double()).sum().backward()

    def run(chains):
        mha = torch.nn.MultiheadAttention(E, heads, 0.0)
        n1, n2 = torch.nn.LayerNorm(E, eps=1e-8), torch.nn.LayerNorm(E, eps=1e-8)
        with torch.no_grad():
            mha.in_proj_weight.copy_(P["in_w"]); mha.in_proj_bias.copy_(P["in_b"])
            mha.out_proj.weight.copy_(P["out_w"]); mha.out_proj.bias.copy_(P["out_b"])
            n1.weight.copy_(P["ln1_w"]); n1.bias.copy_(P["ln1_b"]); n2.weight.copy_(P["ln2_w"]); n2.bias.copy_(P["ln2_b"])
        mha, n1, n2 = mha.cuda(), n1.cuda(), n2.cuda()
        ffn = [P[k].clone().cuda().requires_grad_(True) for k in ("w1", "b1", "w2", "b2")]
        ec = e.clone().cuda().requires_grad_(True)
        kc = keep.cuda()
        old = ops.config.seqblock_bwd
        ops.config.seqblock_bwd = bool(one_pass_bwd)
        try:
            if chains:
                assert ops.seqblock_supported(ec, mha, False)
                out = ops.sasrec_block(ec, n1, mha, n2, ffn[0], ffn[1], ffn[2], ffn[3], kc)
            else:
                x = ops.sasrec_attention_sublayer(ec, n1, mha)
                out = ops.sasrec_ffn_sublayer(x, n2, ffn[0], ffn[1], ffn[2], ffn[3], kc, keep_is_mask=True)
            (out * R.cuda()).sum().backward()
        finally:
            ops.config.seqblock_bwd = old
        grads = dict(ln1_w=n1.weight.grad, ln1_b=n1.bias.grad, in_w=mha.in_proj_weight.grad, in_b=mha.in_proj_bias.grad,
                     out_w=mha.out_proj.weight.grad, out_b=mha.out_proj.bias.grad, ln2_w=n2.weight.grad, ln2_b=n2.bias.grad,
                     w1=ffn[0].grad, b1=ffn[1].grad, w2=ffn[2].grad, b2=ffn[3].grad)
        return out.detach(), ec.grad, grads

    out, de, grads = run(True)
    assert_close(out, out64, TOL, "out")
    assert_close(de, e64.grad, TOL * max(1.0, float(e64.grad.abs().max())), "de")
    for k in names:
        w = P64[k].grad
        assert_close(grads[k], w, TOL * max(1.0, float(w.abs().max())), "d" + k)
    out_s, de_s, grads_s = run(False)
    assert_close(out, out_s, 1e-5, "out vs sub-layer nodes")
    assert_close(de, de_s, 1e-5 * max(1.0, float(de_s.abs().max())), "de vs sub-layer nodes")
    for k in names:
        assert_close(grads[k], grads_s[k], 1e-5 * max(1.0, float(grads_s[k].abs().max())), "d%s vs sub-layer nodes" % k)


@pytest.mark.parametrize("B,L", [(64, 200), (45, 200)])
def test_input_stage_inside_the_first_block_equals_the_two_nodes(B, L):
    """ops.sasrec_block(input_stage=(pos_rows, alpha)) -- SASRec's ``(sqrt(D) e + position) * ~timeline_mask`` (sasrec.py:68-77)
    formed by the first block's node, its backward folded into the store of the block's last backward pass -- against
    ops.sasrec_input followed by ops.sasrec_block: output, the gradient of the raw item block, of the position rows and of
    every parameter."""
    from recbox_amd import ops
    P, g = _params(7)
    e = torch.randn(B, L, E, generator=g) * 0.1
    pos = torch.randn(L + 3, E, generator=g) * 0.1
    keep = (torch.rand(B, L, generator=g) > 0.3).float()
    R = torch.randn(B, L, E, generator=g)

    def run(inside):
        mha = torch.nn.MultiheadAttention(E, 1, 0.0)
        n1, n2 = torch.nn.LayerNorm(E, eps=1e-8), torch.nn.LayerNorm(E, eps=1e-8)
        with torch.no_grad():
            mha.in_proj_weight.copy_(P["in_w"]); mha.in_proj_bias.copy_(P["in_b"])
            mha.out_proj.weight.copy_(P["out_w"]); mha.out_proj.bias.copy_(P["out_b"])
            n1.weight.copy_(P["ln1_w"]); n1.bias.copy_(P["ln1_b"]); n2.weight.copy_(P["ln2_w"]); n2.bias.copy_(P["ln2_b"])
        mha, n1, n2 = mha.cuda(), n1.cuda(), n2.cuda()
        ffn = [P[k].clone().cuda().requires_grad_(True) for k in ("w1", "b1", "w2", "b2")]
        ec = e.clone().cuda().requires_grad_(True)
        pc = pos.clone().cuda().requires_grad_(True)
        kc = keep.cuda()
        if inside:
            out = ops.sasrec_block(ec, n1, mha, n2, ffn[0], ffn[1], ffn[2], ffn[3], kc, input_stage=(pc[:L], 8.0))
        else:
            x = ops.sasrec_input(ec, pc[:L], kc, alpha=8.0)
            out = ops.sasrec_block(x, n1, mha, n2, ffn[0], ffn[1], ffn[2], ffn[3], kc)
        (out * R.cuda()).sum().backward()
        return [out.detach(), ec.grad, pc.grad, n1.weight.grad, n1.bias.grad, mha.in_proj_weight.grad, mha.in_proj_bias.grad,
                mha.out_proj.weight.grad, n2.weight.grad] + [t.grad for t in ffn]

    a, b = run(True), run(False)
    names = ("out", "de_raw", "dpos", "dln1_w", "dln1_b", "din_w", "din_b", "dout_w", "dln2_w", "dw1", "db1", "dw2", "db2")
    for n, x, y in zip(names, a, b):
        assert_close(x, y, 1e-5 * max(1.0, float(y.abs().max())), n)
    assert float(a[2][L:].abs().max()) == 0.0                   # rows of the position table beyond L receive nothing
