"""GPU parity of the matching side (recbox.core EmbeddingLayer, rechub-style layers and the
DSSM / YoutubeDNN / DeepFM / SASRec models), the dense tower GEMM and the attention core against
the golden fixtures of the live reference.  Everything goes through the C ABI."""
from collections import OrderedDict

import pytest
import torch
import torch.nn.functional as F

from conftest import Fixture, assert_close, assert_grads_close, load_params
from test_oracle_golden import CRITEO_SMALL_VOCABS, _MFM, matching_specs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cuda(d):
    return OrderedDict((k, v.cuda()) for k, v in d.items())


def test_matching_embedding_golden():
    import recbox_amd.core.pytorch.layers as C
    fx = Fixture("matching_embedding")
    layer = load_params(C.EmbeddingLayer(_MFM(matching_specs()), 8), fx["p"]).cuda()
    X = _cuda(fx.tensors("in"))
    u = layer(X, feature_source="user")
    i = layer(X, feature_source="item")
    assert i.dim() == 2 and tuple(u.shape) == (7, 3, 8)
    assert_close(u, fx["out"]["user"], TOL)
    assert_close(i, fx["out"]["item"], TOL)
    ((u * X["Ru"]).sum() + (i * X["Ri"]).sum()).backward()
    assert_grads_close(layer, fx["g"], TOL)
    el = layer.embedding_layer.embedding_layers
    assert el["item_id"] is el["user_hist"] and type(el["user_id"]) is torch.nn.Embedding


def _rh():
    from recbox_amd.rechub.basic import features as Fe, layers as La
    return Fe, La


def test_rechub_embedding_golden():
    Fe, La = _rh()
    Sp, Sq, De = Fe.SparseFeature, Fe.SequenceFeature, Fe.DenseFeature
    D = 8
    feats = [Sp("uid", 13, D), Sq("hist_mean", 21, D, pooling="mean", shared_with="iid", padding_idx=0),
             Sq("hist_sum", 21, D, pooling="sum", shared_with="iid", padding_idx=0),
             Sp("iid", 21, D), De("price"), De("age"), Sq("tags", 9, D, pooling="mean")]
    fx = Fixture("rechub_embedding")
    layer = load_params(La.EmbeddingLayer(feats), fx["p"]).cuda()
    X = _cuda(fx.tensors("in"))
    sq = layer(X, feats, squeeze_dim=True)
    ns = layer(X, [f for f in feats if not isinstance(f, De)], squeeze_dim=False)
    assert_close(sq, fx["out"]["squeezed"], TOL)
    assert_close(ns, fx["out"]["stacked"], TOL)
    ((sq * X["Rq"]).sum() + (ns * X["Rn"]).sum()).backward()
    assert_grads_close(layer, fx["g"], TOL)
    cfe = [Sq("seq", 11, 8, pooling="concat"), Sq("pos", 11, 8, pooling="concat", shared_with="seq")]
    cl = load_params(La.EmbeddingLayer(cfe), fx["pc"]).cuda()
    co = cl({"seq": X["c_seq"], "pos": X["c_pos"]}, cfe)
    assert tuple(co.shape) == (7, 2, 5, 8)
    assert_close(co, fx["out"]["concat"], TOL)
    (co * X["Rc"]).sum().backward()
    assert_grads_close(cl, fx["gc"], TOL)           # the pad row DOES receive gradient here
    with pytest.raises(ValueError):
        layer(X, [De("price")], squeeze_dim=False)


def test_rechub_pooling_modules_golden():
    Fe, La = _rh()
    fx = Fixture("pooling")
    t = _cuda(fx.tensors("in"))
    keep = t["keep"].unsqueeze(1).float()
    cases = {"rechub_avg": lambda e: La.AveragePooling()(e, keep), "rechub_sum": lambda e: La.SumPooling()(e, keep),
             "rechub_avg_nomask": lambda e: La.AveragePooling()(e), "rechub_sum_nomask": lambda e: La.SumPooling()(e)}
    for key, fn in cases.items():
        e = t["E"].clone().requires_grad_(True)
        o = fn(e)
        assert_close(o, fx["out"][key], TOL, key)
        (o * t["R"]).sum().backward()
        assert_close(e.grad, fx["g"][key], TOL, "grad " + key)


def test_mlp_golden():
    import recbox_amd.core.pytorch.layers as C
    import recbox_amd.ranking.pytorch.layers as L
    Fe, La = _rh()
    fx = Fixture("mlp")
    x = fx.tensors("in")["x"].cuda()
    mods = {"core": C.MLP_Layer(12, output_dim=3, hidden_units=[16, 8], hidden_activations="ReLU",
                                dropout_rates=[0, 0], batch_norm=True),
            "block": L.MLP_Block(12, hidden_units=[16, 8], hidden_activations="ReLU", output_dim=1, batch_norm=False),
            "rechub": La.MLP(12, output_layer=True, dims=[16, 8], dropout=0, activation="relu")}
    for key, m in mods.items():
        load_params(m, {k[len(key) + 1:]: v for k, v in fx["p"].items() if k.startswith(key + ".")}).cuda().train()
        assert all(type(c) is not torch.nn.Linear or True for c in m.mlp)
        xi = x.clone().requires_grad_(True)
        o = m(xi)
        assert_close(o, fx["out"][key], TOL, key)
        (o * torch.from_numpy(fx["out"]["R_" + key]).cuda()).sum().backward()
        assert_close(xi.grad, fx["g"][key + ".x"], TOL)
        for n, p in m.named_parameters():
            assert_close(p.grad, fx["g"][key + "." + n], TOL, key + "." + n)


@pytest.mark.parametrize("M,N,K,act", [(1, 1, 1, None), (65, 33, 17, "relu"), (300, 130, 257, None),
                                       (9000, 40, 24, "relu"), (8192, 16, 2496, None),
                                       # tall and narrow (SASRec's [B*L, 64] x [64, 64]): the streaming dW / db kernel,
                                       # full and ragged quadrants, a row count that is no multiple of anything
                                       (20001, 64, 64, None), (20011, 64, 64, "relu"), (8192, 33, 64, "relu"),
                                       # 64 -> 128 and 128 -> 64 (the fused K | V projection and its dx): weights in LDS
                                       (20011, 128, 64, "relu"), (70007, 64, 128, None), (12345, 64, 20, None),
                                       (9999, 7, 5, None), (10000, 192, 64, None), (8200, 130, 33, "relu"), (8192, 256, 64, None),
                                       # logit heads (n == 1): the streaming GEMV / outer-product / weighted column-sum kernels
                                       (70001, 1, 400, None), (5000, 1, 1664, None), (3001, 1, 37, "relu"), (2, 1, 7, None),
                                       (66000, 1, 30, None),
                                       # tiles on the matrix edge in every role (rows / columns of the forward, of dx and of the
                                       # split-K dW): 1, 2, 3 and 4 live 32-row / 32-column blocks, narrow tails of one and two
                                       # blocks riding in the main launch, cfg 4's [*, 1677] x [400, 1677] with K >= 4096 rows
                                       (4200, 400, 1677, None), (4130, 168, 198, "relu"), (4100, 228, 168, None),
                                       (4099, 198, 400, None), (4160, 130, 140, None),
                                       # batch >= 8192 with a compute-bound layer: dW on the split-operand kernel with
                                       # transposed staging (K of that GEMM = the batch, split over workgroups), ragged everywhere
                                       (16384, 400, 300, None), (16500, 130, 257, "relu"), (9001, 264, 1677, None)])
def test_linear_matches_torch_fp32(M, N, K, act):
    """The MFMA GEMM (all three operand layouts, split-K weight grad) against torch fp32 on CPU."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y0 = F.linear(xr.double(), wr.double(), br.double())
    y0 = torch.relu(y0) if act else y0
    (y0 * R.double()).sum().backward()
    xc, wc, bc = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    y1 = ops.linear(xc, wc, bc, act)
    (y1 * R.cuda()).sum().backward()
    assert_close(y1, y0.float(), 2e-5 * max(1.0, K ** 0.5 / 8), "y")
    assert_close(xc.grad, xr.grad.float(), 1e-4, "dx")
    assert_close(wc.grad, wr.grad.float(), 1e-4 * max(1.0, M ** 0.5 / 16), "dw")
    assert_close(bc.grad, br.grad.float(), 1e-4 * max(1.0, M ** 0.5 / 16), "db")


@pytest.mark.parametrize("M,N,K", [(4100, 168, 70), (300, 400, 64), (4224, 256, 48), (129, 130, 17), (8192, 64, 64),
                                   (70007, 64, 64), (20011, 128, 64), (20011, 64, 128)])
def test_linear_epilogue_operands_on_every_tile_kind(M, N, K):
    """The optional tail of the GEMM epilogue -- y = (act(x W^T + b) + residual) * row_scale[row] (rbx_linear_fwd_fused) and
    dx = ((dy W) o [mask > 0]) + residual (rbx_linear_dx_fused) -- on interior tiles (operands fetched four outputs at a
    time, untested), on edge tiles, and on a narrow column tail that rides in the main launch, against float64."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    rs = (torch.rand(M, generator=g) > 0.3).float() * 1.5
    want = (torch.relu(x.double() @ w.double().t() + b.double()) + res.double()) * rs.double()[:, None]
    got = ops._lin_fwd(x.cuda(), w.cuda(), b.cuda(), act=1, residual=res.cuda(), row_scale=rs.cuda())
    assert_close(got, want.float(), 2e-5 * max(1.0, K ** 0.5 / 8), "fused forward")
    dy = torch.randn(M, N, generator=g)
    mask = torch.randn(M, K, generator=g)
    res2 = torch.randn(M, K, generator=g)
    want = (dy.double() @ w.double()) * (mask.double() > 0) + res2.double()
    got = ops._lin_dx(dy.cuda(), w.cuda(), mask=mask.cuda(), residual=res2.cuda())
    assert_close(got, want.float(), 2e-5 * max(1.0, N ** 0.5 / 8), "fused dx")
    got = ops._lin_dx(dy.cuda(), w.cuda())
    assert_close(got, (dy.double() @ w.double()).float(), 2e-5 * max(1.0, N ** 0.5 / 8), "plain dx")
    # each operand alone (the 64 -> 64 streaming kernel is instantiated per operand set)
    lin = x.double() @ w.double().t() + b.double()
    got = ops._lin_fwd(x.cuda(), w.cuda(), b.cuda(), act=0, residual=res.cuda())
    assert_close(got, (lin + res.double()).float(), 2e-5 * max(1.0, K ** 0.5 / 8), "forward + residual")
    got = ops._lin_fwd(x.cuda(), w.cuda(), b.cuda(), act=1, row_scale=rs.cuda())
    assert_close(got, (torch.relu(lin) * rs.double()[:, None]).float(), 2e-5 * max(1.0, K ** 0.5 / 8), "forward * row scale")
    got = ops._lin_dx(dy.cuda(), w.cuda(), mask=mask.cuda())
    assert_close(got, ((dy.double() @ w.double()) * (mask.double() > 0)).float(), 2e-5 * max(1.0, N ** 0.5 / 8), "masked dx")
    got = ops._lin_dx(dy.cuda(), w.cuda(), residual=res2.cuda())
    assert_close(got, ((dy.double() @ w.double()) + res2.double()).float(), 2e-5 * max(1.0, N ** 0.5 / 8), "dx + residual")


@pytest.mark.parametrize("M,N,K", [(8192, 400, 1677), (4100, 130, 257), (16384, 256, 512)])
def test_split_bf16_gemm_runs_and_is_as_accurate_as_the_f32_mfma(M, N, K):
    """y = x W^T and dx = dy W of a compute-bound tower layer run on the bf16 matrix cores (weights split into three bf16
    planes by rbx_split_bf16, activations inside gemm_bx6_kernel, six products per f32 product): the launch counter moves,
    and the error against float64 is of the size of the f32-MFMA kernel's own -- also for one-signed operands (ReLU
    outputs against positive weights), where a biased split would show."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.rand(M, K, generator=g)                      # one-signed, like ReLU outputs
    w = torch.rand(N, K, generator=g) / K
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    y64 = x.double() @ w.double().t() + b.double()
    dx64 = dy.double() @ w.double()
    xc, wc, bc, dyc = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    old = ops.config.gemm_bx6
    try:
        ops.config.gemm_bx6 = False
        y32, dx32 = ops._lin_fwd(xc, wc, bc), ops._lin_dx(dyc, wc)
        ops.config.gemm_bx6 = True
        n0 = ops.gemm_bx6_count()
        y6, dx6 = ops._lin_fwd(xc, wc, bc), ops._lin_dx(dyc, wc)
        assert ops.gemm_bx6_count() == n0 + int(K >= 256 and N >= 128) + int(N >= 256 and K >= 128)   # (ops._with_split_weights)
    finally:
        ops.config.gemm_bx6 = old
    for name, got6, got32, want in (("y", y6, y32, y64), ("dx", dx6, dx32, dx64)):
        e6 = float((got6.double().cpu() - want).abs().max())
        e32 = float((got32.double().cpu() - want).abs().max())
        scale = float(want.abs().max())
        assert e6 <= max(4.0 * e32, 2e-6 * scale), (name, e6, e32, scale)
        assert e6 <= 1e-5 * scale, (name, e6, scale)


def test_weight_gradients_beside_the_next_backward_node_are_the_same_bits():
    """ops.config.dw_beside_lookup (round 5): a tower layer's dW / db on a side stream beside whatever consumes its dx, joined
    at the end of the backward pass -- the same kernels on the same data: bit-identical gradients to the in-line order, eager
    and replayed from a hipGraph; a parameter that still holds a gradient takes the in-line path (autograd adds in place)."""
    from recbox_amd import dense, ops
    from recbox_amd.graph import GraphedStep
    torch.manual_seed(5)
    M, K = 8192, 300
    mods = torch.nn.Sequential(torch.nn.Linear(K, 256), torch.nn.BatchNorm1d(256), torch.nn.ReLU(),
                               torch.nn.Linear(256, 384), torch.nn.BatchNorm1d(384), torch.nn.ReLU(),
                               torch.nn.Linear(384, 1)).cuda().train()
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    r = torch.randn(M, 1, device="cuda")

    def step(accumulate=False):
        if not accumulate:
            for p in mods.parameters():
                p.grad = None
            x.grad = None
        (dense.run_sequential(mods, x) * r).sum().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in mods.parameters()] + [x.grad.clone()]

    old, old_check = ops.config.dw_beside_lookup, ops.config.check_ids
    try:
        ops.config.check_ids = False
        ops.config.dw_beside_lookup = False
        want = step()
        want2 = step(accumulate=True)
        ops.config.dw_beside_lookup = True
        got = step()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        got2 = step(accumulate=True)                     # gradients survive: the in-line path, sums as before
        for a, b in zip(got2, want2):
            assert torch.equal(a, b)

        def one():
            for p in mods.parameters():
                p.grad = None
            x.grad = None
            loss = (dense.run_sequential(mods, x) * r).sum()
            loss.backward()
            return loss
        g = GraphedStep(one, warmup=2)
        for _ in range(3):
            g()
        torch.cuda.synchronize()
        for a, b in zip([p.grad for p in mods.parameters()] + [x.grad], want):
            assert torch.equal(a, b)
    finally:
        ops.config.dw_beside_lookup, ops.config.check_ids = old, old_check


def test_weight_gradients_of_non_leaf_shared_and_hooked_weights_stay_in_line():
    """the side stream for dW / db (ops.config.dw_beside_lookup) is only for LEAF parameters that autograd merely
    stores.  A non-leaf weight (a slice / permute of a parameter: CrossNetMix's experts, the in-projection halves of
    nn.MultiheadAttention), a leaf served by two nodes of one pass, and a parameter watched by a tensor hook all compute in
    line: no launch on the side stream for them, gradients the bits of the in-line order, repeatably."""
    from recbox_amd import ops
    from recbox_amd.ranking.pytorch.layers.interactions import CrossNetMix
    torch.manual_seed(11)
    M, K = 8192, 96
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    r = torch.randn(M, K, device="cuda")
    W = torch.nn.Parameter(torch.randn(2 * K, K, device="cuda") * 0.1)
    b = torch.nn.Parameter(torch.randn(2 * K, device="cuda") * 0.1)
    shared = torch.nn.Linear(K, K).cuda()
    hooked = torch.nn.Linear(K, K).cuda()
    seen = []
    hooked.weight.register_hook(lambda g: seen.append(float(g.abs().sum())))
    mix = CrossNetMix(K, layer_num=2, low_rank=16, num_experts=3).cuda()
    params = [W, b] + list(shared.parameters()) + list(hooked.parameters()) + list(mix.parameters())

    def step():
        for p in params:
            p.grad = None
        x.grad = None
        h = ops.linear(x, W[:K], b[:K]) + ops.linear(x, W[K:], b[K:])                  # non-leaf halves of one parameter
        h = ops.linear(torch.tanh(ops.linear(h, shared.weight, shared.bias)), shared.weight, shared.bias)     # one leaf, two nodes
        h = ops.linear(h, hooked.weight, hooked.bias)
        h = mix(h)
        (h * r).sum().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params] + [x.grad.clone()]

    calls = []
    real = ops._run_beside
    old = ops.config.dw_beside_lookup
    try:
        ops.config.dw_beside_lookup = False
        want = step()
        ops.config.dw_beside_lookup = True
        ops._run_beside = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        for _ in range(3):
            got = step()
            for a, w in zip(got, want):
                assert torch.equal(a, w)
        # what went beside: only gate / bias-free leaves of CrossNetMix that are served once?  none here is -- the gate
        # weights are concatenated (non-leaf), U / V / C permuted or selected, `shared` is served twice (the FIRST node may
        # go beside; the second joins before it returns), `hooked` has a hook
        assert len(calls) <= 3 * 1, calls
        assert len(seen) == 4 and all(v == seen[0] for v in seen)
        assert not ops._beside_seen                                      # (cleared by the end-of-pass callback)
    finally:
        ops._run_beside = real
        ops.config.dw_beside_lookup = old


def test_a_backward_pass_that_raises_leaves_no_state_behind_for_the_next_one():
    """The per-pass registries of ops (parameters served in this pass, id lists of this pass) are cleared by autograd's
    end-of-pass callback -- which never runs for a pass that ends in an exception.  The next pass recognises the leftovers by
    the graph task id and starts clean: the side stream is used again and the gradients are the in-line ones."""
    from recbox_amd import ops

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    torch.manual_seed(2)
    lin = torch.nn.Linear(64, 48).cuda()
    x = torch.randn(8192, 64, device="cuda", requires_grad=True)
    calls = []
    real = ops._run_beside
    old = ops.config.dw_beside_lookup
    try:
        ops.config.dw_beside_lookup = False
        ops.linear(x, lin.weight, lin.bias).sum().backward()
        want = [lin.weight.grad.clone(), lin.bias.grad.clone()]
        ops.config.dw_beside_lookup = True
        ops._run_beside = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        lin.zero_grad(set_to_none=True)
        with pytest.raises(RuntimeError, match="boom"):
            ops.linear(Boom.apply(x), lin.weight, lin.bias).sum().backward()      # the Linear's node ran, then the pass died
        torch.cuda.synchronize()
        ops.join_beside()
        lin.zero_grad(set_to_none=True)
        n0 = len(calls)
        ops.linear(x, lin.weight, lin.bias).sum().backward()
        torch.cuda.synchronize()
        assert len(calls) == n0 + 1                                   # not mistaken for a parameter met twice in one pass
        assert torch.equal(lin.weight.grad, want[0]) and torch.equal(lin.bias.grad, want[1])
        assert not ops._beside_seen
    finally:
        ops._run_beside = real
        ops.config.dw_beside_lookup = old


def test_sasrec_in_projection_gradients_with_long_batches_equal_the_in_line_order():
    """rechub SASRec's `_mha` hands slices of in_proj_weight to ops.linear when the fused block chain is off; with
    B L >= 4096 those (non-leaf) weights must not take the side stream: gradients equal the in-line order bit for bit."""
    from recbox_amd import ops
    from recbox_amd.rechub.basic.features import SequenceFeature, SparseFeature
    from recbox_amd.rechub.models.matching import SASRec
    torch.manual_seed(3)
    V, D, L, B = 500, 32, 64, 128                                     # B L = 8192 rows per projection
    feats = [SequenceFeature("seq", V, D, pooling="concat", padding_idx=0),
             SequenceFeature("pos", V, D, pooling="concat", shared_with="seq"),
             SequenceFeature("neg", V, D, pooling="concat", shared_with="seq")]
    model = SASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1).cuda().train()
    g = torch.Generator().manual_seed(4)
    batch = {k: torch.randint(1, V, (B, L), generator=g).cuda() for k in ("seq", "pos", "neg")}

    def step():
        for p in model.parameters():
            p.grad = None
        pos, neg = model(batch)
        (pos.sum() - neg.sum()).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() if p.grad is not None else None for p in model.parameters()]

    old = (ops.config.dw_beside_lookup, ops.config.fuse_sublayers)
    try:
        ops.config.fuse_sublayers = False                               # the layer-by-layer path: _mha -> ops.linear(w[:E]) ...
        ops.config.dw_beside_lookup = False
        want = step()
        ops.config.dw_beside_lookup = True
        for _ in range(3):
            for a, w in zip(step(), want):
                assert (a is None and w is None) or torch.equal(a, w)
    finally:
        ops.config.dw_beside_lookup, ops.config.fuse_sublayers = old


@pytest.mark.parametrize("M,affine", [(8192, False), (12345, False), (8192, True)])
def test_tower_with_batchnorms_vs_float64(M, affine):
    """Linear -> BatchNorm1d -> ReLU -> Linear -> BatchNorm1d -> ReLU -> Linear(.., 1) in training mode (rechub's MLP,
    basic/layers.py:250-266) on the HIP path (split-operand GEMMs, rbx_batchnorm_fwd/bwd with the ReLU fused): output, running
    statistics and every gradient against the same tower in torch float64.  With ``affine`` the BatchNorms carry gammas of
    either sign (some exactly zero) and random betas; an activation that is zero to within float rounding may then fall on
    the other side of the ReLU than in float64, which moves one row of a gradient by a visible amount, so that case is held
    to a relative error of the whole tensor instead of the largest element's."""
    from recbox_amd import dense
    torch.manual_seed(M)
    K = 300
    mods = torch.nn.Sequential(torch.nn.Linear(K, 256), torch.nn.BatchNorm1d(256), torch.nn.ReLU(),
                               torch.nn.Linear(256, 384), torch.nn.BatchNorm1d(384), torch.nn.ReLU(),
                               torch.nn.Linear(384, 1))
    with torch.no_grad():                       # affine terms of either sign, and columns whose gamma is exactly zero
        for bn in ((mods[1], mods[4]) if affine else ()):
            bn.weight.copy_(torch.randn_like(bn.weight))
            bn.bias.copy_(0.5 * torch.randn_like(bn.bias))
            bn.weight[::37] = 0.0
            bn.weight[-1] = 0.0
    ref = __import__("copy").deepcopy(mods).double()
    x = torch.randn(M, K)
    r = torch.randn(M, 1)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * r.double()).sum().backward()
    m1 = __import__("copy").deepcopy(mods).cuda().train()
    x1 = x.cuda().requires_grad_(True)
    y1 = dense.run_sequential(m1, x1)
    (y1 * r.cuda()).sum().backward()

    def same(got, want, what):
        want = want.float()
        if affine:
            err = float((got.cpu() - want).norm()) / max(float(want.norm()), 1e-6)
            assert err < 2e-3 or float((got.cpu() - want).abs().max()) < 1e-4, "%s: relative error %.2e" % (what, err)
        else:
            assert_close(got, want, 2e-4 * max(1.0, float(want.abs().max())), what)

    for (n, p), (_, p1) in zip(ref.named_parameters(), m1.named_parameters()):
        same(p1.grad, p.grad, n)
    same(x1.grad, xr.grad, "dx")
    same(y1, yr, "y")
    for (n, b), (_, b1) in zip(ref.named_buffers(), m1.named_buffers()):
        if b.dtype.is_floating_point:
            assert_close(b1, b.float(), 1e-5 * max(1.0, float(b.abs().max())), "running " + n)


def test_sdpa_and_losses_golden():
    import recbox_amd.ranking.pytorch.layers as L
    fx = Fixture("attention_losses")
    t = _cuda(fx.tensors("in"))
    q, k, v = (t[n].clone().requires_grad_(True) for n in ("Q", "K", "V"))
    o, a = L.ScaledDotProductAttention(0.0)(q, k, v, scale=8 ** 0.5, mask=t["mask"])
    assert_close(o, fx["out"]["attn_out"], TOL)
    assert_close(a, fx["out"]["attn"], TOL)
    (o * t["R"]).sum().backward()
    for n, gr in (("Q", q.grad), ("K", k.grad), ("V", v.grad)):
        assert_close(gr, fx["g"][n], TOL, n)


@pytest.mark.parametrize("D,n_sets,width", [(64, 2, 1), (16, 1, 5), (128, 2, 3), (20, 2, 1), (6, 1, 2)])
def test_gather_dot_vs_torch(D, n_sets, width):
    """K7 rbx_gatherdot fwd/bwd == (x.unsqueeze(1) * embedding(ids)).sum(-1) in fp32 torch (the reference's
    sasrec.py:98-105 op sequence): logits, dx and the dense table gradient, shared and separate tables,
    duplicate ids (small vocab), a padding row that must get no gradient."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(D * 10 + n_sets)
    R, V = 777, 50
    x = torch.randn(R, D, generator=g)
    tabs = [torch.randn(V, D, generator=g) * 0.5 for _ in range(n_sets if D != 64 else 1)]
    sets = [torch.randint(0, V, (R,) if width == 1 else (R, width), generator=g) for _ in range(n_sets)]
    gout = torch.randn(R, n_sets * width, generator=g)
    pad = 0 if D == 16 else None
    # fp32 torch reference on the CPU
    xr = x.clone().requires_grad_(True)
    tr = [t.clone().requires_grad_(True) for t in tabs]
    cols = []
    for s, ids in enumerate(sets):
        e = torch.nn.functional.embedding(ids.reshape(R, -1), tr[s % len(tr)], padding_idx=pad)      # [R, w, D]
        cols.append((xr.unsqueeze(1) * e).sum(-1) * 0.25)
    ref = torch.cat(cols, dim=1)
    (ref * gout).sum().backward()
    xc = x.cuda().requires_grad_(True)
    tc = [t.cuda().requires_grad_(True) for t in tabs]
    out = ops.gather_dot(xc, [i.cuda() for i in sets], tc[0] if len(tc) == 1 else tc, scale=0.25, padding_idx=pad)
    assert_close(out, ref.detach(), 1e-5, "logits")
    (out * gout.cuda()).sum().backward()
    assert_close(xc.grad, xr.grad, 1e-4, "dx")
    for a, b in zip(tc, tr):
        assert_close(a.grad, b.grad, 1e-4, "dW")


@pytest.mark.parametrize("rows,cols,relu", [(64, 32, False), (1000, 400, True), (513, 37, True), (2, 8, False),
                                            (70001, 400, False), (300011, 37, False), (40000, 1030, False),
                                            (9000, 400, True)])
def test_batch_norm_vs_torch(rows, cols, relu):
    """rbx_batchnorm_fwd/bwd == nn.BatchNorm1d (+ReLU) of torch CPU fp32: outputs, input / affine gradients, running
    statistics and num_batches_tracked in training mode over two steps, then eval mode.  (The large shapes sweep the
    element-wise kernels several times per lane; they run without ReLU because among 10^7 pre-activations one lands within
    an ulp of zero and its mask legitimately differs between two fp32 implementations.)"""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    ref = torch.nn.BatchNorm1d(cols)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(cols, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(cols, generator=g) * 0.1)
    dut = torch.nn.BatchNorm1d(cols)
    dut.load_state_dict(ref.state_dict())
    dut.cuda()
    for step in range(2):
        x = torch.randn(rows, cols, generator=g) * 2.0 + 3.0          # mean far from 0: E[x^2]-E[x]^2 would lose digits
        r = torch.randn(rows, cols, generator=g)
        xr = x.clone().requires_grad_(True)
        zr = ref(xr)
        yr = torch.relu(zr) if relu else zr
        (yr * r).sum().backward()
        xc = x.cuda().requires_grad_(True)
        yc = ops.batch_norm(xc, dut, relu=relu)
        (yc * r.cuda()).sum().backward()
        assert_close(yc, yr.detach(), 1e-5, "y")
        if relu:
            # a pre-activation within a few ulp of zero may fall on either side of the ReLU in two fp32 implementations
            # (the CPU reference's reduction order depends on its thread count): the columns that hold such an element are
            # left out of the gradient comparison
            edge = zr.detach().abs() < 1e-5          # (a few dozen of 3.6 M standard-normal values)
            keep_cols = ~edge.any(dim=0)             # (a flipped element moves its whole column's dx through the batch means)
            assert_close(xc.grad.cpu()[:, keep_cols], xr.grad[:, keep_cols], TOL, "dx")
        else:
            assert_close(xc.grad, xr.grad, TOL, "dx")
        cols_ok = ~(zr.detach().abs() < 1e-5).any(dim=0) if relu else torch.ones(cols, dtype=torch.bool)
        assert_close(dut.weight.grad.cpu()[cols_ok], ref.weight.grad[cols_ok], 1e-4 * max(1.0, rows ** 0.5 / 4), "dgamma")
        assert_close(dut.bias.grad.cpu()[cols_ok], ref.bias.grad[cols_ok], 1e-4 * max(1.0, rows ** 0.5 / 4), "dbeta")
        assert_close(dut.running_mean, ref.running_mean, 1e-6, "running_mean")
        assert_close(dut.running_var, ref.running_var, 1e-5, "running_var")
        assert int(dut.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
        ref.zero_grad()
        dut.zero_grad()
    ref.eval()
    dut.eval()
    x = torch.randn(rows, cols, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.sum().backward()
    xc = x.cuda().requires_grad_(True)
    yc = ops.batch_norm(xc, dut)
    yc.sum().backward()
    assert_close(yc, yr.detach(), 1e-5, "eval y")
    assert_close(xc.grad, xr.grad, 1e-5, "eval dx")
    assert int(dut.num_batches_tracked) == 2


@pytest.mark.parametrize("shape", [(7, 9, 64), (300, 16), (5, 3, 200), (33, 7), (2, 1024)])
def test_layer_norm_vs_torch(shape):
    """rbx_layernorm_fwd/bwd == nn.LayerNorm(D, eps=1e-8) of torch CPU fp32 (output, dx, dgamma, dbeta)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    D = shape[-1]
    ref = torch.nn.LayerNorm(D, eps=1e-8)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(D, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(D, generator=g) * 0.1)
    dut = torch.nn.LayerNorm(D, eps=1e-8)
    dut.load_state_dict(ref.state_dict())
    dut.cuda()
    x = torch.randn(*shape, generator=g) * 1.5 + 2.0
    r = torch.randn(*shape, generator=g)
    xr = x.clone().requires_grad_(True)
    (ref(xr) * r).sum().backward()
    xc = x.cuda().requires_grad_(True)
    yc = ops.layer_norm(xc, dut)
    (yc * r.cuda()).sum().backward()
    assert_close(yc, ref(x).detach(), 2e-5, "y")
    assert_close(xc.grad, xr.grad, 1e-4, "dx")
    rows = x.numel() // D
    assert_close(dut.weight.grad, ref.weight.grad, 1e-4 * max(1.0, rows ** 0.5 / 4), "dgamma")
    assert_close(dut.bias.grad, ref.bias.grad, 1e-4 * max(1.0, rows ** 0.5 / 4), "dbeta")


def _dssm_feats(Fe, D=16):
    Sp, Sq = Fe.SparseFeature, Fe.SequenceFeature
    uf = [Sp("user_id", 61, D), Sp("gender", 3, D),
          Sq("hist_movie_id", 38, D, pooling="mean", shared_with="movie_id", padding_idx=0)]
    itf = [Sp("movie_id", 38, D), Sp("cate_id", 7, D)]
    return uf, itf


def test_dssm_golden():
    """BASELINE.json configs[0] (DSSM two-tower, MovieLens-shaped ids) in miniature."""
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import DSSM
    fx = Fixture("rechub_dssm")
    uf, itf = _dssm_feats(Fe)
    model = DSSM(uf, itf, {"dims": [32, 16], "activation": "prelu"}, {"dims": [32, 16], "activation": "prelu"},
                 temperature=0.02)
    load_params(model, fx["p"]).cuda().train()
    X = _cuda(fx.tensors("in"))
    p = model(X)
    assert_close(p, fx["out"]["y"], TOL)
    loss = F.binary_cross_entropy(p, X["label"])
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)
    model.mode = "user"
    assert tuple(model(X).shape) == (64, 16)


def test_youtubednn_golden():
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import YoutubeDNN
    Sp, Sq = Fe.SparseFeature, Fe.SequenceFeature
    D = 16
    uf = [Sp("user_id", 61, D), Sq("hist_movie_id", 38, D, pooling="mean", shared_with="movie_id", padding_idx=0)]
    itf = [Sp("movie_id", 38, D)]
    ngf = [Sq("neg_items", 38, D, pooling="concat", shared_with="movie_id")]
    fx = Fixture("rechub_youtubednn")
    model = load_params(YoutubeDNN(uf, itf, ngf, {"dims": [32, 16]}, temperature=0.02), fx["p"]).cuda().train()
    X = _cuda(fx.tensors("in"))
    y = model(X)
    assert tuple(y.shape) == (64, 4)
    assert_close(y, fx["out"]["y"], TOL, "logits")   # cosine / 0.02 (50x amplification): 1.9e-5 achieved (parity ledger)
    loss = F.cross_entropy(y, torch.zeros(64, dtype=torch.long, device="cuda"))
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)          # 5.7e-6 achieved


def test_deepfm_golden():
    Fe, La = _rh()
    from recbox_amd.rechub.models.ranking import DeepFM
    dense = [Fe.DenseFeature("I%d" % i) for i in range(1, 4)]
    sparse = [Fe.SparseFeature("C%d" % (i + 1), v + 1, 16) for i, v in enumerate(CRITEO_SMALL_VOCABS[:8])]
    fx = Fixture("rechub_deepfm")
    model = DeepFM(sparse + dense, sparse, {"dims": [32, 16], "dropout": 0.0, "activation": "relu"})
    load_params(model, fx["p"]).cuda().train()
    X = _cuda(fx.tensors("in"))
    p = model(X)
    assert_close(p, fx["out"]["y"], TOL)
    loss = F.binary_cross_entropy(p, X["label"])
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_sasrec_golden():
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import SASRec
    Sq = Fe.SequenceFeature
    fe = [Sq("seq", 31, 8, pooling="concat"), Sq("pos", 31, 8, pooling="concat", shared_with="seq"),
          Sq("neg", 31, 8, pooling="concat", shared_with="seq")]
    fx = Fixture("rechub_sasrec")
    model = load_params(SASRec(fe, max_len=12, dropout_rate=0.0, num_blocks=2, num_heads=1), fx["p"]).cuda().train()
    X = _cuda(fx.tensors("in"))
    pl, nl = model(X)
    assert_close(pl, fx["out"]["pos_logits"], TOL)
    assert_close(nl, fx["out"]["neg_logits"], TOL)
    m = (X["pos"] != 0).float()
    loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_sasrec_d64_golden_runs_the_mfma_attention():
    """Live-reference fixture at cfg 5's shape (D = 64, one head => head_dim 64, L = 200): the attention of this model
    is the MFMA kernel (rbx_attn_mfma.hip), the small fixture above (head_dim 8) only reaches the VALU one."""
    Fe, La = _rh()
    from recbox_amd import ops
    from recbox_amd.rechub.models.matching import SASRec
    Sq = Fe.SequenceFeature
    fe = [Sq("seq", 97, 64, pooling="concat"), Sq("pos", 97, 64, pooling="concat", shared_with="seq"),
          Sq("neg", 97, 64, pooling="concat", shared_with="seq")]
    fx = Fixture("rechub_sasrec_d64")
    model = load_params(SASRec(fe, max_len=200, dropout_rate=0.0, num_blocks=2, num_heads=1), fx["p"]).cuda().train()
    X = _cuda(fx.tensors("in"))
    seen = []
    timer = ops.KernelTimer(lambda m: seen.append(m) or False)
    ops.kernel_timer = timer
    try:
        pl, nl = model(X)
        m = (X["pos"] != 0).float()
        loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
        loss.backward()
    finally:
        ops.kernel_timer = None
    assert ("attn_bwd", 2, 200, 64) in seen                    # head_dim 64, L = 200: the MFMA path's shape
    assert_close(pl, fx["out"]["pos_logits"], TOL)
    assert_close(nl, fx["out"]["neg_logits"], TOL)
    assert_close(loss, fx["out"]["loss"], TOL)
    assert_grads_close(model, fx["g"], TOL)


@pytest.mark.parametrize("L,D,causal", [(200, 64, True), (37, 32, True), (256, 64, False), (64, 32, False),
                                        (1, 64, True), (200, 16, True)])
def test_attention_matches_torch(L, D, causal):
    """The attention core (MFMA path for d in {32, 64}, VALU path otherwise) against torch fp64 on CPU;
    (200, 64, causal) is the cfg-5 shape."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H = 3, 2
    q, k, v = (torch.randn(B, H, L, D, generator=g) for _ in range(3))
    R = torch.randn(B, H, L, D, generator=g)
    qr, kr, vr = (t.clone().double().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-1, -2)) * D ** -0.5
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    o0 = s.softmax(-1) @ vr
    (o0 * R.double()).sum().backward()
    qc, kc, vc = (t.clone().cuda().requires_grad_(True) for t in (q, k, v))
    o1, _ = ops.attention(qc, kc, vc, scale=D ** -0.5, causal=causal, fill=float("-inf"))
    (o1 * R.cuda()).sum().backward()
    assert_close(o1, o0.float(), TOL)
    for a, b, n in ((qc.grad, qr.grad, "dq"), (kc.grad, kr.grad, "dk"), (vc.grad, vr.grad, "dv")):
        assert_close(a, b.float(), TOL, n)


@pytest.mark.parametrize("Lq,Lk,D,causal,use_mask", [(512, 512, 64, True, False), (1000, 1000, 64, True, False),
                                                      (300, 700, 32, False, True), (513, 513, 16, True, True),
                                                      (257, 257, 64, False, False), (200, 200, 64, True, True)])
def test_attention_of_any_length_vs_float64(Lq, Lk, D, causal, use_mask):
    """Round 5: the VALU kernels stream K / V (the backward: Q / dO) through LDS in chunks, so any sequence length runs --
    nn.MultiheadAttention has no limit (sasrec.py:81-94; VERDICT r4 missing #5).  L = 512 / 1000 (several chunks, several query
    blocks per sequence), rectangular masked attention with returned probabilities, and the cfg-5 shape WITH an explicit mask
    (which keeps it off the matrix-core path: two chunks of 184 keys) against torch float64: output, probabilities, dq, dk, dv;
    and dropout evaluates the same mask in the backward as in the forward at these lengths."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(Lq + Lk)
    BH = 3
    q = torch.randn(BH, 1, Lq, D, generator=g)
    k, v = (torch.randn(BH, 1, Lk, D, generator=g) for _ in range(2))
    R = torch.randn(BH, 1, Lq, D, generator=g)
    mask = (torch.rand(BH, 1, Lq, Lk, generator=g) < 0.8).float() if use_mask else None
    if mask is not None:
        mask[..., 0] = 1.0                                        # every row keeps its first key (also under the causal mask)
    qr, kr, vr = (t.clone().double().requires_grad_(True) for t in (q, k, v))
    sc = (qr @ kr.transpose(-1, -2)) * D ** -0.5
    if mask is not None:
        sc = sc.masked_fill(mask == 0, -1.0e9)
    if causal:
        sc = sc.masked_fill(~torch.tril(torch.ones(Lq, Lk, dtype=torch.bool)), float("-inf"))
    p0 = sc.softmax(-1)
    o0 = p0 @ vr
    (o0 * R.double()).sum().backward()
    qc, kc, vc = (t.clone().cuda().requires_grad_(True) for t in (q, k, v))
    o1, p1 = ops.attention(qc, kc, vc, mask=None if mask is None else mask.cuda(), scale=D ** -0.5, causal=causal,
                           fill=-1.0e9 if mask is not None else float("-inf"), need_probs=use_mask)
    (o1 * R.cuda()).sum().backward()
    assert_close(o1, o0.float(), TOL, "o")
    if use_mask:
        assert_close(p1, p0.float(), 1e-5, "probabilities")
    for a, b, n in ((qc.grad, qr.grad, "dq"), (kc.grad, kr.grad, "dk"), (vc.grad, vr.grad, "dv")):
        assert_close(a, b.float(), TOL * max(1.0, float(b.abs().max())), n)
    # dropout: dv = P_dropped^T dO with the forward's mask -- checked through the linearity of o in v
    qd, kd, vd = (t.clone().cuda().requires_grad_(True) for t in (q, k, v))
    od, _ = ops.attention(qd, kd, vd, mask=None if mask is None else mask.cuda(), scale=D ** -0.5, causal=causal,
                          fill=-1.0e9 if mask is not None else float("-inf"), dropout_p=0.3, seed=11)
    (od * R.cuda()).sum().backward()
    assert_close((vd.grad * v.cuda()).sum(), (od.detach() * R.cuda()).sum(), 2e-3 * float(od.detach().abs().sum()) / od.numel() * 100 + 1e-2,
                 "<dv, v> == <o, R> under dropout")


def test_sasrec_longer_than_the_matrix_core_kernels_take():
    """SASRec with max_len 300 (> 256: no packed / MFMA attention) against the oracle's restatement: logits and gradients."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd.rechub.models.matching import SASRec
    V, D, L, B = 5000, 32, 300, 6
    feats = bench._sasrec_features(V, D)
    with torch.device("cuda"):
        model = SASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=2)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.1)
    model.train()
    x = bench._sasrec_batch(B, V, L, 5, "cuda")
    ref = R.RefSASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=2).train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    xc = {k: v.cpu() for k, v in x.items()}
    pl, nl = model(x)
    pl0, nl0 = ref(xc)
    assert_close(pl, pl0, TOL, "pos logits")
    assert_close(nl, nl0, TOL, "neg logits")
    (pl.sum() - nl.sum()).backward()
    (pl0.sum() - nl0.sum()).backward()
    wantp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        want = wantp[n].grad
        assert_close(p.grad, want, 1e-4 * max(1.0, float(want.abs().max())), "grad " + n)


def test_multi_head_target_attention_golden():
    import recbox_amd.ranking.pytorch.layers as L
    fx = Fixture("target_attention_losses")
    att = load_params(L.MultiHeadTargetAttention(input_dim=16, attention_dim=16, num_heads=2), fx["p"]).cuda()
    t = _cuda(fx.tensors("in"))
    tgt = t["target"].clone().requires_grad_(True)
    hist = t["history"].clone().requires_grad_(True)
    out = att(tgt, hist, t["mask"])
    assert_close(out, fx["out"]["attn"], TOL)
    (out * t["R"]).sum().backward()
    assert_close(tgt.grad, fx["g"]["target"], TOL)
    assert_close(hist.grad, fx["g"]["history"], TOL)
    for n, p in att.named_parameters():
        assert_close(p.grad, fx["g"]["p." + n], TOL, n)


def _matching_model_and_batches():
    """Two matching EmbeddingLayers (user side only: one lookup per step, so every table feeds exactly one op) and a few
    batches of different sizes: id + padded mean-pooled history + numeric feature."""
    import recbox_amd.core.pytorch.layers as C

    class UserTower(torch.nn.Module):
        def __init__(self):
            super(UserTower, self).__init__()
            self.emb = C.EmbeddingLayer(_MFM(matching_specs()), 8)
            self.head = torch.nn.Linear(3 * 8, 1)

        def forward(self, X):
            u = self.emb(X, feature_source="user")
            return self.head(u.flatten(start_dim=1))

    batches = []
    for k, B in enumerate([40, 40, 200, 9]):
        g = torch.Generator().manual_seed(300 + k)
        hist = torch.randint(0, 19, (B, 6), generator=g)
        hist[torch.arange(6)[None, :] >= torch.randint(0, 7, (B, 1), generator=g)] = 19        # pad tail (some rows empty)
        X = OrderedDict(user_id=torch.randint(0, 13, (B,), generator=g), user_hist=hist,
                        age=torch.rand(B, generator=g), item_id=torch.randint(0, 19, (B,), generator=g))
        batches.append((X, (torch.rand(B, 1, generator=g) < 0.4).float()))
    return UserTower().cuda(), UserTower().cuda(), batches, (lambda m, X: m(X))


@pytest.mark.parametrize("which", ["youtubednn", "deepfm"])
def test_model_mirrors_with_persistent_gradients(which):
    """ops.config.reuse_grad_buffers = "all" on the model mirrors whose tables each feed ONE lookup per training step
    (YoutubeDNN: user id + history + item + negatives in a single gather with a shared item table; DeepFM: one gather
    for the FM, LR and deep parts): three steps on different batches, bit-identical to fresh gradients."""
    from recbox_amd import ops
    from recbox_amd.rechub.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from recbox_amd.rechub.models.matching import YoutubeDNN
    from recbox_amd.rechub.models.ranking import DeepFM
    V, D = 5000, 16

    def build():
        if which == "youtubednn":
            uf = [SparseFeature("user_id", 300, 8),
                  SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)]
            return YoutubeDNN(uf, [SparseFeature("item", V, D)],
                              [SequenceFeature("neg_items", V, D, pooling="concat", shared_with="item")],
                              {"dims": [32, D], "activation": "relu"}, temperature=0.05).cuda()
        dense = [DenseFeature("I%d" % i) for i in range(3)]
        sparse = [SparseFeature("C%d" % i, v, D) for i, v in enumerate([7, 400, V])]
        return DeepFM(dense + sparse, sparse, {"dims": [24, 24], "dropout": 0.0, "activation": "relu"}).cuda()

    def batch(B, seed):
        g = torch.Generator().manual_seed(seed)
        if which == "youtubednn":
            L = 9
            lens = torch.randint(0, L + 1, (B,), generator=g)
            hist = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
            x = {"user_id": torch.randint(0, 300, (B,), generator=g), "hist": hist,
                 "item": torch.randint(1, V, (B,), generator=g), "neg_items": torch.randint(1, V, (B, 3), generator=g)}
            return {k: v.cuda() for k, v in x.items()}, torch.zeros(B, dtype=torch.long, device="cuda")
        x = {"I%d" % i: torch.rand(B, generator=g) for i in range(3)}
        for i, v in enumerate([7, 400, V]):
            x["C%d" % i] = torch.randint(0, v, (B,), generator=g)
        return {k: v.cuda() for k, v in x.items()}, (torch.rand(B, generator=g) < 0.3).float().cuda()

    fresh, reuse = build(), build()
    with torch.no_grad():
        for p in fresh.parameters():
            p.normal_(0, 0.1)
    reuse.load_state_dict(fresh.state_dict())
    loss_fn = F.cross_entropy if which == "youtubednn" else F.binary_cross_entropy
    old = ops.config.reuse_grad_buffers
    try:
        for k, B in enumerate([64, 64, 257]):
            x, y = batch(B, 500 + k)
            for model, flag in ((fresh, False), (reuse, "all")):
                ops.config.reuse_grad_buffers = flag
                model.train()
                model.zero_grad(set_to_none=True)
                loss_fn(model(x), y).backward()
            for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
                assert (p0.grad is None) == (p1.grad is None), n
                if p0.grad is not None:
                    assert torch.equal(p1.grad, p0.grad), "%s step %d: %s" % (which, k, n)
    finally:
        ops.config.reuse_grad_buffers = old


@pytest.mark.parametrize("shape,with_add,alpha", [((5, 9, 64), True, 8.0), ((4096, 7), False, 1.0), ((3, 1, 4), True, 1.0),
                                                  ((0, 16), False, 1.0)])
def test_row_scale_vs_torch(shape, with_add, alpha):
    """rbx_rowscale: (alpha x + add) * scale per row -- SASRec's embedding prologue and timeline mask in one pass --
    forward and both gradients against the three ATen expressions the reference writes."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    add = torch.randn(shape, generator=g) if with_add else None
    s = (torch.rand(shape[:-1], generator=g) < 0.6).float()
    R = torch.randn(shape, generator=g)
    xr = x.clone().requires_grad_(True)
    ar = add.clone().requires_grad_(True) if with_add else None
    ref = xr * alpha
    if with_add:
        ref = ref + ar
    ref = ref * s.unsqueeze(-1)
    (ref * R).sum().backward()
    xc = x.cuda().requires_grad_(True)
    ac = add.cuda().requires_grad_(True) if with_add else None
    out = ops.row_scale(xc, s.cuda(), add=ac, alpha=alpha)
    (out * R.cuda()).sum().backward()
    assert_close(out, ref, 1e-6, "out")
    assert_close(xc.grad, xr.grad, 1e-6, "dx")
    if with_add:
        assert_close(ac.grad, ar.grad, 1e-6, "dadd")


@pytest.mark.parametrize("rows,cols,n,copies", [(300, 1677, 1664, 2), (257, 40, 24, 2), (64, 37, 5, 2), (1, 8, 8, 1),
                                                (5000, 1680, 1664, 2)])
def test_shared_prefix_vs_plain_slicing(rows, cols, n, copies):
    """ops.shared_prefix (rbx_sum_prefix) == x, x[:, :n], x[:, :n] with autograd's own fan-out: same forward values,
    same input gradient, also when a reader is absent or hands back a broadcast gradient (DeepFM's shared block,
    deepfm.py:34-39)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g)
    w = [torch.randn(rows, cols, generator=g), torch.randn(rows, n, generator=g), torch.randn(rows, n, generator=g)]
    padded = ops._padded_rows(rows, cols, "cuda")
    padded.copy_(x)                                           # row stride a multiple of 4 floats, like the lookup's output
    for xin in (padded, x.cuda()):
        for used in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (0, 0, 1)):
            if copies == 1 and used[2]:
                continue
            xr = x.clone().requires_grad_(True)
            parts_r = (xr, xr[:, :n], xr[:, :n])
            xc = xin.detach().requires_grad_(True)
            parts_c = ops.shared_prefix(xc, n, copies=copies)
            for a, b in zip(parts_c, parts_r):
                assert torch.equal(a.detach().cpu(), b.detach())
            lr = sum((p * wt).sum() for p, wt, u in zip(parts_r, w, used) if u)
            lc = sum((p * wt.cuda()).sum() for p, wt, u in zip(parts_c, w, used) if u)
            lr.backward()
            lc.backward()
            assert_close(xc.grad, xr.grad, 1e-6, "dx %s" % (used,))
    xr = x.clone().requires_grad_(True)
    xc = x.cuda().requires_grad_(True)
    parts_c = ops.shared_prefix(xc, n, copies=copies)
    (xr.sum() * 2.0 + xr[:, :n].sum()).backward()             # broadcast (stride-0) gradients
    (parts_c[0].sum() * 2.0 + parts_c[1].sum()).backward()
    assert_close(xc.grad, xr.grad, 1e-6, "dx broadcast")


def test_youtubednn_with_dense_user_features_matches_oracle():
    """a squeezed gather puts DenseFeature values BEHIND every embedding of the call, so YoutubeDNN's
    single-gather training path may not be used when the user side has dense features (youtube_dnn.py:46-56 embeds
    ``user_features`` on their own).  Against the CPU oracle of the reference's op sequence: logits and every gradient."""
    from oracle import torch_ref as R
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import YoutubeDNN
    V, D, B, L, n_neg = 53, 8, 37, 6, 3

    def feats(Sp, Sq, De):
        uf = [Sp("user_id", 19, D), De("age"), Sq("hist", V, D, pooling="mean", shared_with="item", padding_idx=0), De("act")]
        return uf, [Sp("item", V, D)], [Sq("neg_items", V, D, pooling="concat", shared_with="item")]

    ref = R.RefYoutubeDNN(*feats(R.RefSparseFeature, R.RefSequenceFeature, R.RefDenseFeature), {"dims": [16, D]},
                          temperature=0.1)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    dut = YoutubeDNN(*feats(Fe.SparseFeature, Fe.SequenceFeature, Fe.DenseFeature), {"dims": [16, D]}, temperature=0.1)
    dut.load_state_dict(ref.state_dict())
    dut.cuda().train()
    ref.train()
    lens = torch.randint(0, L + 1, (B,), generator=g)
    x = {"user_id": torch.randint(0, 19, (B,), generator=g), "age": torch.rand(B, generator=g),
         "act": torch.randn(B, generator=g),
         "hist": torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None]),
         "item": torch.randint(1, V, (B,), generator=g), "neg_items": torch.randint(1, V, (B, n_neg), generator=g)}
    y0 = ref(x)
    F.cross_entropy(y0, torch.zeros(B, dtype=torch.long)).backward()
    y1 = dut(_cuda(x))
    assert tuple(y1.shape) == (B, 1 + n_neg)
    assert_close(y1, y0, TOL)
    F.cross_entropy(y1, torch.zeros(B, dtype=torch.long, device="cuda")).backward()
    want = dict(ref.named_parameters())
    for n, p in dut.named_parameters():
        assert_close(p.grad, want[n].grad, TOL, "grad " + n)


@pytest.mark.parametrize("L,D,causal,use_mask,p,BH", [(200, 64, True, False, 0.5, 6), (37, 32, True, False, 0.1, 6),
                                                      (64, 64, False, False, 0.3, 6), (9, 8, False, True, 0.5, 6),
                                                      (23, 16, True, False, 0.25, 6),
                                                      (200, 64, True, False, 0.5, 260)])   # (the forward that loops over sequences)
def test_attention_dropout_vs_restatement_with_the_kernels_mask(L, D, causal, use_mask, p, BH):
    """Dropout on the attention probabilities inside the fused kernels (nn.MultiheadAttention(dropout) of rechub SASRec,
    sasrec.py:29,56; `attention = self.dropout(attention)` of dot_product_attention.py:40-41): out = (keep o P / (1 - p)) V.
    The kernels' keep mask (a counter-based function, re-evaluated by the backward) is read back through
    rbx_attn_dropout_mask and injected into a torch fp64 restatement: output, returned probabilities and dQ / dK / dV."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(L + D)
    q, k, v = (torch.randn(BH, L, D, generator=g) for _ in range(3))
    R = torch.randn(BH, L, D, generator=g)
    mask = (torch.rand(BH, L, L, generator=g) > 0.2).float() if use_mask else None
    if mask is not None:
        mask[:, :, 0] = 1.0
    seed = 1234567 + L
    keep = ops.attention_dropout_mask(BH, L, L, p, seed)
    p_eff = round(p * 65536) / 65536.0
    sigma = (p_eff * (1 - p_eff) / keep.numel()) ** 0.5
    assert abs(float(keep.float().mean()) - (1 - p_eff)) < 4 * sigma + 1e-3   # the rate is what was asked for
    assert not torch.equal(keep, ops.attention_dropout_mask(BH, L, L, p, seed + 1))
    assert torch.equal(keep, ops.attention_dropout_mask(BH, L, L, p, seed))
    qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
    s = (qd @ kd.transpose(1, 2)) * (D ** -0.5)
    if mask is not None:
        s = s.masked_fill(mask.double() == 0, -1.0e9)
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    P = torch.softmax(s, dim=-1)
    Pd = P * keep.cpu().double() / (1 - p_eff)
    want = Pd @ vd
    (want * R.double()).sum().backward()
    qc, kc, vc = (t.cuda().requires_grad_() for t in (q, k, v))
    out, probs = ops.attention(qc, kc, vc, mask=mask.cuda() if mask is not None else None, scale=D ** -0.5, causal=causal,
                               fill=-1.0e9, need_probs=use_mask, dropout_p=p, seed=seed)
    assert_close(out, want, TOL, "output")
    if use_mask:
        assert_close(probs, Pd, TOL, "returned (dropped) probabilities")
    (out * R.cuda()).sum().backward()
    assert_close(qc.grad, qd.grad, TOL, "dQ")
    assert_close(kc.grad, kd.grad, TOL, "dK")
    assert_close(vc.grad, vd.grad, TOL, "dV")


@pytest.mark.parametrize("L,BH", [(65, 7), (96, 301), (129, 6), (200, 7), (224, 5)])
def test_streamed_attention_forward_pairs_of_sequences_vs_float64(L, BH):
    """Causal sequences of 3..7 key tiles at head_dim 64 take the forward kernel that streams K and V through a ring of
    32-key tiles (rbx_attn_stream.h): a workgroup serves a PAIR of sequences -- the even one walks its key tiles upwards, the
    odd one downwards -- and an odd count leaves the last workgroup with one sequence.  Output, log-sum-exp (through the
    backward kernels, which rebuild P from it) and gradients against float64, for the first, the last and a middle sequence."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(7 * L + BH)
    D = 64
    q, k, v = (torch.randn(BH, 1, L, D, generator=g) for _ in range(3))
    R = torch.randn(BH, 1, L, D, generator=g)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o, _ = ops.attention(qc, kc, vc, scale=D ** -0.5, causal=True, fill=float("-inf"))
    (o * R.cuda()).sum().backward()
    tri = torch.tril(torch.ones(L, L, dtype=torch.bool))
    for b in sorted({0, 1, BH // 2, BH - 2, BH - 1}):
        qd, kd, vd = (t[b, 0].double().requires_grad_(True) for t in (q, k, v))
        s = ((qd @ kd.t()) * D ** -0.5).masked_fill(~tri, float("-inf"))
        want = s.softmax(-1) @ vd
        (want * R[b, 0].double()).sum().backward()
        assert_close(o[b, 0], want.float(), TOL, "out of sequence %d" % b)
        assert_close(qc.grad[b, 0], qd.grad.float(), TOL, "dq of sequence %d" % b)
        assert_close(kc.grad[b, 0], kd.grad.float(), TOL, "dk of sequence %d" % b)
        assert_close(vc.grad[b, 0], vd.grad.float(), TOL, "dv of sequence %d" % b)
    # a sequence's result does not depend on its neighbour: the same rows as the even and as the odd member of a pair
    q2 = torch.cat([q[:1], q[:1], q[1:2]]).cuda()
    k2 = torch.cat([k[:1], k[:1], k[1:2]]).cuda()
    v2 = torch.cat([v[:1], v[:1], v[1:2]]).cuda()
    o2, _ = ops.attention(q2, k2, v2, scale=D ** -0.5, causal=True, fill=float("-inf"))
    assert_close(o2[1], o2[0], 2e-6, "the same sequence walked upwards and downwards")
    assert torch.equal(o2[0], o.detach()[0])


@pytest.mark.parametrize("L", [64, 160, 200, 224, 256])
def test_every_form_of_the_causal_forward_vs_float64(L):
    """head_dim 64, causal: L <= 64 runs the resident kernel, 3-6 key tiles (L = 160) the streamed f32 ring
    (rbx_attn_stream.h), 7 tiles (L = 200, 224: BASELINE cfg 5) the bf16-plane ring (rbx_attn_planes.h), L = 256 the looping
    resident kernel: each against softmax(Q K^T / 8 + causal) V in float64, and its dropout variant against the same product
    with the mask the no-dropout probabilities imply (kept entries scaled by 1 / (1 - p), rows still sum to the kept mass)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(L)
    q, k, v = (torch.randn(9, 1, L, 64, generator=g) for _ in range(3))
    o, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), scale=0.125, causal=True, fill=float("-inf"))
    s = (q.double() @ k.double().transpose(-1, -2)) * 0.125
    s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    want = torch.softmax(s, dim=-1) @ v.double()
    assert_close(o, want.float(), 2e-6 * max(1.0, float(want.abs().max())), "forward, L = %d" % L)
    od, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), scale=0.125, causal=True, fill=float("-inf"), dropout_p=0.25, seed=5)
    od2, _ = ops.attention(q.cuda(), k.cuda(), v.cuda(), scale=0.125, causal=True, fill=float("-inf"), dropout_p=0.25, seed=5)
    assert torch.equal(od, od2) and not torch.equal(od, o)                 # a function of the seed; and it drops something
    # E[dropout output] = the plain output: the mean over the 9 x L rows is close, row by row it is not
    assert float((od - o).abs().mean()) < 0.5 * float(o.abs().mean()) + 0.2


@pytest.mark.parametrize("L", [192, 200, 256])
def test_attention_forward_that_loops_over_sequences_equals_one_workgroup_per_sequence(L):
    """More than 256 sequences of L > 160 at head_dim 64 take the forward kernel that keeps one workgroup per CU, loops over
    the sequences and prefetches the next K, V into registers (6, 7 or 8 float4 per thread at L = 192, 200, 256); the same
    sequences in two launches of at most 256 take the one-workgroup-per-sequence kernel.  Same arithmetic in the same order:
    bit-identical outputs and gradients.  (Since round 4 the causal L = 192 and 200 cases run the streamed kernel of
    rbx_attn_stream.h whatever the count -- the split below keeps every sequence's place in its pair --, L = 256 the two
    resident forms.)"""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(L)
    BH, D = 300, 64
    q, k, v = (torch.randn(BH, 1, L, D, generator=g).cuda() for _ in range(3))
    R = torch.randn(BH, 1, L, D, generator=g).cuda()

    def run(lo, hi):
        qq, kk, vv = (t[lo:hi].clone().requires_grad_(True) for t in (q, k, v))
        o, _ = ops.attention(qq, kk, vv, scale=D ** -0.5, causal=True, fill=float("-inf"))
        (o * R[lo:hi]).sum().backward()
        return o.detach(), qq.grad, kk.grad, vv.grad

    whole = run(0, BH)
    parts = [run(0, 150), run(150, BH)]
    for i, name in enumerate(("out", "dq", "dk", "dv")):
        assert torch.equal(whole[i], torch.cat([p[i] for p in parts])), name
    # and it is right: one sequence against float64
    qd, kd, vd = (t[BH - 1, 0].double().cpu() for t in (q, k, v))
    s = (qd @ kd.t()) * D ** -0.5
    s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    assert_close(whole[0][BH - 1, 0], (s.softmax(-1) @ vd).float(), TOL, "last sequence")


def test_sasrec_trains_with_the_reference_default_dropout():
    """rechub's SASRec defaults to dropout_rate = 0.5 (sasrec.py:29): embedding dropout, attention dropout (inside the
    fused kernel) and the feed-forward dropouts all active.  The step runs, is finite, differs from the dropout-free
    output, reproduces under the same torch seed, and .eval() switches every dropout off."""
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import SASRec
    Sq = Fe.SequenceFeature
    fe = [Sq("seq", 97, 64, pooling="concat"), Sq("pos", 97, 64, pooling="concat", shared_with="seq"),
          Sq("neg", 97, 64, pooling="concat", shared_with="seq")]
    fx = Fixture("rechub_sasrec_d64")
    model = load_params(SASRec(fe, max_len=200), fx["p"]).cuda()
    assert model.attention_layers[0].dropout == 0.5
    X = _cuda(fx.tensors("in"))

    def run():
        for p in model.parameters():
            p.grad = None
        pl, nl = model(X)
        m = (X["pos"] != 0).float()
        loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
        loss.backward()
        return pl.detach().clone(), model.attention_layers[0].in_proj_weight.grad.clone()

    model.train()
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    a, ga = run()
    torch.manual_seed(5)
    torch.cuda.manual_seed(5)
    b, gb = run()
    assert torch.isfinite(a).all() and torch.isfinite(ga).all()
    assert torch.equal(a, b) and torch.equal(ga, gb)
    c, _ = run()
    assert not torch.equal(a, c)                                            # another draw
    model.eval()
    e, _ = run()
    assert_close(e, fx["out"]["pos_logits"], TOL)                           # dropout off == the dropout-free fixture


def test_table_read_by_two_lookups_gets_one_gradient_tensor():
    """SASRec's item table feeds the embedding layer (item sequence) and gather_dot (candidates).  With
    ops.config.share_table_grads the second backward node adds its rows into the dense gradient the first one returned:
    the same gradient as autograd's sum of two dense tensors (within rounding: another order of the same sums), identical
    from run to run, and the parameter's gradient IS the tensor the first node made (no aten::add of two 256 MB tensors)."""
    from recbox_amd import ops
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import SASRec
    Sq = Fe.SequenceFeature
    fx = Fixture("rechub_sasrec_d64")
    X = _cuda(fx.tensors("in"))

    def run(share):
        fe = [Sq("seq", 97, 64, pooling="concat"), Sq("pos", 97, 64, pooling="concat", shared_with="seq"),
              Sq("neg", 97, 64, pooling="concat", shared_with="seq")]
        model = load_params(SASRec(fe, max_len=200, dropout_rate=0.0), fx["p"]).cuda().train()
        old = ops.config.share_table_grads
        ops.config.share_table_grads = share
        made = []
        real = ops._publish_grads

        def spy(ctx, params, grads):
            made.extend(g.data_ptr() for g in grads if g is not None)
            return real(ctx, params, grads)
        ops._publish_grads = spy
        try:
            pl, nl = model(X)
            m = (X["pos"] != 0).float()
            loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
            loss.backward()
        finally:
            ops.config.share_table_grads = old
            ops._publish_grads = real
        table = model.item_emb.embed_dict["seq"].weight
        return table.grad.clone(), table.grad.data_ptr() in made, {n: p.grad.clone() for n, p in model.named_parameters()}

    g_sum, _, all_sum = run(False)
    g_one, same_tensor, all_one = run(True)
    g_two, _, _ = run(True)
    assert same_tensor                                   # AccumulateGrad received the first node's tensor alone
    assert torch.equal(g_one, g_two)
    assert_close(g_one, g_sum, 1e-6 * max(1.0, float(g_sum.abs().max())), "item table gradient")
    assert_close(g_one, fx["g"]["item_emb.embed_dict.seq.weight"], TOL, "vs the live-reference fixture")
    for n in all_sum:
        if "item_emb" not in n:
            assert torch.equal(all_sum[n], all_one[n]), n
    assert not ops._pending_grads                        # every published gradient was adopted (the position table, read
                                                         # once, was never published)


def test_sasrec_sublayers_as_one_node_equal_the_composed_ops():
    """ops.sasrec_attention_sublayer / sasrec_ffn_sublayer (residual adds, timeline mask, ReLU backward and the gradient sums
    of tensors with two readers folded into GEMM epilogues) against the same block composed from layer_norm / linear /
    attention_packed / row_scale: logits and every parameter gradient, in eval mode and with attention dropout under one seed."""
    from recbox_amd import ops
    Fe, La = _rh()
    from recbox_amd.rechub.models.matching import SASRec
    Sq = Fe.SequenceFeature
    fx = Fixture("rechub_sasrec_d64")
    X = _cuda(fx.tensors("in"))

    def run(fused, train):
        # (new feature objects per model: a rechub feature caches its nn.Embedding, two models built from the same
        #  features share the table -- and its .grad)
        fe = [Sq("seq", 97, 64, pooling="concat"), Sq("pos", 97, 64, pooling="concat", shared_with="seq"),
              Sq("neg", 97, 64, pooling="concat", shared_with="seq")]
        model = load_params(SASRec(fe, max_len=200, dropout_rate=0.0), fx["p"]).cuda()
        for m in model.attention_layers:
            m.dropout = 0.25 if train else 0.0                               # dropout on the attention probabilities only
        model.train(train)
        old = ops.config.fuse_sublayers
        ops.config.fuse_sublayers = fused
        try:
            torch.manual_seed(11)
            torch.cuda.manual_seed(11)
            pl, nl = model(X)
            m = (X["pos"] != 0).float()
            loss = -((F.logsigmoid(pl) + F.logsigmoid(-nl)) * m).sum() / m.sum()
            loss.backward()
        finally:
            ops.config.fuse_sublayers = old
        return pl.detach(), nl.detach(), dict((n, p.grad.clone()) for n, p in model.named_parameters())

    for train in (False, True):
        pa, na, ga = run(True, train)
        pb, nb, gb = run(False, train)
        assert_close(pa, pb, 1e-5, "pos logits (train=%s)" % train)
        assert_close(na, nb, 1e-5, "neg logits (train=%s)" % train)
        bad = ["%s %.2e (|ref| %.2e)" % (n, (ga[n] - gb[n]).abs().max().item(), gb[n].abs().max().item()) for n in gb
               if (ga[n] - gb[n]).abs().max().item() > 1e-5 + 1e-5 * gb[n].abs().max().item()]
        assert not bad, "train=%s: %s" % (train, "; ".join(bad))
        if not train:
            assert_close(pa, fx["out"]["pos_logits"], TOL)                  # and == the live-reference fixture
            for n in ga:
                assert_close(ga[n], fx["g"][n], TOL, "grad " + n)


@pytest.mark.parametrize("B,L,D", [(300, 200, 64), (37, 50, 64), (5, 7, 6), (1, 3, 4)])
def test_sasrec_input_reads_position_rows_in_place(B, L, D):
    """ops.sasrec_input == (alpha e + position_emb(tile(arange(L)))) * keep (sasrec.py:68-77) and its two gradients, against
    torch float64; the position rows' gradient is a deterministic column sum over the batch (bit-identical when repeated)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + L)
    e = torch.randn(B, L, D, generator=g)
    table = torch.randn(L + 3, D, generator=g)
    keep = (torch.rand(B, L, generator=g) > 0.25).float()
    R = torch.randn(B, L, D, generator=g)
    alpha = D ** 0.5
    ed, td = e.double().requires_grad_(True), table.double().requires_grad_(True)
    pos = torch.arange(L).unsqueeze(0).expand(B, -1)
    want = (ed * alpha + torch.nn.functional.embedding(pos, td)) * keep.double().unsqueeze(-1)
    (want * R.double()).sum().backward()

    def run():
        ec, tc = e.cuda().requires_grad_(True), table.cuda().requires_grad_(True)
        out = ops.sasrec_input(ec, tc[:L], keep.cuda(), alpha=alpha)
        (out * R.cuda()).sum().backward()
        return out.detach(), ec.grad, tc.grad

    out, de, dt = run()
    assert_close(out, want.detach().float(), 1e-5 * max(1.0, float(want.abs().max())), "out")
    assert_close(de, ed.grad.float(), 1e-5 * max(1.0, float(ed.grad.abs().max())), "de")
    assert_close(dt, td.grad.float(), 1e-4 * max(1.0, float(td.grad.abs().max())), "d position rows")
    assert float(dt[L:].abs().max()) == 0.0
    again = run()
    assert torch.equal(dt, again[2]) and torch.equal(de, again[1])


def test_ffn_sublayer_backward_scales_masked_rows_inside_its_gemms():
    """ops.sasrec_ffn_sublayer with a 0 / 1 keep mask at a size the slab kernels take (12 800 rows of 64): the backward that
    scales the rows inside the dW kernel and the dx epilogues (rbx_linear_dwdb_scaled / rbx_linear_dx_scaled) against the one
    that first writes dout * keep, and both against the block in torch float64."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(5)
    B, L, E = 64, 200, 64
    e = torch.randn(B, L, E, generator=g)
    keep = (torch.rand(B, L, generator=g) > 0.3)
    R = torch.randn(B, L, E, generator=g)
    norm = torch.nn.LayerNorm(E, eps=1e-8)
    w1, b1, w2, b2 = (torch.randn(E, E, generator=g) * 0.2, torch.randn(E, generator=g) * 0.1,
                      torch.randn(E, E, generator=g) * 0.2, torch.randn(E, generator=g) * 0.1)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(E, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(E, generator=g) * 0.1)

    def reference():
        nd = __import__("copy").deepcopy(norm).double()
        ps = [t.double().requires_grad_(True) for t in (e, w1, b1, w2, b2)]
        n = nd(ps[0])
        out = (n + torch.relu(n @ ps[1].t() + ps[2]) @ ps[3].t() + ps[4]) * keep.double().unsqueeze(-1)
        (out * R.double()).sum().backward()
        return [out.detach()] + [p.grad for p in ps] + [nd.weight.grad, nd.bias.grad]

    def run(in_gemms, as_mask):
        nc = __import__("copy").deepcopy(norm).cuda()
        ps = [t.cuda().requires_grad_(True) for t in (e, w1, b1, w2, b2)]
        old = ops.config.ffn_mask_in_gemms
        ops.config.ffn_mask_in_gemms = in_gemms
        try:
            out = ops.sasrec_ffn_sublayer(ps[0], nc, ps[1], ps[2], ps[3], ps[4], keep.cuda() if as_mask else keep.float().cuda())
            (out * R.cuda()).sum().backward()
        finally:
            ops.config.ffn_mask_in_gemms = old
        return [out.detach()] + [p.grad for p in ps] + [nc.weight.grad, nc.bias.grad]

    want = reference()
    names = ("out", "de", "dw1", "db1", "dw2", "db2", "dgamma", "dbeta")
    for got in (run(True, True), run(True, False), run(False, True)):
        for n, a, b in zip(names, got, want):
            assert_close(a, b.float(), TOL * max(1.0, float(b.abs().max())), n)


def test_deepfm_cfg4_full_size_sampled_rows_vs_fp64_oracle():
    """BASELINE.json cfg 4 at full size (Criteo-sized tables, D = 64, MLP 3 x 400, B = 65 536): the predictions of 512
    sampled rows against the oracle's restatement evaluated in float64 on those rows (BatchNorm in eval mode, so that a
    row does not depend on the rest of the batch), and the first-order / tower gradients against the oracle on the
    sampled sub-batch."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd.rechub.models.ranking import DeepFM
    B, D = 65536, 64
    dense, sparse = bench._deepfm_features(D)
    mlp = {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"}
    with torch.device("cuda"):
        model = DeepFM(dense + sparse, sparse, mlp)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.05)
    g = torch.Generator().manual_seed(3)
    for m in model.modules():                      # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    model.eval()
    x = bench._deepfm_batch(B, 77, "uniform", "cuda")
    with torch.no_grad():
        pred = model(x)
    pick = torch.randint(0, B, (512,), generator=g)
    ref = R.RefDeepFM(dense + sparse, sparse, mlp).double().eval()
    sd = {k: v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu() for k, v in model.state_dict().items()}
    ref.load_state_dict(sd)
    xs = {k: (v[pick.cuda()].cpu().double() if v.is_floating_point() else v[pick.cuda()].cpu()) for k, v in x.items()}
    with torch.no_grad():
        want = ref(xs)
    assert_close(pred[pick.cuda()], want, TOL, "predictions of the sampled rows")
    # gradients on the sampled sub-batch (eval-mode BatchNorm): every dense parameter, and the rows of one large table
    sub = {k: v[pick.cuda()].contiguous() for k, v in x.items()}
    F.binary_cross_entropy(model(sub), sub["label"], reduction="sum").backward()
    F.binary_cross_entropy(ref(xs), xs["label"].double(), reduction="sum").backward()
    wantp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        if "embed_dict" in n and p.shape[0] > 100000 and "C2" not in n:
            continue                                # (one 1 M-row table is enough; the rest are 256 MB of zeros each)
        assert_close(p.grad, wantp[n].grad, TOL, "grad " + n)       # 3e-6 achieved (parity ledger, profiles/r04)


def test_sasrec_cfg5_full_size_vs_oracle():
    """BASELINE.json cfg 5's shape (1 M items, D = 64, L = 200, 2 blocks, one head) at B = 512 against the oracle's
    restatement on the CPU: both logit blocks and the gradients of every block parameter; the item-table gradient through
    its touched rows."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd.rechub.models.matching import SASRec
    V, D, L, B = 1_000_000, 64, 200, 512
    feats = bench._sasrec_features(V, D)
    with torch.device("cuda"):
        model = SASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.1)
    model.train()
    x = bench._sasrec_batch(B, V, L, 5, "cuda")
    ref = R.RefSASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1).train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    xc = {k: v.cpu() for k, v in x.items()}
    pl, nl = model(x)
    pl0, nl0 = ref(xc)
    assert_close(pl, pl0, TOL, "pos logits")        # logits are O(10): sums of 64 products of O(1) values; 1.2e-7 achieved
    assert_close(nl, nl0, TOL, "neg logits")
    w = x["weight"] * float((x["seq"] != 0).sum())                           # 1 on real positions
    (-((F.logsigmoid(pl) + F.logsigmoid(-nl)) * w).sum()).backward()
    (-((F.logsigmoid(pl0) + F.logsigmoid(-nl0)) * w.cpu()).sum()).backward()
    wantp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        want = wantp[n].grad
        tol = 1e-4 * max(1.0, float(want.abs().max()))                        # sums over up to 100 000 positions
        if n.endswith("embed_dict.seq.weight"):
            rows = torch.unique(torch.cat([xc["seq"].reshape(-1), xc["pos"].reshape(-1), xc["neg"].reshape(-1)]))[:20000]
            assert_close(p.grad[rows.cuda()], want[rows], tol, "item rows")
            continue
        assert_close(p.grad, want, tol, "grad " + n)


def test_deepfm_step_with_weight_gradients_beside_the_lookup_backward_is_the_same_bits():
    """ops.config.dw_beside_lookup on DeepFM at the benchmarked shape (B = 8 192, D = 64, 3 x 400): the input stage's dW1 / db1
    and first-order gradients and every tower layer's dW / db run on the side stream beside the embedding lookup's backward /
    the BatchNorm backward below -- every gradient (dense tables included) bit-identical to the in-line order, over two steps
    with different batches (the second one re-uses the allocator's blocks the first one freed)."""
    import bench
    from recbox_amd import ops
    from recbox_amd.rechub.models.ranking import DeepFM
    B, D = 8192, 64
    dense, sparse = bench._deepfm_features(D)
    mlp = {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"}
    with torch.device("cuda"):
        model = DeepFM(dense + sparse, sparse, mlp)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.05)
    model.train()
    old = ops.config.dw_beside_lookup
    got = {}
    try:
        for beside in (False, True):
            ops.config.dw_beside_lookup = beside
            grads = []
            for k in range(2):
                x = bench._deepfm_batch(B, 90 + k, "uniform", "cuda")
                for p in model.parameters():
                    p.grad = None
                ops.binary_cross_entropy(model(x), x["label"]).backward()
                torch.cuda.synchronize()
                grads.append([p.grad.clone() if p.grad is not None else None for p in model.parameters()])
            got[beside] = grads
            with torch.no_grad():                                   # (the BatchNorm running statistics moved: same start for both)
                for m in model.modules():
                    if isinstance(m, torch.nn.BatchNorm1d):
                        m.reset_running_stats()
    finally:
        ops.config.dw_beside_lookup = old
    names = [n for n, _ in model.named_parameters()]
    for k in range(2):
        for n, a, b in zip(names, got[True][k], got[False][k]):
            assert (a is None) == (b is None), n
            if a is not None:
                assert torch.equal(a, b), "step %d: %s" % (k, n)


def test_deepfm_training_step_in_the_benchmarked_mode_vs_fp64_oracle():
    """What ``bench.py --config deepfm`` times, end to end against the oracle: DeepFM at the Criteo
    table sizes, D = 64, the 3 x 400 tower at B = 8 192 -- large enough for the split-operand bf16 GEMMs (gemm_bxp / gemm_bxt:
    K >= 256, N >= 128), BatchNorm in TRAINING mode (batch statistics, running statistics updated) and the fused input
    stage (deepfm_input_stage) -- against ``RefDeepFM`` in float64 on the CPU: every prediction at 1e-4 absolute, the
    BatchNorm running statistics after the step, every dense gradient and the rows of one 1 M-row table at 1e-4 of the
    tensor's largest entry (the backward driven by a random linear functional of the predictions: O(1) gradients)."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd import ops
    from recbox_amd.rechub.models.ranking import DeepFM
    B, D = 8192, 64
    dense, sparse = bench._deepfm_features(D)
    mlp = {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"}
    with torch.device("cuda"):
        model = DeepFM(dense + sparse, sparse, mlp)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.05)
    # BatchNorm scales near their default 1 (not N(0, 0.05)): with |gamma| ~ 0.05 the pre-ReLU values crowd around zero and
    # a handful of the 3.3 M units per layer land within fp32 rounding of the kink -- the fp64 reference then switches a
    # unit the fp32 step does not, and ONE such unit moves a column's dgamma / dbeta by 1e-2 and everything below it by
    # 1e-3 (observed, identically with the f32 and the split-bf16 GEMMs: profiles/r04/INDEX.md); that is the ReLU's
    # discontinuity, not a kernel's error
    gb = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_((1.0 + 0.1 * torch.randn(m.num_features, generator=gb)).cuda())
                m.bias.copy_((0.1 * torch.randn(m.num_features, generator=gb)).cuda())
    ref = R.RefDeepFM(dense + sparse, sparse, mlp).double().train()
    ref.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()
                         for k, v in model.state_dict().items()})
    model.train()
    x = bench._deepfm_batch(B, 78, "uniform", "cuda")
    xs = {k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in x.items()}
    before = ops.gemm_bx6_count()
    # The ReLU decisions of the reference are taken from THIS forward: the fp32 pre-activations carry ~4e-6 of rounding,
    # so of the 3 x 3.3 M units a few lie closer to the kink than that and the fp64 reference would switch them the other
    # way -- ONE such unit moves its column's dgamma / dbeta by 1e-2 and every gradient below it by 1e-3 (measured on bare
    # towers, with the f32 and the split-bf16 GEMMs alike: profiles/r04/INDEX.md).  That is the ReLU's discontinuity, not
    # an error of a kernel; with the masks fixed, the two backward passes differentiate the same piecewise-linear function.
    masks = []
    real_bn = ops.batch_norm

    def recording_bn(xin, module, relu=False, prelu=None):
        out = real_bn(xin, module, relu=relu, prelu=prelu)
        if relu:
            masks.append((out.detach() > 0).cpu())
        return out

    ops.batch_norm = recording_bn
    try:
        pred = model(x)
    finally:
        ops.batch_norm = real_bn
    relus = [i for i, mod in enumerate(ref.mlp.mlp) if isinstance(mod, torch.nn.ReLU)]
    assert len(masks) == len(relus) == 3

    class MaskedReLU(torch.nn.Module):
        def __init__(self, keep):
            super().__init__()
            self.keep = keep.double()

        def forward(self, z):
            return z * self.keep

    for i, keep in zip(relus, masks):
        ref.mlp.mlp[i] = MaskedReLU(keep)
    # The backward is driven by a fixed random functional of the predictions, sum(R * pred): BCE on fp32 PROBABILITIES
    # (the reference's own loss, ctr_trainer.py:33) divides by p (1 - p), which a saturated fp32 prediction holds to a few
    # digits only -- 1e-3 relative on such a sample whatever computes it, the reference's fp32 CPU path included
    R0 = torch.randn(B, 1, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    (pred * R0.float().cuda().view_as(pred)).sum().backward()
    if ops.config.gemm_bx6:
        assert ops.gemm_bx6_count() > before, "the split-operand GEMMs did not run at the tower's shapes"
    want = ref(xs)
    (want * R0.view_as(want)).sum().backward()
    assert_close(pred, want, TOL, "training-mode predictions, all %d rows" % B)
    for (n, b), (_, b0) in zip(model.named_buffers(), ref.named_buffers()):
        if b.is_floating_point():
            assert_close(b, b0, TOL, "BatchNorm buffer " + n)
    wantp = dict(ref.named_parameters())
    bad = []
    for n, p in model.named_parameters():
        big = "embed_dict" in n and p.shape[0] > 100000
        if big and not n.endswith("embed_dict.C2.weight"):
            continue                                # (one 1 M-row table is enough)
        w = wantp[n].grad
        tol = TOL * max(1.0, float(w.abs().max()))
        try:
            if big:
                rows = torch.unique(xs["C2"])
                assert_close(p.grad[rows.cuda()], w[rows], tol, "rows of the 1 M-row table " + n)
                untouched = torch.ones(p.shape[0], dtype=torch.bool)
                untouched[rows] = False
                assert float(p.grad[untouched.cuda()].abs().max()) == 0.0
            else:
                assert_close(p.grad, w, tol, "grad " + n)
        except AssertionError as e:                 # (every parameter is compared before the test fails: the list says where)
            bad.append("%s (|grad|max %.3g): %s" % (n, float(w.abs().max()), str(e).split("\n")[0]))
    assert not bad, "\n".join(bad)


def test_sasrec_at_the_benchmarked_batch_on_sampled_sequences_vs_oracle():
    """``bench.py --config sasrec`` runs B = 4096; the oracle comparison above runs B = 512.  SASRec has
    no coupling between the sequences of a batch (LayerNorm is per position), so the model is run at B = 4096 -- the
    launch shapes the bench times -- and compared with the oracle on 192 sampled sequences: both logit blocks at 1e-4, and,
    with the loss restricted to those sequences, the gradients of every block parameter and of the touched item rows."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd.rechub.models.matching import SASRec
    V, D, L, B, S = 1_000_000, 64, 200, 4096, 192
    feats = bench._sasrec_features(V, D)
    with torch.device("cuda"):
        model = SASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1)
    bench.init_weights_device(model, torch.device("cuda"), 0, 0, std=0.1)
    model.train()
    x = bench._sasrec_batch(B, V, L, 6, "cuda")
    g = torch.Generator().manual_seed(8)
    pick = torch.randperm(B, generator=g)[:S].sort().values
    ref = R.RefSASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1).train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    xc = {k: v[pick.cuda()].cpu() for k, v in x.items()}
    pl, nl = model(x)
    pl0, nl0 = ref(xc)
    assert_close(pl[pick.cuda()], pl0, TOL, "pos logits of the sampled sequences (model run at B = 4096)")
    assert_close(nl[pick.cuda()], nl0, TOL, "neg logits of the sampled sequences")
    w = torch.zeros(B, L, device="cuda")
    w[pick.cuda()] = (x["seq"][pick.cuda()] != 0).float()
    (-((F.logsigmoid(pl) + F.logsigmoid(-nl)) * w).sum()).backward()
    (-((F.logsigmoid(pl0) + F.logsigmoid(-nl0)) * w[pick.cuda()].cpu()).sum()).backward()
    wantp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        want = wantp[n].grad
        tol = TOL * max(1.0, float(want.abs().max()))
        if n.endswith("embed_dict.seq.weight"):
            rows = torch.unique(torch.cat([xc["seq"].reshape(-1), xc["pos"].reshape(-1), xc["neg"].reshape(-1)]))[:20000]
            assert_close(p.grad[rows.cuda()], want[rows], tol, "item rows")
            continue
        assert_close(p.grad, want, tol, "grad " + n)


@pytest.mark.parametrize("rows,n", [(1, 5), (257, 5), (65536, 5), (1000, 1), (300, 101)])
def test_softmax_cross_entropy_epilogue_vs_torch(rows, n):
    """K7's loss epilogue (rbx_softmax_ce_*): -log softmax(y)[:, 0] mean (softmax_crossentropy_loss.py:19-22) and the
    general-target CrossEntropyLoss, forward and gradient, against torch in fp64; a column view of a wider block is read
    in place; an out-of-range target raises; bit-identical on repeat."""
    from recbox_amd import ops
    import recbox_amd.core.pytorch.losses as Ls
    g = torch.Generator().manual_seed(rows + n)
    wide = (torch.randn(rows, n + 3, generator=g) * 4).cuda()
    y = wide[:, :n]                                                       # row stride n + 3
    for target in (None, torch.randint(0, n, (rows,), generator=g).cuda()):
        x = y.detach().clone().requires_grad_() if False else y.detach().requires_grad_()
        loss = ops.softmax_cross_entropy(x, target)
        loss.backward()
        xd = y.detach().double().cpu().requires_grad_()
        t = torch.zeros(rows, dtype=torch.long) if target is None else target.cpu()
        want = F.cross_entropy(xd, t)
        want.backward()
        assert_close(loss, want, 1e-5, "loss")
        assert_close(x.grad, xd.grad, 1e-6, "dlogits")
        again = ops.softmax_cross_entropy(y.detach(), target)
        assert torch.equal(again, loss.detach())
    crit = Ls.SoftmaxCrossEntropyLoss()
    assert_close(crit(y.detach(), None), -torch.log(torch.softmax(y.detach().double().cpu(), 1)[:, 0]).mean(), 1e-5)
    if n > 1:
        with pytest.raises(IndexError):
            ops.softmax_cross_entropy(y.detach(), torch.full((rows,), n, device="cuda"))


@pytest.mark.parametrize("shape,weighted", [((7, 12), True), ((512, 200), True), ((1, 1), False), ((4096, 200), True)])
def test_pair_logsigmoid_epilogue_vs_torch(shape, weighted):
    from recbox_amd import ops
    g = torch.Generator().manual_seed(shape[0])
    pos, neg = (torch.randn(shape, generator=g) * 6 for _ in range(2))
    w = None
    if weighted:
        keep = (torch.rand(shape, generator=g) > 0.3).float()
        w = keep / keep.sum().clamp(min=1)
    pc, nc = pos.cuda().requires_grad_(), neg.cuda().requires_grad_()
    loss = ops.pair_logsigmoid_loss(pc, nc, w.cuda() if w is not None else None, scale=0.5)
    loss.backward()
    pd, nd = pos.double().requires_grad_(), neg.double().requires_grad_()
    ww = w.double() if w is not None else 1.0
    want = 0.5 * (-((F.logsigmoid(pd) + F.logsigmoid(-nd)) * ww).sum())
    want.backward()
    assert_close(loss, want, 1e-5, "loss", rtol=1e-5)
    assert_close(pc.grad, pd.grad, 1e-6, "dpos")
    assert_close(nc.grad, nd.grad, 1e-6, "dneg")


@pytest.mark.parametrize("B,L,H,hd,causal,p", [(3, 200, 1, 64, True, 0.0), (2, 37, 2, 32, True, 0.0), (4, 64, 4, 32, False, 0.0),
                                               (3, 200, 1, 64, True, 0.5), (2, 50, 2, 64, False, 0.25)])
def test_attention_on_packed_operands_equals_the_copying_path(B, L, H, hd, causal, p):
    """rbx_attn_packed_*: Q [B, L, E] and K | V [B, L, 2 E] read where the projections left them, O / dQ / dK | dV written
    in place -- bit-identical to splitting, transposing to [B, H, L, hd], calling rbx_attn_* and transposing back (same
    kernels, same (b * H + h) numbering, hence also the same dropout mask for a given seed)."""
    from recbox_amd import ops
    E = H * hd
    g = torch.Generator().manual_seed(B * L + H)
    q = torch.randn(B, L, E, generator=g).cuda()
    kv = torch.randn(B, L, 2 * E, generator=g).cuda()
    R = torch.randn(B, L, E, generator=g).cuda()
    assert ops.attention_packed_supported(L, hd)
    q1, kv1 = q.clone().requires_grad_(), kv.clone().requires_grad_()
    o1 = ops.attention_packed(q1, kv1, H, hd ** -0.5, causal=causal, dropout_p=p, seed=99)
    (o1 * R).sum().backward()
    q2, kv2 = q.clone().requires_grad_(), kv.clone().requires_grad_()
    qh = q2.view(B, L, H, hd).transpose(1, 2)
    kh = kv2[..., :E].reshape(B, L, H, hd).transpose(1, 2)
    vh = kv2[..., E:].reshape(B, L, H, hd).transpose(1, 2)
    o2, _ = ops.attention(qh, kh, vh, scale=hd ** -0.5, causal=causal, fill=float("-inf"), dropout_p=p, seed=99)
    o2 = o2.transpose(1, 2).reshape(B, L, E)
    (o2 * R).sum().backward()
    assert torch.equal(o1, o2)
    assert torch.equal(q1.grad, q2.grad)
    assert torch.equal(kv1.grad, kv2.grad)


def test_l2_normalize_reads_rows_inside_a_wider_block():
    """ops.l2_normalize of a [B, n, D] VIEW behind the first columns of a [B, width] block (rbx_l2norm_fwd_strided) and
    ops.split_last(views=True): the values and gradients of the contiguous route, without the copies."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(9)
    B, n, D, lead = 301, 5, 32, 48
    base = torch.randn(B, lead + n * D + 3, generator=g).cuda()            # (+3: the row stride is not the row length)
    base[7, lead:lead + D] = 0.0                                            # a zero row: the eps clamp
    block = base[:, :lead + n * D].clone().requires_grad_(True)
    wide = base.clone().requires_grad_(True)
    a0, b0 = ops.split_last(block, lead)
    a1, b1 = ops.split_last(wide[:, :lead + n * D], lead, views=True)
    assert b1.data_ptr() != b0.data_ptr() and not b1.is_contiguous()
    y0 = ops.l2_normalize(b0.view(B, n, D))
    y1 = ops.l2_normalize(b1.unflatten(1, (n, D)))
    assert torch.equal(y0, y1) and torch.equal(a0, a1)
    w = torch.randn(B, n, D, generator=g).cuda()
    ((y0 * w).sum() + a0.sum()).backward()
    ((y1 * w).sum() + a1.sum()).backward()
    assert torch.equal(block.grad, wide.grad[:, :lead + n * D])
    assert float(wide.grad[:, lead + n * D:].abs().max()) == 0.0
    ref = F.normalize(b0.detach().view(B, n, D).double(), dim=-1).float()
    assert_close(y1, ref, 1e-6)


def test_cos_dot_fuses_the_normalisation_and_mirrors_the_gradient_layout():
    """ops.cos_dot(u, v) == scale * <u, F.normalize(v)> (values, both gradients, the eps clamp on a zero row) for contiguous
    rows and for rows read behind the leading columns of a wider block; in the second case the gradient comes back in a block
    of the same shape and ops.split_last(views=True) returns it whole -- same numbers as the concatenating route."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(21)
    B, N, D, lead = 257, 5, 32, 64
    u0 = F.normalize(torch.randn(B, D, generator=g), dim=-1)
    blk0 = torch.randn(B, lead + N * D, generator=g)
    blk0[3, lead + D:lead + 2 * D] = 0.0                                    # a zero candidate row
    w = torch.randn(B, N, generator=g)
    # float64 restatement
    ur = u0.double().requires_grad_(True)
    br = blk0.double().requires_grad_(True)
    vr = br[:, lead:].unflatten(1, (N, D))
    outr = 7.0 * (ur.unsqueeze(1) * F.normalize(vr, p=2, dim=-1, eps=1e-12)).sum(-1)
    ((outr * w.double()).sum() + (br[:, :lead] * 0.5).sum()).backward()
    for views in (True, False):
        u = u0.cuda().requires_grad_(True)
        blk = blk0.cuda().requires_grad_(True)
        a, b = ops.split_last(blk, lead, views=views)
        out = ops.cos_dot(u, b.unflatten(1, (N, D)), eps=1e-12, scale=7.0)
        ((out * w.cuda()).sum() + (a * 0.5).sum()).backward()
        assert_close(out, outr.float(), 1e-5, "values (views=%s)" % views)
        assert_close(u.grad, ur.grad.float(), 1e-5, "du (views=%s)" % views)
        assert_close(blk.grad, br.grad.float(), 1e-5, "d block (views=%s)" % views)
    # the view route really skipped the concatenation: its gradient block is the buffer cos_dot wrote into
    blk = blk0.cuda().requires_grad_(True)
    a, b = ops.split_last(blk, lead, views=True)
    seen = {}
    def remember(gr):
        seen["ptr"] = gr.untyped_storage().data_ptr()

    hook = b.register_hook(remember)
    ops.cos_dot(u0.cuda(), b.unflatten(1, (N, D)), scale=1.0).sum().backward()
    hook.remove()
    assert blk.grad.untyped_storage().data_ptr() == seen["ptr"]


@pytest.mark.parametrize("B,N,D", [(1, 1, 4), (257, 5, 128), (130, 7, 256), (65, 3, 512), (300, 9, 20), (33, 4, 1024),
                                   (4100, 6, 64), (100, 2, 6)])
def test_cos_dot_row_widths(B, N, D):
    """ops.cos_dot over the row widths its kernels split differently: float4 rows on one lane group (D = 4 .. 256), several
    float4 per lane (D = 512, 1024), the scalar kernels (D % 4 != 0 or no power-of-two group), more candidates than one chunk
    of four -- values and both gradients against float64, a zero row included."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(B + N + D)
    u0 = torch.randn(B, D, generator=g)
    v0 = torch.randn(B, N, D, generator=g)
    v0[B // 2, N - 1] = 0.0
    w = torch.randn(B, N, generator=g)
    ur, vr = u0.double().requires_grad_(True), v0.double().requires_grad_(True)
    outr = 3.0 * (ur.unsqueeze(1) * F.normalize(vr, p=2, dim=-1, eps=1e-12)).sum(-1)
    (outr * w.double()).sum().backward()
    u, v = u0.cuda().requires_grad_(True), v0.cuda().requires_grad_(True)
    out = ops.cos_dot(u, v, eps=1e-12, scale=3.0)
    (out * w.cuda()).sum().backward()
    tol = 2e-5 * max(1.0, D ** 0.5 / 8)
    assert_close(out, outr.float(), tol, "values")
    assert_close(u.grad, ur.grad.float(), tol * max(1.0, N ** 0.5), "du")
    assert_close(v.grad, vr.grad.float(), tol, "dv")


def test_deepfm_input_stage_as_one_node_equals_the_three_readers():
    """ops.deepfm_input_stage (tower's first Linear + FM + first-order Linear over one gathered block; the block's gradient out
    of ONE GEMM, rbx_linear_dx_deepfm) against the three separate readers: predictions and every gradient, and against the
    live-reference fixture."""
    from recbox_amd import ops
    Fe, La = _rh()
    from recbox_amd.rechub.models.ranking import DeepFM
    fx = Fixture("rechub_deepfm")
    X = _cuda(fx.tensors("in"))

    def run(fused):
        dense = [Fe.DenseFeature("I%d" % i) for i in range(1, 4)]
        sparse = [Fe.SparseFeature("C%d" % (i + 1), v + 1, 16) for i, v in enumerate(CRITEO_SMALL_VOCABS[:8])]
        model = DeepFM(sparse + dense, sparse, {"dims": [32, 16], "dropout": 0.0, "activation": "relu"})
        load_params(model, fx["p"]).cuda().train()
        old = ops.config.fuse_deepfm_input
        ops.config.fuse_deepfm_input = fused
        try:
            p = model(X)
            F.binary_cross_entropy(p, X["label"]).backward()
        finally:
            ops.config.fuse_deepfm_input = old
        return p.detach(), dict((n, q.grad.clone()) for n, q in model.named_parameters())

    pa, ga = run(True)
    pb, gb = run(False)
    assert_close(pa, pb, 1e-6, "predictions")
    for n in gb:
        assert_close(ga[n], gb[n], 1e-6, "grad " + n)
    assert_close(pa, fx["out"]["y"], TOL)
    for n in ga:
        assert_close(ga[n], fx["g"][n], TOL, "grad " + n)


@pytest.mark.parametrize("rows,cols,per_column", [(257, 40, False), (257, 40, True), (4099, 401, False)])
def test_batch_norm_with_fused_prelu_vs_torch(rows, cols, per_column):
    """rbx_batchnorm_prelu_fwd/bwd == nn.BatchNorm1d -> nn.PReLU of torch CPU: outputs, dx, dgamma / dbeta, the slope's
    gradient (one parameter, or one per column), running statistics; slopes of BOTH signs (with a <= 0 the sign of y no
    longer tells the sign of the pre-activation: the kernels rebuild z from x); training and eval mode; and through a
    rechub MLP(activation="prelu") tower (third_party/rechub/basic/layers.py:255-263), where the walker fuses the pair."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(rows + cols + int(per_column))
    ref_bn, ref_act = torch.nn.BatchNorm1d(cols), torch.nn.PReLU(cols if per_column else 1)
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(cols, generator=g) + 0.5)
        ref_bn.bias.copy_(torch.randn(cols, generator=g) * 0.3)
        ref_act.weight.copy_(torch.randn(ref_act.weight.shape, generator=g) * 0.5 if per_column else torch.tensor([-0.3]))
    dut_bn, dut_act = torch.nn.BatchNorm1d(cols), torch.nn.PReLU(cols if per_column else 1)
    dut_bn.load_state_dict(ref_bn.state_dict())
    dut_act.load_state_dict(ref_act.state_dict())
    dut_bn.cuda()
    dut_act.cuda()
    for step in range(3):
        if step == 2:
            ref_bn.eval()
            dut_bn.eval()
        x = torch.randn(rows, cols, generator=g) * 2.0 + 1.0
        r = torch.randn(rows, cols, generator=g)
        xr = x.clone().requires_grad_(True)
        zr = ref_bn(xr)
        yr = ref_act(zr)
        (yr * r).sum().backward()
        xc = x.cuda().requires_grad_(True)
        yc = ops.batch_norm(xc, dut_bn, prelu=dut_act)
        (yc * r.cuda()).sum().backward()
        assert_close(yc, yr.detach(), 1e-5, "y")
        keep = ~(zr.detach().abs() < 1e-5).any(dim=0)          # columns without a pre-activation within an ulp of zero
        scale = max(1.0, rows ** 0.5 / 4)
        assert_close(xc.grad.cpu()[:, keep], xr.grad[:, keep], TOL, "dx")
        assert_close(dut_bn.weight.grad.cpu()[keep], ref_bn.weight.grad[keep], 1e-4 * scale, "dgamma")
        assert_close(dut_bn.bias.grad.cpu()[keep], ref_bn.bias.grad[keep], 1e-4 * scale, "dbeta")
        if per_column:
            assert_close(dut_act.weight.grad.cpu()[keep], ref_act.weight.grad[keep], 1e-4 * scale, "dslope")
        elif bool(keep.all()):
            assert_close(dut_act.weight.grad, ref_act.weight.grad, 1e-4 * scale * cols ** 0.5, "dslope")
        assert_close(dut_bn.running_mean, ref_bn.running_mean, 1e-6, "running_mean")
        assert_close(dut_bn.running_var, ref_bn.running_var, 1e-5, "running_var")
        for m in (ref_bn, ref_act, dut_bn, dut_act):
            m.zero_grad()
    if rows > 1000:
        return
    # the tower: Linear -> BatchNorm1d -> PReLU -> Dropout(0) twice + Linear(*, 1)
    from recbox_amd.rechub.basic.layers import MLP
    torch.manual_seed(0)
    tower = MLP(cols, output_layer=True, dims=[24, 16], activation="prelu").cuda().train()
    twin = torch.nn.Sequential(*[type(m)(*(([m.in_features, m.out_features]) if isinstance(m, torch.nn.Linear) else
                                          ([m.num_features] if isinstance(m, torch.nn.BatchNorm1d) else
                                           ([m.p] if isinstance(m, torch.nn.Dropout) else []))))
                                 for m in tower.mlp]).train()
    twin.load_state_dict(dict((k, v.cpu()) for k, v in tower.mlp.state_dict().items()))
    x = torch.randn(rows, cols, generator=g)
    out = tower(x.cuda())
    want = twin(x)
    assert_close(out, want.detach(), 1e-4, "MLP(prelu) output")
    out.sum().backward()
    want.sum().backward()
    for (n, p), (_, q) in zip(tower.mlp.named_parameters(), twin.named_parameters()):
        assert_close(p.grad, q.grad, TOL * max(1.0, float(q.grad.abs().max())), "MLP(prelu) grad " + n)


def test_hip_path_vs_second_oracle_recbole_layers():
    """The HIP path against the SECOND oracle (RecBole's FM and attention layers, fixture from the live reference:
    oracle/gen_golden_recbole.py; third_party/recbole/model/layers.py:127-205,380-470): the FM term and vector through
    rbx_interaction_*, and the attention layer composed from the path's own ops -- rbx_linear for the four projections,
    rbx_attn (causal, 2 heads of 16) and rbx_layernorm -- outputs and every gradient at 1e-4."""
    from recbox_amd import ops
    fx = Fixture("recbole_layers")
    fm = fx.tensors("fm", "cuda")
    table = fm["p.table"].clone().requires_grad_()
    e = table[fm["in.ids"] + fm["in.offsets"].unsqueeze(0)]
    out = ops.interaction(e, "product_sum")
    assert_close(out, fm["out.sum"], TOL, "FM term")
    assert_close(ops.interaction(e.detach(), "bi_interaction"), fm["out.vec"], TOL, "FM vector")
    (out * fm["in.R"]).sum().backward()
    assert_close(table.grad, fm["g.table"], TOL, "FM table gradient")
    m = fx.tensors("mha", "cuda")
    heads = int(m["in.heads"])
    x = m["in.x"].clone().requires_grad_()
    p = dict((k[2:], v.clone().requires_grad_()) for k, v in m.items() if k.startswith("p."))
    B, L, H = x.shape
    hd = H // heads

    def split(t):
        return t.view(B, L, heads, hd).transpose(1, 2).reshape(B * heads, L, hd)
    q = split(ops.linear(x, p["query.weight"], p["query.bias"]))
    k = split(ops.linear(x, p["key.weight"], p["key.bias"]))
    v = split(ops.linear(x, p["value.weight"], p["value.bias"]))
    o = ops.attention(q, k, v, scale=1.0 / hd ** 0.5, causal=True)
    o = o[0] if isinstance(o, tuple) else o
    ctx = o.view(B, heads, L, hd).transpose(1, 2).reshape(B, L, H)
    ln = torch.nn.LayerNorm(H, eps=1e-12).cuda()
    ln.weight, ln.bias = torch.nn.Parameter(p["LayerNorm.weight"].detach().clone()), torch.nn.Parameter(p["LayerNorm.bias"].detach().clone())
    y = ops.layer_norm(ops.linear(ctx, p["dense.weight"], p["dense.bias"]) + x, ln)
    assert_close(y, m["out.y"], TOL, "attention layer output")
    (y * m["in.R"]).sum().backward()
    assert_close(x.grad, m["g.x"], TOL, "attention layer dx")
    for name, t in p.items():
        got = {"LayerNorm.weight": ln.weight.grad, "LayerNorm.bias": ln.bias.grad}.get(name, t.grad)
        assert_close(got, m["g." + name], TOL, "attention layer grad " + name)
