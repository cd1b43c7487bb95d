"""GPU parity of the ranking-side hot path (FeatureEmbedding / FM / LR /
InnerProductInteraction / pooling) against the golden fixtures of the live
reference and against the oracle on seeded inputs.  All calls go through the
C ABI (ctypes -> librecbox_hip.so).  Tolerance: BASELINE.json's 1e-4 on fp32
outputs and gradients; row selection is exact."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import Fixture, assert_close, assert_grads_close, load_params
from test_oracle_golden import (CRITEO_SMALL_VOCABS, _FM, criteo_small_features, ranking_embedding_features)

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cuda(d):
    return OrderedDict((k, v.cuda()) for k, v in d.items())


def _layers():
    import recbox_amd.ranking.pytorch.layers as L
    return L


def test_library_is_native():
    from recbox_amd import _lib
    assert _lib.lib.rbx_version() >= 100
    assert torch.cuda.is_available()


def test_feature_embedding_golden():
    L = _layers()
    fx = Fixture("ranking_feature_embedding")
    fm = _FM(ranking_embedding_features())
    layer = load_params(L.FeatureEmbedding(fm, 8), fx["p"]).cuda()
    X = _cuda(fx.tensors("in"))
    out = layer(X)
    assert_close(out, fx["out"]["emb"], TOL, "emb")
    (out * X["R"]).sum().backward()
    assert_grads_close(layer, fx["g"], TOL)
    assert_close(layer(X, feature_source=["user"]), fx["out"]["emb_user"], TOL)
    assert_close(layer(X, feature_type="categorical"), fx["out"]["emb_cat"], TOL)
    assert_close(layer(X, dynamic_emb_dim=True), fx["out"]["emb_dyn"], TOL)
    el = layer.embedding_layer.embedding_layers
    assert el["c2"] is el["hist"] and type(el["c1"]) is torch.nn.Embedding and type(el["n1"]) is torch.nn.Linear


def test_dict2tensor_stacks_the_current_values_after_an_edit():
    """models post-process entries of the embedding dict (``feature_emb_dict[f] = gated``) and then call
    ``dict2tensor``; the reference stacks the CURRENT values (feature_embedding.py:169-186).  The fused block of the
    layer call must not be handed out once the dict was edited -- and gradients must flow through the edit."""
    L = _layers()
    fm = _FM(criteo_small_features())
    layer = L.FeatureEmbedding(fm, 16).cuda()
    X = _cuda(Fixture("ranking_fm").tensors("in"))
    inner = layer.embedding_layer
    d = inner(X)
    whole = inner.dict2tensor(d)
    assert whole.data_ptr() == d.fused.data_ptr()                       # untouched dict: the block itself
    names = list(d.keys())
    gate = torch.rand(whole.shape[0], 16, device="cuda", requires_grad=True)
    d[names[20]] = d[names[20]] * gate
    got = inner.dict2tensor(d)
    want = whole.clone()
    want[:, 20] = whole[:, 20] * gate.detach()
    assert torch.equal(got, want)
    got.sum().backward()
    assert torch.equal(gate.grad, whole[:, 20].detach())
    d2 = inner(X)
    d2.pop(names[3])
    assert torch.equal(inner.dict2tensor(d2), torch.cat([whole[:, :3], whole[:, 4:]], dim=1))


def test_fm_golden():
    L = _layers()
    fx = Fixture("ranking_fm")
    fm = _FM(criteo_small_features())
    model = torch.nn.ModuleDict({"embedding_layer": L.FeatureEmbedding(fm, 16), "fm": L.FactorizationMachine(fm)})
    load_params(model, fx["p"]).cuda()
    X = _cuda(fx.tensors("in"))
    emb = model["embedding_layer"](X)
    assert tuple(emb.shape) == (64, 39, 16)
    assert_close(emb, fx["out"]["feature_emb"], TOL)
    logit = model["fm"](X, emb)
    assert_close(logit, fx["out"]["logit"], TOL)
    loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), X["label"], reduction="mean")
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


def test_inner_product_golden_and_kats():
    L = _layers()
    fx = Fixture("inner_product")
    t = _cuda(fx.tensors("in"))
    for mode in ("product_sum", "bi_interaction", "inner_product", "elementwise_product"):
        e = t["E"].clone().requires_grad_(True)
        o = L.InnerProductInteraction(6, output=mode).cuda()(e)
        assert_close(o, fx["out"][mode], TOL, mode)
        (o * t["R_" + mode]).sum().backward()
        assert_close(e.grad, fx["g"][mode], TOL, "grad " + mode)
    x = torch.arange(12.0).reshape(2, 3, 2).cuda()
    assert L.InnerProductInteraction(3, "product_sum")(x).tolist() == [[31.0], [427.0]]
    assert L.InnerProductInteraction(3, "bi_interaction")(x).tolist() == [[8.0, 23.0], [188.0, 239.0]]
    assert L.InnerProductInteraction(3, "inner_product")(x).tolist() == [[3.0, 5.0, 23.0], [111.0, 137.0, 179.0]]
    with pytest.raises(ValueError):
        L.InnerProductInteraction(3, "outer")


def test_pooling_golden_and_kats():
    L = _layers()
    import recbox_amd.core.pytorch.layers as C
    fx = Fixture("pooling")
    t = _cuda(fx.tensors("in"))
    cases = {"core_avg": lambda e: C.MaskedAveragePooling()(e), "core_sum": lambda e: C.MaskedSumPooling()(e),
             "rank_avg": lambda e: L.MaskedAveragePooling()(e),
             "rank_avg_mask": lambda e: L.MaskedAveragePooling()(e, mask=t["keep"]),
             "rank_sum": lambda e: L.MaskedSumPooling()(e)}
    for key, fn in cases.items():
        e = t["E"].clone().requires_grad_(True)
        o = fn(e)
        assert_close(o, fx["out"][key], TOL, key)
        (o * t["R"]).sum().backward()
        assert_close(e.grad, fx["g"][key], TOL, "grad " + key)
    s = torch.tensor([[[1., 2.], [3., 4.], [0., 0.]], [[0., 0.], [0., 0.], [0., 0.]]]).cuda()
    assert L.MaskedAveragePooling()(s).tolist() == [[2.0, 3.0], [0.0, 0.0]]
    assert L.MaskedAveragePooling()(torch.tensor([[[1., -1.], [3., 4.]]]).cuda()).tolist() == [[4.0, 3.0]]


def test_backward_kat_shared_table_and_padding():
    L = _layers()
    f = OrderedDict()
    f["h"] = {"source": "", "type": "sequence", "vocab_size": 6, "padding_idx": 0, "max_len": 4,
              "feature_encoder": "layers.MaskedAveragePooling()"}
    f["c"] = {"source": "", "type": "categorical", "vocab_size": 6, "padding_idx": 0, "share_embedding": "h"}
    layer = L.FeatureEmbedding(_FM(f), 2).cuda()
    tab = layer.embedding_layer.embedding_layers["h"]
    with torch.no_grad():
        tab.weight.copy_(torch.arange(12.0).reshape(6, 2))
        tab.weight[0].zero_()
    X = {"h": torch.tensor([[1, 2, 2, 0], [0, 0, 0, 0]]).cuda(), "c": torch.tensor([2, 0]).cuda()}
    out = layer(X)
    assert_close(out, torch.tensor([[[10 / 3, 13 / 3], [4., 5.]], [[0., 0.], [0., 0.]]]), 1e-6)
    out.sum().backward()
    want = torch.zeros(6, 2)
    want[1], want[2] = 1 / 3, 5 / 3
    assert_close(tab.weight.grad, want, 1e-6)


def _criteo_like(B, vocabs, D, seed, zipf=False):
    """Seeded model + batch; ids as float64 columns like the hstacked ranking loader."""
    g = torch.Generator().manual_seed(seed)
    feats = OrderedDict()
    for i in range(13):
        feats["I%d" % (i + 1)] = {"source": "", "type": "numeric"}
    for i, v in enumerate(vocabs):
        feats["C%d" % (i + 1)] = {"source": "", "type": "categorical", "vocab_size": v + 1, "padding_idx": 0}
    X = OrderedDict()
    for i in range(13):
        X["I%d" % (i + 1)] = torch.rand(B, generator=g, dtype=torch.float64)
    for i, v in enumerate(vocabs):
        if zipf:
            u = torch.rand(B, generator=g, dtype=torch.float64)
            ids = (torch.floor(v ** u)).clamp(1, v)          # log-uniform: heavy head, long tail
        else:
            ids = torch.randint(1, v + 1, (B,), generator=g).double()
        X["C%d" % (i + 1)] = ids
    y = (torch.rand(B, 1, generator=g) < 0.25).float()
    return _FM(feats), X, y


def _run_fm(model, X, y):
    emb = model["embedding_layer"](X)
    logit = model["fm"](X, emb)
    loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), y, reduction="mean")
    loss.backward()
    return emb, logit, loss


@pytest.mark.parametrize("B,zipf", [(1, False), (7, False), (257, True), (4096, False), (5000, True)])
def test_fm_matches_oracle_seeded(B, zipf):
    """Duplicate-heavy tables (V=3..101 at B up to 5000 => runs crossing many reduce chunks)."""
    from oracle import torch_ref as R
    L = _layers()
    fm, X, y = _criteo_like(B, CRITEO_SMALL_VOCABS + [], 16, seed=B, zipf=zipf)
    ref = torch.nn.ModuleDict({"embedding_layer": R.RefFeatureEmbedding(fm, 16), "fm": R.RefFactorizationMachine(fm)})
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for m in ref.modules():
            if isinstance(m, torch.nn.Embedding) and m.padding_idx is not None:
                m.weight[m.padding_idx].zero_()
    dut = torch.nn.ModuleDict({"embedding_layer": L.FeatureEmbedding(fm, 16), "fm": L.FactorizationMachine(fm)})
    dut.load_state_dict(ref.state_dict())
    dut.cuda()
    e0, l0, loss0 = _run_fm(ref, X, y)
    e1, l1, loss1 = _run_fm(dut, _cuda(X), y.cuda())
    assert torch.equal(e1.cpu(), e0), "gathered rows must be bit-identical (pure copies / one fp32 multiply)"
    assert_close(l1, l0, TOL, "logit")
    assert_close(loss1, loss0, TOL, "loss")
    for (n0, p0), (n1, p1) in zip(ref.named_parameters(), dut.named_parameters()):
        assert n0 == n1
        # mean-reduced BCE scales grads by 1/B: compared at the scale of the summed gradient, to 1e-4 of the tensor's largest
        # entry (a hot row of a 3-row table sums thousands of contributions; until round 4 this was an absolute B / 64 slack)
        want = p0.grad * B
        assert_close(p1.grad * B, want, TOL * max(1.0, float(want.abs().max())), "grad " + n0)


def test_fused_fm_model_golden():
    """FM model body fused into rbx_fm_fwd / rbx_fm_bwd against the live-reference fixture."""
    from recbox_amd.ranking.pytorch.models import FM
    fx = Fixture("ranking_fm")
    fm = _FM(criteo_small_features())
    model = load_params(FM(fm, 16), fx["p"]).cuda()
    X = _cuda(fx.tensors("in"))
    logit = model.logits(X)
    assert_close(logit, fx["out"]["logit"], TOL)
    prob = model(X)["y_pred"]
    assert_close(prob, fx["out"]["prob"], TOL)
    loss = torch.nn.functional.binary_cross_entropy(prob, X["label"], reduction="mean")
    assert_close(loss, fx["out"]["loss"], TOL)
    loss.backward()
    assert_grads_close(model, fx["g"], TOL)


@pytest.mark.parametrize("B,zipf", [(1, False), (33, True), (4096, False), (5000, True)])
def test_fused_fm_matches_layer_path_and_oracle(B, zipf):
    from oracle import torch_ref as R
    from recbox_amd.ranking.pytorch.models import FM
    fm, X, y = _criteo_like(B, CRITEO_SMALL_VOCABS + [], 16, seed=100 + B, zipf=zipf)
    ref = R.RefFMModel(fm, 16)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for m in ref.modules():
            if isinstance(m, torch.nn.Embedding) and m.padding_idx is not None:
                m.weight[m.padding_idx].zero_()
    fused, plain = FM(fm, 16, fused=True), FM(fm, 16, fused=False)
    fused.load_state_dict(ref.state_dict())
    plain.load_state_dict(ref.state_dict())
    fused.cuda(), plain.cuda()
    Xc, yc = _cuda(X), y.cuda()
    outs = []
    for model, inp, lab in ((ref, X, y), (fused, Xc, yc), (plain, Xc, yc)):
        logit = model.logits(inp) if hasattr(model, "logits") else model(inp)
        loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), lab, reduction="sum")
        loss.backward()
        outs.append(logit)
    assert_close(outs[1], outs[0], TOL, "fused logit vs oracle")
    assert_close(outs[2], outs[0], TOL, "layer-path logit vs oracle")
    for (n0, p0), (_, p1), (_, p2) in zip(ref.named_parameters(), fused.named_parameters(),
                                          plain.named_parameters()):
        scale = max(1.0, float(p0.grad.abs().max()))     # summed loss: 1e-4 of the tensor's largest entry (hot rows grow with B)
        assert_close(p1.grad, p0.grad, TOL * scale, "fused grad " + n0)
        assert_close(p2.grad, p0.grad, TOL * scale, "layer grad " + n0)


@pytest.mark.parametrize("dim", [8, 32, 64, 128])
def test_fused_fm_of_other_dims_matches_the_oracle(dim):
    """The fused FM body at embedding dims other than the benchmark's 16: the general forward kernel, the sorted backward for
    every table (tier A is for rows of up to 16 floats) and the numeric features' reductions in whichever form fits the
    workgroup's LDS (512-sample units up to dim ~40, fm_numeric_partial_kernel beyond) -- logits and every gradient against
    the float64-free oracle restatement of the model."""
    from oracle import torch_ref as R
    from recbox_amd.ranking.pytorch.models import FM
    B = 1500
    fm, X, y = _criteo_like(B, CRITEO_SMALL_VOCABS + [20000], dim, seed=300 + dim, zipf=(dim == 32))
    ref = R.RefFMModel(fm, dim)
    g = torch.Generator().manual_seed(dim)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for m in ref.modules():
            if isinstance(m, torch.nn.Embedding) and m.padding_idx is not None:
                m.weight[m.padding_idx].zero_()
    fused = FM(fm, dim, fused=True)
    fused.load_state_dict(ref.state_dict())
    fused.cuda()
    outs = []
    for model, inp, lab in ((ref, X, y), (fused, _cuda(X), y.cuda())):
        logit = model.logits(inp) if hasattr(model, "logits") else model(inp)
        torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), lab, reduction="sum").backward()
        outs.append(logit)
    assert_close(outs[1], outs[0], TOL * max(1.0, float(outs[0].detach().abs().max())), "fused logit vs oracle")
    for (n0, p0), (_, p1) in zip(ref.named_parameters(), fused.named_parameters()):
        assert_close(p1.grad, p0.grad, TOL * max(1.0, float(p0.grad.abs().max())), "fused grad " + n0)


def _fm_pair(seed, vocabs=None, dim=16):
    from recbox_amd.ranking.pytorch.models import FM
    fm, _, _ = _criteo_like(4, vocabs or (CRITEO_SMALL_VOCABS + [70000]), dim, seed=seed)
    a, b = FM(fm, dim, fused=True).cuda(), FM(fm, dim, fused=True).cuda()
    with torch.no_grad():
        for p in a.parameters():
            p.normal_(0, 0.1)
    b.load_state_dict(a.state_dict())
    return fm, a, b


def _bce_step(model, X, y):
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(model.logits(X)), y, reduction="mean")
    loss.backward()
    return loss


def test_ops_backward_equals_loss_backward_bit_for_bit():
    """``ops.backward(loss)`` hands autograd the persistent constant 1 as the loss's gradient (no ones_like fill, and the
    fused sigmoid + BCE returns dL/dlogit unscaled): the same gradients as ``loss.backward()``, bit for bit."""
    from recbox_amd import ops
    from recbox_amd.ranking.pytorch.torch_utils import get_loss
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, a, b = _fm_pair(47, vocabs)
    _, X, y = _criteo_like(777, vocabs, 16, seed=5)
    Xc, yc = _cuda(X), y.cuda()
    loss_fn = get_loss("binary_crossentropy")
    for model, how in ((a, "autograd"), (b, "ops")):
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(Xc)["y_pred"], yc, reduction="mean")
        if how == "ops":
            ops.backward(loss)
        else:
            loss.backward()
    for (n, p0), (_, p1) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p0.grad, p1.grad), n


def test_embedding_regulariser_under_persistent_gradients_and_graph_replay():
    """the reference's own harness adds an Lp term over every "embedding_layer" parameter to the loss
    (ranking_model.py:72-87: ``add_regularization``) -- its backward writes ALL rows of a table's gradient, in place, into
    whatever arrived first.  With persistent gradient buffers (GraphedStep's default) the op's tensor arrives first and
    aliases the buffer of which only the looked-up rows are ever cleared: the guard (ops._GradPool: a post-accumulate hook
    per pooled parameter) notices the second writer in the first warm-up step and clears the buffer in full from then on.
    Replays of the captured step over rotating batches == eager steps with fresh gradients, never stale rows."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, fresh, reuse = _fm_pair(43, vocabs)
    lam = 0.03
    old, old_check = ops.config.reuse_grad_buffers, ops.config.check_ids
    ops.config.check_ids = False
    try:
        def step_of(model, Xc, yc):
            params = list(model.parameters())
            emb = [p for n, p in model.named_parameters() if "embedding_layer" in n]

            def fn():
                for p in params:
                    p.grad = None
                loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(model.logits(Xc)), yc, reduction="mean")
                reg = sum((p ** 2).sum() for p in emb)                     # emb_reg with p = 2 (ranking_model.py:82-84)
                (loss + lam / 2 * reg).backward()
                return loss
            return fn

        batches = []
        for k in range(3):
            _, X, y = _criteo_like(600, vocabs, 16, seed=90 + k, zipf=bool(k % 2))
            batches.append((_cuda(X), y.cuda()))
        ops.config.reuse_grad_buffers = False
        want = []
        for Xc, yc in batches:
            step_of(fresh, Xc, yc)()
            want.append([p.grad.clone() for p in fresh.parameters()])
        graphs = []
        for Xc, yc in batches:
            graphs.append(GraphedStep(step_of(reuse, Xc, yc), warmup=2, reuse_grads=True, params=list(reuse.parameters()),
                                      pool=graphs[0].pool() if graphs else None))
        for rnd in range(2):
            for k in (0, 1, 2, 1, 0):
                graphs[k]()
                torch.cuda.synchronize()
                worst = 0.0
                for (n, p), g in zip(reuse.named_parameters(), want[k]):
                    err = float((p.grad - g).abs().max())
                    worst = max(worst, err)
                    assert err <= 1e-6, "round %d batch %d: %s differs by %g (stale rows?)" % (rnd, k, n, err)
        print("regulariser under persistent gradients: max |grad - fresh| = %.3g" % worst)
        # the other order of arrival: a regulariser term built BEFORE the forward runs its backward AFTER the lookup's (the
        # engine runs later-built nodes first), so the op's tensor arrives first -- it aliases the persistent buffer -- and
        # the dense Lp gradient is then added INTO the buffer.  Two such steps, then clean ones: no row of an earlier
        # step may survive
        fm2, fresh2, reuse2 = _fm_pair(44, vocabs)

        def reg_first_step(model, Xc, yc, with_reg):
            model.zero_grad(set_to_none=True)
            emb = [p for n, p in model.named_parameters() if "embedding_layer" in n]
            reg = sum((p ** 2).sum() for p in emb) if with_reg else 0.0
            loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(model.logits(Xc)), yc, reduction="mean")
            (loss + lam / 2 * reg).backward()

        for k, (Xc, yc) in enumerate(batches + batches[:2]):
            ops.config.reuse_grad_buffers = False
            reg_first_step(fresh2, Xc, yc, k < 2)
            ops.config.reuse_grad_buffers = True
            reg_first_step(reuse2, Xc, yc, k < 2)
            for (n, p0), (_, p1) in zip(fresh2.named_parameters(), reuse2.named_parameters()):
                assert float((p1.grad - p0.grad).abs().max()) <= 1e-6, "step %d: %s keeps rows of an earlier step" % (k, n)
    finally:
        ops.config.reuse_grad_buffers, ops.config.check_ids = old, old_check


def test_reuse_grad_buffers_equals_fresh_grads():
    """ops.config.reuse_grad_buffers (persistent dense grads, rbx_fm_rezero clears only the rows the previous step wrote)
    leaves bit-identical gradients to the zero-filled path over steps with different ids and batch sizes (smaller,
    larger than ever before, back), after a training forward without backward, and when two forwards precede the
    backwards; a step that does not start from p.grad None is refused."""
    from recbox_amd import ops
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, fresh, reuse = _fm_pair(41, vocabs)
    old = ops.config.reuse_grad_buffers
    try:
        for k, B in enumerate([513, 513, 100, 2000, 1, 513]):
            _, X, y = _criteo_like(B, vocabs, 16, seed=50 + k, zipf=bool(k % 2))
            Xc, yc = _cuda(X), y.cuda()
            ops.config.reuse_grad_buffers = False
            _bce_step(fresh, Xc, yc)
            ops.config.reuse_grad_buffers = True
            if k == 3:
                reuse.logits(Xc)                             # a training forward whose backward never runs
            _bce_step(reuse, Xc, yc)
            for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "step %d: %s" % (k, n)
        # two forwards, then both backwards: the first one's sorted ids were overwritten -> it re-sorts, fresh grads
        _, X1, y1 = _criteo_like(300, vocabs, 16, seed=71)
        _, X2, y2 = _criteo_like(300, vocabs, 16, seed=72)
        want = []
        ops.config.reuse_grad_buffers = False
        for X, y in ((X1, y1), (X2, y2)):
            _bce_step(fresh, _cuda(X), y.cuda())
            want.append([p.grad.clone() for p in fresh.parameters()])
        ops.config.reuse_grad_buffers = True
        reuse.zero_grad(set_to_none=True)
        l1 = reuse.logits(_cuda(X1))
        l2 = reuse.logits(_cuda(X2))
        for logit, y, w in ((l2, y2, want[1]), (l1, y1, want[0])):
            reuse.zero_grad(set_to_none=True)
            torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), y.cuda(), reduction="mean").backward()
            for p, g in zip(reuse.parameters(), w):
                assert torch.equal(p.grad, g)
        # two forwards feeding ONE backward pass (autograd sums their gradients in place): neither may use the persistent
        # buffer; the step after it, with a fresh batch, is still exact
        reuse.zero_grad(set_to_none=True)
        fresh.zero_grad(set_to_none=True)
        for model in (fresh, reuse):
            ops.config.reuse_grad_buffers = model is reuse
            la, lb = model.logits(_cuda(X1)), model.logits(_cuda(X2))
            (torch.nn.functional.binary_cross_entropy(torch.sigmoid(la), y1.cuda()) +
             torch.nn.functional.binary_cross_entropy(torch.sigmoid(lb), y2.cuda())).backward()
        for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
            assert_close(p1.grad, p0.grad, 1e-6, "two forwards, one backward: " + n)
        _, X3, y3 = _criteo_like(400, vocabs, 16, seed=73)
        for model in (fresh, reuse):
            ops.config.reuse_grad_buffers = model is reuse
            for _ in range(2):
                _bce_step(model, _cuda(X3), y3.cuda())
        for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
            assert torch.equal(p1.grad, p0.grad), "after the combined backward: " + n
        with pytest.raises(RuntimeError, match="reuse_grad_buffers"):
            logit = reuse.logits(_cuda(X1))                  # p.grad still set from the step above
            logit.sum().backward()
    finally:
        ops.config.reuse_grad_buffers = old


@pytest.mark.parametrize("kind", ["fm_dim7", "fm_dim4", "lr_only", "frozen_table"])
def test_reuse_grad_buffers_other_shapes(kind):
    """The persistent-gradient path where the kernels take other branches: scalar (D = 7) and one-float4 (D = 4) rows,
    the LR-only body (LogisticRegression: dim-1 tables, no embedding part), and a model with a frozen table -- several
    steps each (the row re-zero only runs from the second step on), bit-identical to fresh gradients."""
    from recbox_amd import ops
    vocabs = [37, 5, 3001, 211, 70000]
    if kind == "lr_only":
        L = _layers()
        fm, _, _ = _criteo_like(4, vocabs, 16, seed=5)
        fresh, reuse = L.LogisticRegression(fm, use_bias=True).cuda(), L.LogisticRegression(fm, use_bias=True).cuda()
        with torch.no_grad():
            for p in fresh.parameters():
                p.normal_(0, 0.1)
        reuse.load_state_dict(fresh.state_dict())
        run = lambda m, X: m(X)
    else:
        dim = {"fm_dim7": 7, "fm_dim4": 4}.get(kind, 16)
        fm, fresh, reuse = _fm_pair(77, vocabs, dim=dim)
        if kind == "frozen_table":
            for m in (fresh, reuse):
                m.embedding_layer.embedding_layer.embedding_layers["C3"].weight.requires_grad_(False)
        run = lambda m, X: m.logits(X)
    old = ops.config.reuse_grad_buffers
    try:
        for k, B in enumerate([300, 300, 1000, 77]):
            _, X, y = _criteo_like(B, vocabs, 16, seed=90 + k, zipf=bool(k % 2))
            Xc, yc = _cuda(X), y.cuda()
            for model, flag in ((fresh, False), (reuse, True)):
                ops.config.reuse_grad_buffers = flag
                model.zero_grad(set_to_none=True)
                torch.nn.functional.binary_cross_entropy(torch.sigmoid(run(model, Xc)), yc, reduction="mean").backward()
            for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
                if p0.grad is None:
                    assert p1.grad is None, n
                else:
                    assert torch.equal(p1.grad, p0.grad), "%s step %d: %s" % (kind, k, n)
    finally:
        ops.config.reuse_grad_buffers = old


@pytest.mark.parametrize("model_kind", ["fm_layers", "sequence_pooling"])
def test_reuse_grad_buffers_all_covers_the_generic_lookup(model_kind):
    """ops.config.reuse_grad_buffers = "all": the generic lookup (FeatureEmbedding of the layer-composed FM; a matching
    EmbeddingLayer with a shared table, a pad row and mean-pooled histories) keeps persistent gradients too -- several
    steps with different ids and batch sizes, bit-identical to fresh zero-filled gradients."""
    from recbox_amd import ops
    if model_kind == "fm_layers":
        from recbox_amd.ranking.pytorch.models import FM
        vocabs = [37, 5, 3001, 211, 70000]
        fm, _, _ = _criteo_like(4, vocabs, 16, seed=7)
        fresh, reuse = FM(fm, 16, fused=False).cuda(), FM(fm, 16, fused=False).cuda()
        batches = [_criteo_like(B, vocabs, 16, seed=120 + k, zipf=bool(k % 2))[1:] for k, B in enumerate([300, 300, 900, 50])]
        run = lambda m, X: m.logits(X)
    else:
        from test_gpu_matching import _matching_model_and_batches
        fresh, reuse, batches, run = _matching_model_and_batches()
    with torch.no_grad():
        for p in fresh.parameters():
            p.normal_(0, 0.1)
    reuse.load_state_dict(fresh.state_dict())
    old = ops.config.reuse_grad_buffers
    try:
        for k, (X, y) in enumerate(batches):
            Xc, yc = _cuda(X), y.cuda()
            for model, flag in ((fresh, False), (reuse, "all")):
                ops.config.reuse_grad_buffers = flag
                model.zero_grad(set_to_none=True)
                out = run(model, Xc)
                torch.nn.functional.binary_cross_entropy(torch.sigmoid(out.reshape(yc.shape)), yc, reduction="mean").backward()
            for (n, p0), (_, p1) in zip(fresh.named_parameters(), reuse.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "%s step %d: %s" % (model_kind, k, n)
    finally:
        ops.config.reuse_grad_buffers = old


def test_lookups_over_the_same_ids_share_one_sort():
    """FeatureEmbedding and LogisticRegression of the layer-composed FM look up the same id columns with the same table
    layout: the second lookup copies the first one's sorted (row, sample) pairs (rbx_sort_share) -- gradients bit-identical
    to sorting twice; a lookup over other ids, or after an in-place write to an id tensor, sorts for itself."""
    from recbox_amd import ops
    from recbox_amd.ranking.pytorch.models import FM
    vocabs = [37, 5, 3001, 211, 70000]
    fm, X, y = _criteo_like(700, vocabs, 16, seed=9, zipf=True)
    a, b = FM(fm, 16, fused=False).cuda(), FM(fm, 16, fused=False).cuda()
    with torch.no_grad():
        for p in a.parameters():
            p.normal_(0, 0.1)
    b.load_state_dict(a.state_dict())
    Xc, yc = _cuda(X), y.cuda()
    old = ops.config.share_sorts
    try:
        ops.config.share_sorts = False
        _bce_step(a, Xc, yc)
        ops.config.share_sorts = True
        before = dict(ops.sort_counts)
        _bce_step(b, Xc, yc)
        assert ops.sort_counts["sorted"] - before["sorted"] == 1 and ops.sort_counts["shared"] - before["shared"] == 1
        for (n, p0), (_, p1) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p1.grad, p0.grad), n
        # other ids in one of the columns between the two lookups' tensors: no sharing, still right
        # (the memo would also serve a NEW step over the unchanged batch -- same ids, same pairs --, so start clean)
        ops._sort_memo.clear()
        before = dict(ops.sort_counts)
        b.zero_grad(set_to_none=True)
        emb = b.embedding_layer(Xc)
        X2 = type(Xc)((k, v.clone()) for k, v in Xc.items())
        X2["C3"] = X2["C3"].flip(0).contiguous()
        lr_out = b.fm.lr_layer(X2)
        (emb.sum() + lr_out.sum()).backward()
        assert ops.sort_counts["shared"] == before["shared"] and ops.sort_counts["sorted"] - before["sorted"] == 2
        # an in-place write to an id tensor after the first lookup invalidates the memo
        ops._sort_memo.clear()
        before = dict(ops.sort_counts)
        b.zero_grad(set_to_none=True)
        emb = b.embedding_layer(Xc)
        Xc["C1"].copy_(Xc["C1"].flip(0))
        lr_out = b.fm.lr_layer(Xc)
        assert ops.sort_counts["shared"] == before["shared"]
    finally:
        ops.config.share_sorts = old


@pytest.mark.parametrize("reuse", [False, "all"])
def test_graphed_layer_path_replays_follow_the_batch(reuse):
    """The layer-composed FM (two lookups, the second one sharing the first one's id sort) inside a hipGraph: after the
    static batch is refilled, a replay must sort the NEW ids (a shared copy baked into the graph may only come from the
    sort captured just before it) -- loss and gradients of the eager step on that batch, three batches in a row."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    from recbox_amd.ranking.pytorch.models import FM
    vocabs = [37, 5, 3001, 211, 70000]
    fm, X0, y0 = _criteo_like(600, vocabs, 16, seed=70)
    eager, graphed = FM(fm, 16, fused=False).cuda(), FM(fm, 16, fused=False).cuda()
    with torch.no_grad():
        for p in eager.parameters():
            p.normal_(0, 0.1)
    graphed.load_state_dict(eager.state_dict())
    Xs, ys = _cuda(X0), y0.cuda()
    old = (ops.config.check_ids, ops.config.reuse_grad_buffers)
    ops.config.check_ids = False
    try:
        step = GraphedStep(lambda: _bce_step(graphed, Xs, ys), warmup=3, reuse_grads=reuse)
        for k in range(3):
            _, X, y = _criteo_like(600, vocabs, 16, seed=71 + k, zipf=(k == 1))
            for n in Xs:
                Xs[n].copy_(X[n])
            ys.copy_(y)
            loss = step()
            ops.config.reuse_grad_buffers = False
            want = _bce_step(eager, _cuda(X), y.cuda())
            torch.cuda.synchronize()
            assert_close(loss.reshape(1), want.reshape(1), 1e-6, "loss")
            for (n, p0), (_, p1) in zip(eager.named_parameters(), graphed.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "replay %d: %s" % (k, n)
    finally:
        ops.config.check_ids, ops.config.reuse_grad_buffers = old


def test_graphed_step_replays_equal_eager_steps():
    """GraphedStep (whole step in one hipGraph, persistent gradients re-zeroed by row): after refilling the static batch,
    a replay leaves the loss and gradients of the eager step on that batch -- three different batches in a row."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, eager, graphed = _fm_pair(43, vocabs)
    B = 777
    _, X0, y0 = _criteo_like(B, vocabs, 16, seed=60)
    Xs, ys = _cuda(X0), y0.cuda()
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        step = GraphedStep(lambda: _bce_step(graphed, Xs, ys), warmup=3)
        for k in range(3):
            _, X, y = _criteo_like(B, vocabs, 16, seed=61 + k, zipf=(k == 1))
            for n in Xs:
                Xs[n].copy_(X[n])
            ys.copy_(y)
            loss = step()
            want = _bce_step(eager, _cuda(X), y.cuda())
            torch.cuda.synchronize()
            assert_close(loss.reshape(1), want.reshape(1), 1e-6, "loss")
            for (n, p0), (_, p1) in zip(eager.named_parameters(), graphed.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "replay %d: %s" % (k, n)
    finally:
        ops.config.check_ids = old


def test_one_graph_per_resident_batch_replayed_in_rotation():
    """bench.py's rotation: K batches stay on the device, each with a captured step of its own over ONE model (shared
    intermediate memory, one persistent gradient buffer); replayed in any order, a call leaves the loss and EVERY p.grad
    (the small dense ones live in per-graph memory: GraphedStep(params=...) re-points them) of the eager step on its batch."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, eager, graphed = _fm_pair(47, vocabs)
    B = 513
    data = []
    for k in range(3):
        _, X, y = _criteo_like(B, vocabs, 16, seed=90 + k, zipf=(k == 1))
        data.append((_cuda(X), y.cuda()))
    params = list(graphed.parameters())
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        steps = []
        for Xk, yk in data:
            steps.append(GraphedStep((lambda Xk=Xk, yk=yk: _bce_step(graphed, Xk, yk)), warmup=2, params=params,
                                     pool=steps[0].pool() if steps else None))
        for k in (0, 1, 2, 1, 1, 0, 2):
            loss = steps[k]()
            want = _bce_step(eager, data[k][0], data[k][1])
            torch.cuda.synchronize()
            assert_close(loss.reshape(1), want.reshape(1), 1e-6, "loss")
            for (n, p0), (_, p1) in zip(eager.named_parameters(), graphed.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "graph %d: %s" % (k, n)
    finally:
        ops.config.check_ids = old


def test_several_steps_in_one_captured_graph_leave_the_last_steps_gradients():
    """bench.py --steps-per-graph (round 6): S consecutive steps, one per resident batch, captured in ONE hipGraph (a replay pays
    the graph-to-graph latency once per S steps).  Every step inside the graph ends with both backward chains joined, the next
    one clears exactly the rows its predecessor stored: after a replay the loss and every p.grad are those of the eager step on
    the LAST batch of the group, whichever group ran before it -- two groups of two and of three batches, replayed in any order."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, eager, graphed = _fm_pair(59, vocabs)
    B = 600
    data = []
    for k in range(5):
        _, X, y = _criteo_like(B, vocabs, 16, seed=140 + k, zipf=(k in (1, 4)))
        data.append((_cuda(X), y.cuda()))
    params = list(graphed.parameters())
    groups = [(0, 1), (2, 3, 4)]
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        steps = []
        for grp in groups:
            def fn(grp=grp):
                out = None
                for k in grp:
                    out = _bce_step(graphed, data[k][0], data[k][1])
                return out
            steps.append(GraphedStep(fn, warmup=2, params=params, pool=steps[0].pool() if steps else None))
        for g in (0, 1, 1, 0, 1):
            loss = steps[g]()
            k = groups[g][-1]
            want = _bce_step(eager, data[k][0], data[k][1])
            torch.cuda.synchronize()
            assert_close(loss.reshape(1), want.reshape(1), 1e-6, "loss")
            for (n, p0), (_, p1) in zip(eager.named_parameters(), graphed.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "group %d: %s" % (g, n)
    finally:
        ops.config.check_ids = old


def test_id_sort_made_one_step_ahead_eager_and_captured():
    """FM.presort / forward(presorted=...): the id sort of batch i + 1 runs on the side stream while step i runs (what a loop
    whose loader is one batch ahead does), with persistent gradients.  Eagerly the gradient pool remembers whose rows to
    clear; captured steps (one per resident batch, replayed in ring order) are told through ``previous``.  Every step leaves
    the loss and the gradients of the plain eager step on its batch."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, eager, fast = _fm_pair(53, vocabs)
    B, K = 640, 3
    data = []
    for k in range(K):
        _, X, y = _criteo_like(B, vocabs, 16, seed=120 + k, zipf=(k == 2))
        data.append((_cuda(X), y.cuda()))
    params = list(fast.parameters())
    side = ops.side_stream(torch.device("cuda"))

    def make(k, handles):
        Xk, yk = data[k]

        def one_step():
            for p in params:
                p.grad = None
            cur = torch.cuda.current_stream()
            start = cur.record_event()
            prob = fast(Xk, presorted=handles[k])["y_pred"]
            side.wait_event(start)
            with torch.cuda.stream(side):
                fast.presort(data[(k + 1) % K][0], into=handles[(k + 1) % K])
            loss = ops.binary_cross_entropy(prob, yk)
            loss.backward()
            cur.wait_stream(side)
            return loss
        return one_step

    def check(k, loss):
        eager.zero_grad(set_to_none=True)                      # the plain step: same model code, its own sort inside the step
        want = ops.binary_cross_entropy(eager(data[k][0])["y_pred"], data[k][1])
        want.backward()
        torch.cuda.synchronize()
        assert_close(loss.reshape(1), want.reshape(1), 1e-6, "loss of batch %d" % k)
        for (n, p0), (_, p1) in zip(eager.named_parameters(), fast.named_parameters()):
            assert torch.equal(p1.grad, p0.grad), "batch %d: %s" % (k, n)

    old = (ops.config.check_ids, ops.config.reuse_grad_buffers)
    ops.config.check_ids = False
    try:
        ops.config.reuse_grad_buffers = True
        handles = [fast.presort(Xk) for Xk, _ in data]
        steps = [make(k, handles) for k in range(K)]
        for k in (0, 1, 2, 0, 1):                                  # eager: any order that follows the ring
            check(k, steps[k]())
        for k in range(K):
            handles[k].previous = handles[k - 1]
        graphs = []
        for k in range(K):
            graphs.append(GraphedStep(steps[k], warmup=2, params=params, pool=graphs[0].pool() if graphs else None))
        ops.config.reuse_grad_buffers = False                      # (the eager reference steps use fresh gradients)
        for k in (0, 1, 2, 0, 1, 2, 0):
            check(k, graphs[k]())
    finally:
        ops.config.check_ids, ops.config.reuse_grad_buffers = old


def test_bench_configuration_graph_replays_equal_fresh_eager_steps():
    """BASELINE.json configs[1] exactly as bench.py runs it (26 Criteo-sized tables + 13 numeric features, D = 16,
    B = 65 536, ids as float64 columns, hipGraph replay with persistent gradients): two replays on two different batches
    leave bit-identical loss and gradients to eager steps that zero-fill fresh gradients."""
    import bench
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    from recbox_amd.ranking.pytorch.models import FM
    from recbox_amd.ranking.pytorch.torch_utils import get_loss
    dev = torch.device("cuda", 0)
    fmw = bench.CriteoFeatureMap(16)
    eager, graphed = FM(fmw.fm, 16, fused=True).to(dev), FM(fmw.fm, 16, fused=True).to(dev)
    bench.init_weights(eager)
    graphed.load_state_dict(eager.state_dict())
    loss_fn = get_loss("binary_crossentropy")
    static = bench.synthetic_batch(65536, 1, "uniform", dev)
    Xs, ys = bench.slice_inputs(fmw.fm, static)
    ys = ys.clone()

    def step(model, X, y):
        for p in model.parameters():
            p.grad = None
        loss = loss_fn(model(X)["y_pred"], y, reduction="mean")
        loss.backward()
        return loss

    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        replay = GraphedStep(lambda: step(graphed, Xs, ys), warmup=3)
        for seed, dist in ((2, "uniform"), (3, "zipf")):
            batch = bench.synthetic_batch(65536, seed, dist, dev)
            static.copy_(batch)
            ys.copy_(bench.slice_inputs(fmw.fm, batch)[1])
            loss = replay()
            X, y = bench.slice_inputs(fmw.fm, batch)
            want = step(eager, X, y)
            torch.cuda.synchronize()
            assert torch.equal(loss, want), (float(loss), float(want))
            for (n, p0), (_, p1) in zip(eager.named_parameters(), graphed.named_parameters()):
                assert torch.equal(p1.grad, p0.grad), "%s ids: %s" % (dist, n)
    finally:
        ops.config.check_ids = old


def test_bench_configuration_matches_cpu_oracle_at_full_size():
    """The bench step itself (B = 65 536, Criteo-sized tables, summed BCE so that gradients are O(1)) against the CPU
    oracle (the reference's op sequence on ATen CPU kernels): logits within 1e-4, every gradient within 1e-4 of the
    largest magnitude of its tensor (rows of the 3-row table sum 21 845 terms)."""
    import bench
    from oracle import torch_ref as R
    from recbox_amd.ranking.pytorch.models import FM
    fmw = bench.CriteoFeatureMap(16)
    ref = R.RefFMModel(fmw.fm, 16)
    bench.init_weights(ref)
    dut = FM(fmw.fm, 16, fused=True)
    dut.load_state_dict(ref.state_dict())
    dut.cuda()
    batch = bench.synthetic_batch(65536, 5, "uniform", "cpu")
    X, y = bench.slice_inputs(fmw.fm, batch)
    Xc, yc = bench.slice_inputs(fmw.fm, batch.cuda())
    torch.set_num_threads(min(16, torch.get_num_threads()))
    lr = ref(X)
    torch.nn.functional.binary_cross_entropy(torch.sigmoid(lr), y, reduction="sum").backward()
    ld = dut.logits(Xc)
    torch.nn.functional.binary_cross_entropy(torch.sigmoid(ld), yc, reduction="sum").backward()
    assert_close(ld, lr, TOL, "logit")
    for (n, p0), (_, p1) in zip(ref.named_parameters(), dut.named_parameters()):
        scale = max(1.0, float(p0.grad.abs().max()))
        assert_close(p1.grad / scale, p0.grad / scale, TOL, "grad " + n)


def test_logistic_regression_fused_and_frozen_tables():
    L = _layers()
    from oracle import torch_ref as R
    fm, X, y = _criteo_like(300, CRITEO_SMALL_VOCABS[:6], 8, seed=5)
    ref = R.RefLogisticRegression(fm)
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, 0.1)
    dut = L.LogisticRegression(fm)
    dut.load_state_dict(ref.state_dict())
    dut.cuda()
    frozen = "embedding_layer.embedding_layer.embedding_layers.C2.weight"
    for m in (ref, dut):
        dict(m.named_parameters())[frozen].requires_grad_(False)
    o0 = ref(X)
    o1 = dut(_cuda(X))
    assert_close(o1, o0, TOL)
    (o0 * y).sum().backward()
    (o1 * y.cuda()).sum().backward()
    for (n, p0), (_, p1) in zip(ref.named_parameters(), dut.named_parameters()):
        if n == frozen:
            assert p1.grad is None
        else:
            assert_close(p1.grad, p0.grad, TOL, n)


@pytest.mark.parametrize("factor", [None, 1.5])
def test_sharded_fm_equals_fm_on_one_gpu(factor):
    """ShardedFM (remote rows through the exchange path, here a world of one; exact and padded exchange)
    == FM: logits and every gradient."""
    from recbox_amd.ranking.pytorch.models import FM, ShardedFM
    vocabs = [50, 7, 400, 31, 300, 9]
    fm, X, y = _criteo_like(513, vocabs, 16, seed=21, zipf=True)
    ref = FM(fm, 16).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, 0.1)
    dut = ShardedFM(fm, 16, shard_min_vocab=300, capacity_factor=factor).cuda()   # C3 and C5 become row-sharded
    assert dut.sharded_names == ["C3", "C5"]
    sd = ref.state_dict()
    e_pre = "embedding_layer.embedding_layer.embedding_layers.%s.weight"
    l_pre = "fm.lr_layer.embedding_layer.embedding_layer.embedding_layers.%s.weight"
    dut.load_state_dict({k: v for k, v in sd.items() if not any(("." + n + ".") in k for n in dut.sharded_names)},
                        strict=False)
    dut.tables.load_full_tables([sd[e_pre % n] for n in dut.sharded_names], [sd[l_pre % n] for n in dut.sharded_names])
    Xc, yc = _cuda(X), y.cuda()
    outs = []
    for m in (ref, dut):
        logit = m.logits(Xc)
        torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), yc, reduction="sum").backward()
        outs.append(logit)
    assert_close(outs[1], outs[0], 1e-5, "logit")
    ref_g = dict((n, p.grad) for n, p in ref.named_parameters())
    for n, p in dut.named_parameters():
        if n.startswith("tables."):
            continue
        assert_close(p.grad, ref_g[n], 1e-4 * 8, n)
    for t, name in enumerate(dut.sharded_names):
        sl, owned = dut.tables.local_rows_of(t)
        got = dut.tables.weight.grad[sl]
        assert_close(got[1:, :16], ref_g[e_pre % name][owned.cuda()][1:], 1e-4 * 8, "shard emb " + name)
        assert_close(got[1:, 16], ref_g[l_pre % name][owned.cuda()][1:, 0], 1e-4 * 8, "shard lr " + name)


@pytest.mark.parametrize("world,cap", [(1, 4096), (3, 1024), (8, 320), (8, 64)])
def test_route_kernel_equals_torch_restatement(world, cap):
    """rbx_route (stable counting sort by owner, wire slots, dump slot + overflow byte) == the torch restatement
    padded_route (tests/test_distributed_gloo.py), bit for bit, including a capacity that overflows."""
    from recbox_amd import ops
    from test_distributed_gloo import padded_route
    g = torch.Generator().manual_seed(5)
    B, T = 1000, 3
    vocabs = [5000, 77, 123456]
    ids = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], dim=1)
    ids[::7, 0] = 8                                              # a hot id: one owner gets far more than its share
    counts = torch.tensor([[(v - r + world - 1) // world for v in vocabs] for r in range(world)])
    base = torch.zeros_like(counts)
    base[:, 1:] = counts.cumsum(1)[:, :-1]
    owner = ids % world
    row = base[owner, torch.arange(T)[None, :].expand_as(ids)] + ids // world
    of_ref = torch.zeros((), dtype=torch.bool)
    slot_ref, send_ref = padded_route(owner.reshape(-1), row.reshape(-1), cap, world, of_ref)
    of = torch.zeros((), dtype=torch.bool, device="cuda")
    send, slot = ops.route(ids.cuda(), world, cap, base.cuda(), of)
    torch.cuda.synchronize()
    assert torch.equal(slot.cpu().reshape(-1).long(), slot_ref)
    assert torch.equal(send.cpu(), send_ref)
    assert bool(of.cpu()) == bool(of_ref)
    of32 = torch.zeros((), dtype=torch.bool, device="cuda")              # the same with 32-bit row numbers on the wire
    send32, slot32 = ops.route(ids.cuda(), world, cap, base.cuda(), of32, wire=torch.int32)
    torch.cuda.synchronize()
    assert send32.dtype == torch.int32 and torch.equal(send32.cpu().long(), send_ref)
    assert torch.equal(slot32, slot) and bool(of32.cpu()) == bool(of_ref)
    if cap == 64:
        assert bool(of_ref)                                      # this case is meant to overflow


@pytest.mark.parametrize("graphs", [False, True])
def test_sharded_fm_step_pieces(graphs):
    """ShardedFMStep (route / serve / local / settle pieces with the collectives between them, eager and as
    hipGraph replays) leaves the same loss and gradients as the autograd step of ShardedFM."""
    from recbox_amd import ops
    from recbox_amd.graph import ShardedFMStep
    from recbox_amd.ranking.pytorch.models import ShardedFM
    vocabs = [50, 7, 400, 31, 300, 9]
    fm, X, y = _criteo_like(513, vocabs, 16, seed=23, zipf=True)
    ref = ShardedFM(fm, 16, shard_min_vocab=300, capacity_factor=1.5).cuda()
    dut = ShardedFM(fm, 16, shard_min_vocab=300, capacity_factor=1.5).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, 0.1)
    dut.load_state_dict(ref.state_dict())
    Xc, yc = _cuda(X), y.cuda()
    prob = ref(Xc)["y_pred"]
    loss_ref = torch.nn.functional.binary_cross_entropy(prob, yc, reduction="mean")
    loss_ref.backward()
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        step = ShardedFMStep(dut, Xc, yc, graphs=graphs)
        for _ in range(2):                       # replays are idempotent: grads are rebuilt, not accumulated
            loss = step()
    finally:
        ops.config.check_ids = old
    torch.cuda.synchronize()
    assert not bool(dut.tables.overflow)
    assert_close(loss.reshape(1), loss_ref.reshape(1), 1e-6, "loss")
    ref_g = dict((n, p.grad) for n, p in ref.named_parameters())
    for n, p in dut.named_parameters():
        assert p.grad is not None, n
        assert_close(p.grad, ref_g[n], 1e-6, n)


@pytest.mark.parametrize("graphs", [False, True])
def test_sharded_fm_step_clears_the_rows_of_the_previous_batch(graphs):
    """The shard's dense gradient is ONE buffer from step to step (HipLocalOps.persistent): after a step on batch A, a step on
    batch B leaves exactly B's gradient -- the rows only A touched are zero again."""
    from recbox_amd import ops
    from recbox_amd.graph import ShardedFMStep
    from recbox_amd.ranking.pytorch.models import ShardedFM
    vocabs = [50, 7, 4000, 31, 3000, 9]
    fm, XA, yA = _criteo_like(257, vocabs, 16, seed=31, zipf=False)
    _, XB, yB = _criteo_like(257, vocabs, 16, seed=32, zipf=False)
    ref = ShardedFM(fm, 16, shard_min_vocab=300, capacity_factor=1.5).cuda()
    dut = ShardedFM(fm, 16, shard_min_vocab=300, capacity_factor=1.5).cuda()
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, 0.1)
    dut.load_state_dict(ref.state_dict())
    Xc, yc = _cuda(XA), yA.cuda()
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        step = ShardedFMStep(dut, Xc, yc, graphs=graphs)
        step()
        grad_a = dut.tables.weight.grad.clone()
        for k, v in _cuda(XB).items():               # the step reads its inputs in place
            Xc[k].copy_(v)
        yc.copy_(yB.cuda())
        loss = step()
    finally:
        ops.config.check_ids = old
    torch.cuda.synchronize()
    prob = ref(_cuda(XB))["y_pred"]
    loss_ref = torch.nn.functional.binary_cross_entropy(prob, yB.cuda(), reduction="mean")
    loss_ref.backward()
    assert_close(loss.reshape(1), loss_ref.reshape(1), 1e-6, "loss")
    ref_g = dict((n, p.grad) for n, p in ref.named_parameters())
    for n, p in dut.named_parameters():
        assert_close(p.grad, ref_g[n], 1e-6, n)
    only_a = (grad_a.abs().sum(1) > 0) & (ref_g["tables.weight"].abs().sum(1) == 0)
    assert int(only_a.sum()) > 100                   # (the case is meant to have rows that only batch A touched)
    assert float(dut.tables.weight.grad[only_a].abs().max()) == 0.0


@pytest.mark.parametrize("version", ["v1", "v2"])
def test_cross_net_golden(version):
    """CrossNet / CrossNetV2 against the live-reference fixture: output, dx and every parameter grad."""
    L = _layers()
    fx = Fixture("cross_net")
    x = fx.tensors("in")["x"]
    R = fx.tensors("in")["R"]
    cls = L.CrossNet if version == "v1" else L.CrossNetV2
    net = load_params(cls(x.shape[1], 3), fx["p_" + version]).cuda()
    xc = x.cuda().requires_grad_(True)
    out = net(xc)
    assert_close(out, fx["out_" + version]["y"], TOL, "y")
    (out * R.cuda()).sum().backward()
    assert_close(xc.grad, fx["out_" + version]["dx"], TOL, "dx")
    assert_grads_close(net, fx["g_" + version], TOL)
    if version == "v1":                                       # the per-layer module is usable on its own, like the reference's
        one = net.cross_net[0]
        want = one.weight(xc.detach()) * xc.detach() + one.bias
        assert_close(one(xc.detach(), xc.detach()), want.detach(), TOL, "CrossInteraction")


@pytest.mark.parametrize("n", [1, 513, 65536])
def test_bce_mean_vs_torch(n):
    """rbx_bce_mean_fwd/bwd == F.binary_cross_entropy(reduction='mean') of torch CPU, including the -100 log clamp at
    p = 0 / 1 and the 1e-12 clamp of the backward."""
    from recbox_amd.ranking.pytorch.torch_utils import get_loss
    g = torch.Generator().manual_seed(n)
    p = torch.rand(n, 1, generator=g)
    y = (torch.rand(n, 1, generator=g) < 0.3).float()
    if n > 4:
        p[0], p[1], y[0], y[1] = 0.0, 1.0, 1.0, 0.0           # saturated wrong predictions: clamped, not inf
        p[2], p[3] = 1e-9, 1.0 - 1e-7
    pr = p.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy(pr, y, reduction="mean")
    (ref * 3.0).backward()
    pc = p.cuda().requires_grad_(True)
    out = get_loss("binary_crossentropy")(pc, y.cuda(), reduction="mean")
    (out * 3.0).backward()
    assert_close(out.reshape(1), ref.detach().reshape(1), 1e-5 * max(1.0, float(ref.detach())), "loss")
    scale = float(pr.grad.abs().max())
    assert float((pc.grad.cpu() - pr.grad).abs().max()) <= 1e-5 * max(1.0, scale)
    a = get_loss("bce")(pc.detach(), y.cuda())
    b = get_loss("bce")(pc.detach(), y.cuda())
    assert torch.equal(a, b)                                   # fixed-order reduction: bit-identical
    with pytest.raises(NotImplementedError):
        get_loss("no_such_loss")
    assert get_loss("mse_loss") is torch.nn.functional.mse_loss


@pytest.mark.parametrize("n", [1, 777, 65536])
def test_sigmoid_bce_vs_torch_chain(n):
    """rbx_sigmoid_bce_mean == sigmoid -> F.binary_cross_entropy(mean) -> backward of torch CPU: loss, y_pred and
    dL/dlogit (with a gradient scale), including saturated logits (log clamp at -100, 1e-12 clamp of the backward)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 1, generator=g) * 3
    y = (torch.rand(n, 1, generator=g) < 0.3).float()
    if n > 6:
        x[0], x[1], y[0], y[1] = -120.0, 120.0, 1.0, 0.0          # saturated and wrong
        x[2], x[3], y[2], y[3] = -30.0, 30.0, 0.0, 1.0            # saturated and right
        x[4], x[5] = 17.0, -17.0
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy(torch.sigmoid(xr), y, reduction="mean")
    (ref / 4.0).backward()
    loss, dx, prob = ops.sigmoid_bce(x.cuda(), y.cuda(), grad_scale=0.25, want_prob=True)
    assert_close(loss.reshape(1), ref.detach().reshape(1), 1e-6 * max(1.0, float(ref.detach())), "loss")
    assert_close(prob, torch.sigmoid(x), 1e-6, "y_pred")
    assert float((dx.cpu() - xr.grad).abs().max()) <= 1e-6 * max(1.0, float(xr.grad.abs().max())) / max(1, n) * n
    again = ops.sigmoid_bce(x.cuda(), y.cuda(), grad_scale=0.25)
    assert torch.equal(again[0], loss) and torch.equal(again[1], dx)


@pytest.mark.parametrize("kind", ["field_all", "field_each", "field_interaction"])
def test_bilinear_golden(kind):
    """BilinearInteraction / BilinearInteractionV2 against the live-reference fixture (the reference's
    two spellings agree to 1e-6 with each other): output, dx and dW."""
    L = _layers()
    fx = Fixture("bilinear")
    x, R = fx.tensors("in")["x"], fx.tensors("in")["R"]
    B, F, D = x.shape
    for ver, cls in (("v1", L.BilinearInteraction), ("v2", L.BilinearInteractionV2)):
        ref = fx["out_%s_%s" % (kind, ver)]
        layer = cls(F, D, bilinear_type=kind).cuda()
        assert tuple(layer.bilinear_W.shape) == ref["W"].shape
        with torch.no_grad():
            layer.bilinear_W.copy_(torch.from_numpy(ref["W"]))
        xc = x.cuda().requires_grad_(True)
        out = layer(xc)
        assert_close(out, ref["y"], TOL, "y")
        (out * R.cuda()).sum().backward()
        assert_close(xc.grad, ref["dx"], TOL, "dx")
        assert_close(layer.bilinear_W.grad, ref["dW"], TOL, "dW")
    with pytest.raises(NotImplementedError):
        L.BilinearInteractionV2(F, D, bilinear_type="nope")


def test_cross_net_mix_golden():
    """CrossNetMix (experts batched into three GEMMs per layer + rbx_cross) against the live-reference fixture: same
    state_dict keys, output, dx and every parameter gradient."""
    L = _layers()
    fx = Fixture("cross_net_mix")
    x, R = fx.tensors("in")["x"], fx.tensors("in")["R"]
    net = load_params(L.CrossNetMix(x.shape[1], layer_num=2, low_rank=6, num_experts=3), fx["p"]).cuda()
    xc = x.cuda().requires_grad_(True)
    out = net(xc)
    assert_close(out, fx["out"]["y"], TOL, "y")
    (out * R.cuda()).sum().backward()
    assert_close(xc.grad, fx["out"]["dx"], TOL, "dx")
    assert_grads_close(net, fx["g"], TOL)


@pytest.mark.parametrize("tag,order,bn", [("o2", 2, False), ("o5bn", 5, True)])
def test_interaction_machine_golden(tag, order, bn):
    L = _layers()
    fx = Fixture("im_hfm")
    x = fx.tensors("in")["x"]
    net = L.InteractionMachine(x.shape[2], order=order, batch_norm=bn)
    net.load_state_dict(fx.tensors("p_" + tag), strict=False)          # (running statistics are not part of the fixture)
    net.cuda().train()
    xc = x.cuda().requires_grad_(True)
    out = net(xc)
    assert_close(out, fx["out_" + tag]["y"], TOL, "y")
    (out * fx.tensors("out_" + tag)["R"].cuda()).sum().backward()
    assert_close(xc.grad, fx["out_" + tag]["dx"], TOL, "dx")
    assert_grads_close(net, fx["g_" + tag], TOL)


@pytest.mark.parametrize("kind", ["hadamard_product", "circular_convolution", "circular_correlation"])
def test_holographic_interaction_golden(kind):
    L = _layers()
    fx = Fixture("im_hfm")
    x = fx.tensors("in")["x"]
    layer = L.HolographicInteraction(x.shape[1], interaction_type=kind).cuda()
    xc = x.cuda().requires_grad_(True)
    out = layer(xc)
    assert_close(out, fx["hfm_" + kind]["y"], TOL, "y")
    (out * fx.tensors("hfm_" + kind)["R"].cuda()).sum().backward()
    assert_close(xc.grad, fx["hfm_" + kind]["dx"], TOL, "dx")
    with pytest.raises(ValueError):
        L.HolographicInteraction(3, interaction_type="nope")(xc)


@pytest.mark.parametrize("act", ["ReLU", "Sigmoid"])
def test_squeeze_excitation_golden(act):
    L = _layers()
    fx = Fixture("senet_din")
    x = fx.tensors("in")["x"]
    net = load_params(L.SqueezeExcitation(x.shape[1], reduction_ratio=3, excitation_activation=act), fx["p_se_" + act]).cuda()
    xc = x.cuda().requires_grad_(True)
    out = net(xc)
    assert_close(out, fx["out_se_" + act]["y"], TOL, "y")
    (out * fx.tensors("out_se_" + act)["R"].cuda()).sum().backward()
    assert_close(xc.grad, fx["out_se_" + act]["dx"], TOL, "dx")
    assert_grads_close(net, fx["g_se_" + act], TOL)


@pytest.mark.parametrize("tag,acts,soft", [("relu", "ReLU", False), ("dice_soft", "Dice", True)])
def test_din_attention_golden(tag, acts, soft):
    L = _layers()
    fx = Fixture("senet_din")
    t = fx.tensors("in")
    net = L.DIN_Attention(embedding_dim=t["target"].shape[1], attention_units=[12, 6], hidden_activations=acts,
                          use_softmax=soft)
    net.load_state_dict(fx.tensors("p_din_" + tag), strict=False)      # (running statistics are not in the fixture)
    net.cuda().train()
    tc, hc = t["target"].cuda().requires_grad_(True), t["hist"].cuda().requires_grad_(True)
    out = net(tc, hc, t["mask"].cuda())
    assert_close(out, fx["out_din_" + tag]["y"], TOL, "y")
    (out * fx.tensors("out_din_" + tag)["R"].cuda()).sum().backward()
    assert_close(tc.grad, fx["out_din_" + tag]["dt"], TOL, "d target")
    assert_close(hc.grad, fx["out_din_" + tag]["dh"], TOL, "d history")
    assert_grads_close(net, fx["g_din_" + tag], TOL)


def test_kmax_pooling_matches_its_definition():
    L = _layers()
    x = torch.randn(5, 9, 4, generator=torch.Generator().manual_seed(3)).cuda().requires_grad_(True)
    out = L.KMaxPooling(3, dim=1)(x)
    want = torch.stack([torch.stack([x[b, :, d][torch.sort(torch.topk(x[b, :, d], 3).indices).values] for d in range(4)], 1)
                        for b in range(5)])
    assert torch.equal(out, want)
    out.sum().backward()
    assert float(x.grad.sum()) == 5 * 3 * 4


def test_cin_golden():
    """CompressedInteractionNet against the live-reference fixture: same state_dict keys, output, dx, grads."""
    L = _layers()
    fx = Fixture("cin")
    x, R = fx.tensors("in")["x"], fx.tensors("in")["R"]
    net = load_params(L.CompressedInteractionNet(x.shape[1], [6, 3], output_dim=2), fx["p"]).cuda()
    xc = x.cuda().requires_grad_(True)
    out = net(xc)
    assert_close(out, fx["out"]["y"], TOL, "y")
    (out * R.cuda()).sum().backward()
    assert_close(xc.grad, fx["out"]["dx"], TOL, "dx")
    assert_grads_close(net, fx["g"], TOL)


def test_backward_is_deterministic_and_linear():
    """Full-size property checks (B = 65 536, 26 fields): two backward passes are
    bit-identical (no float atomics) and column sums of dW equal column sums of dY."""
    L = _layers()
    B = 65536
    vocabs = [1460, 583, 100000, 50000, 305, 24, 12517, 633, 3, 93145, 5683, 100000, 3194, 27, 14992, 100000, 10,
              5652, 2173, 4, 100000, 18, 15, 28618, 105, 14257]
    fm, X, y = _criteo_like(B, vocabs, 16, seed=11, zipf=True)
    layer = L.FeatureEmbedding(fm, 16).cuda()
    with torch.no_grad():
        for p in layer.parameters():
            p.normal_(0, 0.1)
    Xc = _cuda(X)
    R = torch.randn(B, 39, 16, device="cuda")
    grads = []
    for _ in range(2):
        layer.zero_grad(set_to_none=True)
        (layer(Xc) * R).sum().backward()
        grads.append([p.grad.clone() for p in layer.parameters()])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    names = [n for n, _ in layer.named_parameters()]
    for f, name in enumerate(fm.features):
        gidx = names.index("embedding_layer.embedding_layers.%s.weight" % name)
        gw = grads[0][gidx]
        if fm.features[name]["type"] == "numeric":
            want = (R[:, f, :].double() * Xc[name].double()[:, None]).sum(0)
            assert_close(gw.view(-1), want, 2e-2, name)
        else:
            keep = (Xc[name] != 0).double()[:, None]
            want = (R[:, f, :].double() * keep).sum(0)
            got = gw.double().sum(0)
            assert_close(got, want, 2e-2, name)
            assert float(gw[0].abs().max()) == 0.0            # padding row stays zero


def test_out_of_range_id_raises_index_error():
    L = _layers()
    f = OrderedDict([("c", {"source": "", "type": "categorical", "vocab_size": 5})])
    layer = L.FeatureEmbedding(_FM(f), 4).cuda()
    with pytest.raises(IndexError):
        layer({"c": torch.tensor([1, 7]).cuda()})


def test_cpu_tensor_is_rejected_not_silently_computed():
    L = _layers()
    f = OrderedDict([("c", {"source": "", "type": "categorical", "vocab_size": 5})])
    layer = L.FeatureEmbedding(_FM(f), 4).cuda()
    with pytest.raises(RuntimeError):
        layer({"c": torch.tensor([1, 2])})


def test_empty_batch():
    L = _layers()
    f = OrderedDict([("c", {"source": "", "type": "categorical", "vocab_size": 5})])
    layer = L.FeatureEmbedding(_FM(f), 4).cuda()
    out = layer({"c": torch.zeros(0, dtype=torch.long).cuda()})
    assert tuple(out.shape) == (0, 1, 4)   # the ranking flavour always stacks, even one feature


def test_fm_accepts_the_flat_batch_tensor_in_any_id_dtype():
    """f-3 wire format: the same model fed with the float64 [B, cols] batch, a float32 one and an int32 one
    (ids) gives the same logits; columns are read in place (views), sequence features take max_len columns."""
    from recbox_amd.ranking.features import FeatureMap
    from recbox_amd.ranking.pytorch.models import FM, inputs_from_batch
    fm = FeatureMap("t", "/tmp")
    fm.features = OrderedDict([("C1", {"source": "", "type": "categorical", "vocab_size": 50, "padding_idx": 0}),
                               ("C2", {"source": "", "type": "categorical", "vocab_size": 9, "padding_idx": 0})])
    fm.labels, fm.num_fields = ["y"], 2
    fm.set_column_index()
    model = FM(fm, 8).cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.normal_(0, 0.1)
    g = torch.Generator().manual_seed(3)
    ids = torch.stack([torch.randint(1, 50, (33,), generator=g), torch.randint(1, 9, (33,), generator=g),
                       torch.zeros(33, dtype=torch.long)], dim=1)
    outs = [model(ids.to(dt).cuda())["y_pred"] for dt in (torch.float64, torch.float32, torch.int32, torch.int64)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    X = inputs_from_batch(fm, ids.double().cuda())
    assert X["C1"].data_ptr() == ids.double().cuda().data_ptr() or X["C1"].stride(0) == 3   # a strided view, not a copy
    fm.features["S"] = {"source": "", "type": "sequence", "vocab_size": 9, "max_len": 2}
    fm.set_column_index()
    assert tuple(inputs_from_batch(fm, torch.zeros(4, 5).cuda())["S"].shape) == (4, 2)


@pytest.mark.parametrize("row_floats", [None, 20])
def test_fm_with_packed_tables_is_bit_identical(row_floats):
    """FM.pack_tables(): every (embedding, LR) table pair behind one packed [V, stride] storage (one 128-byte request per
    lookup in the fused forward).  Same parameter names, shapes and types; logits, loss and every dense gradient bit for bit
    equal to the unpacked model; strict state_dict round trip both ways; a hipGraph replay keeps working."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    from recbox_amd.ranking.pytorch.models import FM
    vocabs = CRITEO_SMALL_VOCABS + [70000]
    fm, X0, y0 = _criteo_like(1500, vocabs, 16, seed=5, zipf=True)
    plain, packed = FM(fm, 16).cuda(), FM(fm, 16).cuda()
    with torch.no_grad():
        for p in plain.parameters():
            p.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())) * 0.1)
    packed.load_state_dict(plain.state_dict())
    n = packed.pack_tables(row_floats)
    assert n == len(vocabs)
    el = packed.embedding_layer.embedding_layer.embedding_layers
    ll = packed.fm.lr_layer.embedding_layer.embedding_layer.embedding_layers
    stride = row_floats or 32
    assert type(el["C1"]) is torch.nn.Embedding and el["C1"].weight.stride() == (stride, 1)
    assert ll["C1"].weight.data_ptr() == el["C1"].weight.data_ptr() + 16 * 4 and ll["C1"].weight.stride(0) == stride
    assert [k for k in packed.state_dict()] == [k for k in plain.state_dict()]
    for (k, a), (_, b) in zip(plain.state_dict().items(), packed.state_dict().items()):
        assert torch.equal(a, b), k
    X, y = _cuda(X0), y0.cuda()

    def step(model):
        for p in model.parameters():
            p.grad = None
        logit = model.logits(X)
        loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), y)
        loss.backward()
        return logit.detach().clone(), loss.detach().clone()

    l0, loss0 = step(plain)
    l1, loss1 = step(packed)
    assert torch.equal(l0, l1) and torch.equal(loss0, loss1)
    for (k, p0), (_, p1) in zip(plain.named_parameters(), packed.named_parameters()):
        assert p1.grad.is_contiguous() and torch.equal(p0.grad, p1.grad), k
    # an optimiser step through the strided parameters, then back into a fresh unpacked model
    with torch.no_grad():
        for p0, p1 in zip(plain.parameters(), packed.parameters()):
            p0.add_(p0.grad, alpha=-0.5)
            p1.add_(p1.grad, alpha=-0.5)
    fresh = FM(fm, 16).cuda()
    fresh.load_state_dict(packed.state_dict())
    for (k, a), (_, b) in zip(plain.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        replay = GraphedStep(lambda: step(packed), warmup=3)
        want = step(plain)
        got = replay()
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        for (k, p0), (_, p1) in zip(plain.named_parameters(), packed.named_parameters()):
            assert torch.equal(p0.grad, p1.grad), k
    finally:
        ops.config.check_ids = old


def test_out_of_range_ids_are_reported_late_when_the_eager_check_is_off():
    """With ops.config.check_ids = False (hipGraph capture, benchmarks) an id outside its table no longer raises at the
    lookup -- the reference's nn.Embedding would (IndexError) -- but it is not silent either: every kernel ORs into one
    persistent device word, ops.check_deferred_ids() reports it, and GraphedStep(check_every=N) calls that every N replays."""
    from recbox_amd import ops
    from recbox_amd.graph import GraphedStep
    from recbox_amd.ranking.pytorch.models import FM
    vocabs = CRITEO_SMALL_VOCABS[:6]
    fm, X0, y0 = _criteo_like(300, vocabs, 16, seed=9)
    model = FM(fm, 16).cuda()
    Xs, ys = _cuda(X0), y0.cuda()
    old = ops.config.check_ids
    ops.config.check_ids = False
    try:
        ops.check_deferred_ids()                                   # clear whatever earlier tests left behind
        step = GraphedStep(lambda: _bce_step(model, Xs, ys), warmup=3, check_every=2)
        step()
        step()                                                     # replay 2: checked, clean
        Xs["C2"][17] = vocabs[1] + 5                               # beyond the table (vocab_size = vocabs[1] + 1)
        step()                                                     # replay 3: flagged on the device, not read yet
        with pytest.raises(IndexError):
            step()                                                 # replay 4: the periodic check reports it
        Xs["C2"][17] = 1
        step()
        ops.check_deferred_ids()                                   # the word was cleared: clean again
    finally:
        ops.config.check_ids = old


@pytest.mark.gpu
def test_sigmoid_output_bce_matches_the_unfused_chain():
    """ops.binary_cross_entropy(ops.sigmoid_output(logit), y): loss and dL/dlogit from one pass over the logits
    (rbx_sigmoid_bce_mean + rbx_scale_by_scalar) == torch's sigmoid -> binary_cross_entropy chain, clamps included; y_pred stays
    an ordinary differentiable tensor for every other reader."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(4099, 1, generator=g) * 4
    x0[:6, 0] = torch.tensor([60.0, -60.0, 110.0, -110.0, 0.0, 17.0])
    y = (torch.rand(4099, 1, generator=g) < 0.3).float()
    y[:4, 0] = torch.tensor([0.0, 1.0, 0.0, 1.0])                       # saturated AND wrong: the -100 clamp is hit
    xr = x0.double().requires_grad_(True)
    pr = torch.sigmoid(xr.float())
    lr_ = torch.nn.functional.binary_cross_entropy(pr, y) * 3.0 + (pr * 0.25).sum()
    lr_.backward()
    x = x0.cuda().requires_grad_(True)
    p = ops.sigmoid_output(x)
    assert torch.equal(p.detach().cpu(), torch.sigmoid(x0.cuda()).cpu())
    loss = ops.binary_cross_entropy(p, y.cuda()) * 3.0 + (p * 0.25).sum()
    loss.backward()
    assert abs(loss.item() - lr_.item()) <= 1e-4 * max(1.0, abs(lr_.item()))
    assert (x.grad.cpu().double() - xr.grad).abs().max().item() <= 1e-6
    # the same tensors with the fusion switched off: the two-step kernels
    old = ops.config.fuse_sigmoid_bce
    ops.config.fuse_sigmoid_bce = False
    try:
        x2 = x0.cuda().requires_grad_(True)
        l2 = ops.binary_cross_entropy(ops.sigmoid_output(x2), y.cuda()) * 3.0
        l2.backward()
    finally:
        ops.config.fuse_sigmoid_bce = old
    x3 = x0.cuda().requires_grad_(True)
    l3 = ops.binary_cross_entropy(ops.sigmoid_output(x3), y.cuda()) * 3.0
    l3.backward()
    assert abs(l2.item() - l3.item()) <= 1e-5 * max(1.0, abs(l2.item()))
    assert (x2.grad - x3.grad).abs().max().item() <= 1e-7
    with pytest.raises(ValueError):
        ops.binary_cross_entropy(ops.sigmoid_output(x3.detach()), y.cuda()[:10])


@pytest.mark.gpu
def test_tables_of_mixed_sizes_sort_in_their_own_number_of_passes():
    """Tables of 4, 200, 300, 70 000 and 1 200 000 rows in ONE lookup (1-, 2- and 3-pass segments of the backward's sort, two
    shared tables among them, a padding row): gradients == the dense index_add restatement in float64."""
    from recbox_amd import _embed_host as host
    from recbox_amd._lib import FIELD_CATEGORICAL
    torch.manual_seed(3)
    B, D = 5000, 8
    vocabs = [4, 200, 300, 70000, 1200000]
    mods = [torch.nn.Embedding(v, D, padding_idx=(0 if k == 3 else None)).cuda() for k, v in enumerate(vocabs)]
    uses = [0, 1, 2, 3, 4, 2, 0]                                            # field -> table (tables 2 and 0 feed two fields)
    ids = [torch.randint(0, vocabs[t], (B,), device="cuda") for t in uses]
    plan = host.Plan([host.Lookup("f%d" % k, FIELD_CATEGORICAL, mods[t], dim=D) for k, t in enumerate(uses)])
    out = plan.run(ids)
    assert out.shape == (B, len(uses) * D)
    w = torch.randn(B, len(uses) * D, device="cuda")
    (out * w).sum().backward()
    for t, m in enumerate(mods):
        want = torch.zeros(vocabs[t], D, dtype=torch.float64, device="cuda")
        for k, u in enumerate(uses):
            if u != t:
                continue
            g = w[:, k * D:(k + 1) * D].double()
            if m.padding_idx is not None:
                g = g * (ids[k] != m.padding_idx).unsqueeze(1)
            want.index_add_(0, ids[k], g)
            assert torch.equal(out[:, k * D:(k + 1) * D], m.weight.detach()[ids[k]])
        assert (m.weight.grad.double() - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item()), t


# ---- round 5: rbx_fm_fwd's kernel for ids that are the columns of one batch tensor (csrc/rbx_fm_quad.hip) ----------------
def _flat_fm_case(n_num, vocabs, B, seed, dtype=torch.float64, integer_values=False):
    """An FM over ``n_num`` numeric + len(vocabs) categorical features and the flat [B, F + 1] batch tensor the reference's
    loader would deliver for it (h5_dataloader.py:36-47), label in the last column."""
    from recbox_amd.ranking.features import FeatureMap
    from recbox_amd.ranking.pytorch.models import FM
    g = torch.Generator().manual_seed(seed)
    fm = FeatureMap("t", "/tmp")
    fm.features = OrderedDict()
    for i in range(n_num):
        fm.features["I%d" % (i + 1)] = {"source": "", "type": "numeric"}
    for i, v in enumerate(vocabs):
        fm.features["C%d" % (i + 1)] = {"source": "", "type": "categorical", "vocab_size": v + 1, "padding_idx": 0}
    fm.labels, fm.num_fields = ["y"], n_num + len(vocabs)
    fm.set_column_index()
    cols = []
    for i in range(n_num):
        x = torch.rand(B, generator=g, dtype=torch.float64)
        cols.append(torch.floor(x * 7) if integer_values else x)
    for v in vocabs:
        cols.append(torch.randint(0, v + 1, (B,), generator=g).double())
    cols.append((torch.rand(B, generator=g) < 0.25).double())
    batch = torch.stack(cols, dim=1).to(dtype).cuda()
    model = FM(fm, 16).cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * 0.1).cuda())
    return fm, model, batch


@pytest.mark.parametrize("n_num,n_cat,B,dtype", [
    (13, 26, 65536, torch.float64), (13, 26, 1000, torch.float32), (13, 26, 5, torch.int64), (13, 26, 257, torch.int32),
    (0, 1, 1, torch.float64), (3, 9, 333, torch.float64), (2, 62, 4097, torch.float64), (0, 4, 64, torch.int32),
    (5, 8, 100, torch.float64), (16, 0, 77, torch.float32), (1, 15, 63, torch.int64)])
def test_fm_quad_forward_equals_the_general_kernel(n_num, n_cat, B, dtype):
    """rbx_fm_fwd on a flat batch tensor (dim 16) runs fm_quad_fwd_kernel (lane group of 4 per sample, the owner lane decodes
    its columns, DPP quad broadcast of the row offsets); it performs the general kernel's floating-point operations in the
    same order: logits, sigmoid outputs, the kept sums S (through the gradients) are BIT-identical with the path forced back
    (rbx_fm_quad(0)), for every id dtype, feature counts 1 .. 64 and batch sizes off the tile size."""
    from recbox_amd._lib import lib
    from recbox_amd.ranking.pytorch.models import inputs_from_batch
    vocabs = (CRITEO_SMALL_VOCABS * 3)[:n_cat]
    integer = dtype in (torch.int32, torch.int64)
    fm, model, batch = _flat_fm_case(n_num, vocabs, B, seed=B + n_cat, dtype=dtype, integer_values=integer)
    y = batch[:, -1:].float()
    was = lib.rbx_fm_quad(-1)
    res = []
    try:
        for on in (1, 0):
            lib.rbx_fm_quad(on)
            model.zero_grad(set_to_none=True)
            logit = model.logits(inputs_from_batch(fm, batch))
            out = model(batch)["y_pred"]
            loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(logit), y, reduction="sum")
            loss.backward()
            torch.cuda.synchronize()
            res.append([logit.detach().clone(), out.detach().clone()] + [p.grad.clone() for p in model.parameters()])
    finally:
        lib.rbx_fm_quad(was)
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # and against the float64 restatement of the model on the same inputs
    from oracle import torch_ref as R
    ref = R.RefFMModel(fm, 16)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    X = OrderedDict((k, v.double().cpu()) for k, v in inputs_from_batch(fm, batch).items() if k != "y")
    assert_close(res[0][0], ref(X), TOL, "quad logit vs oracle")


def test_fm_quad_out_of_range_ids_read_as_zero_rows_and_raise():
    """An id outside its table (too large, negative, NaN) reads as a zero row in both kernels and sets the status word
    (ops.config.check_ids turns it into the reference's IndexError); with the check off the logits agree bit for bit."""
    from recbox_amd import ops
    from recbox_amd._lib import lib
    from recbox_amd.ranking.pytorch.models import inputs_from_batch
    fm, model, batch = _flat_fm_case(2, [5, 9, 1000], 130, seed=9)
    bad = batch.clone()
    bad[3, 2] = 6.0          # C1 has 6 rows (0..5)
    bad[64, 3] = -1.0
    bad[129, 4] = float("nan")
    was, chk = lib.rbx_fm_quad(-1), ops.config.check_ids
    try:
        outs = []
        for on in (1, 0):
            lib.rbx_fm_quad(on)
            ops.config.check_ids = True
            with pytest.raises(IndexError):
                model.logits(inputs_from_batch(fm, bad))
                torch.cuda.synchronize()
            ops.config.check_ids = False
            with torch.no_grad():
                outs.append(model.logits(inputs_from_batch(fm, bad)).clone())
            with pytest.raises(IndexError):
                ops.check_deferred_ids()                 # reported late when the check is off
        assert torch.equal(outs[0], outs[1])
        good = outs[0][torch.tensor([i for i in range(130) if i not in (3, 64, 129)]).cuda()]
        with torch.no_grad():
            ref = model.logits(inputs_from_batch(fm, batch))[torch.tensor([i for i in range(130) if i not in (3, 64, 129)]).cuda()]
        assert torch.equal(good, ref)
    finally:
        lib.rbx_fm_quad(was)
        ops.config.check_ids = chk


@pytest.mark.parametrize("B,F,D,M", [(1, 2, 4, None), (33, 5, 8, None), (33, 5, 8, 7), (257, 39, 16, None), (64, 39, 16, 24)])
def test_cin_outer_product_kernel_vs_the_einsum(B, F, D, M):
    """rbx_cin_outer_fwd/bwd against compressed_interaction_net.py:41-43 (einsum + view), laid out for the channel GEMM:
    z[(b, d), h M + m]; M None = the first layer (X_k = X_0: both uses of X_0 receive gradient)."""
    from recbox_amd import ops
    g = torch.Generator().manual_seed(B + F)
    x0 = torch.randn(B, F, D, generator=g, dtype=torch.float64)
    xk = None if M is None else torch.randn(B * D, M, generator=g, dtype=torch.float64)
    R = torch.randn(B * D, F * (M or F), generator=g, dtype=torch.float64)
    a = x0.clone().requires_grad_()
    k = None if xk is None else xk.clone().requires_grad_()
    kk = a if k is None else k.view(B, D, -1).transpose(1, 2)                  # [B, M, D]
    want = torch.einsum("bhd,bmd->bhmd", a, kk).reshape(B, -1, D).transpose(1, 2).reshape(B * D, -1)
    (want * R).sum().backward()
    ad = x0.float().cuda().requires_grad_()
    kd = None if xk is None else xk.float().cuda().requires_grad_()
    z = ops.cin_outer(ad, kd)
    assert_close(z, want.float(), 1e-5, "z")
    (z * R.float().cuda()).sum().backward()
    assert_close(ad.grad, a.grad.float(), 1e-4, "dx0")
    if kd is not None:
        assert_close(kd.grad, k.grad.float(), 1e-4, "dxk")
