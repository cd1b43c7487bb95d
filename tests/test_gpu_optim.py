"""recbox_amd.optim: the sparse-row optimiser step (rbx_embed_sparse_update / rbx_fm_sparse_update) against the rules of
torch.optim's sparse branches applied, in plain torch, to exactly the rows the batch looked up ("or, opt-in, a
sparse-row update path"; the reference's loop: ranking/pytorch/models/ranking_model.py:191-197)."""
import math
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_ranking import _criteo_like, _cuda  # noqa: E402


def _reference_step(rule, p, g, rows, st, hp):
    """The sparse branch of torch.optim.{SGD, Adagrad, SparseAdam} on ``rows`` of ``p`` (fp32, same formulas)."""
    st["step"] += 1
    t = st["step"]
    gr = g[rows]
    if rule == "sgd":
        p[rows] -= hp["lr"] * gr
    elif rule == "adagrad":
        st["sum"][rows] += gr * gr
        p[rows] -= hp["lr"] * gr / (st["sum"][rows].sqrt() + hp["eps"])
    else:
        b1, b2 = hp["betas"]
        st["m"][rows] = b1 * st["m"][rows] + (1 - b1) * gr
        st["v"][rows] = b2 * st["v"][rows] + (1 - b2) * gr * gr
        step = hp["lr"] * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        p[rows] -= step * st["m"][rows] / (st["v"][rows].sqrt() + hp["eps"])


def _make(rule, params):
    from recbox_amd import optim
    if rule == "sgd":
        return optim.SparseSGD(params, lr=0.05), {"lr": 0.05}
    if rule == "adagrad":
        return optim.SparseAdagrad(params, lr=0.05, eps=1e-10), {"lr": 0.05, "eps": 1e-10}
    return optim.SparseAdam(params, lr=0.01, betas=(0.9, 0.999), eps=1e-8), {"lr": 0.01, "betas": (0.9, 0.999), "eps": 1e-8}


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("rule", ["sgd", "adagrad", "adam"])
def test_sparse_row_step_equals_torch_rule_on_touched_rows(rule, fused):
    """Four steps of an FM over tables of 5 .. 70 000 rows (the fused body: tier A tables by bitmap, tier B by sorted run
    heads; the layer-composed model: rbx_embed_sparse_update for FeatureEmbedding + the LR-only fused call): every table
    and its optimiser state equal the torch rule applied to the rows the batch looked up; rows it did not look up are
    bit-identical to their initial values."""
    from recbox_amd import ops, optim
    from recbox_amd.ranking.pytorch.models import FM
    vocabs = [37, 5, 3001, 211, 70000]
    fm, _, _ = _criteo_like(4, vocabs, 16, seed=3)
    model = FM(fm, 16, fused=fused).cuda()
    with torch.no_grad():
        for p in model.parameters():
            p.normal_(0, 0.1)
    tables, rest = optim.split_parameters(model)
    assert len(tables) == 2 * len(vocabs)
    opt, hp = _make(rule, tables)
    names = dict((id(p), n) for n, p in model.named_parameters())
    # torch-side copies of every table + state
    ref = dict((id(p), p.detach().clone()) for p in tables)
    init = dict((id(p), p.detach().clone()) for p in tables)
    state = dict((id(p), {"step": 0, "sum": torch.zeros_like(p), "m": torch.zeros_like(p), "v": torch.zeros_like(p)}) for p in tables)
    ever = dict((id(p), torch.zeros(p.shape[0], dtype=torch.bool, device="cuda")) for p in tables)
    feat_of = {}
    for holder in (model.embedding_layer.embedding_layer.embedding_layers, model.fm.lr_layer.embedding_layer.embedding_layer.embedding_layers):
        for fname, mod in holder.items():
            if isinstance(mod, torch.nn.Embedding):
                feat_of[id(mod.weight)] = fname
    old = ops.config.reuse_grad_buffers
    try:
        for flag in (False, True):                      # fresh gradients, then the persistent buffers
            ops.config.reuse_grad_buffers = flag
            for k, B in enumerate([300, 700, 64, 700]):
                _, X, y = _criteo_like(B, vocabs, 16, seed=50 + k + 10 * int(flag), zipf=bool(k % 2))
                Xc, yc = _cuda(X), y.cuda()
                opt.zero_grad()
                for p in rest:
                    p.grad = None
                loss = torch.nn.functional.binary_cross_entropy(torch.sigmoid(model.logits(Xc)), yc, reduction="mean")
                loss.backward()
                grads = dict((id(p), p.grad.detach().clone()) for p in tables)
                opt.step()
                for p in tables:
                    rows = torch.unique(Xc[feat_of[id(p)]].long())
                    rows = rows[rows != 0]              # padding_idx = 0: never looked up by these batches, never stepped
                    ever[id(p)][rows] = True
                    _reference_step(rule, ref[id(p)], grads[id(p)], rows, state[id(p)], hp)
                    assert torch.allclose(p.detach(), ref[id(p)], atol=2e-6, rtol=0), \
                        "%s step %d (%s): %g" % (names[id(p)], k, flag, (p.detach() - ref[id(p)]).abs().max())
        # every step went through the C ABI (fused: one rbx_fm_sparse_update; layers: rbx_embed_sparse_update + the LR call)
        assert opt.calls["dense"] == 0 and opt.calls["rows"] == 8 * (1 if fused else 2), opt.calls
        for p in tables:
            untouched = ~ever[id(p)]
            assert torch.equal(p.detach()[untouched], init[id(p)][untouched]), names[id(p)]
            st = opt.state[id(p)]
            if rule == "adagrad":
                assert torch.allclose(st["s"][0], state[id(p)]["sum"], atol=1e-6)
            if rule == "adam":
                assert torch.allclose(st["s"][0], state[id(p)]["m"], atol=1e-6) and torch.allclose(st["s"][1], state[id(p)]["v"], atol=1e-6)
    finally:
        ops.config.reuse_grad_buffers = old
        ops.config.track_touched_rows = False


def test_sparse_adam_on_a_rechub_model_and_dense_fallback():
    """YoutubeDNN (rechub mirror: one gather over user id, history, item and negatives): the item table is stepped through
    its lookup's sorted ids; a parameter without such a record (a tower weight handed to the same optimiser) takes the
    dense fallback -- both equal the torch rule."""
    import torch.nn.functional as F
    from recbox_amd import ops, optim
    from test_gpu_shard import _rh, _seed_params, _youtube_batch, _youtube_feats
    from recbox_amd.rechub.models.matching import YoutubeDNN
    Fe = _rh()
    V, D, B, L, n_neg = 997, 16, 129, 9, 3
    model = YoutubeDNN(*_youtube_feats(Fe, V, D, True), {"dims": [32, D]}, temperature=0.1).cuda()
    _seed_params(model)
    tables, rest = optim.split_parameters(model)
    tower = rest[0]
    opt = optim.SparseAdam(tables + [tower], lr=0.01)
    hp = {"lr": 0.01, "betas": (0.9, 0.999), "eps": 1e-8}
    ref = dict((id(p), p.detach().clone()) for p in tables + [tower])
    state = dict((id(p), {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}) for p in tables + [tower])
    tgt = torch.zeros(B, dtype=torch.long, device="cuda")
    try:
        for k in range(3):
            x = {kk: v.cuda() for kk, v in _youtube_batch(B, V, L, n_neg, 40 + k).items()}
            model.zero_grad(set_to_none=True)
            F.cross_entropy(model(x), tgt).backward()
            grads = dict((id(p), p.grad.detach().clone()) for p in tables + [tower])
            opt.step()
            for p in tables + [tower]:
                g = grads[id(p)]
                rows = (g.reshape(g.shape[0], -1) != 0).any(dim=1).nonzero().reshape(-1)
                _reference_step("adam", ref[id(p)], g, rows, state[id(p)], hp)
                assert torch.allclose(p.detach(), ref[id(p)], atol=2e-6, rtol=0), (k, tuple(p.shape))
        assert opt.calls == {"rows": 3, "dense": 3}, opt.calls       # tables through their lookup's ids, the tower weight densely
    finally:
        ops.config.track_touched_rows = False
