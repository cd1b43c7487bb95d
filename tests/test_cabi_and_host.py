"""CPU-only checks: the C-ABI library loads and exports every symbol declared in
include/recbox_hip.h (no compute calls: there is no GPU here), and the host-side mirrors of the
reference's schema / layer API behave like the reference (names, shapes, exceptions)."""
import ctypes
import json
import os
import re
from collections import OrderedDict

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "recbox_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rbx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from recbox_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "librecbox_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "recbox_amd/_lib.py has no ctypes signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names          # and nothing bound that the header does not declare
    assert _lib.lib.rbx_version() == 124
    assert ctypes.sizeof(_lib.rbx_field_t) == 104    # matches the C layout (8-byte aligned; table_stride appended in round 2)


def test_every_c_symbol_has_a_python_caller():
    """The host layer (recbox_amd/*.py) calls every compute entry point of the C ABI, and the op surface the
    mirrors are built on is complete (guards against a half-written ops.py reaching the GPU box)."""
    from recbox_amd import _lib, ops
    src = ""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "recbox_amd")):
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                src += open(os.path.join(dirpath, f)).read()
    for n in _lib.SIGNATURES:
        if n in ("rbx_last_error", "rbx_version"):
            continue
        assert ("lib.%s(" % n) in src, "no Python caller of %s" % n
    for name in ("embed_lookup", "interaction", "pool", "fm_fused", "route", "linear", "l2_normalize", "pair_dot",
                 "gather_dot", "attention", "interaction_rowsum", "KernelTimer", "EmbedPlan", "FieldSpec"):
        assert hasattr(ops, name), "recbox_amd.ops.%s is missing" % name


def test_product_path_refuses_cpu_tensors():
    """No silent CPU/PyTorch fallback: a CPU call raises."""
    import recbox_amd.ranking.pytorch.layers as L
    from test_oracle_golden import _FM
    f = OrderedDict([("c", {"source": "", "type": "categorical", "vocab_size": 5})])
    layer = L.FeatureEmbedding(_FM(f), 4)
    with pytest.raises(RuntimeError):
        layer({"c": torch.tensor([1, 2])})
    with pytest.raises(RuntimeError):
        L.InnerProductInteraction(3)(torch.zeros(2, 3, 4))


def test_interaction_rejects_unknown_mode_like_reference():
    import recbox_amd.ranking.pytorch.layers as L
    with pytest.raises(ValueError, match="is not supported"):
        L.InnerProductInteraction(4, output="concat")
    m = L.InnerProductInteraction(4, output="inner_product")
    assert m.interaction_units == 6 and tuple(m.triu_mask.shape) == (4, 4)


def test_ranking_feature_map_roundtrip_and_columns(tmp_path):
    from recbox_amd.ranking.features import FeatureMap
    fm = FeatureMap("ds", str(tmp_path))
    fm.features = OrderedDict([
        ("I1", {"source": "u", "type": "numeric"}),
        ("C1", {"source": "u", "type": "categorical", "vocab_size": 10, "padding_idx": 0}),
        ("S1", {"source": "i", "type": "sequence", "vocab_size": 7, "max_len": 3, "embedding_dim": 4}),
        ("M", {"source": "u", "type": "meta"})])
    fm.labels = ["y"]
    fm.num_fields = 3
    fm.set_column_index()
    assert fm.column_index == {"I1": 0, "C1": 1, "S1": [2, 3, 4], "M": 5, "y": 6} and fm.input_length == 6
    assert fm.get_num_fields() == 3 and fm.get_num_fields("i") == 1
    fm.default_emb_dim = 8
    assert fm.sum_emb_out_dim() == 8 + 8 + 4
    path = os.path.join(str(tmp_path), "ds", "feature_map.json")
    fm.save(path)
    blob = json.load(open(path))
    assert [list(d)[0] for d in blob["features"]] == ["I1", "C1", "S1", "M"]
    fm2 = FeatureMap("ds", str(tmp_path))
    fm2.load(path, {"embedding_dim": 8, "use_features": ["C1", "S1"],
                    "feature_specs": [{"name": ["C1"], "embedding_dim": 2}]})
    assert list(fm2.features) == ["C1", "S1"] and fm2.features["C1"]["embedding_dim"] == 2
    assert fm2.column_index["S1"] == [1, 2, 3]
    with pytest.raises(RuntimeError):
        FeatureMap("other", str(tmp_path)).load(path, {})


def test_matching_feature_map_roundtrip(tmp_path):
    from recbox_amd.matching.features import FeatureMap
    fm = FeatureMap("ml", str(tmp_path), "user_id", "item_id", "label")
    fm.feature_specs = OrderedDict([("user_id", {"source": "user", "type": "categorical", "vocab_size": 5}),
                                    ("item_id", {"source": "item", "type": "categorical", "vocab_size": 9})])
    fm.num_fields = 2
    path = os.path.join(str(tmp_path), "ml", "feature_map.json")
    fm.save(path)
    fm2 = FeatureMap("ml", str(tmp_path), None, None, None)
    fm2.load(path)
    assert list(fm2.feature_specs) == ["user_id", "item_id"] and fm2.get_num_fields("user") == 1


def test_layer_parameter_contract_matches_reference_names():
    """state_dict keys / module types that the reference harness relies on."""
    import recbox_amd.ranking.pytorch.layers as L
    from recbox_amd.ranking.pytorch.models import FM
    from test_oracle_golden import _FM, criteo_small_features, ranking_embedding_features
    layer = L.FeatureEmbedding(_FM(ranking_embedding_features()), 8)
    keys = list(layer.state_dict())
    assert keys == ["embedding_layer.embedding_layers.%s.weight" % n for n in ("n1", "c1", "n2", "hist", "c2", "c3")]  # the alias c2 is listed too, as in the reference
    el = layer.embedding_layer.embedding_layers
    assert type(el["c1"]) is torch.nn.Embedding and type(el["n1"]) is torch.nn.Linear and el["c2"] is el["hist"]
    assert float(el["c1"].weight[0].abs().sum()) == 0.0 and float(el["c1"].weight[1:].std()) < 1e-3
    assert isinstance(layer.embedding_layer.feature_encoders["hist"], L.MaskedAveragePooling)
    lr = L.LogisticRegression(_FM(ranking_embedding_features()))
    assert lr.embedding_layer.embedding_layer.embedding_layers["hist"].embedding_dim == 1
    assert isinstance(lr.embedding_layer.embedding_layer.feature_encoders["hist"], L.MaskedSumPooling)
    fx_keys = set(FM(_FM(criteo_small_features()), 16).state_dict())
    assert "fm.lr_layer.bias" in fx_keys and "fm.lr_layer.embedding_layer.embedding_layer.embedding_layers.C1.weight" in fx_keys
    with pytest.raises(ValueError):
        L.FeatureEmbedding(_FM(OrderedDict([("s", {"source": "", "type": "sequence", "vocab_size": 4,
                                                    "feature_encoder": "layers.NoSuchPooling()"})])), 4)


def test_rechub_layer_contract():
    from recbox_amd.rechub.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from recbox_amd.rechub.basic.layers import MLP, EmbeddingLayer, PredictionLayer
    from recbox_amd.rechub.models.matching import SASRec
    a, b = SparseFeature("a", 10, 4), SequenceFeature("h", 10, 4, pooling="mean", shared_with="a")
    layer = EmbeddingLayer([a, b, DenseFeature("d")])
    assert list(layer.state_dict()) == ["embed_dict.a.weight"] and layer.n_dense == 1
    assert a.get_embedding_layer() is layer.embed_dict["a"] and layer.embed_dict["a"].padding_idx is None
    assert SparseFeature("x", 10000).embed_dim == 60          # 6 * V^(1/4)
    m = MLP(8, output_layer=True, dims=[4], dropout=0.1, activation="prelu")
    assert [type(c).__name__ for c in m.mlp] == ["Linear", "BatchNorm1d", "PReLU", "Dropout", "Linear"]
    with pytest.raises(ValueError):
        PredictionLayer("ranking")
    fe = [SequenceFeature("seq", 17, 8, pooling="concat"), SequenceFeature("pos", 17, 8, pooling="concat", shared_with="seq"),
          SequenceFeature("neg", 17, 8, pooling="concat", shared_with="seq")]
    keys = set(SASRec(fe, max_len=4, dropout_rate=0.0).state_dict())
    assert {"item_emb.embed_dict.seq.weight", "position_emb.weight", "attention_layers.0.in_proj_weight",
            "forward_layers.1.conv2.bias", "last_layernorm.weight"} <= keys


def test_embed_plan_layout_is_pure_host_logic():
    """Slot offsets / dedup of shared tables are computed without touching the GPU."""
    from recbox_amd import _embed_host as host
    from recbox_amd._lib import FIELD_CATEGORICAL, FIELD_DENSE, FIELD_NUMERIC, POOL_CONCAT, POOL_MEAN_ID
    t1, t2, w = torch.nn.Embedding(10, 4), torch.nn.Embedding(6, 4), torch.nn.Linear(1, 4, bias=False)
    plan = host.Plan([host.Lookup("a", FIELD_CATEGORICAL, t1, 4),
                      host.Lookup("n", FIELD_NUMERIC, w, 4),
                      host.Lookup("h", FIELD_CATEGORICAL, t1, 4, pool=POOL_MEAN_ID, seq_len=5, mask_id=0, eps=1e-16),
                      host.Lookup("s", FIELD_CATEGORICAL, t2, 4, pool=POOL_CONCAT, seq_len=3),
                      host.Lookup("d", FIELD_DENSE, None, 1)])
    assert [s.out_off for s in plan.specs] == [0, 4, 8, 12, 24] and plan.width == 25
    assert plan.modules == [t1, w, t2] and [s.param for s in plan.specs] == [0, 1, 0, 2, -1]
    assert plan.plan.arr[2].vocab == 10 and plan.plan.arr[2].seq_len == 5 and plan.plan.needs_row_scale
    with pytest.raises(NotImplementedError):
        host.Plan([host.Lookup("x%d" % i, FIELD_DENSE, None, 1) for i in range(65)])


def test_listwise_losses_match_reference_fixture():
    """The six y_pred[B, 1+negs] losses are plain ATen expressions (SURVEY a-13): checked on CPU."""
    from conftest import Fixture, assert_close
    from recbox_amd.core.pytorch import losses
    fx = Fixture("target_attention_losses")
    t = fx.tensors("in")
    cases = {"pairwise_logistic": losses.PairwiseLogisticLoss(), "pairwise_margin": losses.PairwiseMarginLoss(0.5),
             "mse": losses.MSELoss(), "ccl": losses.CosineContrastiveLoss(0.1),
             "ccl_weighted": losses.CosineContrastiveLoss(0.1, negative_weight=2.0)}
    for name, fn in cases.items():
        y = t["y_pred"].clone().requires_grad_(True)
        v = fn(y, t["y_true"])
        assert_close(v, fx["out"][name], 1e-6, name)
        v.backward()
        assert_close(y.grad, fx["g"][name], 1e-6, "grad " + name)
    fx2 = Fixture("attention_losses")
    t2 = fx2.tensors("in")
    assert_close(losses.SoftmaxCrossEntropyLoss()(t2["y_pred"], t2["y_true"]), fx2["out"]["softmax_ce"], 1e-6)
    assert_close(losses.SigmoidCrossEntropyLoss()(t2["y_pred"], t2["y_true"]), fx2["out"]["sigmoid_ce"], 1e-5)


def test_fused_dict_drops_its_block_when_the_caller_edits_it():
    """``dict2tensor`` may hand out the fused [B, width] block only while the dict still holds what the
    layer put there; the FuxiCTR idiom ``feature_emb_dict[f] = new_emb`` (and pop / del / update) must make it stack the
    CURRENT values, as feature_embedding.py:169-186 and core/pytorch/layers/embedding.py:109-114 do."""
    from collections import OrderedDict
    from recbox_amd._embed_host import FusedDict
    import recbox_amd.ranking.pytorch.layers as RL
    import recbox_amd.core.pytorch.layers as CL

    class Spec(object):
        pool = 0

    class Plan(object):
        specs = [Spec(), Spec()]
        uniform_dim = 4

    def fresh():
        block = torch.arange(24.0).reshape(3, 8)
        d = FusedDict()
        d["a"], d["b"] = block[:, :4], block[:, 4:]
        d.seal(block, Plan(), ["a", "b"])
        return d, block

    for edit in ("set", "pop", "del", "update", "popitem", "clear", "move"):
        d, block = fresh()
        assert d.fused is block
        if edit == "set":
            d["b"] = torch.ones(3, 4)
        elif edit == "pop":
            d.pop("a")
        elif edit == "del":
            del d["a"]
        elif edit == "update":
            d.update({"b": torch.ones(3, 4)})
        elif edit == "popitem":
            d.popitem()
        elif edit == "clear":
            d.clear()
        else:
            d.move_to_end("a")
        assert d.fused is None and d.plan is None and d.names == (), edit

    class FMap(object):
        features = OrderedDict((n, {"source": "", "type": "categorical", "vocab_size": 5}) for n in ("a", "b"))
        feature_specs = features
        num_fields = 2

    rl = RL.FeatureEmbeddingDict(FMap(), 4)
    cl = CL.EmbeddingDictLayer(FMap(), 4)
    for layer in (rl, cl):
        d, block = fresh()
        assert layer.dict2tensor(d).data_ptr() == block.data_ptr()            # untouched: the block itself, viewed
        new_b = torch.full((3, 4), -1.0)
        d["b"] = new_b
        got = layer.dict2tensor(d)
        assert torch.equal(got, torch.stack([block[:, :4], new_b], dim=1))    # the replaced value is what gets stacked
        d, block = fresh()
        d.pop("a")
        got = layer.dict2tensor(d)
        want = block[:, 4:] if layer is cl else block[:, 4:].unsqueeze(1)     # core: a single entry is returned un-stacked
        assert torch.equal(got, want)


def test_tower_with_too_few_per_layer_settings_raises_like_the_reference():
    """``MLP_Layer(hidden_units=[8, 4])`` with the default ``dropout_rates=[]`` is an IndexError in the
    reference (mlp.py:25-37 indexes ``dropout_rates[idx]``); it must not build a shorter tower silently."""
    import recbox_amd.core.pytorch.layers as CL
    import recbox_amd.ranking.pytorch.layers as RL
    with pytest.raises(IndexError):
        CL.MLP_Layer(6, hidden_units=[8, 4])
    with pytest.raises(IndexError):
        RL.MLP_Block(6, hidden_units=[8, 4], hidden_activations=["relu"])
    assert len(CL.MLP_Layer(6, hidden_units=[8, 4], dropout_rates=[0, 0]).mlp) == 4
    assert len(RL.MLP_Block(6, hidden_units=[8, 4]).mlp) == 4                 # scalar default dropout broadcasts
