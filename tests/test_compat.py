"""recbox_amd.compat.install(): every dotted path of SURVEY.md 8(b) -- the reference's own (``recbox.*``), FuxiCTR's
(``fuxictr.*``: recbox.ranking imports itself under that name, feature_embedding.py:24-25) and Torch-RecHub's
(``torch_rechub.*``: sasrec.py:13-14) -- imports and yields the HIP-backed mirror class.  Runs in a child process: the
aliases live in sys.modules."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT] + ([extra_path] if extra_path else []))
    proc = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stdout
    return proc.stdout


def test_every_reference_path_imports_the_mirror():
    out = _run("""
        import importlib, sys
        import recbox_amd.compat as C
        assert "recbox" not in sys.modules and "fuxictr" not in sys.modules and "torch_rechub" not in sys.modules
        rep = C.install()
        table = C.alias_table()
        assert len(table) >= 80
        for path, (target, names) in table.items():
            mod = importlib.import_module(path)
            src = importlib.import_module(target)
            for n in (names or []):
                assert getattr(mod, n) is getattr(src, n), (path, n)
        # the import forms the reference's own files use
        from recbox.ranking.pytorch.layers.embeddings.feature_embedding import FeatureEmbedding, FeatureEmbeddingDict
        from recbox.ranking.pytorch.layers.interactions.inner_product import InnerProductInteraction
        from recbox.ranking.pytorch.layers.blocks.factorization_machine import FactorizationMachine
        from recbox.ranking.pytorch.layers.blocks.logistic_regression import LogisticRegression
        from recbox.ranking.pytorch.layers import MLP_Block, MaskedAveragePooling, ScaledDotProductAttention
        from recbox.core.pytorch.layers.embedding import EmbeddingLayer, EmbeddingDictLayer
        from recbox.core.pytorch.layers import MLP_Layer
        from recbox.core.pytorch.losses import SoftmaxCrossEntropyLoss
        from recbox.matching.features import FeatureMap as MatchingMap
        from recbox.ranking.features import FeatureMap as RankingMap
        from fuxictr.pytorch import layers                     # feature_embedding.py:25: `from fuxictr.pytorch import layers`
        from fuxictr.pytorch.layers import FeatureEmbedding as FE2
        from fuxictr.features import FeatureMap as FM2
        from torch_rechub.basic.layers import EmbeddingLayer as RechubEmbedding, MLP, FM       # sasrec.py:13
        from torch_rechub.basic.features import SparseFeature, SequenceFeature, DenseFeature   # sasrec.py:14
        from torch_rechub.models.matching import DSSM, YoutubeDNN, SASRec
        from torch_rechub.models.ranking import DeepFM
        from recbox.third_party.rechub.models.matching.sasrec import SASRec as S2
        assert FE2 is FeatureEmbedding and S2 is SASRec and FM2 is RankingMap and layers.FeatureEmbedding is FeatureEmbedding
        for cls in (FeatureEmbedding, InnerProductInteraction, FactorizationMachine, LogisticRegression, MLP_Block,
                    EmbeddingLayer, MLP_Layer, RechubEmbedding, MLP, DSSM, YoutubeDNN, SASRec, DeepFM, MatchingMap):
            assert cls.__module__.startswith("recbox_amd."), cls
        C.uninstall(rep)
        assert "recbox" not in sys.modules and "fuxictr" not in sys.modules and "torch_rechub" not in sys.modules
        print("ok", len(table))
    """)
    assert out.strip().startswith("ok")


def test_overlay_rebinds_the_layers_of_an_installed_reference(tmp_path):
    """With the reference importable, its modules stay and only the hot-path names are re-bound (and restored by uninstall)."""
    pkg = tmp_path / "recbox" / "ranking" / "pytorch" / "layers" / "embeddings"
    pkg.mkdir(parents=True)
    for d in (tmp_path / "recbox", tmp_path / "recbox" / "ranking", tmp_path / "recbox" / "ranking" / "pytorch",
              tmp_path / "recbox" / "ranking" / "pytorch" / "layers", pkg):
        (d / "__init__.py").write_text("")
    (pkg / "feature_embedding.py").write_text("class FeatureEmbedding(object):\n    reference = True\nKEEP = 7\n")
    out = _run("""
        import recbox_amd.compat as C
        import recbox.ranking.pytorch.layers.embeddings.feature_embedding as ref_mod
        ref_cls = ref_mod.FeatureEmbedding
        rep = C.install(prefixes=("recbox",))
        import recbox.ranking.pytorch.layers.embeddings.feature_embedding as again
        assert again is ref_mod and again.KEEP == 7                       # the installed module itself, still there
        assert again.FeatureEmbedding.__module__.startswith("recbox_amd.") and not hasattr(again.FeatureEmbedding, "reference")
        assert "recbox.ranking.pytorch.layers.embeddings.feature_embedding" not in rep["created"]
        from recbox.ranking.pytorch.layers.blocks.logistic_regression import LogisticRegression   # absent in the stub: created
        assert LogisticRegression.__module__.startswith("recbox_amd.")
        C.uninstall(rep)
        assert ref_mod.FeatureEmbedding is ref_cls
        print("ok")
    """, extra_path=str(tmp_path))
    assert out.strip() == "ok"
