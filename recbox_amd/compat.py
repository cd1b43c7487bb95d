"""One call that puts the HIP-backed mirrors under the reference's own import paths.

    import recbox_amd.compat
    recbox_amd.compat.install()
    from recbox.ranking.pytorch.layers.embeddings.feature_embedding import FeatureEmbedding      # the HIP-backed class
    from fuxictr.pytorch.layers import InnerProductInteraction                                     # (FuxiCTR's own name)
    from torch_rechub.basic.layers import EmbeddingLayer, MLP                                      # (Torch-RecHub's)

The reference (reczoo/RecBox v0.0.4) has no plugin interface: its model zoo imports the layers by dotted path --
``recbox.ranking`` is FuxiCTR under another top-level name and still imports itself as ``fuxictr``
(/root/reference/recbox/ranking/pytorch/layers/embeddings/feature_embedding.py:24-25), the vendored Torch-RecHub models
import ``torch_rechub.basic.layers`` (third_party/rechub/models/matching/sasrec.py:13-14).  ``install()`` registers, for every
dotted path of SURVEY.md section 8(b) -- packages, the deep per-class modules, and the three top-level spellings
``recbox`` / ``fuxictr`` / ``torch_rechub`` (+ ``recbox.third_party.rechub``) -- a module that exposes the mirror classes of
this package under the reference's names.

Two situations:
* the reference is NOT installed: the paths are created from nothing (packages are empty namespaces carrying only the
  hot-path modules);
* the reference IS importable (or already imported): its modules stay what they are -- harness, trainers, datasets keep
  working -- and only the hot-path names are re-bound on them (``overlay``), so a model class defined afterwards, or one
  that looks the layer up at call time (``layers.FeatureEmbedding``), builds on the HIP kernels.  Classes that captured a
  reference layer at import time (``from ... import FeatureEmbedding`` executed earlier) keep the one they captured: call
  ``install()`` before importing the model zoo.

Nothing here touches the GPU; the mirrors themselves refuse CPU tensors (no fallback).
"""
import importlib
import sys
import types

# reference module (relative to its package root) -> (module of this package, names or None = the module's __all__ / public names)
_RANKING = {
    "features": ("recbox_amd.ranking.features", None),
    "pytorch.torch_utils": ("recbox_amd.ranking.pytorch.torch_utils", None),
    "pytorch.models": ("recbox_amd.ranking.pytorch.models", ["FM", "ShardedFM", "inputs_from_batch"]),
    "pytorch.layers": ("recbox_amd.ranking.pytorch.layers", None),
    "pytorch.layers.pooling": ("recbox_amd.ranking.pytorch.layers.pooling", None),
    "pytorch.layers.activations": ("recbox_amd.ranking.pytorch.layers.attentions", ["Dice", "GELU"]),
    "pytorch.layers.embeddings": ("recbox_amd.ranking.pytorch.layers.embeddings", ["FeatureEmbedding", "FeatureEmbeddingDict"]),
    "pytorch.layers.embeddings.feature_embedding": ("recbox_amd.ranking.pytorch.layers.embeddings",
                                                    ["FeatureEmbedding", "FeatureEmbeddingDict"]),
    "pytorch.layers.interactions": ("recbox_amd.ranking.pytorch.layers.interactions", None),
    "pytorch.layers.interactions.inner_product": ("recbox_amd.ranking.pytorch.layers.interactions", ["InnerProductInteraction"]),
    "pytorch.layers.interactions.cross_net": ("recbox_amd.ranking.pytorch.layers.interactions",
                                              ["CrossInteraction", "CrossNet", "CrossNetV2", "CrossNetMix"]),
    "pytorch.layers.interactions.compressed_interaction_net": ("recbox_amd.ranking.pytorch.layers.interactions",
                                                               ["CompressedInteractionNet"]),
    "pytorch.layers.interactions.bilinear_interaction": ("recbox_amd.ranking.pytorch.layers.interactions",
                                                         ["BilinearInteraction", "BilinearInteractionV2"]),
    "pytorch.layers.interactions.holographic_interaction": ("recbox_amd.ranking.pytorch.layers.interactions",
                                                            ["HolographicInteraction"]),
    "pytorch.layers.interactions.interaction_machine": ("recbox_amd.ranking.pytorch.layers.interactions", ["InteractionMachine"]),
    "pytorch.layers.blocks": ("recbox_amd.ranking.pytorch.layers.blocks", ["LogisticRegression", "FactorizationMachine", "MLP_Block"]),
    "pytorch.layers.blocks.logistic_regression": ("recbox_amd.ranking.pytorch.layers.blocks", ["LogisticRegression"]),
    "pytorch.layers.blocks.factorization_machine": ("recbox_amd.ranking.pytorch.layers.blocks", ["FactorizationMachine"]),
    "pytorch.layers.blocks.mlp_block": ("recbox_amd.ranking.pytorch.layers.blocks", ["MLP_Block"]),
    "pytorch.layers.attentions": ("recbox_amd.ranking.pytorch.layers.attentions",
                                  ["ScaledDotProductAttention", "MultiHeadTargetAttention", "SqueezeExcitation", "DIN_Attention"]),
    "pytorch.layers.attentions.dot_product_attention": ("recbox_amd.ranking.pytorch.layers.attentions", ["ScaledDotProductAttention"]),
    "pytorch.layers.attentions.target_attention": ("recbox_amd.ranking.pytorch.layers.attentions",
                                                   ["MultiHeadTargetAttention", "DIN_Attention"]),
    "pytorch.layers.attentions.squeeze_excitation": ("recbox_amd.ranking.pytorch.layers.attentions", ["SqueezeExcitation"]),
}
_CORE = {
    "metrics": ("recbox_amd.core.metrics", None),
    "pytorch.layers": ("recbox_amd.core.pytorch.layers", None),
    "pytorch.layers.embedding": ("recbox_amd.core.pytorch.layers.embedding", ["EmbeddingLayer", "EmbeddingDictLayer"]),
    "pytorch.layers.sequence": ("recbox_amd.core.pytorch.layers.sequence", ["MaskedAveragePooling", "MaskedSumPooling"]),
    "pytorch.layers.mlp": ("recbox_amd.core.pytorch.layers.mlp", ["MLP_Layer"]),
    "pytorch.losses": ("recbox_amd.core.pytorch.losses", None),
    "pytorch.losses.softmax_crossentropy_loss": ("recbox_amd.core.pytorch.losses", ["SoftmaxCrossEntropyLoss"]),
    "pytorch.losses.sigmoid_crossentropy_loss": ("recbox_amd.core.pytorch.losses", ["SigmoidCrossEntropyLoss"]),
    "pytorch.losses.pairwise_logistic_loss": ("recbox_amd.core.pytorch.losses", ["PairwiseLogisticLoss"]),
    "pytorch.losses.pairwise_margin_loss": ("recbox_amd.core.pytorch.losses", ["PairwiseMarginLoss"]),
    "pytorch.losses.mse_loss": ("recbox_amd.core.pytorch.losses", ["MSELoss"]),
    "pytorch.losses.cosine_contrastive_loss": ("recbox_amd.core.pytorch.losses", ["CosineContrastiveLoss"]),
}
_MATCHING = {
    "features": ("recbox_amd.matching.features", None),
    "pytorch.dataloaders": ("recbox_amd.matching.pytorch.dataloaders", None),
    "pytorch.dataloaders.h5_generator": ("recbox_amd.matching.pytorch.dataloaders.h5_generator", None),
}
_RECHUB = {
    "basic.features": ("recbox_amd.rechub.basic.features", ["SparseFeature", "SequenceFeature", "DenseFeature"]),
    "basic.initializers": ("recbox_amd.rechub.basic.initializers", None),
    "basic.layers": ("recbox_amd.rechub.basic.layers", None),
    "basic.activation": ("recbox_amd.rechub.basic.layers", ["Dice", "activation_layer"]),
    "models.matching": ("recbox_amd.rechub.models.matching", ["DSSM", "YoutubeDNN", "SASRec"]),
    "models.matching.dssm": ("recbox_amd.rechub.models.matching", ["DSSM"]),
    "models.matching.youtube_dnn": ("recbox_amd.rechub.models.matching", ["YoutubeDNN"]),
    "models.matching.sasrec": ("recbox_amd.rechub.models.matching", ["SASRec", "PointWiseFeedForward"]),
    "models.ranking": ("recbox_amd.rechub.models.ranking", ["DeepFM"]),
    "models.ranking.deepfm": ("recbox_amd.rechub.models.ranking", ["DeepFM"]),
}


def alias_table():
    """{reference dotted path: (module of this package, names or None)} for every path ``install()`` serves."""
    table = {}
    for root, sub in (("recbox.ranking", _RANKING), ("fuxictr", _RANKING), ("recbox.core", _CORE), ("recbox.matching", _MATCHING),
                      ("torch_rechub", _RECHUB), ("recbox.third_party.rechub", _RECHUB)):
        for rel, target in sub.items():
            table[root + "." + rel] = target
    return table


def _public(mod, names):
    if names is None:
        names = getattr(mod, "__all__", None) or [n for n in vars(mod) if not n.startswith("_")]
    return [n for n in names if hasattr(mod, n) and not isinstance(getattr(mod, n), types.ModuleType)]


def _ensure_module(name, created):
    """The module registered under ``name``: the real one if the reference is importable, else an empty package."""
    mod = sys.modules.get(name)
    if mod is not None:
        return mod
    try:
        mod = importlib.import_module(name)
        return mod
    except Exception:                                   # noqa: BLE001 -- not installed, or its own imports are missing here
        sys.modules.pop(name, None)
    mod = types.ModuleType(name, "registered by recbox_amd.compat.install(): HIP-backed mirrors under the reference's path")
    mod.__path__ = []                                   # a package: `import a.b.c` walks through it
    mod.__recbox_amd__ = True
    sys.modules[name] = mod
    created.append(name)
    if "." in name:
        parent, _, leaf = name.rpartition(".")
        setattr(_ensure_module(parent, created), leaf, mod)
    return mod


def install(prefixes=("recbox", "fuxictr", "torch_rechub"), overlay=True):
    """Register the mirrors under the reference's dotted paths (see the module docstring).  ``prefixes``: which top-level
    spellings to serve.  ``overlay``: when a path resolves to a module of an installed reference, re-bind the hot-path
    names on it (False: leave installed modules alone, only fill in what cannot be imported).  Returns
    {"created": [...], "patched": {path: [names]}}; ``uninstall(report)`` undoes it."""
    created, patched, saved = [], {}, {}
    for path, (target, names) in sorted(alias_table().items()):
        if not any(path == p or path.startswith(p + ".") for p in prefixes):
            continue
        src = importlib.import_module(target)
        n_before = len(created)
        mod = _ensure_module(path, created)
        fresh = path in created[n_before:]
        if not fresh and not overlay and not getattr(mod, "__recbox_amd__", False):
            continue
        bound = []
        for n in _public(src, names):
            if not fresh and hasattr(mod, n):
                saved.setdefault(path, {})[n] = getattr(mod, n)
            setattr(mod, n, getattr(src, n))
            bound.append(n)
        if fresh or getattr(mod, "__recbox_amd__", False):
            mod.__all__ = sorted(set(getattr(mod, "__all__", [])) | set(bound))
        patched[path] = bound
    return {"created": created, "patched": patched, "saved": saved}


def uninstall(report):
    """Undo an ``install()``: drop the modules it created, restore the names it re-bound on installed ones."""
    for path, names in report.get("patched", {}).items():
        mod = sys.modules.get(path)
        if mod is None or path in report["created"]:
            continue
        for n in names:
            old = report.get("saved", {}).get(path, {}).get(n)
            if old is not None:
                setattr(mod, n, old)
            elif hasattr(mod, n):
                delattr(mod, n)
    for name in sorted(report.get("created", []), key=len, reverse=True):
        mod = sys.modules.pop(name, None)
        if mod is not None and "." in name:
            parent, _, leaf = name.rpartition(".")
            pm = sys.modules.get(parent)
            if pm is not None and getattr(pm, leaf, None) is mod:
                delattr(pm, leaf)
