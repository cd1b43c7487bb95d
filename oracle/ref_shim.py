"""Import shim for the READ-ONLY reference at /root/reference (dev container only).

TEST INFRASTRUCTURE -- never imported by the product path (recbox_amd/).

The reference (reczoo/RecBox v0.0.4) is pure Python on PyTorch; it imports a few
binary deps that are absent here and never used on the fwd/bwd path (SURVEY.md
section 8c).  This module stubs those, installs the two un-vendored aliases the
reference relies on (``fuxictr`` -> ``recbox.ranking``, ``torch_rechub`` ->
``recbox.third_party.rechub``) and returns the imported package.

It is only used by ``oracle/gen_golden.py`` (fixture generation) and by the
``not gpu`` tests that cross-check the oracle against the live reference when
/root/reference exists.  Nothing here travels to, or runs on, the GPU box.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("RECBOX_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "recbox"))


def _stub(name):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__dict__["__stub__"] = True
        sys.modules[name] = m
    return sys.modules[name]


def import_reference():
    """Return the reference ``recbox`` package (imported from REFERENCE_ROOT)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # binary deps that are imported at module scope but unused on the hot path
    for name in ("h5py", "faiss", "annoy", "pymilvus"):
        try:
            importlib.import_module(name)
        except Exception:
            _stub(name)
    if getattr(sys.modules.get("annoy"), "__stub__", False):
        sys.modules["annoy"].AnnoyIndex = object
    if getattr(sys.modules.get("pymilvus"), "__stub__", False):
        for n in ("Collection", "CollectionSchema", "DataType", "FieldSchema", "connections", "utility"):
            setattr(sys.modules["pymilvus"], n, object)
    recbox = importlib.import_module("recbox")
    # recbox.ranking is FuxiCTR under another top-level name
    if "fuxictr" not in sys.modules:
        sys.modules["fuxictr"] = importlib.import_module("recbox.ranking")
    if "torch_rechub" not in sys.modules:
        sys.modules["torch_rechub"] = importlib.import_module("recbox.third_party.rechub")
        sys.modules["torch_rechub.basic"] = importlib.import_module("recbox.third_party.rechub.basic")
        sys.modules["torch_rechub.basic.features"] = importlib.import_module(
            "recbox.third_party.rechub.basic.features")
        sys.modules["torch_rechub.basic.layers"] = importlib.import_module(
            "recbox.third_party.rechub.basic.layers")
    return recbox
