"""End-to-end fwd+bwd steps of BASELINE.json configs 3-5 on ONE GPU (config 2 is bench.py): the model mirrors
of recbox_amd.rechub built from the kept API, synthetic inputs per SURVEY.md 8d, no optimiser step.
    python profiles/ubench/models_bench.py [youtube] [deepfm] [sasrec]
Prints ms/step and samples/s for eager launches and (where the step captures) a hipGraph replay, next to the
CPU probe of the live reference quoted in SURVEY.md section 6 (8 host cores, indicative only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops  # noqa: E402
from recbox_amd.graph import GraphedStep  # noqa: E402
from recbox_amd.rechub.basic.features import DenseFeature, SequenceFeature, SparseFeature  # noqa: E402
from recbox_amd.rechub.models.matching import SASRec, YoutubeDNN  # noqa: E402
from recbox_amd.rechub.models.ranking import DeepFM  # noqa: E402

ops.config.check_ids = False
CRITEO = [1460, 583, 1000000, 1000000, 305, 24, 12517, 633, 3, 93145, 5683, 1000000, 3194, 27, 14992, 1000000, 10, 5652,
          2173, 4, 1000000, 18, 15, 286181, 105, 142572]


def init(model, std=0.1):
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))


def measure(name, step, batch, cpu_probe, steps=20, graph=True, persistent=False):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    line = "%-44s eager %8.3f ms  %10.0f samples/s" % (name, eager * 1e3, batch / eager)
    if graph:
        try:
            g = GraphedStep(step, warmup=2)
            for _ in range(3):
                g()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g()
            torch.cuda.synchronize()
            rep = (time.perf_counter() - t0) / steps
            line += "   hipGraph %8.3f ms  %10.0f samples/s" % (rep * 1e3, batch / rep)
        except Exception as exc:                                  # a step that cannot be captured stays eager
            line += "   (graph capture failed: %s)" % type(exc).__name__
            torch.cuda.synchronize()
        if persistent:
            # every table of this model feeds exactly ONE lookup per step: its dense gradients may stay in a persistent
            # buffer of which only the previous step's rows are cleared (ops.config.reuse_grad_buffers = "all")
            g = GraphedStep(step, warmup=2, reuse_grads="all")
            for _ in range(3):
                g()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g()
            torch.cuda.synchronize()
            rep = (time.perf_counter() - t0) / steps
            line += "   hipGraph + persistent grads %8.3f ms  %10.0f samples/s" % (rep * 1e3, batch / rep)
    print(line + "   | reference CPU probe (SURVEY 6): %s" % cpu_probe, flush=True)


def youtube(B=65536, V=10_000_000, D=128, L=50, n_neg=4):
    g = torch.Generator().manual_seed(1)
    uf = [SparseFeature("user_id", 1_000_000, 16),
          SequenceFeature("hist", V, D, pooling="mean", shared_with="item", padding_idx=0)]
    itf = [SparseFeature("item", V, D)]
    negf = [SequenceFeature("neg_items", V, D, pooling="concat", shared_with="item")]
    model = YoutubeDNN(uf, itf, negf, {"dims": [256, 128], "activation": "relu"}, temperature=0.02).cuda()
    init(model)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    hist = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None, :] < lens[:, None])
    x = {"user_id": torch.randint(0, 1_000_000, (B,), generator=g).cuda(), "hist": hist.cuda(),
         "item": torch.randint(1, V, (B,), generator=g).cuda(), "neg_items": torch.randint(1, V, (B, n_neg), generator=g).cuda()}
    target = torch.zeros(B, dtype=torch.long, device="cuda")

    def step():
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), target)
        loss.backward()
        return loss

    measure("cfg3 YoutubeDNN 10Mx128 one GPU, B=%d" % B, step, B, "4322 ms -> 15.2 k samples/s", persistent=True)


def deepfm(B=65536, D=64):
    g = torch.Generator().manual_seed(1)
    dense = [DenseFeature("I%d" % i) for i in range(13)]
    sparse = [SparseFeature("C%d" % i, v + 1, D) for i, v in enumerate(CRITEO)]
    model = DeepFM(dense + sparse, sparse, {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"}).cuda()
    init(model)
    x = {}
    for i in range(13):
        x["I%d" % i] = torch.rand(B, generator=g).cuda()
    for i, v in enumerate(CRITEO):
        x["C%d" % i] = torch.randint(1, v + 1, (B,), generator=g).cuda()
    y = (torch.rand(B, generator=g) < 0.25).float().cuda()

    def step():
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.binary_cross_entropy(model(x), y)
        loss.backward()
        return loss

    measure("cfg4 DeepFM D=64 MLP 3x400 (BatchNorm on), B=%d" % B, step, B, "3044 ms -> 21.5 k samples/s", persistent=True)


def sasrec(B=4096, V=1_000_000, D=64, L=200):
    g = torch.Generator().manual_seed(1)
    feats = [SequenceFeature("seq", V, D, pooling="concat"),
             SequenceFeature("pos", V, D, pooling="concat", shared_with="seq"),
             SequenceFeature("neg", V, D, pooling="concat", shared_with="seq")]
    model = SASRec(feats, max_len=L, dropout_rate=0.0, num_blocks=2, num_heads=1).cuda()
    init(model)
    lens = torch.randint(20, L + 1, (B,), generator=g)
    keep = torch.arange(L)[None, :] < lens[:, None]
    seq = torch.randint(1, V, (B, L), generator=g) * keep
    x = {"seq": seq.cuda(), "pos": (torch.randint(1, V, (B, L), generator=g) * keep).cuda(),
         "neg": (torch.randint(1, V, (B, L), generator=g) * keep).cuda()}
    mask = keep.cuda()

    def step():
        model.zero_grad(set_to_none=True)
        pos, neg = model(x)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pos[mask], torch.ones_like(pos[mask])) + \
            torch.nn.functional.binary_cross_entropy_with_logits(neg[mask], torch.zeros_like(neg[mask]))
        loss.backward()
        return loss

    measure("cfg5 SASRec V=1M D=64 L=200 2 blocks, B=%d" % B, step, B, "B=512: 1184 ms -> 430 samples/s", graph=False)


if __name__ == "__main__":
    which = sys.argv[1:] or ["youtube", "deepfm", "sasrec"]
    for name in which:
        globals()[name]()
        torch.cuda.empty_cache()
