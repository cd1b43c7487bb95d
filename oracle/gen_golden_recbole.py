"""Second oracle: fixtures from the RecBole layers the reference vendors under
recbox/third_party/recbole/model/layers.py -- ``FMEmbedding`` + ``BaseFactorizationMachine`` (:127-205) and the
``MultiHeadAttention`` layer of its ``TransformerEncoder`` (:380-470) -- a SECOND, independently written statement of the
FM interaction and of the self-attention step than the FuxiCTR / rechub modules every other fixture comes from.

TEST INFRASTRUCTURE.  Runs in the dev container only (it imports /root/reference through oracle/ref_shim.py, with empty
stand-ins for the logging / table-printing packages RecBole's utils import at module scope and never use here):
    python oracle/gen_golden_recbole.py          -> tests/golden/recbole_layers.npz
The fixture holds inputs, the modules' state_dict, outputs and gradients: data, no reference source.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402


def recbole_layers():
    ref_shim.import_reference()
    for n in ("colorlog", "colorama", "texttable", "torch.utils.tensorboard"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["colorama"].init = getattr(sys.modules["colorama"], "init", lambda *a, **k: None)
    for mod, name in (("texttable", "Texttable"), ("colorlog", "ColoredFormatter"), ("torch.utils.tensorboard", "SummaryWriter")):
        if not hasattr(sys.modules[mod], name):
            setattr(sys.modules[mod], name, object)
    sys.modules.setdefault("recbole", importlib.import_module("recbox.third_party.recbole"))
    return importlib.import_module("recbox.third_party.recbole.model.layers")


def main():
    L = recbole_layers()
    out = {}
    # ---- FM: token fields with one shared table + offsets, then the FM term ----
    g = torch.Generator().manual_seed(1)
    field_dims = [7, 30, 5, 101, 13]
    offsets = np.array((0, *np.cumsum(field_dims)[:-1]), dtype=np.int64)
    B, D = 37, 8
    emb = L.FMEmbedding(field_dims, offsets, D)
    with torch.no_grad():
        emb.embedding.weight.copy_(torch.randn(sum(field_dims), D, generator=g) * 0.3)
    ids = torch.stack([torch.randint(0, v, (B,), generator=g) for v in field_dims], dim=1)
    e = emb(ids)
    fm_sum = L.BaseFactorizationMachine(reduce_sum=True)(e)
    fm_vec = L.BaseFactorizationMachine(reduce_sum=False)(e)
    R = torch.randn(B, 1, generator=g)
    (fm_sum * R).sum().backward()
    out.update({"fm.in.ids": ids.numpy(), "fm.in.offsets": offsets, "fm.in.R": R.numpy(),
                "fm.p.table": emb.embedding.weight.detach().numpy(), "fm.out.sum": fm_sum.detach().numpy(),
                "fm.out.vec": fm_vec.detach().numpy(), "fm.g.table": emb.embedding.weight.grad.numpy()})
    # ---- the attention layer of RecBole's TransformerEncoder (SASRec, recbole/model/sequential_recommender/sasrec.py) ----
    Bn, Ln, H, heads = 3, 20, 32, 2
    mha = L.MultiHeadAttention(n_heads=heads, hidden_size=H, hidden_dropout_prob=0.0, attn_dropout_prob=0.0, layer_norm_eps=1e-12)
    with torch.no_grad():
        for p in mha.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        mha.LayerNorm.weight.copy_(1.0 + 0.1 * torch.randn(H, generator=g))
    mha.eval()
    x = (torch.randn(Bn, Ln, H, generator=g)).requires_grad_()
    causal = torch.tril(torch.ones(Ln, Ln))
    mask = ((1.0 - causal) * -10000.0).view(1, 1, Ln, Ln).expand(Bn, 1, Ln, Ln)      # get_attention_mask's additive form
    y = mha(x, mask)
    R2 = torch.randn(Bn, Ln, H, generator=g)
    (y * R2).sum().backward()
    out.update({"mha.in.x": x.detach().numpy(), "mha.in.R": R2.numpy(), "mha.in.heads": np.array(heads),
                "mha.out.y": y.detach().numpy(), "mha.g.x": x.grad.numpy()})
    for k, v in mha.state_dict().items():
        out["mha.p." + k] = v.numpy()
    for k, p in mha.named_parameters():
        out["mha.g." + k] = p.grad.numpy()
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "recbole_layers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, sorted(out))


if __name__ == "__main__":
    main()
