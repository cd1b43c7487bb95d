"""Bisect the training-mode DeepFM gradient mismatch (r04): a rechub MLP tower on the GPU path against the same modules in
float64 on the CPU, by depth and batch size, backward driven by a random linear functional."""
import sys
sys.path.insert(0, ".")
import torch
from recbox_amd.rechub.basic.layers import MLP


def run(B, K, dims, out_layer, seed=0, gamma=1.0):
    torch.manual_seed(seed)
    m = MLP(K, output_layer=out_layer, dims=dims, dropout=0.0, activation="relu").cuda()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_((torch.randn(p.shape, generator=g) * 0.05).cuda())
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_((gamma + 0.1 * torch.randn(mod.num_features, generator=g)).cuda())
                mod.bias.copy_((0.1 * torch.randn(mod.num_features, generator=g)).cuda())
    import copy
    ref = torch.nn.Sequential(*[copy.deepcopy(x).cpu().double() for x in m.mlp]).train()
    m.train()
    x = torch.randn(B, K, generator=g)
    xc = x.cuda().requires_grad_()
    xr = x.double().requires_grad_()
    y = m(xc)
    R = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * R.float().cuda()).sum().backward()
    yr = ref(xr)
    (yr * R).sum().backward()
    out = ["B=%d K=%d dims=%s out=%s gamma=%g | y %.2e | dx %.2e (max %.2g)" % (
        B, K, dims, out_layer, gamma, float((y.detach().cpu().double() - yr).abs().max()),
        float((xc.grad.cpu().double() - xr.grad).abs().max()), float(xr.grad.abs().max()))]
    for (n, p), (_, q) in zip(m.mlp.named_parameters(), ref.named_parameters()):
        out.append("   %-10s err %.2e / max %.2g" % (n, float((p.grad.cpu().double() - q.grad).abs().max()), float(q.grad.abs().max())))
    print("\n".join(out), flush=True)


for B in (1000, 8192):
    run(B, 400, [400], False)
    run(B, 400, [400, 400], False)
    run(B, 1677, [400, 400, 400], True)
run(8192, 400, [400, 400], False, gamma=0.05)
import os
os.environ["RECBOX_AMD_GEMM_BX6"] = "0"
