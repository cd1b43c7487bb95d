import copy, torch, sys
sys.path.insert(0, "/root/repo")
from recbox_amd import ops, dense
M, K = 8192, 300
for variant in ("plain", "neg", "zero", "beta"):
    torch.manual_seed(M)
    mods = torch.nn.Sequential(torch.nn.Linear(K, 256), torch.nn.BatchNorm1d(256), torch.nn.ReLU(),
                               torch.nn.Linear(256, 384), torch.nn.BatchNorm1d(384), torch.nn.ReLU(), torch.nn.Linear(384, 1))
    with torch.no_grad():
        for bn in (mods[1], mods[4]):
            if variant == "neg": bn.weight.copy_(torch.randn_like(bn.weight))
            if variant == "zero": bn.weight[::37] = 0.0
            if variant == "beta": bn.bias.copy_(0.5 * torch.randn_like(bn.bias))
    ref = copy.deepcopy(mods).double()
    x = torch.randn(M, K); r = torch.randn(M, 1)
    xr = x.double().requires_grad_(True)
    (ref(xr) * r.double()).sum().backward()
    for fused in (True, False, "fwd"):
        m = copy.deepcopy(mods).cuda().train()
        xc = x.cuda().requires_grad_(True)
        ops.config.bn_in_gemm = fused
        y = dense.run_sequential(m, xc)
        (y * r.cuda()).sum().backward()
        errs = []
        for (n, p), (_, p1) in zip(ref.named_parameters(), m.named_parameters()):
            errs.append("%s %.1e/%.1e" % (n, (p1.grad.cpu() - p.grad.float()).abs().max(), p.grad.abs().max()))
        print(variant, fused, " ".join(errs), "dx %.1e" % (xc.grad.cpu() - xr.grad.float()).abs().max())
