"""The towers' stand-alone activations on the HIP path (csrc/rbx_act.hip): nn.PReLU without a BatchNorm in front,
nn.Dropout(p > 0) in training, Dice -- against the reference's own module compositions
(/root/reference/recbox/core/pytorch/layers/activations.py:23-33, ranking/pytorch/layers/blocks/mlp_block.py:42-58)
restated with torch CPU float64 operations."""
import pytest
import torch
from torch import nn

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


class _RefDice(nn.Module):
    """activations.py:23-33, verbatim arithmetic, float64."""

    def __init__(self, input_dim, eps=1e-9):
        super().__init__()
        self.bn = nn.BatchNorm1d(input_dim, affine=False, eps=eps, momentum=0.01)
        self.alpha = nn.Parameter(torch.zeros(input_dim))

    def forward(self, X):
        p = torch.sigmoid(self.bn(X))
        return p * X + self.alpha * (1 - p) * X


@pytest.mark.parametrize("rows,cols", [(1, 5), (7, 3), (256, 64), (1000, 400), (4097, 33)])
def test_dice_matches_the_reference_composition(rows, cols):
    from recbox_amd.ranking.pytorch.layers.attentions import Dice
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g, dtype=torch.float64) * 2 + 0.3
    w = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    ref = _RefDice(cols).double()
    dut = Dice(cols).cuda()
    with torch.no_grad():
        ref.alpha.copy_(torch.randn(cols, generator=g, dtype=torch.float64) * 0.5)
        dut.alpha.copy_(ref.alpha.float().cuda())
    for training in ((True, False) if rows > 1 else (False,)):       # (BatchNorm1d refuses one row in training, both sides)
        ref.train(training), dut.train(training)
        xr = x.clone().requires_grad_()
        xd = x.float().cuda().requires_grad_()
        ref.zero_grad(), dut.zero_grad()
        yr, yd = ref(xr), dut(xd)
        assert_close(yd, yr.float(), TOL, "dice forward (training=%s)" % training)
        (yr * w).sum().backward()
        (yd * w.float().cuda()).sum().backward()
        scale = max(1.0, float(xr.grad.abs().max()))
        assert_close(xd.grad / scale, (xr.grad / scale).float(), TOL, "dice dx (training=%s)" % training)
        scale = max(1.0, float(ref.alpha.grad.abs().max()))
        assert_close(dut.alpha.grad / scale, (ref.alpha.grad / scale).float(), TOL, "dice dalpha")
    assert_close(dut.bn.running_mean, ref.bn.running_mean.float(), TOL, "running mean")
    assert_close(dut.bn.running_var, ref.bn.running_var.float(), TOL, "running var")
    assert int(dut.bn.num_batches_tracked) == int(ref.bn.num_batches_tracked)


@pytest.mark.parametrize("rows,cols,n_slope", [(1, 1, 1), (300, 17, 1), (300, 17, 17), (5000, 400, 400), (64, 256, 1)])
def test_standalone_prelu_matches_torch(rows, cols, n_slope):
    from recbox_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    w = torch.randn(rows, cols, generator=g, dtype=torch.float64)
    ref = nn.PReLU(n_slope).double()
    dut = nn.PReLU(n_slope).cuda()
    with torch.no_grad():
        ref.weight.copy_(torch.rand(n_slope, generator=g, dtype=torch.float64) - 0.3)
        dut.weight.copy_(ref.weight.float().cuda())
    xr, xd = x.clone().requires_grad_(), x.float().cuda().requires_grad_()
    yr, yd = ref(xr), ops.prelu(xd, dut)
    assert_close(yd, yr.float(), 1e-6, "prelu forward")
    (yr * w).sum().backward()
    (yd * w.float().cuda()).sum().backward()
    assert_close(xd.grad, xr.grad.float(), 1e-6, "prelu dx")
    scale = max(1.0, float(ref.weight.grad.abs().max()))
    assert_close(dut.weight.grad / scale, (ref.weight.grad / scale).float(), TOL, "prelu dslope")


def test_dropout_keeps_the_rate_scales_and_replays_its_mask_in_the_backward():
    from recbox_amd import ops
    torch.manual_seed(5)
    x = torch.ones(1 << 20, device="cuda").requires_grad_()
    for p in (0.1, 0.5, 0.9):
        y = ops.dropout(x, p, True, seed=1234)
        kept = y != 0
        rate = float(kept.float().mean())
        assert abs(rate - (1 - p)) < 4e-3, (p, rate)                         # 2^20 draws: sigma <= 5e-4
        assert_close(y[kept], torch.full_like(y[kept], 1.0 / (1 - p)), 1e-6, "scale of the kept values")
        x.grad = None
        (y * 3.0).sum().backward()
        assert torch.equal(x.grad != 0, kept)                                 # the backward evaluates the same mask
        assert_close(x.grad[kept], torch.full_like(y[kept], 3.0 / (1 - p)), 1e-6, "dx")
        assert torch.equal(ops.dropout(x, p, True, seed=1234), y)             # a function of (seed, element) only
        assert not torch.equal(ops.dropout(x, p, True, seed=1235), y)
    assert ops.dropout(x, 0.3, False) is x and ops.dropout(x, 0.0, True) is x   # evaluation / p = 0: nothing is launched
    # neighbouring elements are independent: the lag-1 correlation of the keep mask is ~0
    k = (ops.dropout(x, 0.5, True, seed=77) != 0).float()
    corr = float(((k[1:] - 0.5) * (k[:-1] - 0.5)).mean() / 0.25)
    assert abs(corr) < 5e-3, corr


@pytest.mark.parametrize("act,bn,drop", [("Dice", False, 0.0), ("PReLU", False, 0.0), ("ReLU", True, 0.25), ("Dice", True, 0.2)])
def test_mlp_block_with_these_activations_runs_no_aten_activation_kernels(act, bn, drop):
    """ranking MLP_Block (mlp_block.py:23-61) with Dice / stand-alone PReLU / Dropout: evaluation-mode and training outputs
    against the float64 composition (Dropout: evaluation exact; training statistically through its own test above), and the
    kernels the training step launches are the library's -- no at::native activation / dropout / BatchNorm kernel."""
    from recbox_amd.ranking.pytorch.layers.attentions import Dice
    from recbox_amd.ranking.pytorch.layers.blocks import MLP_Block
    torch.manual_seed(3)
    B, D, H = 512, 48, [64, 32]
    acts = act                    # by NAME: fuxictr's get_activation builds Dice(units) / nn.PReLU(units, init=0.1) per layer
    dut = MLP_Block(D, hidden_units=H, hidden_activations=acts, output_dim=1, dropout_rates=drop, batch_norm=bn).cuda()
    # the float64 twin: the same module list with the reference's compositions
    mods = []
    for m in dut.mlp:
        if type(m) is nn.Linear:
            t = nn.Linear(m.in_features, m.out_features, bias=m.bias is not None).double()
            t.load_state_dict({k: v.double().cpu() for k, v in m.state_dict().items()})
        elif type(m) is nn.BatchNorm1d:
            t = nn.BatchNorm1d(m.num_features).double()
        elif isinstance(m, Dice):
            t = _RefDice(m.alpha.numel()).double()
            with torch.no_grad():
                m.alpha.copy_(torch.rand_like(m.alpha) - 0.5)
                t.alpha.copy_(m.alpha.double().cpu())
        elif type(m) is nn.PReLU:
            assert m.weight.numel() == mods[-1].out_features and float(m.weight[0].detach()) == pytest.approx(0.1)   # one slope per unit
            t = nn.PReLU(m.weight.numel(), init=0.1).double()
        elif type(m) is nn.Dropout:
            t = nn.Dropout(m.p)
        else:
            t = type(m)()
        mods.append(t)
    ref = nn.Sequential(*mods)
    x = torch.randn(B, D, dtype=torch.float64)
    if drop == 0.0:
        xr, xd = x.clone().requires_grad_(), x.float().cuda().requires_grad_()
        yr, yd = ref(xr), dut(xd)
        assert_close(yd, yr.float(), TOL, "training forward")
        yr.sum().backward(), yd.sum().backward()
        assert_close(xd.grad, xr.grad.float(), TOL, "dx")
    else:
        dut(x.float().cuda()).sum().backward()                        # (moves the BatchNorm statistics on the HIP side only)
        for a, b in zip(ref, dut.mlp):
            if type(a) is nn.BatchNorm1d:
                a.load_state_dict({k: v.double().cpu() if v.is_floating_point() else v.cpu() for k, v in b.state_dict().items()})
    ref.eval(), dut.eval()
    with torch.no_grad():
        assert_close(dut(x.float().cuda()), ref(x).float(), TOL, "evaluation forward")
    dut.train()
    from torch.profiler import ProfilerActivity, profile
    xd = x.float().cuda().requires_grad_()
    dut(xd).sum().backward()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        dut(xd).sum().backward()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if e.device_type is not None and "Memcpy" not in e.key and "Memset" not in e.key]
    bad = [n for n in names if "rbx::" not in n
           and any(s in n for s in ("dropout", "Dropout", "prelu", "PRelu", "batch_norm", "sigmoid", "threshold"))]
    assert not bad, bad
    assert any("rbx::" in n for n in names), names
