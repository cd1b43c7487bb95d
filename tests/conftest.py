import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


class Fixture(object):
    """A golden .npz split into its groups: in / p / out / g (+ extras)."""

    def __init__(self, name):
        self.name = name
        raw = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.groups = {}
        for key in raw.files:
            grp, rest = key.split(".", 1)
            self.groups.setdefault(grp, {})[rest] = raw[key]

    def __getitem__(self, grp):
        return self.groups[grp]

    def tensors(self, grp, device="cpu"):
        import torch
        return {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in self.groups[grp].items()}


@pytest.fixture
def golden():
    return Fixture


def load_params(module, params):
    """Load a golden 'p' group (numpy) into a module, strictly."""
    import torch
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return module


# Achieved errors of every assert_close of the session, per test: written to gpurun_out/parity_errors.txt (and, on request,
# RECBOX_PARITY_LEDGER=<path>) when the session ends, so that the margin under each tolerance is on file, not just "passed"
# ("smoke reports 3.6e-7, so the kernels are probably far inside -- the tests do not show it").
LEDGER = {}


def _note(what, err, tol):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    ent = LEDGER.setdefault(test, {})
    key = (what or "-", float(tol))
    ent[key] = max(ent.get(key, 0.0), float(err))


def pytest_sessionfinish(session, exitstatus):
    if not LEDGER:
        return
    paths = [os.environ.get("RECBOX_PARITY_LEDGER")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "parity_errors.txt"))
    lines = ["# test :: what : max abs error achieved / tolerance asserted (every assert_close of the session)"]
    for test in sorted(LEDGER):
        for (what, tol), err in sorted(LEDGER[test].items()):
            lines.append("%s :: %s : %.3e / %.1e" % (test, what, err, tol))
    for path in paths:
        if path:
            try:
                with open(path, "a") as fh:
                    fh.write("\n".join(lines) + "\n")
            except OSError:
                pass


def assert_close(a, b, tol=1e-4, what="", rtol=1e-5):
    import torch
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    if a.numel() == 0:
        return
    # absolute bar `tol` on O(1) values; fp32-relative slack for huge ones (e.g. x / 1e-12)
    excess = ((a - b).abs() - rtol * b.abs()).max().item()
    _note(what, (a - b).abs().max().item(), tol)
    assert excess <= tol, "%s max abs err %.3e > %.1e" % (what, (a - b).abs().max().item(), tol)


def assert_grads_close(module, ggroup, tol=1e-4):
    for name, p in module.named_parameters():
        got = p.grad if p.grad is not None else p.new_zeros(p.shape)
        assert_close(got, ggroup[name], tol, "grad " + name)
