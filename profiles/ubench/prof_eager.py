import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import CriteoFeatureMap, synthetic_batch, slice_inputs, init_weights
from recbox_amd import ops
from recbox_amd.ranking.pytorch.models import FM
ops.config.check_ids = False
fmw = CriteoFeatureMap(16)
model = FM(fmw.fm, 16).cuda()
init_weights(model)
batch = synthetic_batch(65536, 1, "uniform", "cuda")
X, y = slice_inputs(fmw.fm, batch)
def step():
    model.zero_grad(set_to_none=True)
    prob = model(X)["y_pred"]
    loss = torch.nn.functional.binary_cross_entropy(prob, y, reduction="mean")
    loss.backward()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
