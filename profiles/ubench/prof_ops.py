import sys, os
sys.path.insert(0, "/root/repo")
import torch
from bench import CriteoFeatureMap, synthetic_batch, slice_inputs, init_weights
from recbox_amd import ops
from recbox_amd.ranking.pytorch.models import FM
from recbox_amd.ranking.pytorch.torch_utils import get_loss
ops.config.check_ids = False
fmw = CriteoFeatureMap(16)
model = FM(fmw.fm, 16).cuda(); init_weights(model)
batch = synthetic_batch(65536, 1, "uniform", "cuda")
X, y = slice_inputs(fmw.fm, batch)
loss_fn = get_loss("bce")
def step():
    model.zero_grad(set_to_none=True)
    prob = model(X)["y_pred"]
    loss = loss_fn(prob, y, reduction="mean")
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA or "emcpy" in e.name or "copy" in e.name.lower()]
seen = []
for e in prof.events():
    n = e.name
    if ("Memcpy" in n or "copy_" in n or "fill_" in n or "zero_" in n or "aten::to" in n or "contiguous" in n or "clone" in n or "aten::zeros" in n) and e.device_type != torch.autograd.DeviceType.CUDA:
        seen.append((n, [str(s) for s in (e.input_shapes or [])][:2]))
from collections import Counter
for k, v in Counter((n, tuple(s)) for n, s in seen).most_common(25):
    print(v, k)
