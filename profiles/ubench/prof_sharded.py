import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from bench import CriteoFeatureMap, synthetic_batch, slice_inputs, init_weights
from recbox_amd import ops
from recbox_amd.ranking.pytorch.models import ShardedFM
ops.config.check_ids = False
fmw = CriteoFeatureMap(16)
model = ShardedFM(fmw.fm, 16).cuda()
init_weights(model)
batch = synthetic_batch(65536, 1, "uniform", "cuda")
X, y = slice_inputs(fmw.fm, batch)
def step():
    model.zero_grad(set_to_none=True)
    prob = model(X)["y_pred"]
    loss = torch.nn.functional.binary_cross_entropy(prob, y, reduction="mean")
    loss.backward()
    model.sync_grads()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:7000])
dist.destroy_process_group()
