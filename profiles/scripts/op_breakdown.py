"""ATen operators that still launch kernels inside one eager step of a bench config (torch.profiler, device time):
python profiles/scripts/op_breakdown.py youtubednn|deepfm|sasrec"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from recbox_amd import ops

cfg = sys.argv[1]
dev = torch.device("cuda:0")
ops.config.check_ids = False
ops.config.reuse_grad_buffers = "all" if cfg != "sasrec" else True
B = 65536 if cfg != "sasrec" else 4096
if cfg == "youtubednn":
    from recbox_amd.rechub.models.matching import YoutubeDNN
    V = 10_000_000
    with torch.device(dev):
        model = YoutubeDNN(*bench._youtube_features(V, 128), {"dims": [256, 128], "activation": "relu"}, temperature=0.02)
    x = bench._youtube_batch(B, V, 50, 4, 1, "uniform", dev)
    loss_of = lambda: ops.softmax_cross_entropy(model(x))
elif cfg == "deepfm":
    from recbox_amd.rechub.models.ranking import DeepFM
    dense, sparse = bench._deepfm_features(64)
    with torch.device(dev):
        model = DeepFM(dense + sparse, sparse, {"dims": [400, 400, 400], "dropout": 0.0, "activation": "relu"})
    x = bench._deepfm_batch(B, 1, "uniform", dev)
    loss_of = lambda: ops.binary_cross_entropy(model(x), x["label"])
else:
    from recbox_amd.rechub.models.matching import SASRec
    V = 1_000_000
    with torch.device(dev):
        model = SASRec(bench._sasrec_features(V, 64), max_len=200, dropout_rate=0.0, num_blocks=2, num_heads=1)
    x = bench._sasrec_batch(B, V, 200, 1, dev)
    loss_of = lambda: bench._sasrec_loss(model, x)
bench.init_weights_device(model, dev, 0, 0)
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    loss = loss_of()
    loss.backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "device_time_total", None)
    if t is None:
        t = getattr(e, "cuda_time_total", 0)
    self_t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if self_t > 0 and e.key.startswith("aten::"):
        rows.append((self_t, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
print("# %s: aten operators with device time in one eager step (self device us, calls, op, input shapes)" % cfg)
for t, c, k, s in rows[:16]:
    print("%9.1f %4d  %-28s %s" % (t, c, k, s))
