"""EXPERIMENT (world of one through RCCL): ShardedFMStep as 8 hipGraph pieces with the exchanges between them vs the WHOLE step,
exchanges included (rbx_all_to_all = grouped ncclSend/ncclRecv on the capturing stream), in one hipGraph.
    python -u profiles/ubench/whole_graph_sharded.py
Measured on MI355X: 0.642 ms (pieces, direct exchanges) vs 0.586 ms (one graph), identical loss and gradients;
dist.destroy_process_group() then hangs (RCCL teardown after a capture), so the script leaves through os._exit."""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from recbox_amd import comm, ops
from recbox_amd.graph import ShardedFMStep
from recbox_amd.ranking.pytorch.models import ShardedFM
ops.config.check_ids = False
comm.direct.enable(True)
fmw = bench.CriteoFeatureMap(16)
X, y = bench.slice_inputs(fmw.fm, bench.synthetic_batch(65536, 1, "uniform", dev))
res = {}
for mode in (True, "whole"):
    model = ShardedFM(fmw.fm, 16, shard_min_vocab=100000, capacity_factor=1.25).to(dev)
    bench.init_weights(model)
    if mode == "whole":
        eager = ShardedFMStep(model, X, y, graphs=False)          # the same pieces, captured together below
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                             # warm up off the legacy stream, as GraphedStep does
            for _ in range(2):
                eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            eager._run(eager.pieces)

        def step():
            g.replay()
            return eager.loss
    else:
        step = ShardedFMStep(model, X, y, graphs=mode)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e3
    res[mode] = (dt, float(loss), [p.grad.clone() for p in model.parameters()])
    print("graphs=%r: %.3f ms/step loss %.6f" % (mode, dt, float(loss)))
same = all(torch.equal(a, b) for a, b in zip(res[True][2], res["whole"][2]))
print("gradients identical:", same)
sys.stdout.flush()
os._exit(0)        # destroy_process_group() hangs after RCCL kernels were captured
