"""Does the attention forward want more workgroups per CU?  Same total work, sequence lengths whose K, V fit the LDS once (L = 200,
224: 104-116 KB, one 8-wave workgroup per CU) or twice (L = 128, 96: 50-66 KB); useful causal FLOPs = 2 * 2 * (L (L + 1) / 2) * d."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from recbox_amd import ops

def run(B, L, D=64):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, L, D, generator=g).cuda().view(B, 1, L, D).requires_grad_(True)
    k = torch.randn(B, L, D, generator=g).cuda().view(B, 1, L, D).requires_grad_(True)
    v = torch.randn(B, L, D, generator=g).cuda().view(B, 1, L, D).requires_grad_(True)
    f = lambda: ops.attention(q, k, v, scale=D ** -0.5, causal=True, fill=float("-inf"))[0]
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): o = f()
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 10 * 1e-3
    go = torch.randn_like(o)
    for _ in range(2): o = f(); o.backward(go)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        o = f(); o.backward(go)
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / 5 * 1e-3 - tf
    useful = 4.0 * B * (L * (L + 1) / 2) * D
    nT = (L + 31) // 32
    executed = 4.0 * B * (nT * (nT + 1) / 2) * 32 * 32 * D
    print("L=%3d B=%5d  fwd %7.1f us  useful %5.1f TF (%.2f)  executed %5.1f TF (%.2f)   bwd %7.1f us  useful %5.1f TF" % (
        L, B, tf * 1e6, useful / tf / 1e12, useful / tf / 157.3e12, executed / tf / 1e12, executed / tf / 157.3e12,
        tb * 1e6, 2.5 * useful / tb / 1e12))

for L, B in ((224, 3657), (200, 4096), (192, 4267), (128, 6400), (96, 8533), (64, 12800)):
    run(B, L)
