import torch, time
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it*1e-3
for (M,K,N) in [(65536,400,400),(65536,1677,400),(65536,1680,400),(819200,64,64),(1024,128,1000000),(8192,4096,4096)]:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda')
    s=t(lambda: torch.nn.functional.linear(x,w))
    print("rocBLAS/hipBLASLt fp32 linear [%d,%d]x[%d,%d]^T  %8.1f us  %6.1f TFLOP/s" % (M,K,N,K,s*1e6, 2.0*M*K*N/s/1e12))
    g=torch.randn(M,N,device='cuda')
    s=t(lambda: g.t() @ x)
    print("   dW = g^T x                                   %8.1f us  %6.1f TFLOP/s" % (s*1e6, 2.0*M*K*N/s/1e12))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recbox_amd import ops
for (M,K,N) in [(65536,400,400),(65536,400,384),(65536,400,512),(65536,1680,400),(819200,64,64),(1024,128,1000000),(8192,4096,4096)]:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(N,K,device='cuda')
    s=t(lambda: ops.linear(x,w))
    print("rbx_linear_fwd [%d,%d]x[%d,%d]^T  %8.1f us  %6.1f TFLOP/s" % (M,K,N,K,s*1e6, 2.0*M*K*N/s/1e12))
