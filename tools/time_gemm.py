"""Time acmil_gemm_{f32,f16x3,bf16x3} on the GEMM shapes of the TransMIL forward and the GA backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops

SHAPES = [  # name, M, N, K, trans_a, trans_b
    ("transmil qkv   x W^T", 100608, 1152, 384, False, True),
    ("transmil fc1   x W^T", 100000, 384, 768, False, True),
    ("transmil out   x W^T", 100490, 384, 384, False, True),
    ("ga bwd G  h W^T     ", 50000, 128, 256, False, True),
    ("ga bwd dh dS W      ", 50000, 256, 128, False, False),
    ("ga bwd dWv dS^T h   ", 128, 256, 50000, True, False),
    ("ga bwd dW1 dh^T x   ", 256, 512, 50000, True, False),
]
dev = torch.device("cuda", 0)
for name, M, N, K, ta, tb in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    line = "%s M=%6d N=%5d K=%6d:" % (name, M, N, K)
    for prec in ("fp32", "f16x3", "bf16x3"):
        for _ in range(3):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, precision=prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, precision=prec)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += "  %s %7.1f us (%5.0f TF)" % (prec, us, 2.0 * M * N * K / us / 1e6)
    print(line)
