"""Timing of the packed-weight Linear kernel vs the generic split-f16 GEMM on the TransMIL cfg4 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops
for (m, k, n) in [(100000, 768, 384), (100608, 384, 1152), (100490, 384, 384), (50000, 512, 256), (50000, 1024, 512)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.03
    packed = ops.linear_pack(w)
    y = torch.empty(m, n, device="cuda")
    for name, fn in (("linear", lambda: ops.linear_f16x3(x, packed, n, out=y)), ("gemm  ", lambda: ops.gemm(x, w, trans_b=True, precision="f16x3", out=y))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%s M=%d K=%d N=%d: %.1f us  %.0f TF algorithmic (%.0f executed)" % (name, m, k, n, us, 2.0 * m * k * n / us / 1e6, 6.0 * m * k * n / us / 1e6), flush=True)
